cd $GRAFT_REPO_ROOT
TAG=${1:-r03p}
timeout 900 python -m pytest tests -m gpu -x -q -k "fri or plonk or stark_prove or segment_proof_matches_oracle or fuzz" 2>&1 | tail -2
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
for rep in 1 2; do
python bench.py $QUICK 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('2^20', round(b['ms_per_step'],2))"
python bench.py $QUICK --log-ns realistic 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('real', round(b['ms_per_step'],2))"
done
hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/hostperm tools/scratch/hostperm.hip 2>/dev/null && /tmp/hostperm
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/zktrace && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/zktrace -o tr -- python "$GRAFT_REPO_ROOT/bench.py" --log-ns realistic --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-secondary --commit-steps 0 --in-flight 1 --no-dist-selftest > /dev/null 2>&1
F=$(find /tmp/zktrace -name "*kernel_trace.csv" | head -1)
gzip -c "$F" > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace.csv.gz"
python "$GRAFT_REPO_ROOT/tools/gap_analysis.py" "$F" 0.5 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_gaps_realistic.txt" 2>&1
head -12 "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_gaps_realistic.txt"
