cd $GRAFT_REPO_ROOT
TAG=${1:-r03g}
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/${TAG}_gputests.log
for rep in 1 2; do
python bench.py $QUICK 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('2^20', round(b['ms_per_step'],2), {k: round(v,4) for k,v in b['segment_timing_s'].items()})"
python bench.py $QUICK --log-ns realistic 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('real', round(b['ms_per_step'],2), {k: round(v,4) for k,v in b['segment_timing_s'].items()})"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/zktrace && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/zktrace -o tr -- python "$GRAFT_REPO_ROOT/bench.py" --log-ns realistic --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-secondary --commit-steps 0 --in-flight 1 --no-dist-selftest > /dev/null 2>&1
F=$(find /tmp/zktrace -name "*kernel_trace.csv" | head -1)
if [ -n "$F" ]; then python "$GRAFT_REPO_ROOT/tools/gap_analysis.py" "$F" 0.5 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_gaps_realistic.txt" 2>&1; head -16 "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_gaps_realistic.txt"; fi
