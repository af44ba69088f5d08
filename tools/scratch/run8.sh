cd $GRAFT_REPO_ROOT
TAG=${1:-r03l}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/zktrace && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/zktrace -o tr -- python "$GRAFT_REPO_ROOT/bench.py" --log-ns realistic --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-secondary --commit-steps 0 --in-flight 1 --no-dist-selftest > /dev/null 2>&1
F=$(find /tmp/zktrace -name "*kernel_trace.csv" | head -1)
gzip -c "$F" > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace.csv.gz"
python "$GRAFT_REPO_ROOT/tools/gap_analysis.py" "$F" 0.5 __none__ 8 8 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_gaps_ctx.txt" 2>&1
head -20 "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_gaps_ctx.txt"
cd $GRAFT_REPO_ROOT
ZK_HOST_PROFILE=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest --log-ns realistic 2>&1 | grep "zk host\|ms_per_step" | cut -c1-300 | head -40
hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/hostperm tools/scratch/hostperm.hip 2>/dev/null && /tmp/hostperm; grep -m1 "model name" /proc/cpuinfo; nproc
