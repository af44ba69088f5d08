cd $GRAFT_REPO_ROOT
TAG=${1:-r03c}
QUICK="--steps 5 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/${TAG}_gputests.log
timeout 900 python bench.py $QUICK --no-pmc > gpurun_out/${TAG}_bench_quick.json 2> gpurun_out/${TAG}_bench_quick.err
python tools/scratch/show.py gpurun_out/${TAG}_bench_quick.json | head -3
ZK_NTT_NT=1 timeout 900 python bench.py $QUICK > gpurun_out/${TAG}_bench_ntt_nt.json 2> gpurun_out/${TAG}_bench_ntt_nt.err
echo "---- ZK_NTT_NT=1"; python tools/scratch/show.py gpurun_out/${TAG}_bench_ntt_nt.json | grep -v "quotient\|   " 
timeout 600 python bench.py $QUICK --no-pmc --log-ns realistic > gpurun_out/${TAG}_bench_realistic.json 2>/dev/null
echo "---- realistic"; python tools/scratch/show.py gpurun_out/${TAG}_bench_realistic.json | head -3
