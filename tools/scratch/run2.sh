cd $GRAFT_REPO_ROOT
TAG=${1:-r03b}
if [ "${2:-tests}" = tests ]; then
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -15 gpurun_out/${TAG}_gputests.log
fi
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 > gpurun_out/${TAG}_bench_quick.json 2> gpurun_out/${TAG}_bench_quick.err
python tools/scratch/show.py gpurun_out/${TAG}_bench_quick.json
