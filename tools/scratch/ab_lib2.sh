cd $GRAFT_REPO_ROOT
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${1:-r03m}_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/${1:-r03m}_gputests.log
cp zk_evm_amd/libzkstark_hip.so /tmp/new.so
for rep in 1 2; do
  for V in base new; do
    if [ $V = base ]; then cp tools/scratch/libzkstark_hip_base.so zk_evm_amd/libzkstark_hip.so; else cp /tmp/new.so zk_evm_amd/libzkstark_hip.so; fi
    python bench.py $QUICK 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('2^20 $V', round(b['ms_per_step'],2), {k: round(v,4) for k,v in b['segment_timing_s'].items()})"
    python bench.py $QUICK --log-ns realistic 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('real $V', round(b['ms_per_step'],2))"
  done
done
cp /tmp/new.so zk_evm_amd/libzkstark_hip.so
