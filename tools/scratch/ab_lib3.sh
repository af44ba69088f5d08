cd $GRAFT_REPO_ROOT
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
timeout 900 python -m pytest tests -m gpu -x -q -k "fri or plonk or stark_prove or segment_proof_matches_oracle" 2>&1 | tail -2
cp zk_evm_amd/libzkstark_hip.so /tmp/new.so
for rep in 1 2 3; do
  for V in base new; do
    if [ $V = base ]; then cp tools/scratch/libzkstark_hip_base.so zk_evm_amd/libzkstark_hip.so; else cp /tmp/new.so zk_evm_amd/libzkstark_hip.so; fi
    python bench.py $QUICK 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('2^20 $V', round(b['ms_per_step'],2))"
  done
done
cp /tmp/new.so zk_evm_amd/libzkstark_hip.so
