# r03v: PLONK quotient accumulators in registers / global loads in the wide-gate evaluator
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "plonk" 2>&1 | tail -2
cp zk_evm_amd/libzkstark_hip.so /tmp/new.so
for rep in 1 2 3; do
  for V in base5 new; do
    if [ $V = base5 ]; then cp tools/scratch/libzkstark_hip_base5.so zk_evm_amd/libzkstark_hip.so; else cp /tmp/new.so zk_evm_amd/libzkstark_hip.so; fi
    echo -n "$V 2^13: "; timeout 300 python tools/plonk_trace.py 13 20 2>/dev/null | tail -1
    echo -n "$V 2^12: "; timeout 300 python tools/plonk_trace.py 12 20 2>/dev/null | tail -1
  done
done
for V in base5 new; do
  if [ $V = base5 ]; then cp tools/scratch/libzkstark_hip_base5.so zk_evm_amd/libzkstark_hip.so; else cp /tmp/new.so zk_evm_amd/libzkstark_hip.so; fi
  echo -n "$V 2^13 x4 workers: "; timeout 300 python tools/plonk_trace.py 13 20 4 2>/dev/null | tail -1
  echo -n "$V 2^14: "; timeout 300 python tools/plonk_trace.py 14 10 2>/dev/null | tail -1
done
cp /tmp/new.so zk_evm_amd/libzkstark_hip.so
