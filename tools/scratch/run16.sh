# r03t visit 5: (a) which of the three NTT changes costs the values -> coefficients direction, (b) cooperative permutation with three layers per step
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 000 001 010 011 100 101 110 111; do
  echo -n "tw/tile/lds=$v 116x2^20: "; tools/scratch/kb/kbench_$v 116 20 5 | head -1
done
done
for v in 000 111 100 010; do echo -n "tw/tile/lds=$v 2431x2^17: "; tools/scratch/kb/kbench_$v 2431 17 3 | head -1; done
for v in 000 111; do echo -n "tw/tile/lds=$v 30x2^21: "; tools/scratch/kb/kbench_$v 30 21 5 | head -1; done
for v in 000 111; do tools/scratch/kb/kbench_$v 116 20 1 | tail -1; done
# (b) coop
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=b.get('segment_timing_s',{}); c=b.get('commit_stages_ms_per_step',{}); print('$1', round(b['ms_per_step'],2), 'tree', round(c.get('tree',0),2), 'tables', round(sum(v for k,v in t.items() if k.startswith('prove')),4))"; }
timeout 900 python -m pytest tests -m gpu -x -q -k "kat or commit or merkle or primitives or fri or plonk or segment_proof_matches_oracle" 2>&1 | tail -2
cp zk_evm_amd/libzkstark_hip.so /tmp/new.so
for rep in 1 2; do
  for V in base3 new; do
    if [ $V = base3 ]; then cp tools/scratch/libzkstark_hip_base3.so zk_evm_amd/libzkstark_hip.so; else cp /tmp/new.so zk_evm_amd/libzkstark_hip.so; fi
    python bench.py $QUICK --log-ns realistic 2>/dev/null | line "real $V"
    timeout 300 python tools/plonk_trace.py 13 20 2>/dev/null | tail -1
  done
done
for V in base3 new; do
  if [ $V = base3 ]; then cp tools/scratch/libzkstark_hip_base3.so zk_evm_amd/libzkstark_hip.so; else cp /tmp/new.so zk_evm_amd/libzkstark_hip.so; fi
  python bench.py $QUICK 2>/dev/null | line "2^20 $V"
done
cp /tmp/new.so zk_evm_amd/libzkstark_hip.so
