# batch sizes once more: helper 4 / 2, singles 16 / 32 / 8
cd $GRAFT_REPO_ROOT
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=b.get('segment_timing_s',{}); print('$1', round(b['ms_per_step'],2), 'ctl', round(t.get('compute CTL data',0),4), 'tables', round(sum(v for k,v in t.items() if k.startswith('prove')),4), 'arith', round(t.get('prove arithmetic_stark STARK',0),4))"; }
cp zk_evm_amd/libzkstark_hip.so /tmp/orig.so
for rep in 1 2 3; do
for V in hb4 hb2 sb32 sb8; do
  cp tools/scratch/libzk_$V.so zk_evm_amd/libzkstark_hip.so
  python bench.py $QUICK 2>/dev/null | line "2^20 $V"
  python bench.py $QUICK --log-ns realistic 2>/dev/null | line "real $V"
done
done
for V in hb2 sb8; do
  cp tools/scratch/libzk_$V.so zk_evm_amd/libzkstark_hip.so
  timeout 600 python -m pytest tests -m gpu -x -q -k "stark_aux or segment_proof_matches_oracle or stark_prove" 2>&1 | tail -1
done
cp /tmp/orig.so zk_evm_amd/libzkstark_hip.so
