# checks kernel: both entries of a pair with their loads together (centry_pair_fast)
cd $GRAFT_REPO_ROOT
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=b.get('segment_timing_s',{}); print('$1', round(b['ms_per_step'],2), 'tables', round(sum(v for k,v in t.items() if k.startswith('prove')),4), 'sponge', round(t.get('prove keccak_sponge_stark STARK',0),4), 'bytep', round(t.get('prove byte_packing_stark STARK',0),4))"; }
cp zk_evm_amd/libzkstark_hip.so /tmp/orig.so
cp tools/scratch/libzk_pf1.so zk_evm_amd/libzkstark_hip.so
timeout 900 python -m pytest tests -m gpu -x -q -k "stark_prove or segment_proof_matches_oracle or fuzz or stark_verify" 2>&1 | tail -2
for rep in 1 2 3; do
for V in orig pf0 pf1; do
  if [ $V = orig ]; then cp /tmp/orig.so zk_evm_amd/libzkstark_hip.so; else cp tools/scratch/libzk_$V.so zk_evm_amd/libzkstark_hip.so; fi
  python bench.py $QUICK 2>/dev/null | line "2^20 $V"
  python bench.py $QUICK --log-ns realistic 2>/dev/null | line "real $V"
done
done
cp tools/scratch/libzk_pf1.so zk_evm_amd/libzkstark_hip.so
timeout 300 python tools/soak_segment.py 20 2 2>/dev/null | tail -c 200
cp /tmp/orig.so zk_evm_amd/libzkstark_hip.so
