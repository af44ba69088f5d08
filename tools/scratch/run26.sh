# checks kernel: entry loop unrolled by 2 / 4 (compiler-interleaved load chains, as the unrolled helper batch got)
cd $GRAFT_REPO_ROOT
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=b.get('segment_timing_s',{}); print('$1', round(b['ms_per_step'],2), 'tables', round(sum(v for k,v in t.items() if k.startswith('prove')),4), 'sponge', round(t.get('prove keccak_sponge_stark STARK',0),4))"; }
cp zk_evm_amd/libzkstark_hip.so /tmp/orig.so
for rep in 1 2; do
for V in cu1 cu2 cu4; do
  cp tools/scratch/libzk_$V.so zk_evm_amd/libzkstark_hip.so
  python bench.py $QUICK 2>/dev/null | line "2^20 $V"
  python bench.py $QUICK --log-ns realistic 2>/dev/null | line "real $V"
done
done
for V in cu2 cu4; do
  cp tools/scratch/libzk_$V.so zk_evm_amd/libzkstark_hip.so
  timeout 600 python -m pytest tests -m gpu -x -q -k "stark_prove or segment_proof_matches_oracle" 2>&1 | tail -1
done
cp /tmp/orig.so zk_evm_amd/libzkstark_hip.so
