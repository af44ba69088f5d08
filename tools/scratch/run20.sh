# r03u visit: helper batch 16 / 8 / 4 (singles stay 16); NTT LDS padding with the free addressing (kbench + bench env)
cd $GRAFT_REPO_ROOT
K=tools/scratch/kb/kbench_pad
for rep in 1 2; do
for P in 0 1; do echo -n "ZK_NTT_PAD=$P 116x2^20 | "; ZK_NTT_PAD=$P $K 116 20 5 | head -1; done
done
for P in 0 1; do echo -n "ZK_NTT_PAD=$P 2431x2^17 | "; ZK_NTT_PAD=$P $K 2431 17 3 | head -1; done
for P in 0 1; do echo -n "ZK_NTT_PAD=$P 30x2^21 | "; ZK_NTT_PAD=$P $K 30 21 5 | head -1; done
for P in 0 1; do ZK_NTT_PAD=$P $K 116 20 1 | tail -1; done
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=b.get('segment_timing_s',{}); c=b.get('commit_stages_ms_per_step',{}); print('$1', round(b['ms_per_step'],2), 'ifft', round(c.get('ifft',0),1), 'lde', round(c.get('lde',0),1), 'ctl', round(t.get('compute CTL data',0),4), 'tables', round(sum(v for k,v in t.items() if k.startswith('prove')),4))"; }
cp zk_evm_amd/libzkstark_hip.so /tmp/orig.so
for rep in 1 2; do
for V in nb16 nb8 nb4; do
  cp tools/scratch/libzk_$V.so zk_evm_amd/libzkstark_hip.so
  python bench.py $QUICK 2>/dev/null | line "2^20 $V"
  python bench.py $QUICK --log-ns realistic 2>/dev/null | line "real $V"
done
cp tools/scratch/libzk_nb8.so zk_evm_amd/libzkstark_hip.so
ZK_NTT_PAD=1 python bench.py $QUICK 2>/dev/null | line "2^20 nb8 PAD=1"
ZK_NTT_PAD=1 python bench.py $QUICK --log-ns realistic 2>/dev/null | line "real nb8 PAD=1"
done
cp tools/scratch/libzk_nb4.so zk_evm_amd/libzkstark_hip.so
timeout 600 python -m pytest tests -m gpu -x -q -k "stark_aux or segment_proof_matches_oracle or stark_prove" 2>&1 | tail -1
cp tools/scratch/libzk_nb8.so zk_evm_amd/libzkstark_hip.so
ZK_NTT_PAD=1 timeout 600 python -m pytest tests -m gpu -x -q -k "ntt or commit or segment_proof_matches_oracle or primitives" 2>&1 | tail -1
cp /tmp/orig.so zk_evm_amd/libzkstark_hip.so
