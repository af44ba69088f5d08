# occupancy by launch bounds: checks kernel 5 waves (96 VGPRs + 140 B scratch), FRI combination 8 waves (64 VGPRs + 60 B scratch)
cd $GRAFT_REPO_ROOT
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=b.get('segment_timing_s',{}); print('$1', round(b['ms_per_step'],2), 'tables', round(sum(v for k,v in t.items() if k.startswith('prove')),4))"; }
cp zk_evm_amd/libzkstark_hip.so /tmp/orig.so
for rep in 1 2 3; do
for V in d0 cw5 fw8; do
  cp tools/scratch/libzk_$V.so zk_evm_amd/libzkstark_hip.so
  python bench.py $QUICK 2>/dev/null | line "2^20 $V"
  python bench.py $QUICK --log-ns realistic 2>/dev/null | line "real $V"
done
done
cp /tmp/orig.so zk_evm_amd/libzkstark_hip.so
