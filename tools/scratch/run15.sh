# r03t visit 4: NTT with buffer accesses / one-add tile addresses: parity, then base2 (old ntt.cuh) vs new
cd $GRAFT_REPO_ROOT
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=b.get('segment_timing_s',{}); n=b.get('ntt',{}); c=b.get('commit_stages_ms_per_step',{}); print('$1', round(b['ms_per_step'],2), 'ntt GB/s', round(n.get('achieved_GBs',0),1), 'ifft', round(c.get('ifft',0),1), 'lde', round(c.get('lde',0),1), 'commits', round(t.get('compute all trace commitments',0),4))"; }
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cp zk_evm_amd/libzkstark_hip.so /tmp/new.so
for rep in 1 2 3; do
  for V in base2 new; do
    if [ $V = base2 ]; then cp tools/scratch/libzkstark_hip_base2.so zk_evm_amd/libzkstark_hip.so; else cp /tmp/new.so zk_evm_amd/libzkstark_hip.so; fi
    python bench.py $QUICK 2>/dev/null | line "2^20 $V"
    python bench.py $QUICK --log-ns realistic 2>/dev/null | line "real $V"
  done
done
cp /tmp/new.so zk_evm_amd/libzkstark_hip.so
python bench.py --workload commit --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('commit new', b['ms_per_step'], b.get('stages_ms'), b.get('ntt'))"
cp tools/scratch/libzkstark_hip_base2.so zk_evm_amd/libzkstark_hip.so
python bench.py --workload commit --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('commit base2', b['ms_per_step'], b.get('stages_ms'), b.get('ntt'))"
cp /tmp/new.so zk_evm_amd/libzkstark_hip.so
timeout 300 python tools/soak_segment.py 20 3 2>/dev/null | tail -c 300
