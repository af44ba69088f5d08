# r03t visit 6: settled code (twiddle buffer loads + LDS adds, point tables from two small ones, tail stream): whole suite, base (r03s) vs new
cd $GRAFT_REPO_ROOT
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=b.get('segment_timing_s',{}); c=b.get('commit_stages_ms_per_step',{}); print('$1', round(b['ms_per_step'],2), 'ifft', round(c.get('ifft',0),1), 'lde', round(c.get('lde',0),1), 'tree', round(c.get('tree',0),1), 'commits', round(t.get('compute all trace commitments',0),4), 'tables', round(sum(v for k,v in t.items() if k.startswith('prove')),4))"; }
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cp zk_evm_amd/libzkstark_hip.so /tmp/new.so
for rep in 1 2 3; do
  for V in base new; do
    if [ $V = base ]; then cp tools/scratch/libzkstark_hip_base.so zk_evm_amd/libzkstark_hip.so; else cp /tmp/new.so zk_evm_amd/libzkstark_hip.so; fi
    python bench.py $QUICK 2>/dev/null | line "2^20 $V"
    python bench.py $QUICK --log-ns realistic 2>/dev/null | line "real $V"
  done
done
cp /tmp/new.so zk_evm_amd/libzkstark_hip.so
timeout 300 python tools/soak_segment.py 20 3 2>/dev/null | tail -c 300
