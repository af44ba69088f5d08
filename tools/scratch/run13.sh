# r03t visit 2: whole GPU suite on the new library (new dot_acc_reduce / gl_inv / openings chunks), then base vs new
cd $GRAFT_REPO_ROOT
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=b.get('segment_timing_s',{}); print('$1', round(b['ms_per_step'],2), 'tables', round(sum(v for k,v in t.items() if k.startswith('prove')),4), 'ctl', round(t.get('compute CTL data',0),4))"; }
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
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/zkst && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zkst -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest > /dev/null 2>&1
F=$(find /tmp/zkst -name "*kernel_stats.csv" | head -1); cp $F $GRAFT_REPO_ROOT/gpurun_out/r03t_kernel_stats_3seg.csv; head -40 $F | cut -c1-150
