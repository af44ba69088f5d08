# r03u visit: helper values loaded ahead in the checks kernel + double-buffered column loads in the FRI combination
cd $GRAFT_REPO_ROOT
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=b.get('segment_timing_s',{}); print('$1', round(b['ms_per_step'],2), 'ctl', round(t.get('compute CTL data',0),4), 'tables', round(sum(v for k,v in t.items() if k.startswith('prove')),4))"; }
timeout 900 python -m pytest tests -m gpu -x -q -k "fri or segment_proof_matches_oracle or stark_prove or plonk or stark_aux" 2>&1 | tail -1
cp zk_evm_amd/libzkstark_hip.so /tmp/new.so
for rep in 1 2 3; do
  for V in base4 new; do
    if [ $V = base4 ]; then cp tools/scratch/libzkstark_hip_base4.so zk_evm_amd/libzkstark_hip.so; else cp /tmp/new.so zk_evm_amd/libzkstark_hip.so; fi
    python bench.py $QUICK 2>/dev/null | line "2^20 $V"
    python bench.py $QUICK --log-ns realistic 2>/dev/null | line "real $V"
  done
done
cd /tmp && export TMPDIR=/tmp
for V in base4 new; do
if [ $V = base4 ]; then cp $GRAFT_REPO_ROOT/tools/scratch/libzkstark_hip_base4.so $GRAFT_REPO_ROOT/zk_evm_amd/libzkstark_hip.so; else cp /tmp/new.so $GRAFT_REPO_ROOT/zk_evm_amd/libzkstark_hip.so; fi
rm -rf /tmp/zkst && ZK_LANES=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zkst -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest > /dev/null 2>&1
F=$(find /tmp/zkst -name "*kernel_stats.csv" | head -1); echo "== $V (lanes off)"; grep -E "quotient_checks_kernel|fri_combine_kernel<3, 1>|helper_cols_kernel<2>|eval_columns_partial_kernel<2>" $F | cut -c1-140
done
cp /tmp/new.so $GRAFT_REPO_ROOT/zk_evm_amd/libzkstark_hip.so
