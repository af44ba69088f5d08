# r03t visit 1: (a) parity subset on the new library, (b) openings row chunk 2^9 / 2^10 / 2^11, (c) base vs new library
cd $GRAFT_REPO_ROOT
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=b.get('segment_timing_s',{}); print('$1', round(b['ms_per_step'],2), 'tables', round(sum(v for k,v in t.items() if k.startswith('prove')),4))"; }
timeout 1200 python -m pytest tests -m gpu -x -q -k "fri or segment_proof_matches_oracle or stark_prove or plonk or full_size" 2>&1 | tail -3
for R in 9 10 11; do
  ZK_EVAL_ROWS_LOG=$R timeout 600 python -m pytest tests -m gpu -x -q -k "fri or segment_proof_matches_oracle" 2>&1 | tail -1
done
cp zk_evm_amd/libzkstark_hip.so /tmp/new.so
for rep in 1 2; do
  for R in 9 10 11; do
    ZK_EVAL_ROWS_LOG=$R python bench.py $QUICK 2>/dev/null | line "2^20 new rows_log=$R"
  done
  cp tools/scratch/libzkstark_hip_base.so zk_evm_amd/libzkstark_hip.so
  python bench.py $QUICK 2>/dev/null | line "2^20 BASE"
  cp /tmp/new.so zk_evm_amd/libzkstark_hip.so
done
for R in 9 10 11; do
  ZK_EVAL_ROWS_LOG=$R python bench.py $QUICK --log-ns realistic 2>/dev/null | line "real new rows_log=$R"
done
cp tools/scratch/libzkstark_hip_base.so zk_evm_amd/libzkstark_hip.so
python bench.py $QUICK --log-ns realistic 2>/dev/null | line "real BASE"
cp /tmp/new.so zk_evm_amd/libzkstark_hip.so
# kernel times of the two kernels touched
cd /tmp && export TMPDIR=/tmp
for R in 9 11; do
rm -rf /tmp/zkst && ZK_EVAL_ROWS_LOG=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zkst -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest > /dev/null 2>&1
F=$(find /tmp/zkst -name "*kernel_stats.csv" | head -1); echo "rows_log=$R"; grep -E "eval_columns|fri_combine" $F | cut -c1-200
done
