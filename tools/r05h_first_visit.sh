#!/bin/bash
# The first GPU visit after r05f-h (everything below was written with gpurun closed): what the helper process finds on the chip,
# the tests of the plan machinery, and the headline with the arbitrated plans next to the r04 plans -- in one box, ~12 minutes.
#   tools/r05h_first_visit.sh <tag>
TAG=${1:-r05h}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; cd "$ROOT"
echo "== swap instructions"; timeout 60 tools/probe_permlane 2>&1 | tee "$OUT/${TAG}_probe_permlane.log" | head -8
echo "== the helper, as the library runs it"
( time ZK_NTT_TUNE_INPROC=1 ZK_NTT_SWAP=2 timeout 300 zk_evm_amd/zk_ntt_tune 0 > "$OUT/${TAG}_helper_stdout.txt" 2> "$OUT/${TAG}_helper_stderr.txt"; echo "helper rc=$?" ) 2>&1 | tail -5
head -c 6000 "$OUT/${TAG}_helper_stdout.txt"; tail -5 "$OUT/${TAG}_helper_stderr.txt"
echo "== tests of the plan machinery + commitments + a segment"
timeout 1200 python -m pytest tests/test_gpu_tune.py tests/test_gpu_commit.py tests/test_gpu_segment.py -m gpu -x -q 2>&1 | tail -6 | tee "$OUT/${TAG}_tests_plans.log"
QUICK="--steps 5 --warmup 3 --no-cpu-baseline --no-pmc --no-secondary --commit-steps 0 --in-flight 1"
echo "== headline, arbitrated plans"; ZK_NTT_TUNE_VERBOSE=1 timeout 900 python bench.py $QUICK > "$OUT/${TAG}_bench_quick_auto.json" 2> "$OUT/${TAG}_bench_quick_auto.err"; tail -1 "$OUT/${TAG}_bench_quick_auto.json" | cut -c1-500; grep -c "lane-swap$" "$OUT/${TAG}_bench_quick_auto.err"
echo "== headline, r04 plans"; ZK_NTT_SWAP=0 ZK_TREE_BATCH=0 ZK_NTT_COL_BATCH_MB=0 timeout 900 python bench.py $QUICK 2>/dev/null > "$OUT/${TAG}_bench_quick_r04_plans.json"; tail -1 "$OUT/${TAG}_bench_quick_r04_plans.json" | cut -c1-500
echo "== block-shaped heights, arbitrated / r04"; timeout 900 python bench.py $QUICK --log-ns realistic 2>/dev/null > "$OUT/${TAG}_bench_quick_realistic_auto.json"; tail -1 "$OUT/${TAG}_bench_quick_realistic_auto.json" | cut -c1-400
ZK_NTT_SWAP=0 ZK_TREE_BATCH=0 ZK_NTT_COL_BATCH_MB=0 timeout 900 python bench.py $QUICK --log-ns realistic 2>/dev/null > "$OUT/${TAG}_bench_quick_realistic_r04_plans.json"; tail -1 "$OUT/${TAG}_bench_quick_realistic_r04_plans.json" | cut -c1-400
