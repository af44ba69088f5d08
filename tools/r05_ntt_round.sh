#!/bin/bash
# One GPU-box visit for the r05 NTT work: bit-exactness and speed of the lane-swap kernels (csrc/ntt_swap.cuh) and of the column
# batches against the tile kernels, per-kernel durations under rocprofv3, the parity test.  Through gpurun, from the repo root:
#   tools/r05_ntt_round.sh <tag>
TAG=${1:-r05g}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; cd "$ROOT"
echo "== the swap instructions"; timeout 60 tools/probe_permlane 2>&1 | tee "$OUT/${TAG}_probe_permlane.log" | head -70
echo "== swap A/B"; timeout 1500 tools/ab_ntt_swap.sh 2>&1 | tee "$OUT/${TAG}_ab_ntt_swap.log" | cut -c1-230
echo "== column batches"; timeout 600 tools/ab_ntt_col_batch.sh 2>&1 | tee "$OUT/${TAG}_ab_ntt_col_batch.log" | cut -c1-230
echo "== per-kernel durations (kbench 116 x 2^20, 4 commitments each)"
cd /tmp && export TMPDIR=/tmp
for S in 0 1; do
  rm -rf /tmp/kt_$S
  ZK_NTT_SWAP=$S timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$S -o k -- "$ROOT/tools/kbench" 116 20 3 > /dev/null 2>&1
  f=$(find /tmp/kt_$S -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && { cp "$f" "$OUT/${TAG}_kernel_stats_kbench_116x2p20_swap$S.csv"; echo "-- swap=$S"; grep -i "ntt\|Name" "$f" | cut -d, -f1-4 | cut -c1-150; }
done
cd "$ROOT"
echo "== parity test"; ZK_TEST_UNVALIDATED_PLANS=1 timeout 1500 python -m pytest tests/test_gpu_commit.py -m gpu -x -q -k "lane_swap or fused or persistent" 2>&1 | tail -5 | tee "$OUT/${TAG}_swap_parity_test.log"
echo "== the headline with the plans arbitrated on the box (default) and with the r04 plans forced"
QUICK="--steps 5 --warmup 3 --no-cpu-baseline --no-pmc --no-secondary --commit-steps 0 --in-flight 1"
ZK_NTT_TUNE_VERBOSE=1 timeout 900 python bench.py $QUICK > "$OUT/${TAG}_bench_quick_auto.json" 2> "$OUT/${TAG}_bench_quick_auto.err"; grep "^ntt " "$OUT/${TAG}_bench_quick_auto.err" | cut -c1-220; tail -1 "$OUT/${TAG}_bench_quick_auto.json" | cut -c1-400
ZK_NTT_SWAP=0 ZK_NTT_COL_BATCH_MB=0 timeout 900 python bench.py $QUICK > "$OUT/${TAG}_bench_quick_r04_plans.json" 2> /dev/null; tail -1 "$OUT/${TAG}_bench_quick_r04_plans.json" | cut -c1-400
ZK_NTT_TUNE_VERBOSE=1 timeout 900 python bench.py $QUICK --log-ns realistic > "$OUT/${TAG}_bench_quick_realistic_auto.json" 2> "$OUT/${TAG}_bench_quick_realistic_auto.err"; grep "^ntt " "$OUT/${TAG}_bench_quick_realistic_auto.err" | cut -c1-220; tail -1 "$OUT/${TAG}_bench_quick_realistic_auto.json" | cut -c1-300
ZK_NTT_SWAP=0 ZK_NTT_COL_BATCH_MB=0 timeout 900 python bench.py $QUICK --log-ns realistic > "$OUT/${TAG}_bench_quick_realistic_r04_plans.json" 2> /dev/null; tail -1 "$OUT/${TAG}_bench_quick_realistic_r04_plans.json" | cut -c1-300
ZK_TREE_BATCH=1 timeout 900 python bench.py $QUICK --log-ns realistic > "$OUT/${TAG}_bench_quick_realistic_tree_batch.json" 2> /dev/null; tail -1 "$OUT/${TAG}_bench_quick_realistic_tree_batch.json" | cut -c1-300
