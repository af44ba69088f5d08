#!/bin/bash
# The first GPU visit after rounds 5 and 6 (both written with gpurun closed): hardware evidence for the second forms behind the
# plan table (lane-swap NTT kernels, column batches, batched tree tops), the whole GPU suite, the headline with the compiled-in
# (empty) table next to the table the offline tuner finds on this box, and the kernel traces of both -- in one box, ~35 minutes.
#   tools/r06_first_visit.sh <tag>          (through gpurun, from the repo root; results under gpurun_out/<tag>_*)
# What to do with the results: profiles/README.md "after the first visit".
TAG=${1:-r06a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; cd "$ROOT"
sha256sum zk_evm_amd/libzkstark_hip.so oracle/liboracle.so > "$OUT/${TAG}_library_sha256.txt"
echo "== swap instructions"; timeout 60 tools/probe_permlane 2>&1 | tee "$OUT/${TAG}_probe_permlane.log" | head -8
echo "== the offline tuner: both forms of every shape, compared word for word and timed"
( time timeout 600 zk_evm_amd/zk_ntt_tune 0 > "$OUT/${TAG}_plan_table_tuner_report.txt" 2> "$OUT/${TAG}_plan_table_tuner_stderr.txt"; echo "tuner rc=$? (0 = identical outputs everywhere, 7 = a second form DIFFERED)" ) 2>&1 | tail -5
head -c 7000 "$OUT/${TAG}_plan_table_tuner_report.txt"; tail -5 "$OUT/${TAG}_plan_table_tuner_stderr.txt"
PLANS=$(head -1 "$OUT/${TAG}_plan_table_tuner_report.txt")
echo "== the second forms against the oracle (tests/test_gpu_zz_plans.py)"
timeout 2400 python -m pytest tests/test_gpu_zz_plans.py -m gpu -q 2>&1 | tail -30 | tee "$OUT/${TAG}_tests_zz_plans.log"
echo "== the whole GPU suite"
timeout 3000 python -m pytest tests -m gpu -x -q > "$OUT/${TAG}_gputests.log" 2>&1; echo "gpu tests rc=$?"; tail -5 "$OUT/${TAG}_gputests.log"
QUICK="--steps 5 --warmup 3 --no-cpu-baseline --no-pmc --no-secondary --commit-steps 0 --in-flight 1"
line() { tail -1 "$1" | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('   value', b.get('value'), 'ms/step', b.get('ms_per_step'), 'ntt', b.get('ntt'))" 2>/dev/null || tail -c 300 "$1"; }
echo "== headline, compiled-in table"; timeout 900 python bench.py $QUICK > "$OUT/${TAG}_bench_quick_builtin.json" 2> "$OUT/${TAG}_bench_quick_builtin.err"; line "$OUT/${TAG}_bench_quick_builtin.json"
echo "== headline, the tuner's table: $PLANS"; ZK_NTT_SWAP_PLANS="$PLANS" timeout 900 python bench.py $QUICK > "$OUT/${TAG}_bench_quick_tuned.json" 2>/dev/null; line "$OUT/${TAG}_bench_quick_tuned.json"
for kv in "ZK_NTT_SWAP=1" "ZK_TREE_BATCH=1" "ZK_NTT_COL_BATCH_MB=96" "ZK_NTT_SWAP=0"; do
  echo "== headline, $kv"; env $kv timeout 900 python bench.py $QUICK > "$OUT/${TAG}_bench_quick_${kv%%=*}_${kv##*=}.json" 2>/dev/null; line "$OUT/${TAG}_bench_quick_${kv%%=*}_${kv##*=}.json"
done
echo "== block-shaped heights: compiled-in / tuned / tree tops batched"
timeout 900 python bench.py $QUICK --log-ns realistic > "$OUT/${TAG}_bench_quick_realistic_builtin.json" 2>/dev/null; line "$OUT/${TAG}_bench_quick_realistic_builtin.json"
ZK_NTT_SWAP_PLANS="$PLANS" timeout 900 python bench.py $QUICK --log-ns realistic > "$OUT/${TAG}_bench_quick_realistic_tuned.json" 2>/dev/null; line "$OUT/${TAG}_bench_quick_realistic_tuned.json"
ZK_TREE_BATCH=1 timeout 900 python bench.py $QUICK --log-ns realistic > "$OUT/${TAG}_bench_quick_realistic_tree_batch.json" 2>/dev/null; line "$OUT/${TAG}_bench_quick_realistic_tree_batch.json"
echo "== kernel traces: kbench 116 x 2^20, tile kernels / lane-swap kernels"
cd /tmp && export TMPDIR=/tmp
for S in 0 1; do
  rm -rf /tmp/kt_$S
  ZK_NTT_SWAP=$S timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$S -o k -- "$ROOT/tools/kbench" 116 20 3 > /dev/null 2>&1
  f=$(find /tmp/kt_$S -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && { cp "$f" "$OUT/${TAG}_kernel_stats_kbench_116x2p20_swap$S.csv"; echo "-- swap=$S"; grep -i "ntt\|Name" "$f" | cut -d, -f1-4 | cut -c1-150; }
done
rm -rf /tmp/kt_b; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_b -o k -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-secondary --commit-steps 0 --in-flight 1 > /dev/null 2>&1
f=$(find /tmp/kt_b -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && { cp "$f" "$OUT/${TAG}_kernel_stats_default_bench_3steps.csv"; head -12 "$f" | cut -d, -f1-4 | cut -c1-150; }
cd "$ROOT"
echo "== kbench A/B of every plan (checksums must agree per shape)"; timeout 1500 tools/ab_ntt_swap.sh 2>&1 | tee "$OUT/${TAG}_ab_ntt_swap.log" | cut -c1-230 | tail -40
echo "== the default bench line, full"; timeout 1200 python bench.py > "$OUT/${TAG}_bench_default.json" 2> "$OUT/${TAG}_bench_default.err"; tail -1 "$OUT/${TAG}_bench_default.json" | cut -c1-1500
echo "== the CPU oracle on this box's cores: one block-shaped segment, cached for cpu_baseline (copy tools/cpu_baseline_cache.json back)"
timeout 900 python tools/cpu_baseline_cache.py measure --log-ns realistic > "$OUT/${TAG}_cpu_baseline_segment.json" 2> /dev/null; cut -c1-300 "$OUT/${TAG}_cpu_baseline_segment.json"
cp tools/cpu_baseline_cache.json "$OUT/${TAG}_cpu_baseline_cache.json"
