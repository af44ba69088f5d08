#!/bin/bash
# The committed evidence of a round, from the final code (run through gpurun from the repo root):
#   profiles/<tag>_kernel_stats_default_bench_3steps.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 3`
#   profiles/<tag>_pmc_segment_per_kernel.csv              mean counters per dispatch and kernel (tools/collect_pmc.sh passes)
#   profiles/<tag>_soak_segment_2p20_x10.json              ten identical 2^20 segment proofs: one SHA-256
#   profiles/<tag>_fuzz_parity_{per_table,segment,plonk}.json   randomised differential parity on the final kernels
# Usage: tools/profile_round.sh <tag> [fuzz seconds per mode]
TAG=${1:-r03z}; FUZZ=${2:-60}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
FAST="--no-cpu-baseline --no-pmc --no-secondary --commit-steps 0 --in-flight 1 --no-dist-selftest"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/zkstats && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zkstats -o st -- python "$ROOT/bench.py" --steps 3 --warmup 1 $FAST > "$OUT/${TAG}_bench_line_under_rocprof.json" 2> /dev/null
F=$(find /tmp/zkstats -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" "$OUT/${TAG}_kernel_stats_default_bench_3steps.csv" && head -8 "$F" | cut -c1-160
# the `north_star` shape (per-table heights of scripts/prove_stdio.rs:89-101): its own kernel table, one segment at a time
rm -rf /tmp/zkstats_r && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zkstats_r -o st -- python "$ROOT/bench.py" --log-ns realistic --steps 5 --warmup 2 $FAST > "$OUT/${TAG}_bench_line_realistic_under_rocprof.json" 2> /dev/null
F=$(find /tmp/zkstats_r -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" "$OUT/${TAG}_kernel_stats_realistic_5steps.csv" && head -8 "$F" | cut -c1-160
T=$(find /tmp/zkstats_r -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python "$ROOT/tools/gap_analysis.py" "$T" 0.4 > "$OUT/${TAG}_gaps_realistic.txt" 2>&1
cd "$ROOT"
timeout 1500 tools/collect_pmc.sh "$TAG" --commit-steps 0 --in-flight 1 --no-secondary --no-dist-selftest > /dev/null 2>&1
python tools/pmc_summary.py "$OUT/pmc_$TAG" > "$OUT/${TAG}_pmc_segment_per_kernel.csv" 2> /dev/null; head -4 "$OUT/${TAG}_pmc_segment_per_kernel.csv" | cut -c1-200
timeout 600 python tools/soak_segment.py 20 10 > "$OUT/${TAG}_soak_segment_2p20_x10.json" 2> "$OUT/${TAG}_soak.err"; tail -c 600 "$OUT/${TAG}_soak_segment_2p20_x10.json"
for mode in per_table segment plonk; do
  arg=""; [ $mode != per_table ] && arg=$mode
  timeout $((FUZZ + 240)) python -m tests.fuzz_parity $FUZZ 31337 $arg > "$OUT/${TAG}_fuzz_parity_$mode.json" 2> "$OUT/${TAG}_fuzz_$mode.err"; tail -c 400 "$OUT/${TAG}_fuzz_parity_$mode.json"; echo
done
