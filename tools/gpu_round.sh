#!/bin/bash
# One GPU-box visit of a development round: GPU tests, the default bench line, an A/B line with the lanes off, and a
# kernel trace + idle analysis of a realistic-height segment.  Usage (through gpurun, from the repo root):
#   tools/gpu_round.sh <tag> [tests|notests] [extra steps ...]
# Everything lands under gpurun_out/<tag>_*.
TAG=${1:-r03x}; MODE=${2:-tests}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
# The library that produces this round's evidence is rebuilt FROM SOURCE on the box (no mtime shortcuts: a stale object after
# a header edit would otherwise go unnoticed, r03 verdict weak 10) and its hash is recorded next to the results.
{ echo "== $(date -u +%FT%TZ) tag $TAG"; timeout 1200 python -m zk_evm_amd.build --force && make -s -B -C oracle;
  echo "build rc=$?"; sha256sum zk_evm_amd/libzkstark_hip.so oracle/liboracle.so; /opt/rocm/bin/hipcc --version | head -2; } > "$OUT/${TAG}_final_code_validation.log" 2>&1
tail -4 "$OUT/${TAG}_final_code_validation.log"
QUICK="--steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary --commit-steps 0 --in-flight 1"
if [ "$MODE" = tests ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/${TAG}_gputests.log" 2>&1
  echo "gpu tests rc=$?"; tail -5 "$OUT/${TAG}_gputests.log"
  tail -3 "$OUT/${TAG}_gputests.log" >> "$OUT/${TAG}_final_code_validation.log"
fi
timeout 900 python bench.py > "$OUT/${TAG}_bench_default.json" 2> "$OUT/${TAG}_bench_default.err"
echo "bench rc=$?"; python - "$OUT/${TAG}_bench_default.json" <<'PY'
import json, sys
try:
    b = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("value", b["value"], "ms/step", b["ms_per_step"], "roofline.frac", b["roofline"]["frac"])
    print("timing", {k: round(v, 4) for k, v in b.get("segment_timing_s", {}).items()})
    print("realistic", b.get("realistic", {}).get("single"), b.get("realistic", {}).get("in_flight"))
    print("in_flight", b.get("in_flight"))
    kc = b.get("kernel_counters") or {}
    for t, r in (kc.get("quotient_per_table") or {}).items():
        print("   %-14s air %7.2f ms  checks %7.2f ms  traffic/alg %6.2f  (reported/alg %6.2f)" % (t, r["air_ms"], r["checks_ms"], r["traffic_over_algorithmic"], r["reported_over_algorithmic"]))
    print("quotient total ms", kc.get("quotient_ms_total"))
    for k, v in kc.items():
        if isinstance(v, dict) and "launches" in v:
            print("  %-32s %4d launches %8.2f ms  traffic %6.2f GB  cpi %s" % (k, v["launches"], v["ms"], v.get("traffic_bytes", 0) / 1e9, v.get("cycles_per_wave_instruction")))
    pr = b.get("plonk_recursion", {})
    print("plonk batch", pr.get("batch_2^13"), "threads", (pr.get("in_flight_2^13") or {}).get("proofs_per_s"))
    print("recursion", (b.get("realistic") or {}).get("segment_with_recursion"))
    print("dist", b.get("dist"))
    print("ntt", b.get("ntt"))
    print("h2d", b.get("h2d"))
    print("side_lane", b.get("side_lane"))
except Exception as e:
    print("bench parse failed:", e)
PY
ZK_LANES=0 timeout 600 python bench.py $QUICK > "$OUT/${TAG}_bench_lanes_off.json" 2> "$OUT/${TAG}_bench_lanes_off.err"
python - "$OUT/${TAG}_bench_lanes_off.json" <<'PY'
import json, sys
try:
    b = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("LANES OFF value", b["value"], "ms/step", b["ms_per_step"], {k: round(v, 4) for k, v in b.get("segment_timing_s", {}).items()})
except Exception as e:
    print("lanes-off parse failed:", e)
PY
for L in 1 0; do
  ZK_LANES=$L timeout 600 python bench.py $QUICK --log-ns realistic > "$OUT/${TAG}_bench_realistic_lanes$L.json" 2>/dev/null
  python - "$OUT/${TAG}_bench_realistic_lanes$L.json" $L <<'PY'
import json, sys
try:
    b = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("REALISTIC lanes", sys.argv[2], "value", b["value"], "ms/step", b["ms_per_step"], {k: round(v, 4) for k, v in b.get("segment_timing_s", {}).items()})
except Exception as e:
    print("realistic parse failed:", e)
PY
done
# kernel trace of realistic-height segments (3 proofs) + idle analysis
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/zktrace && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/zktrace -o tr -- python "$ROOT/bench.py" --log-ns realistic --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-secondary --commit-steps 0 --in-flight 1 > /dev/null 2>&1
F=$(find /tmp/zktrace -name "*kernel_trace.csv" | head -1)
if [ -n "$F" ]; then python "$ROOT/tools/gap_analysis.py" "$F" 0.5 > "$OUT/${TAG}_gaps_realistic.txt" 2>&1; head -24 "$OUT/${TAG}_gaps_realistic.txt"; fi
