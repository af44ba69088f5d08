#!/bin/bash
# One parametrised gpurun payload: tools/run_gpu.sh <tag> <step> [<step> ...]   (steps run in order, from the repo root)
#   tests[:<pytest -k expr>]   GPU test suite (or a slice) -> gpurun_out/<tag>_gputests.log
#   bench[:<extra args>]       default bench line -> gpurun_out/<tag>_bench_default.json
#   quick[:<extra args>]       short bench line (no secondaries, no PMC) -> gpurun_out/<tag>_bench_quick.json
#   real                       quick line at the realistic heights -> gpurun_out/<tag>_bench_realistic.json
#   sh:<command>               any shell command, output appended to gpurun_out/<tag>_sh.log
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; cd "$ROOT"
QUICK="--steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary --no-dist-selftest"
show() { python - "$1" <<'PY'
import json, sys
try:
    b = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("contract line:", len(json.dumps(b)), "bytes; cpu_baseline", {k: v for k, v in (b.get("cpu_baseline") or {}).items() if k != "sample"}, "roofline.frac", b.get("roofline", {}).get("frac"))
    import os
    ex = sys.argv[1][:-5] + "_extra.json"
    if os.path.exists(ex):
        full = json.load(open(ex)); full.update({k: v for k, v in b.items() if k not in full}); b = full
    print("value", round(b["value"], 4), "ms/step", round(b["ms_per_step"], 2), "ntt", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in b.get("ntt", {}).items() if not isinstance(v, str)})
    print("timing", {k: round(v, 4) for k, v in b.get("segment_timing_s", {}).items()})
    print("stages", {k: round(v, 2) for k, v in b.get("commit_stages_ms_per_step", {}).items()}, "dist", b.get("dist"), "per_rank", b.get("per_rank_ms_per_step"))
    for k in ("cpu_baseline", "in_flight", "commit_config1", "h2d", "realistic", "block_replay", "from_logs", "plonk_recursion"):
        if k in b:
            v = b[k]
            print(" ", k, (v.get("error") or {kk: vv for kk, vv in v.items() if kk in ("value", "single", "in_flight", "commits_per_s", "seconds", "wall_s", "proofs_identical_to_direct", "block_s", "prove_ms")}) if isinstance(v, dict) else v)
    print("secondary_wall_s", b.get("secondary_wall_s"))
except Exception as e:
    print("bench parse failed:", e)
PY
}
for STEP in "$@"; do
  KIND=${STEP%%:*}; ARG=""; [ "$KIND" != "$STEP" ] && ARG=${STEP#*:}
  case $KIND in
    tests) if [ -n "$ARG" ]; then timeout 2400 python -m pytest tests -m gpu -x -q -k "$ARG" > "$OUT/${TAG}_gputests.log" 2>&1; else timeout 2400 python -m pytest tests -m gpu -x -q > "$OUT/${TAG}_gputests.log" 2>&1; fi
           echo "gpu tests rc=$?"; tail -6 "$OUT/${TAG}_gputests.log" ;;
    bench) timeout 1700 python bench.py $ARG > "$OUT/${TAG}_bench_default.json" 2> "$OUT/${TAG}_bench_default.err"; cp -f bench_extra.json "$OUT/${TAG}_bench_default_extra.json" 2>/dev/null; echo "bench rc=$?"; show "$OUT/${TAG}_bench_default.json"; tail -3 "$OUT/${TAG}_bench_default.err" ;;
    quick) timeout 900 python bench.py $QUICK $ARG > "$OUT/${TAG}_bench_quick.json" 2> "$OUT/${TAG}_bench_quick.err"; cp -f bench_extra.json "$OUT/${TAG}_bench_quick_extra.json" 2>/dev/null; echo "quick rc=$?"; show "$OUT/${TAG}_bench_quick.json" ;;
    real)  timeout 900 python bench.py $QUICK --log-ns realistic $ARG > "$OUT/${TAG}_bench_realistic.json" 2> "$OUT/${TAG}_bench_realistic.err"; cp -f bench_extra.json "$OUT/${TAG}_bench_realistic_extra.json" 2>/dev/null; echo "real rc=$?"; show "$OUT/${TAG}_bench_realistic.json" ;;
    sh)    echo "== $ARG" >> "$OUT/${TAG}_sh.log"; bash -c "$ARG" >> "$OUT/${TAG}_sh.log" 2>&1; echo "sh rc=$?"; tail -30 "$OUT/${TAG}_sh.log" ;;
    *) echo "unknown step $STEP" ;;
  esac
done
