cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03x_gputests.log 2>&1; tail -2 gpurun_out/r03x_gputests.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 300 python tools/soak_segment.py 20 3 2>/dev/null | tail -c 260; echo
timeout 900 python bench.py > gpurun_out/r03x_bench_default.json 2> gpurun_out/r03x_bench_default.err; python - <<'PY'
import json
b = json.loads([l for l in open("gpurun_out/r03x_bench_default.json") if l.startswith("{")][-1])
print("value", b["value"], "ms", b["ms_per_step"], "roofline.frac", b["roofline"]["frac"], "realistic", b["realistic"]["single"], "dist", b["dist"]["ok"])
PY
timeout 200 python -m tests.fuzz_parity 45 4242 2>/dev/null | tail -c 300
