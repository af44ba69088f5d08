# final code of the round: whole GPU suite + smoke + the driver's bench command
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03w_gputests.log 2>&1; tail -2 gpurun_out/r03w_gputests.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03w_bench_driver_cmd.json 2> gpurun_out/r03w_bench_driver_cmd.err; python - <<'PY'
import json
b = json.loads([l for l in open("gpurun_out/r03w_bench_driver_cmd.json") if l.startswith("{")][-1])
print("value", b["value"], "ms", b["ms_per_step"], "steps", b["steps"], "roofline.frac", b["roofline"]["frac"], "cpu_baseline", b["cpu_baseline"]["value"], "dist", b["dist"]["ok"])
print("realistic", b["realistic"]["single"], "plonk", {k: round(v["ms_per_proof"], 3) for k, v in b["plonk_recursion"]["sizes"].items()}, b["plonk_recursion"]["batch_2^13"]["proofs_per_s"])
PY
