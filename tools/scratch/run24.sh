cd $GRAFT_REPO_ROOT
cp zk_evm_amd/libzkstark_hip.so /tmp/new.so
for rep in 1 2 3; do
  for V in base5 new; do
    if [ $V = base5 ]; then cp tools/scratch/libzkstark_hip_base5.so zk_evm_amd/libzkstark_hip.so; else cp /tmp/new.so zk_evm_amd/libzkstark_hip.so; fi
    echo -n "$V x4: "; timeout 300 python tools/plonk_trace.py 13 40 4 2>/dev/null | tail -1
    echo -n "$V x8: "; timeout 300 python tools/plonk_trace.py 13 40 8 2>/dev/null | tail -1
  done
done
cp /tmp/new.so zk_evm_amd/libzkstark_hip.so
python - <<'PY'
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-pmc", "--commit-steps", "0", "--in-flight", "1", "--no-dist-selftest"], capture_output=True, text=True).stdout
b = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
pr = b.get("plonk_recursion", {})
print("new bench plonk:", {k: round(v["ms_per_proof"], 3) for k, v in pr.get("sizes", {}).items()}, pr.get("in_flight_2^13", {}).get("proofs_per_s"), pr.get("batch_2^13", {}).get("proofs_per_s"))
print("recursion:", b.get("realistic", {}).get("segment_with_recursion"))
PY
