import json, sys
b = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value", b["value"], "ms/step", b["ms_per_step"])
print("timing", {k: round(v, 4) for k, v in b.get("segment_timing_s", {}).items()})
kc = b.get("kernel_counters") or {}
for t, r in (kc.get("quotient_per_table") or {}).items():
    print("   %-14s air %7.2f ms  checks %7.2f ms  traffic/alg %6.2f  (reported/alg %6.2f)  %s" % (t, r["air_ms"], r["checks_ms"], r["traffic_over_algorithmic"], r["reported_over_algorithmic"], r["air_kernel"]))
print("quotient total ms", kc.get("quotient_ms_total"))
for k, v in kc.items():
    if isinstance(v, dict) and "launches" in v:
        print("  %-32s %4d launches %8.2f ms  traffic %s GB  cpi %s" % (k, v["launches"], v["ms"], round(v.get("traffic_bytes", 0) / 1e9, 2), v.get("cycles_per_wave_instruction")))
print("ntt", b.get("ntt"))
print("dist", b.get("dist"))
print("side", b.get("side_lane"))
