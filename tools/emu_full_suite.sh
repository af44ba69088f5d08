#!/bin/bash
# Every GPU test against the CPU emulation build (tests/emu/), one process per test under a hard time limit, in order
# and again with deferred streams (HIPEMU_ASYNC=1) -- the record behind tests/emu/quick_slice.txt.  ~1.5 h on 8 cores.
#   tools/emu_full_suite.sh [tag=r06b] [limit seconds=600]      -> profiles/<tag>_emu_gpu_suite.log, profiles/<tag>_emu_gpu_suite_async.log
TAG=${1:-r06b}; LIMIT=${2:-600}
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
python tests/emu/build_emu.py > /dev/null || exit 1
summ() {   # json, title
python - "$1" "$2" <<'PY'
import collections, json, subprocess, sys
d = json.load(open(sys.argv[1]))
c = collections.Counter(v["status"] for v in d.values())
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
print("# %s -- commit %s; NOT a hardware measurement: the library's kernels and host code compiled for the CPU (tests/emu/)" % (sys.argv[2], head))
print("# %d tests: %s" % (len(d), ", ".join("%d %s" % (n, s) for s, n in sorted(c.items()))))
print("# `failed` / `timeout` below are explained in profiles/README.md (tests that need the real library or a real GPU's size)")
for k, v in d.items():
    print("%-8s %7.1f s  %s" % (v["status"], v["s"], k))
    if v["status"] == "failed":
        tail = [ln for ln in v["tail"].splitlines() if ln.startswith("E ")]
        print("         " + (tail[-1][:300] if tail else v["tail"][-300:].replace("\n", " ")))
PY
}
python tools/emu_survey.py --limit "$LIMIT" --out /tmp/${TAG}_sync.json > /tmp/${TAG}_sync.log 2>&1
# second pass: what timed out or failed while two tests shared the machine (the multi-rank tests start 2-8 processes each), alone
python - <<PY
import json
d = json.load(open("/tmp/${TAG}_sync.json"))
open("/tmp/${TAG}_retry_ids.txt", "w").write("\n".join(k for k, v in d.items() if v["status"] != "passed") + "\n")
PY
python tools/emu_survey.py --limit $((LIMIT * 3)) --workers 1 --only /tmp/${TAG}_retry_ids.txt --out /tmp/${TAG}_retry.json > /tmp/${TAG}_retry.log 2>&1
python - <<PY
import json
d, r = json.load(open("/tmp/${TAG}_sync.json")), json.load(open("/tmp/${TAG}_retry.json"))
for k, v in r.items():
    if v["status"] == "passed" or d[k]["status"] == "timeout":
        d[k] = dict(v, second_pass=True)
json.dump(d, open("/tmp/${TAG}_sync.json", "w"), indent=0)
PY
summ /tmp/${TAG}_sync.json "GPU tests on the CPU emulation build, streams in order" > profiles/${TAG}_emu_gpu_suite.log
python - <<PY
import json
d = json.load(open("/tmp/${TAG}_sync.json"))
open("/tmp/${TAG}_async_ids.txt", "w").write("\n".join(k for k, v in d.items() if v["status"] == "passed" and v["s"] <= 200) + "\n")
PY
python tools/emu_survey.py --limit "$LIMIT" --async --only /tmp/${TAG}_async_ids.txt --out /tmp/${TAG}_async.json > /tmp/${TAG}_async.log 2>&1
summ /tmp/${TAG}_async.json "GPU tests on the CPU emulation build, DEFERRED streams (HIPEMU_ASYNC=1: nothing runs until the host waits for it)" > profiles/${TAG}_emu_gpu_suite_async.log
head -3 profiles/${TAG}_emu_gpu_suite.log profiles/${TAG}_emu_gpu_suite_async.log
