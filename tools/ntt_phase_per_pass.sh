#!/bin/bash
# Per-PASS phase isolation: kernel durations (rocprofv3 --kernel-trace) of tools/kbench_dbg with the butterflies skipped (NT=2),
# with loads and stores skipped (NT=12) and complete (NT=0), grouped by (kernel, workgroup size) = pass.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; C=${1:-116}; L=${2:-20}
cd /tmp && export TMPDIR=/tmp
for NT in 0 2 12; do
  rm -rf /tmp/ph_$NT; ZK_NTT_FUSE=0 ZK_NTT_NT=$NT timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ph_$NT -o t -- "$ROOT/tools/kbench_dbg" $C $L 4 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
names = {0: "complete", 2: "memory phases only", 12: "butterflies only"}
res = defaultdict(dict)
for nt in (0, 2, 12):
    acc = defaultdict(list)
    for p in glob.glob("/tmp/ph_%d/**/*kernel_trace.csv" % nt, recursive=True):
        for r in csv.DictReader(open(p)):
            if "ntt_pass_kernel" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0][:40], r["Workgroup_Size_X"], r["Grid_Size_X"] + "x" + r["Grid_Size_Y"])].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    for k, v in acc.items():
        v = sorted(v)[: max(1, len(v) * 3 // 4)]            # drop the slow first repetitions
        res[k][nt] = sum(v) / len(v)
print("pass (kernel, workgroup, grid)".ljust(70), "".join(names[n].rjust(22) for n in (0, 2, 12)), "   (us)")
for k in sorted(res):
    print(str(k).ljust(70), "".join(("%.0f" % res[k].get(n, float("nan"))).rjust(22) for n in (0, 2, 12)))
PY
