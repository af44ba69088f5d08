#!/bin/bash
# Phase isolation of ntt_strided_persist_kernel (tools/kbench_dbg, WRONG results by construction): kernel durations under
# rocprofv3 --kernel-trace with parts of the loop body switched off (ZK_NTT_NT bits: 2 no butterflies, 4 no stores, 8 no tile
# loads, 16 no twiddle loads, 32 no barriers).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; C=${1:-116}; L=${2:-20}
cd /tmp && export TMPDIR=/tmp
VARS="0 2 4 8 12 16 20 24 28 32 60 6 10"
for NT in $VARS; do
  rm -rf /tmp/pp_$NT; ZK_NTT_NT=$NT timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp_$NT -o t -- "$ROOT/tools/kbench_dbg" $C $L 4 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
names = {0: "complete", 2: "no butterflies", 4: "no stores", 8: "no tile loads", 12: "no tile loads/stores", 16: "no twiddle loads",
         20: "no stores, no twiddles", 24: "no tile loads, no twiddles", 28: "no memory at all", 32: "no barriers",
         60: "no memory, no barriers", 6: "no butterflies, no stores", 10: "no butterflies, no tile loads"}
res = defaultdict(dict)
for nt in names:
    acc = defaultdict(list)
    for p in glob.glob("/tmp/pp_%d/**/*kernel_trace.csv" % nt, recursive=True):
        for r in csv.DictReader(open(p)):
            if "ntt_" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0][:44], r["Workgroup_Size_X"], r["Grid_Size_X"] + "x" + r["Grid_Size_Y"])].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    for k, v in acc.items():
        v = sorted(v)[: max(1, len(v) * 3 // 4)]
        res[k][nt] = sum(v) / len(v)
for k in sorted(res):
    print(k)
    for n in names: print("    %-32s %8.0f us" % (names[n], res[k].get(n, float("nan"))))
PY
