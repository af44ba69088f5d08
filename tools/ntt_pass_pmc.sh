#!/bin/bash
# Per-PASS counters of the commitment's NTT (r03 verdict, next-round item 3a: say WHAT the strided pass waits for).
# tools/kbench runs one 116 x 2^20 commitment per repetition: strided values->coefficients pass, fused pass, strided
# coefficients->values pass (ZK_NTT_FUSE=0: the four separate passes).  Each counter group is its own rocprofv3 --pmc run
# (counters only + kernel trace); rows are averaged per (kernel name, grid) = per pass.
# Usage (through gpurun, from the repo root): tools/ntt_pass_pmc.sh <tag> [cols=116] [log_n=20]
TAG=${1:-r04}; C=${2:-116}; L=${3:-20}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_ntt_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; fuse=$2; shift 2
  ZK_NTT_FUSE=$fuse timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/${name}_fuse$fuse" -o "$name" -- "$ROOT/tools/kbench" $C $L 3 > "$OUT/${name}_fuse$fuse.log" 2>&1; }
for F in 1 0; do
  run sq1 $F SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
  run sq2 $F SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_SALU
  run sq3 $F SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES_EQ_64 SQ_INSTS_FLAT GRBM_GUI_ACTIVE
  run tcp $F TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
  run tcc $F TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum
  run fetch $F FETCH_SIZE GRBM_GUI_ACTIVE
  run write $F WRITE_SIZE
done
python3 - "$OUT" <<'PY' > "$ROOT/gpurun_out/${TAG}_ntt_per_pass_pmc.csv"
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
rows = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for d in sorted(glob.glob(os.path.join(root, "*_fuse[01]"))):
    fuse = d[-1]
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            name = r.get("Kernel_Name", "?")
            if "ntt_" not in name:
                continue
            key = (fuse, name.split("(")[0][:48], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")))
            rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if d.endswith("fetch_fuse" + fuse):
        for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                name = r.get("Kernel_Name", "?")
                if "ntt_" in name:
                    key = (fuse, name.split("(")[0][:48], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")))
                    dur[key].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
counters = sorted({c for k in rows for c in rows[k]})
w = csv.writer(sys.stdout)
w.writerow(["fused_plan", "kernel", "grid", "workgroup", "lds", "dispatches", "us_under_pmc"] + counters)
for k in sorted(rows):
    n = max(len(v) for v in rows[k].values())
    w.writerow(list(k) + [n, "%.1f" % (sum(dur[k]) / len(dur[k])) if dur.get(k) else ""] +
               ["%.5g" % (sum(rows[k][c]) / len(rows[k][c])) if c in rows[k] else "" for c in counters])
PY
head -c 3000 "$ROOT/gpurun_out/${TAG}_ntt_per_pass_pmc.csv"
grep -l "rror" "$OUT"/*.log 2>/dev/null | head
