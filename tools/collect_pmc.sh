#!/bin/bash
# Collect rocprofv3 PMC counters for bench.py in separate passes (SQ=8 slots, TCC=4; FETCH_SIZE and
# WRITE_SIZE cannot share a pass).  Writes CSVs under gpurun_out/pmc_<tag>/passN.
# Usage (on the GPU box, from the repo root): tools/collect_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-pmc $*"
run() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE
find "$OUT" -name "*.csv" | head -20
