#!/bin/bash
# The first visit to a node with MORE THAN ONE MI355X, in one command (r05 verdict, next 8): no run of this repository has had two
# RCCL ranks.  Writes ONE JSONL file -- a line per step: the step's name, its exit status, and the bench line it printed with the
# fields that matter here lifted out (`dist.rccl_large_piece_intact_peer`, `dist.library_pieces_intact`, per-rank ms, the level-3
# phase timings) -- plus each step's full output next to it.
#   tools/first_multi_gpu_visit.sh [N=all visible GPUs] [tag]        (results: gpurun_out/<tag>_multi_gpu.jsonl)
# With N = 1 (a one-GPU box: what the world-1 GPU test of this script runs) every step still goes through RCCL, on one rank.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$ROOT"
NGPU=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
N=${1:-$NGPU}; TAG=${2:-multi}
OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; J="$OUT/${TAG}_multi_gpu.jsonl"; : > "$J"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
STEPS=${ZK_VISIT_STEPS:-5}; LOGN=${ZK_VISIT_LOG_N:-18}
step() {      # name, timeout, command...
  local name=$1 limit=$2; shift 2
  local log="$OUT/${TAG}_${name}.log"
  rm -f bench_extra.json
  timeout "$limit" "$@" > "$log" 2> "$log.err"; local rc=$?
  [ -f bench_extra.json ] && cp bench_extra.json "$OUT/${TAG}_${name}_extra.json"
  python - "$name" "$rc" "$log" "$OUT/${TAG}_${name}_extra.json" >> "$J" <<'PY'
import json, os, sys
name, rc, log, extra_path = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
extra = {}
if os.path.exists(extra_path):
    try:
        extra = json.load(open(extra_path))
    except ValueError:
        pass
line = None
for ln in open(log, errors="replace"):
    if ln.startswith("{"):
        try:
            line = json.loads(ln)
        except ValueError:
            pass
rec = {"step": name, "rc": rc, "log": log}
if line:
    d = line.get("dist") or {}
    rec.update({"n_gpus": line.get("n_gpus"), "value": line.get("value"), "unit": line.get("unit"), "ms_per_step": line.get("ms_per_step"),
                "scaling": line.get("scaling"), "dist": d,
                "rccl_large_piece_intact_peer": d.get("rccl_large_piece_intact_peer"), "library_pieces_intact": d.get("library_pieces_intact", d.get("pieces_of_256MiB_intact")),
                "per_rank_ms_per_step": line.get("per_rank_ms_per_step") or extra.get("per_rank_ms_per_step"),
                "segment_timing_s": extra.get("segment_timing_s"),          # the latency modes: the level-2 phases and the level-3 stages (zk_comm_last_timing)
                "comm": extra.get("comm") or d.get("comm")})
print(json.dumps(rec))
PY
  echo "== $name rc=$rc"; tail -1 "$J" | cut -c1-400
}
# level 1: the scaling curve of the headline metric (one independent 2^20 segment per GPU, no data-path collective: "weak")
for n in 1 2 4 8; do
  [ "$n" -le "$N" ] && step "level1_gpus$n" 1800 python bench.py --gpus "$n" --steps "$STEPS" --warmup 2 --no-cpu-baseline --no-pmc --no-secondary --commit-steps 0 --in-flight 1 ${ZK_VISIT_FORCE_DIST:+--force-dist}
done
# level 2: ONE segment's tables over the GPUs (latency mode, "strong"); 2^18 keeps Keccak's 2431 columns inside one GPU
step "level2_table_parallel" 1800 python bench.py --gpus "$N" --mode table_parallel --log-n "$LOGN" --steps "$STEPS" --warmup 2 ${ZK_VISIT_FORCE_DIST:+--force-dist}
# level 2 + 3: the same with Keccak's ROWS over all ranks (two all-to-alls per commitment, next rows from the column owners)
step "level3_keccak_rows" 1800 python bench.py --gpus "$N" --mode table_parallel_keccak_rows --log-n "$LOGN" --steps "$STEPS" --warmup 2 ${ZK_VISIT_FORCE_DIST:+--force-dist}
# the RCCL > 1 GiB drill between two REAL peers, alone
if [ "$N" -ge 2 ]; then
  step "rccl_repro_2_peers" 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_repro.py
fi
echo "wrote $J"
