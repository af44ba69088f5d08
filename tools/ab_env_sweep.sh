#!/bin/bash
# env-knob sweep inside one gpurun call: realistic heights and 2^20, one knob at a time (defaults: side lane 16, tree tail 17, coop 14 / 14)
cd ${GRAFT_REPO_ROOT:-.}
Q="--warmup 2 --no-cpu-baseline --no-pmc --no-secondary --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(b['ms_per_step'],2))"; }
for kv in "X=0" "ZK_SIDE_LANE_MAX_LOG=14" "ZK_SIDE_LANE_MAX_LOG=17" "ZK_SIDE_LANE_MAX_LOG=18" "ZK_TREE_TAIL_LOG=15" "ZK_TREE_TAIL_LOG=19" "ZK_MERKLE_COOP_LOG=12" "ZK_MERKLE_COOP_LOG=15" "ZK_HASH_COOP_LOG=12" "ZK_HASH_COOP_LOG=15" "X=1"; do
  echo -n "$kv realistic: "; env $kv python bench.py $Q --steps 8 --log-ns realistic 2>/dev/null | line
done
for kv in "X=0" "ZK_TREE_TAIL_LOG=15" "ZK_TREE_TAIL_LOG=19" "ZK_MERKLE_COOP_LOG=15" "X=1"; do
  echo -n "$kv 2^20: "; env $kv python bench.py $Q --steps 4 2>/dev/null | line
done
