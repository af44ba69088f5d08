#!/bin/bash
# A/B inside one gpurun call (r04q): the NTT capped to a share of a CU (ZK_NTT_CU_THREADS: extra LDS request) so that the
# other lane's leaf hashing can be resident beside its memory phases (ZK_TRACE_LANES=1), against the default.
cd ${GRAFT_REPO_ROOT:-.}
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(b['ms_per_step'],2), {k: round(v,3) for k,v in list(b['segment_timing_s'].items())[:2]})"; }
echo "== kbench 116 x 2^20 alone: what the cap costs the NTT by itself"
for T in 0 2048 1024 512; do echo -n "cu_threads=$T : "; ZK_NTT_CU_THREADS=$T tools/kbench 116 20 8 | head -1; done
for rep in 1 2; do
  for cfg in "0 0 0" "1 0 0" "1 1024 0" "1 1024 1" "1 512 1" "1 2048 1" "0 1024 0"; do set -- $cfg
    echo -n "2^20 trace_lanes=$1 ntt_cu_threads=$2 side_normal_prio=$3 : "
    ZK_TRACE_LANES=$1 ZK_NTT_CU_THREADS=$2 ZK_SIDE_NORMAL_PRIORITY=$3 python bench.py $Q 2>/dev/null | line
  done
done
