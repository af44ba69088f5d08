#!/bin/bash
# A/B inside one gpurun call: the big tables' trace commitments over both lanes (ZK_TRACE_LANES) x side-lane priority
cd ${GRAFT_REPO_ROOT:-.}
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(b['ms_per_step'],2), {k: round(v,3) for k,v in list(b['segment_timing_s'].items())[:2]}, 'side', {k: round(v,1) for k,v in b['side_lane']['ms_per_step'].items()})"; }
for rep in 1 2; do for TL in 0 1; do for PR in 0 1; do
  echo -n "2^20 trace_lanes=$TL side_normal_prio=$PR : "; ZK_TRACE_LANES=$TL ZK_SIDE_NORMAL_PRIORITY=$PR python bench.py $Q 2>/dev/null | line
done; done; done
for TL in 0 1; do for PR in 0 1; do
  echo -n "realistic trace_lanes=$TL side_normal_prio=$PR : "; ZK_TRACE_LANES=$TL ZK_SIDE_NORMAL_PRIORITY=$PR python bench.py $Q --log-ns realistic 2>/dev/null | line
done; done
