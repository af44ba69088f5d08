# r03v: sweep of the host-side switches on the final kernels (two workloads, one line each)
cd $GRAFT_REPO_ROOT
QUICK="--steps 5 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=b.get('segment_timing_s',{}); print('$1', round(b['ms_per_step'],2), 'commits', round(t.get('compute all trace commitments',0),4), 'ctl', round(t.get('compute CTL data',0),4), 'tables', round(sum(v for k,v in t.items() if k.startswith('prove')),4))"; }
run() { env $1 python bench.py $QUICK 2>/dev/null | line "2^20 $1"; env $1 python bench.py $QUICK --log-ns realistic 2>/dev/null | line "real $1"; }
run "A=0"
run "ZK_CTL_TWINS=0"
run "ZK_LOOKUP_DUAL=0"
run "ZK_FRI_COEFF_COMBINE_MIN_LOG=25"
run "ZK_FRI_COEFF_COMBINE_MIN_LOG=26"
run "ZK_FRI_COEFF_COMBINE_MIN_LOG=29"
run "A=1"
run "ZK_TREE_TAIL_LOG=15"
run "ZK_TREE_TAIL_LOG=19"
run "ZK_TREE_TAIL_LOG=21"
run "ZK_SIDE_LANE_MAX_LOG=14"
run "ZK_SIDE_LANE_MAX_LOG=17"
run "ZK_SIDE_LANE_MAX_LOG=19"
run "ZK_MERKLE_COOP_LOG=13"
run "ZK_MERKLE_COOP_LOG=15"
run "ZK_HASH_COOP_LOG=13"
run "ZK_HASH_COOP_LOG=15"
run "A=2"
