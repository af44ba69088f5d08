cd $GRAFT_REPO_ROOT
QUICK="--steps 10 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
ZK_HOST_PROFILE=1 python bench.py $QUICK --log-ns realistic 2>&1 | grep "zk host\|ms_per_step" | cut -c1-200 | head -30
python bench.py $QUICK 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('2^20', round(b['ms_per_step'],2), {k: round(v,4) for k,v in b['segment_timing_s'].items()})"
ZK_ARITH_TILED=0 python bench.py $QUICK 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('2^20 arith one-lane', round(b['ms_per_step'],2))"
