# A/B of an environment switch inside one gpurun call.  Usage: ab_env.sh VAR "test -k expression"
cd $GRAFT_REPO_ROOT
VAR=${1:-ZK_NTT_SWAP}
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
timeout 900 python -m pytest tests -m gpu -x -q -k "${2:-ntt or commit or kat or segment_proof_matches_oracle}" 2>&1 | tail -2
for rep in 1 2 3; do
  for V in 0 1; do
    env $VAR=$V python bench.py $QUICK 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('2^20 $VAR=$V', round(b['ms_per_step'],2), round(b.get('ntt',{}).get('achieved_GBs',0),1))"
  done
done
env $VAR=0 python bench.py $QUICK --log-ns realistic 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('real $VAR=0', round(b['ms_per_step'],2))"
env $VAR=1 python bench.py $QUICK --log-ns realistic 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('real $VAR=1', round(b['ms_per_step'],2))"
