cd $GRAFT_REPO_ROOT
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
for rep in 1 2 3; do
  for V in 1 0; do
    ZK_STAGE_KERNEL=$V python bench.py $QUICK 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('2^20 stage_kernel=$V', round(b['ms_per_step'],2))"
    ZK_STAGE_KERNEL=$V python bench.py $QUICK --log-ns realistic 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('real stage_kernel=$V', round(b['ms_per_step'],2))"
  done
done
