cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fuzz_slice.py -m gpu -x -q 2>&1 | tail -3
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
for rep in 1 2; do
python bench.py $QUICK 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('2^20', round(b['ms_per_step'],2))"
python bench.py $QUICK --log-ns realistic 2>/dev/null | python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('real', round(b['ms_per_step'],2))"
done
