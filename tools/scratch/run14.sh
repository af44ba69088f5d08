# r03t visit 3: fused tree tops + tail stream: parity, then env A/B
cd $GRAFT_REPO_ROOT
QUICK="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --commit-steps 0 --in-flight 1 --no-pmc --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=b.get('segment_timing_s',{}); print('$1', round(b['ms_per_step'],2), 'commits', round(t.get('compute all trace commitments',0),4), 'tables', round(sum(v for k,v in t.items() if k.startswith('prove')),4))"; }
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
  for V in "1 1" "0 1" "1 0" "0 0"; do
    set -- $V
    ZK_MERKLE_FUSE=$1 ZK_TREE_TAIL=$2 python bench.py $QUICK 2>/dev/null | line "2^20 fuse=$1 tail=$2"
    ZK_MERKLE_FUSE=$1 ZK_TREE_TAIL=$2 python bench.py $QUICK --log-ns realistic 2>/dev/null | line "real fuse=$1 tail=$2"
  done
done
timeout 300 python tools/soak_segment.py 20 3 2>/dev/null | tail -c 300
echo
# PLONK recursion proofs with / without the fused levels
for F in 1 0; do
ZK_MERKLE_FUSE=$F timeout 600 python tools/plonk_trace.py 13 20 2>/dev/null | tail -3
done
