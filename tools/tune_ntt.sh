#!/bin/bash
# On-GPU sweep of the NTT pass-planning tunables (environment overrides read by the library).
for tile in 12 13 14; do for thr in 3 4 5; do for contig in 10 11; do
  out=$(ZK_NTT_TILE_BITS=$tile ZK_NTT_THREADS_SHIFT=$thr ZK_NTT_CONTIG_BITS=$contig python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('%.2f %.2f %.2f %.2f' % (s['ifft'], s['lde'], s['leaf_hash'], s['tree']))")
  echo "tile=$tile thr_shift=$thr contig=$contig : $out"
done; done; done
