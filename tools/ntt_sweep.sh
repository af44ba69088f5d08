#!/bin/bash
# On-GPU sweep of the NTT plan tunables with tools/kbench (commit kernels only; checksums must not change).
# Usage: tools/ntt_sweep.sh [cols=116] [log_n=20]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
C=${1:-116}; L=${2:-20}
echo "== fused vs separate (defaults)"
for F in 1 0 1 0; do echo -n "ZK_NTT_FUSE=$F : "; ZK_NTT_FUSE=$F tools/kbench $C $L 8 | tr '\n' ' '; echo; done
echo "== fused pass threads shift"
for S in 2 3 4; do echo -n "fused_thr_shift=$S : "; ZK_NTT_FUSED_THREADS_SHIFT=$S tools/kbench $C $L 8 | head -1; done
echo "== strided tile / threads / contiguous bits"
for tile in 12 13 14; do for thr in 2 3 4; do for contig in 10 11 12; do for sb in 9 10; do
  echo -n "tile=$tile thr_shift=$thr contig=$contig strided=$sb : "
  ZK_NTT_TILE_BITS=$tile ZK_NTT_THREADS_SHIFT=$thr ZK_NTT_CONTIG_BITS=$contig ZK_NTT_STRIDED_BITS=$sb tools/kbench $C $L 5 | head -1
done; done; done; done
