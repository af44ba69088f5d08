cd ${GRAFT_REPO_ROOT:-.}
for tile in 10 11 12 13 14; do for thr in 2 3 4; do for contig in 9 10 11 12 13; do for sb in 8 9 10; do
  r=$(ZK_NTT_TILE_BITS=$tile ZK_NTT_THREADS_SHIFT=$thr ZK_NTT_CONTIG_BITS=$contig ZK_NTT_STRIDED_BITS=$sb tools/kbench 116 20 4 2>&1 | head -1)
  echo "tile=$tile thr=$thr contig=$contig strided=$sb : $r"
done; done; done; done
