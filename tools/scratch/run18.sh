# r03t visit 7: NTT pass-plan / tile-shape sweep on the current kernels (the values -> coefficients direction follows its
# instruction count now, coefficients -> values does not)
cd $GRAFT_REPO_ROOT
K=tools/scratch/kb/kbench_cur
run() { echo -n "$1 | "; env $1 $K ${2:-116} ${3:-20} 5 | head -1 | sed 's/cols [0-9]* log_n [0-9]* ://'; }
run "A=0"
for tb in 12 13 14; do for sb in 9 10 11; do for cb in 11 12 13; do
  run "ZK_NTT_TILE_BITS=$tb ZK_NTT_STRIDED_BITS=$sb ZK_NTT_CONTIG_BITS=$cb"
done; done; done
for ts in 2 3 4; do run "ZK_NTT_THREADS_SHIFT=$ts"; done
run "ZK_NTT_STRIDED_MOD3=0"
run "ZK_NTT_COLS_FASTEST=0"
run "ZK_NTT_PAD=1"
echo "--- 2431 x 2^17"
run "A=0" 2431 17
for tb in 13 14; do for sb in 9 10 11; do run "ZK_NTT_TILE_BITS=$tb ZK_NTT_STRIDED_BITS=$sb" 2431 17; done; done
echo "--- 30 x 2^21"
run "A=0" 30 21
for tb in 13 14; do for sb in 9 10 11; do run "ZK_NTT_TILE_BITS=$tb ZK_NTT_STRIDED_BITS=$sb" 30 21; done; done
