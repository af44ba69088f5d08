#!/bin/bash
# A/B: the NTT passes through the lane-swap kernels (ZK_NTT_SWAP=1, csrc/ntt_swap.cuh; ZK_NTT_SWAP_CONTIG=0: the strided passes
# only) against the LDS tile kernels, alone and with the column batches of ab_ntt_col_batch.sh.  Checksums must be identical in
# every line of a shape.
cd "$(dirname "$0")/.."
for shape in "116 20" "2431 18" "30 21" "86 19" "9 22" "116 17" "438 13" "64 14" "300 16" "40 10"; do
  for cfg in "0 1 0 1" "1 0 0 1" "1 1 0 1" "0 1 0 1" "1 0 0 1" "1 1 0 1" "0 1 96 1" "1 1 96 1" "0 1 96 2" "1 1 96 2" "1 1 192 2"; do
    set -- $cfg
    echo -n "shape=$shape swap=$1 contig=$2 batch_MB=$3 streams=$4 : "; ZK_NTT_SWAP=$1 ZK_NTT_SWAP_CONTIG=$2 ZK_NTT_COL_BATCH_MB=$3 ZK_NTT_COL_BATCH_STREAMS=$4 timeout 120 tools/kbench $shape 5 | tr '\n' ' '; echo
  done
done
