#!/bin/bash
# A/B: the strided NTT passes through the lane-swap kernel (ZK_NTT_SWAP=1, ntt.cuh ntt_strided_swap_kernel) against the LDS
# tile kernel, alone and with the column batches of ab_ntt_col_batch.sh.  Checksums must be identical in every line.
cd "$(dirname "$0")/.."
for shape in "116 20" "2431 18" "30 21" "86 19" "9 22"; do
  for cfg in "0 0" "1 0" "0 0" "1 0" "0 96" "1 96"; do
    set -- $cfg
    echo -n "shape=$shape swap=$1 batch_MB=$2 : "; ZK_NTT_SWAP=$1 ZK_NTT_COL_BATCH_MB=$2 tools/kbench $shape 8 | tr '\n' ' '; echo
  done
done
