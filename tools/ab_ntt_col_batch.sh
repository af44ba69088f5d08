#!/bin/bash
# A/B: the commitment's NTT passes over column batches that fit the Infinity Cache (ZK_NTT_COL_BATCH_MB) against all columns
# per pass (0).  Checksums must be identical in every line.
cd "$(dirname "$0")/.."
for shape in "116 20" "2431 18" "30 21" "86 19"; do
  for MB in 0 32 64 96 128 192 0; do
    for ST in 1 2; do
      echo -n "shape=$shape batch_MB=$MB streams=$ST : "; ZK_NTT_COL_BATCH_MB=$MB ZK_NTT_COL_BATCH_STREAMS=$ST tools/kbench $shape 5 | tr '\n' ' '; echo
    done
  done
done
