#!/bin/bash
# A/B inside one gpurun call (r04q): the strided NTT passes through the persistent software-pipelined kernel
# (ZK_NTT_PERSIST, default 1) against the generic tile kernel; checksums must be identical.
cd ${GRAFT_REPO_ROOT:-.}
for shape in "116 20" "2431 18" "30 21" "438 19"; do set -- $shape
  for P in 0 1 0 1; do echo -n "cols=$1 log_n=$2 persist=$P : "; ZK_NTT_PERSIST=$P tools/kbench $1 $2 8 | tr '\n' ' '; echo; done
done
echo "== workgroups"
for W in 128 256 512 768; do echo -n "persist wgs=$W : "; ZK_NTT_PERSIST_WGS=$W tools/kbench 116 20 8 | head -1; done
