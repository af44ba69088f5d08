#!/bin/bash
# Where one wave of ntt_strided_persist_kernel spends a body (tools/kbench_dbg, ZK_NTT_NT bit 64 = section clocks)
cd ${GRAFT_REPO_ROOT:-.}
for NT in 64 92 76 68 72 80; do echo "== ZK_NTT_NT=$NT (64 = trace; +4 no stores, +8 no tile loads, +16 no twiddle loads)"; ZK_NTT_NT=$NT tools/kbench_dbg ${1:-116} ${2:-20} 4 | grep -v fnv; done
