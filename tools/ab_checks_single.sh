#!/bin/bash
# A/B inside one gpurun call: the lookup / CTL checks kernel with both challenges per walk (default: 123 VGPRs, 4 waves / SIMD)
# against one challenge per walk (ZK_CTL_TWINS=0 ZK_LOOKUP_DUAL=0: the DUAL = false kernel, 77 VGPRs, 6 waves / SIMD)
cd ${GRAFT_REPO_ROOT:-.}
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary --no-dist-selftest"
line() { python -c "import sys,json; b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=b['segment_timing_s']; print(round(b['ms_per_step'],2), 'KeccakSponge', round(t['prove keccak_sponge_stark STARK']*1e3,2), 'BytePacking', round(t['prove byte_packing_stark STARK']*1e3,2), 'Arithmetic', round(t['prove arithmetic_stark STARK']*1e3,2), 'Logic', round(t['prove logic_stark STARK']*1e3,2), 'Memory', round(t['prove memory_stark STARK']*1e3,2))"; }
for rep in 1 2 3; do for V in "1 1" "0 0"; do set -- $V
  echo -n "2^20 twins=$1 dual=$2 : "; ZK_CTL_TWINS=$1 ZK_LOOKUP_DUAL=$2 python bench.py $Q 2>/dev/null | line
done; done
for V in "1 1" "0 0"; do set -- $V
  echo -n "realistic twins=$1 dual=$2 : "; ZK_CTL_TWINS=$1 ZK_LOOKUP_DUAL=$2 python bench.py $Q --log-ns realistic 2>/dev/null | line
done
timeout 600 python -m pytest tests -m gpu -x -q -k "segment_proof_matches_oracle or stark_prove" 2>&1 | tail -2
ZK_CTL_TWINS=0 ZK_LOOKUP_DUAL=0 timeout 600 python -m pytest tests -m gpu -x -q -k "segment_proof_matches_oracle or stark_prove" 2>&1 | tail -2
