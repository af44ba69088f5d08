#!/bin/bash
# A sanitizer build of the library's HOST side -> zk_evm_amd/csrc/build_san/libzkstark_hip_san.so (device code compiled as
# always: -fno-gpu-sanitize).  Default: UndefinedBehaviorSanitizer (shifts, signed overflow, misaligned / null accesses, bounds of
# sized arrays, bad enum / bool loads, float casts), which needs no preloaded runtime.  `asan` as first argument adds
# AddressSanitizer (`tsan`: ThreadSanitizer instead, for the batch prover's worker threads and the per-slot contexts) -- usable for host-only callers; with a GPU in the process ROCm's ASan runtime intercepts
# hsa_amd_memory_pool_allocate and, without the ASan-built HSA libraries (absent from this image), aborts HIP's first allocation
# ("out of memory: allocator is trying to allocate 0x400000 bytes", r04), so the GPU tests run under UBSan only.
# `tsan` likewise builds, but with its runtime preloaded torch cannot bring up the GPU in this image ("Error in dlopen:
# libcaffe2_nvrtc.so", r04): usable only from a caller without torch.
#   tools/asan_run.sh <pytest args>        (through gpurun)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$ROOT/zk_evm_amd/csrc"
SAN=undefined; [ "$1" = asan ] && SAN=address,undefined; [ "$1" = tsan ] && SAN=thread
mkdir -p build_san
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wno-unused-value -fsanitize=$SAN -fno-sanitize=vptr,function -fno-gpu-sanitize -fno-omit-frame-pointer"
pids=()
for u in zkstark zk_airs_a zk_airs_b zk_airs_c zk_airs_d zk_plonk zk_tracegen; do
  hipcc $FLAGS -c $u.hip -o build_san/$u.o & pids+=($!)
done
rc=0; for p in "${pids[@]}"; do wait $p || rc=1; done
[ $rc = 0 ] || { echo "sanitizer compile failed"; exit 1; }
RTD=$(dirname $(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1))
EXTRA="-L$RTD -lclang_rt.ubsan_standalone-x86_64 -Wl,-rpath,$RTD"     # (clang does not link a sanitizer runtime into a shared object)
[ "$1" = asan ] && EXTRA="-fsanitize=address -shared-libsan $EXTRA"
[ "$1" = tsan ] && EXTRA="-fsanitize=thread -shared-libsan"
hipcc --offload-arch=gfx950 -shared -fPIC -o build_san/libzkstark_hip_san.so build_san/*.o $EXTRA && ls -la build_san/libzkstark_hip_san.so
