#!/bin/bash
# pytest against the sanitizer build (tools/asan_build.sh); every UBSan report goes to stderr with a stack and the run goes on --
# grep the log for "runtime error".
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$ROOT"
export ZK_STARK_LIB=$ROOT/zk_evm_amd/csrc/build_san/libzkstark_hip_san.so
mkdir -p "$ROOT/gpurun_out/ubsan"
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=$ROOT/gpurun_out/ubsan/report      # (pytest captures stderr: reports go to files, one per process)
if nm -D "$ZK_STARK_LIB" | grep -q __asan_init; then
  export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:detect_odr_violation=0
  export LD_PRELOAD=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
fi
if nm -D "$ZK_STARK_LIB" | grep -q __tsan_init; then
  export TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:log_path=$ROOT/gpurun_out/ubsan/tsan
  export LD_PRELOAD=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)
fi
python -m pytest "$@"
