#!/bin/bash
# The GPU tests of tests/emu/quick_slice.txt against the CPU emulation build (tests/emu/) under the sanitizers that
# cannot run beside the GPU runtime (r05 verdict, next 5):
#   asan  AddressSanitizer + UndefinedBehaviorSanitizer: "device" memory is host memory, so every out-of-bounds or use-after-free access
#         of a KERNEL, of the arena, of the staging buffers is caught; stack-use-after-scope across the fiber switches included
#   tsan  ThreadSanitizer: the host side's threads (two contexts proving concurrently, the PLONK batch workers, the emulator's block pool)
# -> profiles/<tag>_emu_asan_ubsan.log, profiles/<tag>_emu_tsan.log (the pytest summary + every sanitizer report, if any)
#   tools/emu_sanitizers.sh [tag=r06c] [asan|tsan|both]
TAG=${1:-r06c}; WHICH=${2:-both}
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
RTD=$(dirname "$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)")
IDS=$(grep "::" tests/emu/quick_slice.txt | grep -v "^#")
run() {   # san, runtime .so, options-variable=value, out file, pytest ids...
  local san=$1 rt=$2 opt=$3 out=$4; shift 4
  python tests/emu/build_emu.py --san "$san" > /dev/null || { echo "build of the $san variant failed"; return 1; }
  local lib=$ROOT/tests/emu/build_$san/libzkstark_emu_$san.so rep=/tmp/emu_${san}_reports_$$
  rm -f "$rep".*
  # (the sanitizers write their reports to files: pytest captures the tests' stderr)
  { echo "# quick slice of the GPU tests on the CPU emulation build under -fsanitize=$san ($(git rev-parse --short HEAD)); NOT a hardware run";
    env ZK_STARK_LIB="$lib" HIPEMU_TORCH_SHIM=1 HIPEMU_THREADS=2 PYTHONPATH="$ROOT/tests/emu/site:$ROOT" LD_PRELOAD="$rt" "$opt:log_path=$rep" \
        ASAN_SYMBOLIZER_PATH=/opt/rocm/lib/llvm/bin/llvm-symbolizer TSAN_SYMBOLIZER_PATH=/opt/rocm/lib/llvm/bin/llvm-symbolizer \
        timeout 10800 python -m pytest "$@" -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -60;
    python tools/sanitizer_report_summary.py "$rep".*; } > "$out"
  tail -4 "$out"
}
if [ "$WHICH" = asan ] || [ "$WHICH" = both ]; then
  run asan "$RTD/libclang_rt.asan-x86_64.so" "ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1" profiles/${TAG}_emu_asan_ubsan.log $IDS
fi
if [ "$WHICH" = tsan ] || [ "$WHICH" = both ]; then
  run tsan "$RTD/libclang_rt.tsan-x86_64.so" "TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:second_deadlock_stack=1:exitcode=0" profiles/${TAG}_emu_tsan.log $IDS
  # host threads: two worker threads, a zk_ctx and a stream each, on two emulated devices in one process
  { echo "# tests/emu/two_devices_driver.py (two scheduler threads, two emulated devices) under TSan:";
    rm -f /tmp/emu_tsan_threads_$$.*
    env ZK_STARK_LIB="$ROOT/tests/emu/build_tsan/libzkstark_emu_tsan.so" HIPEMU_TORCH_SHIM=1 HIPEMU_THREADS=2 HIPEMU_DEVICES=2 PYTHONPATH="$ROOT/tests/emu/site:$ROOT" \
        LD_PRELOAD="$RTD/libclang_rt.tsan-x86_64.so" TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:exitcode=0:log_path=/tmp/emu_tsan_threads_$$" \
        TSAN_SYMBOLIZER_PATH=/opt/rocm/lib/llvm/bin/llvm-symbolizer timeout 3600 python tests/emu/two_devices_driver.py 2>&1 | grep "RESULT"
    python tools/sanitizer_report_summary.py /tmp/emu_tsan_threads_$$.*; } >> profiles/${TAG}_emu_tsan.log
  tail -3 profiles/${TAG}_emu_tsan.log
fi
