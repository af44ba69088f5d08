// TEST INFRASTRUCTURE -- hipemu: a CPU stand-in for <hip/hip_runtime.h>, found instead of the real header because
// tests/emu/hipemu/ is first on the include path of the emulation build (tests/emu/build_emu.py).
//
// Purpose (r05 verdict, "next round" item 5): the WHOLE library -- zk_evm_amd/csrc/*.hip with every host-side include and every
// kernel, from its own sources -- compiled as plain host C++ and run end to end on the CPU, so that zk_commit_* / zk_prove_* can be
// checked against the oracle, and run under ASan / UBSan / TSan, without a GPU.  What it is NOT: a performance model, or a check of
// the gfx950 inline assembly (gl.cuh / poseidon.cuh / fri.cuh keep their portable bodies beside the asm; this build takes those).
//
//   device memory   = host memory (hipMalloc = malloc: ASan sees every out-of-bounds access of a kernel)
//   a kernel launch = a closure; its grid runs block by block on a pool of OS threads, the threads of a block as FIBERS on one OS
//                     thread, switched at __syncthreads() and at the wave operations (shuffles, lane swaps): tests/emu/hipemu/hipemu.cpp
//   streams         = in-order queues.  HIPEMU_ASYNC=0 (default): every operation runs when it is enqueued (a legal execution: an
//                     infinitely fast device).  HIPEMU_ASYNC=1: NOTHING runs until the host forces it (stream / event / device
//                     synchronisation, hipFree, a synchronous copy), and then only what that wait depends on -- the other legal
//                     extreme, which exposes a missing event wait, a host buffer reused too early, a block freed under a kernel.
//   `<<<...>>>`     is rewritten by tests/emu/translate.py into HIPEMU_LAUNCH(...) before compilation (g++ / clang++ as C++ have no
//                     such token); so is `extern __shared__ T name[];` (-> a pointer to the block's dynamic LDS).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

// ---- qualifiers -----------------------------------------------------------------------------------------------------------------
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __constant__
#define __shared__ static thread_local          /* one block at a time per OS thread; all its fibers see the same array */
#ifndef __restrict__
#define __restrict__ __restrict
#endif
#define HIP_SYMBOL(x) x

// ---- dim3 and the built-in indices ----------------------------------------------------------------------------------------------
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
namespace hipemu {
struct Fiber;
struct ThreadState { dim3 threadIdx, blockIdx, blockDim, gridDim; unsigned lane, wave; };
extern thread_local ThreadState *t_cur;          // the fiber running on this OS thread
void *dyn_lds();                                 // the running block's dynamic shared memory
void sync_threads();
void wave_sync();
// every live lane of the wave deposits `mine`; returns the value deposited by lane `src` (a lane that has exited or is out of
// range: the caller's own)
uint64_t wave_exchange(uint64_t mine, unsigned src);
}  // namespace hipemu
#define threadIdx (hipemu::t_cur->threadIdx)
#define blockIdx (hipemu::t_cur->blockIdx)
#define blockDim (hipemu::t_cur->blockDim)
#define gridDim (hipemu::t_cur->gridDim)
static constexpr int warpSize = 64;

static inline void __syncthreads() { hipemu::sync_threads(); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ---- wave operations ------------------------------------------------------------------------------------------------------------
static inline int __shfl(int v, int src, int width = 64) {
    const unsigned lane = hipemu::t_cur->lane, base = lane & ~(unsigned)(width - 1);
    return (int)(uint32_t)hipemu::wave_exchange((uint32_t)v, base + ((unsigned)src & (unsigned)(width - 1)));
}
static inline int __shfl_down(int v, unsigned delta, int width = 64) {
    const unsigned lane = hipemu::t_cur->lane, pos = lane & (unsigned)(width - 1);
    const unsigned src = pos + delta < (unsigned)width ? lane + delta : lane;
    return (int)(uint32_t)hipemu::wave_exchange((uint32_t)v, src);
}
static inline int __shfl_xor(int v, int mask, int width = 64) {
    (void)width;
    return (int)(uint32_t)hipemu::wave_exchange((uint32_t)v, hipemu::t_cur->lane ^ (unsigned)mask);
}
// the value is wave-uniform wherever the library uses this (a scalar-register hint): checked
uint32_t hipemu_readfirstlane_checked(uint32_t v);
#define __builtin_amdgcn_readfirstlane(v) hipemu_readfirstlane_checked((uint32_t)(v))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_wave_barrier() hipemu::wave_sync()
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
// v_permlane{16,32}_swap: lanes whose bit 4 (5) is clear give their `b` and take the partner's `a`.  The builtin returns {new a, new b}.
struct hipemu_u32x2 { uint32_t x, y; uint32_t operator[](int i) const { return i ? y : x; } };
hipemu_u32x2 hipemu_permlane_swap(unsigned lanebit, uint32_t a, uint32_t b);
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) hipemu_permlane_swap(4, (a), (b))
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) hipemu_permlane_swap(5, (a), (b))

// ---- small device-library functions -----------------------------------------------------------------------------------------------
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline unsigned __brev(unsigned x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    return __builtin_bswap32(x);
}
static inline unsigned long long __brevll(unsigned long long x) { return ((unsigned long long)__brev((unsigned)x) << 32) | __brev((unsigned)(x >> 32)); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
static inline unsigned long long wall_clock64() { return 0; }
static inline unsigned long long clock64() { return 0; }
#ifndef __clang__
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#endif

template <class T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicSub(T *p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicExch(T *p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicAnd(T *p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T atomicCAS(T *p, T expected, T desired) {
    __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return expected;
}
template <class T> static inline T atomicMin(T *p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
template <class T> static inline T atomicMax(T *p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline int atomicAdd(int *p, unsigned v) { return __atomic_fetch_add(p, (int)v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long v) { return __atomic_fetch_add(p, (unsigned long long)v, __ATOMIC_RELAXED); }
static inline unsigned long atomicAdd(unsigned long *p, unsigned long long v) { return __atomic_fetch_add(p, (unsigned long)v, __ATOMIC_RELAXED); }

struct ulonglong2 { unsigned long long x, y; };
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

// ---- the runtime API (C linkage: tests/cabi/*.c use it too, through hip_runtime_api.h) ------------------------------------------------
#include "hip_runtime_api.h"

// ---- launches ---------------------------------------------------------------------------------------------------------------------
#include <tuple>
namespace hipemu {
void launch(dim3 grid, dim3 block, size_t dyn_lds_bytes, hipStream_t stream, std::function<void()> body, const char *name);
// The arguments are EVALUATED AND COPIED when the launch is enqueued, as the real runtime does (an argument like `*d_out`, or a
// struct filled in a loop, must not be read when the kernel finally runs); every thread of the grid then calls the kernel with them.
template <class F, class... Args>
void launch_args(dim3 grid, dim3 block, size_t dyn_lds_bytes, hipStream_t stream, const char *name, F f, Args... args) {
    auto packed = std::make_tuple(args...);
    launch(grid, block, dyn_lds_bytes, stream, [f, packed]() { std::apply(f, packed); }, name);
}
}  // namespace hipemu
// (KERNEL) is parenthesised by the translator: template arguments may contain commas
#define HIPEMU_LAUNCH(KERNEL, GRID, BLOCK, LDS, STREAM, ...) \
    hipemu::launch_args(dim3 GRID, dim3 BLOCK, (size_t)(LDS), (hipStream_t)(STREAM), #KERNEL, [](auto... a_) { KERNEL(a_...); }, __VA_ARGS__)
