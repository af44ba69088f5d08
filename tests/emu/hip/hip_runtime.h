// CPU emulation shim for tests/emu/ntt_swap_emu.cpp (TEST INFRASTRUCTURE: lets g++ compile zk_evm_amd/csrc/ntt_swap.cuh -- the
// kernels' own source, not a restatement -- and run it with one OS thread per lane).  Found instead of the real <hip/hip_runtime.h>
// because tests/emu/ is first on the include path.  Only what gl.cuh, ntt_common.cuh and ntt_swap.cuh use.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__                      /* only `extern __shared__ T name[]` occurs: resolves to the harness's global arrays */

struct EmuIdx { unsigned x, y, z; };
extern thread_local EmuIdx threadIdx, blockIdx, blockDim, gridDim;
void __syncthreads();

// builtins of the device compiler that the headers use outside their __HIP_DEVICE_COMPILE__ branches
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
static inline unsigned long long wall_clock64() { return 0; }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

struct ulonglong2 { unsigned long long x, y; };
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { return ulonglong2{x, y}; }
