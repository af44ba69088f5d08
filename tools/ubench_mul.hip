// Micro-benchmark: Goldilocks multiply formulations on gfx950 (cycles per wave-multiply per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_mul tools/ubench_mul.hip && tools/ubench_mul
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../zk_evm_amd/csrc/gl.cuh"

// the carry-chain fold used until r03 (gl.cuh now ends the fold with two v_mad_u64_u32: GL_ASM_HEAD / GL_ASM_TAIL)
#define GL_ASM_REDUCE                                                                            \
    "v_sub_co_u32 %[lo], vcc, %[p0], %[t3]\n\t"                                                  \
    "v_subbrev_co_u32 %[t1], vcc, 0, %[t1], vcc\n\t"                                             \
    "v_cndmask_b32_e64 %[e], 0, -1, vcc\n\t"                                                     \
    "v_sub_co_u32 %[lo], vcc, %[lo], %[e]\n\t"                                                   \
    "v_subbrev_co_u32 %[t1], vcc, 0, %[t1], vcc\n\t"                                             \
    "v_sub_co_u32 %[lo], vcc, %[lo], %[t2]\n\t"                                                  \
    "v_subbrev_co_u32 %[e], vcc, 0, %[t2], vcc\n\t"                                              \
    "v_add_co_u32 %[hi], vcc, %[t1], %[e]\n\t"                                                   \
    "v_cndmask_b32_e64 %[e], 0, -1, vcc\n\t"                                                     \
    "v_add_co_u32 %[lo], vcc, %[lo], %[e]\n\t"                                                   \
    "v_addc_co_u32 %[hi], vcc, 0, %[hi], vcc"

#ifndef ITER
#define ITER 2048
#endif

// v2: chained accumulation -- the 64-bit addend of each v_mad_u64_u32 carries the previous partial product's high
// word ({x, 0} pairs built with one v_mov each): 4 mad + 3 mov + 2 add instead of 4 mad + 6 add.
__device__ __forceinline__ u64 gl_mul_v2(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 P = (u64)a0 * b0;
    u64 M = (u64)a0 * b1 + (P >> 32);
    u64 M2 = (u64)a1 * b0 + (u32)M;
    u64 H = (u64)a1 * b1 + (M >> 32) + (M2 >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    u32 lo, hi, t1 = (u32)M2, t2 = (u32)H, t3 = (u32)(H >> 32), e;
    asm(GL_ASM_REDUCE
        : [lo] "=&v"(lo), [hi] "=&v"(hi), [t1] "+&v"(t1), [t2] "+&v"(t2), [t3] "+&v"(t3), [e] "=&v"(e)
        : [p0] "v"((u32)P)
        : "vcc");
    return ((u64)hi << 32) | lo;
#else
    return gl_reduce128(H, (M2 << 32) | (u32)P);
#endif
}

// v3: the first correction of the fold (borrow of [T1:T0] - T3) happens with probability ~2^-33 per lane: skip its three
// instructions with a wave-uniform forward branch inside the asm block.
__device__ __forceinline__ u64 gl_mul_v3(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 P = (u64)a0 * b0;
    u64 M = (u64)a0 * b1 + (P >> 32);
    u64 M2 = (u64)a1 * b0 + (u32)M;
    u64 H = (u64)a1 * b1 + (M >> 32) + (M2 >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    u32 lo, hi, t1 = (u32)M2, t2 = (u32)H, t3 = (u32)(H >> 32), e;
    asm("v_sub_co_u32 %[lo], vcc, %[p0], %[t3]\n\t"
        "v_subbrev_co_u32 %[t1], vcc, 0, %[t1], vcc\n\t"
        "s_cbranch_vccz 1f\n\t"
        "v_cndmask_b32_e64 %[e], 0, -1, vcc\n\t"
        "v_sub_co_u32 %[lo], vcc, %[lo], %[e]\n\t"
        "v_subbrev_co_u32 %[t1], vcc, 0, %[t1], vcc\n"
        "1:\n\t"
        "v_sub_co_u32 %[lo], vcc, %[lo], %[t2]\n\t"
        "v_subbrev_co_u32 %[e], vcc, 0, %[t2], vcc\n\t"
        "v_add_co_u32 %[hi], vcc, %[t1], %[e]\n\t"
        "v_cndmask_b32_e64 %[e], 0, -1, vcc\n\t"
        "v_add_co_u32 %[lo], vcc, %[lo], %[e]\n\t"
        "v_addc_co_u32 %[hi], vcc, 0, %[hi], vcc"
        : [lo] "=&v"(lo), [hi] "=&v"(hi), [t1] "+&v"(t1), [t2] "+&v"(t2), [t3] "+&v"(t3), [e] "=&v"(e)
        : [p0] "v"((u32)P)
        : "vcc");
    return ((u64)hi << 32) | lo;
#else
    return gl_reduce128(H, (M2 << 32) | (u32)P);
#endif
}

// v4: the same with the branch left to the compiler: the borrow mask leaves the asm in an SGPR pair and the rare
// correction is an [[unlikely]] block (out of line: the common path falls through).
__device__ __forceinline__ u64 gl_mul_v4(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 P = (u64)a0 * b0;
    u64 M = (u64)a0 * b1 + (P >> 32);
    u64 M2 = (u64)a1 * b0 + (u32)M;
    u64 H = (u64)a1 * b1 + (M >> 32) + (M2 >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    u32 lo, hi, t1 = (u32)M2, t2 = (u32)H, t3 = (u32)(H >> 32), e;
    u64 bm;
    asm("v_sub_co_u32 %[lo], vcc, %[p0], %[t3]\n\t"
        "v_subbrev_co_u32 %[t1], %[bm], 0, %[t1], vcc"
        : [lo] "=&v"(lo), [t1] "+&v"(t1), [bm] "=&s"(bm)
        : [p0] "v"((u32)P), [t3] "v"(t3)
        : "vcc");
    if (__builtin_expect(bm != 0, 0)) {
        asm("v_cndmask_b32_e64 %[e], 0, -1, %[bm]\n\t"
            "v_sub_co_u32 %[lo], vcc, %[lo], %[e]\n\t"
            "v_subbrev_co_u32 %[t1], vcc, 0, %[t1], vcc"
            : [lo] "+&v"(lo), [t1] "+&v"(t1), [e] "=&v"(e)
            : [bm] "s"(bm)
            : "vcc");
    }
    asm("v_sub_co_u32 %[lo], vcc, %[lo], %[t2]\n\t"
        "v_subbrev_co_u32 %[e], vcc, 0, %[t2], vcc\n\t"
        "v_add_co_u32 %[hi], vcc, %[t1], %[e]\n\t"
        "v_cndmask_b32_e64 %[e], 0, -1, vcc\n\t"
        "v_add_co_u32 %[lo], vcc, %[lo], %[e]\n\t"
        "v_addc_co_u32 %[hi], vcc, 0, %[hi], vcc"
        : [lo] "+&v"(lo), [hi] "=&v"(hi), [e] "=&v"(e)
        : [t1] "v"(t1), [t2] "v"(t2)
        : "vcc");
    return ((u64)hi << 32) | lo;
#else
    return gl_reduce128(H, (M2 << 32) | (u32)P);
#endif
}

// v6 = v4 with the last correction as ONE 64-bit add (v_lshl_add_u64 of {e, 0}) instead of v_add_co + v_addc
__device__ __forceinline__ u64 gl_mul_v6(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 P = (u64)a0 * b0;
    u64 M = (u64)a0 * b1 + (P >> 32);
    u64 M2 = (u64)a1 * b0 + (u32)M;
    u64 H = (u64)a1 * b1 + (M >> 32) + (M2 >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    u32 lo, hi, t1 = (u32)M2, t2 = (u32)H, t3 = (u32)(H >> 32), e;
    u64 bm;
    asm("v_sub_co_u32 %[lo], vcc, %[p0], %[t3]\n\t"
        "v_subbrev_co_u32 %[t1], %[bm], 0, %[t1], vcc"
        : [lo] "=&v"(lo), [t1] "+&v"(t1), [bm] "=&s"(bm)
        : [p0] "v"((u32)P), [t3] "v"(t3)
        : "vcc");
    if (__builtin_expect(bm != 0, 0)) {
        asm("v_cndmask_b32_e64 %[e], 0, -1, %[bm]\n\t"
            "v_sub_co_u32 %[lo], vcc, %[lo], %[e]\n\t"
            "v_subbrev_co_u32 %[t1], vcc, 0, %[t1], vcc"
            : [lo] "+&v"(lo), [t1] "+&v"(t1), [e] "=&v"(e)
            : [bm] "s"(bm)
            : "vcc");
    }
    asm("v_sub_co_u32 %[lo], vcc, %[lo], %[t2]\n\t"
        "v_subbrev_co_u32 %[e], vcc, 0, %[t2], vcc\n\t"
        "v_add_co_u32 %[hi], vcc, %[t1], %[e]\n\t"
        "v_cndmask_b32_e64 %[e], 0, -1, vcc"
        : [lo] "+&v"(lo), [hi] "=&v"(hi), [e] "=&v"(e)
        : [t1] "v"(t1), [t2] "v"(t2)
        : "vcc");
    return (((u64)hi << 32) | lo) + (u64)e;
#else
    return gl_reduce128(H, (M2 << 32) | (u32)P);
#endif
}

template <int V>
__global__ void k_mul(u64 *out, u64 seed) {
    u64 r[8], w = seed | 1;
    for (int i = 0; i < 8; ++i) r[i] = seed * (threadIdx.x + 7 * i + 1);
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = V == 1 ? gl_mul(r[i], w) : V == 2 ? gl_mul_v2(r[i], w) : V == 4 ? gl_mul_v3(r[i], w) : V == 5 ? gl_mul_v4(r[i], w) : V == 6 ? gl_mul_v6(r[i], w) : gl_sqr(r[i]);
    }
    u64 z = 0;
    for (int i = 0; i < 8; ++i) z ^= gl_canon(r[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = z;
}

template <int V>
static void run(const char *name, u64 *d_out, u64 *ref) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8, threads = 256;
    k_mul<V><<<blocks, threads>>>(d_out, 0x9E3779B97F4A7C15ULL);
    hipEventRecord(a);
    k_mul<V><<<blocks, threads>>>(d_out, 0x9E3779B97F4A7C15ULL);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    u64 h[4];
    hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    const double wave_ops = (double)blocks * threads / 64 * ITER * 8;
    printf("%-10s %.3f ms  %.1f cycles per wave-op per SIMD  out %016llx%s\n", name, ms,
           ms * 1e-3 * 2.4e9 * 1024 / wave_ops, (unsigned long long)h[1], ref && *ref != h[1] ? "  MISMATCH" : "");
    if (ref && !*ref) *ref = h[1];
}

int main() {
    u64 *d;
    hipMalloc(&d, 256 * 8 * 256 * 8);
    u64 ref = 0;
    run<1>("gl_mul", d, &ref);
    run<2>("gl_mul_v2", d, &ref);
    run<3>("gl_sqr", d, nullptr);
    run<4>("gl_mul_v3", d, &ref);
    run<5>("gl_mul_v4", d, &ref);
    run<6>("gl_mul_v6", d, &ref);
    return 0;
}
