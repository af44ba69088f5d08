// Micro-benchmark: gl_mul with the 11-instruction reduce (gl.cuh) vs a 9-instruction variant that folds T2*(2^32-1) into
// one v_mad_u64_u32 with carry-out.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 ubench_reduce.hip -o ubench_reduce
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../zk_evm_amd/csrc/gl.cuh"
__device__ __forceinline__ u64 gl_mul2(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 P = (u64)a0 * b0, Q = (u64)a0 * b1, R = (u64)a1 * b0, S = (u64)a1 * b1;
    u32 t1, t2, t3, x0, x1;
    asm("v_add_co_u32 %[x0], vcc, %[q0], %[r0]\n\t"
        "v_addc_co_u32 %[x1], vcc, %[q1], %[r1], vcc\n\t"
        "v_addc_co_u32 %[t3], vcc, 0, %[s1], vcc\n\t"
        "v_add_co_u32 %[t1], vcc, %[p1], %[x0]\n\t"
        "v_addc_co_u32 %[t2], vcc, %[s0], %[x1], vcc\n\t"
        "v_addc_co_u32 %[t3], vcc, 0, %[t3], vcc"
        : [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [x0] "=&v"(x0), [x1] "=&v"(x1)
        : [p1] "v"((u32)(P >> 32)), [q0] "v"((u32)Q), [q1] "v"((u32)(Q >> 32)),
          [r0] "v"((u32)R), [r1] "v"((u32)(R >> 32)), [s0] "v"((u32)S), [s1] "v"((u32)(S >> 32))
        : "vcc");
    u64 lo64 = ((u64)t1 << 32) | (u32)P;
    u64 r; u32 e;
    asm("v_mad_u64_u32 %[r], vcc, %[t2], -1, %[l]\n\t"
        "v_cndmask_b32_e64 %[e], 0, -1, vcc"
        : [r] "=&v"(r), [e] "=&v"(e) : [t2] "v"(t2), [l] "v"(lo64) : "vcc");
    u32 lo = (u32)r, hi = (u32)(r >> 32);
    asm("v_add_co_u32 %[lo], vcc, %[lo], %[e]\n\t"
        "v_addc_co_u32 %[hi], vcc, 0, %[hi], vcc\n\t"
        "v_sub_co_u32 %[lo], vcc, %[lo], %[t3]\n\t"
        "v_subbrev_co_u32 %[hi], vcc, 0, %[hi], vcc\n\t"
        "v_cndmask_b32_e64 %[e], 0, -1, vcc\n\t"
        "v_sub_co_u32 %[lo], vcc, %[lo], %[e]\n\t"
        "v_subbrev_co_u32 %[hi], vcc, 0, %[hi], vcc"
        : [lo] "+v"(lo), [hi] "+v"(hi), [e] "+v"(e) : [t3] "v"(t3) : "vcc");
    return ((u64)hi << 32) | lo;
}
template <int V> __global__ void k(u64 *x, int n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u64 a[4], b = x[i] | 3;
    for (int j = 0; j < 4; ++j) a[j] = x[i] + j * 0x9E3779B97F4A7C15ull;
    for (int it = 0; it < n; ++it)
        for (int j = 0; j < 4; ++j) a[j] = V ? gl_mul2(a[j], b) : gl_mul(a[j], b);
    x[i] = gl_canon(a[0]) ^ gl_canon(a[1]) ^ gl_canon(a[2]) ^ gl_canon(a[3]);
}
int main() {
    const size_t N = (size_t)1 << 20;
    u64 *d, *h = new u64[N], *h2 = new u64[N];
    hipMalloc(&d, N * 8);
    for (size_t i = 0; i < N; ++i) h[i] = i * 0xD6E8FEB86659FD93ull + 12345;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int v = 0; v < 2; ++v) {
        hipMemcpy(d, h, N * 8, hipMemcpyHostToDevice);
        if (v) k<1><<<N / 256, 256>>>(d, 10); else k<0><<<N / 256, 256>>>(d, 10);
        hipMemcpy(d, h, N * 8, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        if (v) k<1><<<N / 256, 256>>>(d, 2000); else k<0><<<N / 256, 256>>>(d, 2000);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[v], e0, e1);
        hipMemcpy(v ? h2 : h, d, N * 8, hipMemcpyDeviceToHost);
        if (!v) { for (size_t i = 0; i < N; ++i) h2[i] = h[i]; for (size_t i = 0; i < N; ++i) h[i] = i * 0xD6E8FEB86659FD93ull + 12345; }
    }
    // h2 holds variant-1 output; recompute variant 0 for comparison
    hipMemcpy(d, h, N * 8, hipMemcpyHostToDevice); k<0><<<N / 256, 256>>>(d, 2000); hipMemcpy(h, d, N * 8, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < N; ++i) bad += h[i] != h2[i];
    printf("gl_mul (11-instr reduce) %.3f ms   gl_mul2 (mad reduce) %.3f ms   ratio %.3f   mismatches %zu\n", ms[0], ms[1], ms[1] / ms[0], bad);
    return 0;
}
