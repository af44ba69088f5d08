// Cost of one multiply-accumulate term of an alpha-power dot product (cycles per wave-term per SIMD), coefficient
// wave-uniform (SGPR), value per lane:
//   1: acc = gl_add(acc, gl_mul(v, c))                    (17 + 8 VALU)
//   2: DotAcc (fri.cuh): 4 x (v_mad_u64_u32 + v_addc on its carry-out), one fold at the end
//   3: carry-free: c in three 22-bit limbs (scalar), v in 32-bit halves: six v_mad_u64_u32 into six 64-bit
//      accumulators that cannot overflow for <= 1024 terms, one fold at the end
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/scratch/ubench_dot tools/scratch/ubench_dot.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../zk_evm_amd/csrc/gl.cuh"
#include "../../zk_evm_amd/csrc/fri.cuh"

#define TERMS 512

struct Dot6 { u64 a[6]; };
__device__ __forceinline__ void dot6_mac(Dot6 &d, u32 c0, u32 c1, u32 c2, u64 v) {
    const u32 v0 = (u32)v, v1 = (u32)(v >> 32);
    asm("v_mad_u64_u32 %0, vcc, %6, %9, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %7, %9, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %6, %10, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %7, %10, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %8, %10, %5"
        : "+v"(d.a[0]), "+v"(d.a[1]), "+v"(d.a[2]), "+v"(d.a[3]), "+v"(d.a[4]), "+v"(d.a[5])
        : "s"(c0), "s"(c1), "s"(c2), "v"(v0), "v"(v1)
        : "vcc");
}
// value = a0 + a1 2^22 + a2 2^44 + a3 2^32 + a4 2^54 + a5 2^76   (each a_i < 2^64)
__device__ __forceinline__ u64 dot6_reduce(const Dot6 &d) {
    u64 r = d.a[0];
    r = gl_add(r, gl_mul(d.a[1], (u64)1 << 22));
    r = gl_add(r, gl_mul(d.a[2], (u64)1 << 44));
    r = gl_add(r, gl_mul(d.a[3], (u64)1 << 32));
    r = gl_add(r, gl_mul(d.a[4], (u64)1 << 54));
    r = gl_add(r, gl_mul(d.a[5], gl_canon(gl_mul((u64)1 << 38, (u64)1 << 38))));
    return r;
}

template <int V>
__global__ void k_dot(const u64 *coef, u64 *out, u64 seed, int iters) {
    u64 x = seed * (threadIdx.x + 1) + blockIdx.x, res = 0;
    for (int it = 0; it < iters; ++it) {
        u64 acc = 0;
        DotAcc d; dot_acc_init(d);
        Dot6 e; for (int i = 0; i < 6; ++i) e.a[i] = 0;
#pragma unroll 8
        for (int t = 0; t < TERMS; ++t) {
            const u64 c = coef[t];
            const u32 cl = __builtin_amdgcn_readfirstlane((u32)c), ch = __builtin_amdgcn_readfirstlane((u32)(c >> 32));
            x = x * 6364136223846793005ULL + 1442695040888963407ULL;      // a new per-lane value each term (2 mads)
            if (V == 1) acc = gl_add(acc, gl_mul(x, c));
            else if (V == 2) dot_acc_mac(d, cl, ch, x);
            else if (V == 3) dot6_mac(e, cl & 0x3FFFFF, (cl >> 22) | ((ch & 0xFFF) << 10), ch >> 12, x);
        }
        if (V == 2) acc = dot_acc_reduce(d);
        if (V == 3) acc = dot6_reduce(e);
        res ^= gl_canon(acc);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

template <int V>
static void run(const char *name, const u64 *coef, u64 *d_out, u64 *ref) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8, threads = 256, iters = 8;
    k_dot<V><<<blocks, threads>>>(coef, d_out, 0x9E3779B97F4A7C15ULL, 1);
    hipEventRecord(a);
    k_dot<V><<<blocks, threads>>>(coef, d_out, 0x9E3779B97F4A7C15ULL, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    u64 h[4];
    hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    const double wave_terms = (double)blocks * threads / 64 * iters * TERMS;
    printf("%-28s %.3f ms  %.1f cycles per wave-term per SIMD (incl. ~9 for the value generator)  out %016llx%s\n", name, ms,
           ms * 1e-3 * 2.4e9 * 1024 / wave_terms, (unsigned long long)h[1], *ref && *ref != h[1] ? "  MISMATCH" : "");
    if (!*ref) *ref = h[1];
}

int main() {
    u64 *d, *coef, hc[TERMS];
    for (int i = 0; i < TERMS; ++i) hc[i] = (0x9E3779B97F4A7C15ULL * (i + 1)) % 0xFFFFFFFF00000001ULL;
    hipMalloc(&d, 256 * 8 * 256 * 8);
    hipMalloc(&coef, sizeof hc);
    hipMemcpy(coef, hc, sizeof hc, hipMemcpyHostToDevice);
    u64 ref = 0;
    run<1>("gl_mul + gl_add", coef, d, &ref);
    run<2>("DotAcc (mad + addc)", coef, d, &ref);
    run<3>("six carry-free mads", coef, d, &ref);
    return 0;
}
