// Micro-benchmark for the MFMA-f64 Poseidon MDS idea: issue cost of v_mfma_f64_16x16x4_f64, of the
// u32<->f64 conversions around it, and whether MFMA work of one wave overlaps integer VALU work of the other
// waves on the same SIMD.  Reports cycles per wave-instruction per SIMD at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d4 __attribute__((ext_vector_type(4)));
#define ITER 2048

__global__ void k_mfma(double *out, double a, double b) {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    double x = a + threadIdx.x, y = b;
    for (int it = 0; it < ITER; ++it) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c7, 0, 0, 0);
    }
    d4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
}

// NMAD integer mads per MFMA, all independent of the MFMA chain
template <int NMFMA, int NMAD>
__global__ void k_mix(double *out, double a, double b, uint32_t u, uint32_t v) {
    d4 c0 = {0, 0, 0, 0}, c1 = c0;
    double x = a + threadIdx.x, y = b;
    uint64_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = 5, r5 = 6, r6 = 7, r7 = 8;
    uint32_t p = u + threadIdx.x, q = v;
    for (int it = 0; it < ITER; ++it) {
        if (NMFMA > 0) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NMAD / 8; ++k) {
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r0) : "v"(p), "v"(q) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r1) : "v"(p), "v"(q) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r2) : "v"(p), "v"(q) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r3) : "v"(p), "v"(q) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r4) : "v"(p), "v"(q) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r5) : "v"(p), "v"(q) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r6) : "v"(p), "v"(q) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r7) : "v"(p), "v"(q) : "vcc");
        }
        if (NMFMA > 1) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c1, 0, 0, 0);
    }
    d4 s = c0 + c1;
    uint64_t z = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w + (double)(uint32_t)z;
}

#define DEF_F64(NAME, INSN, INIT)                                                                  \
    __global__ void NAME(double *out, double a, double b) {                                        \
        double r0 = INIT, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
        double x = a + threadIdx.x;                                                                 \
        uint32_t u = (uint32_t)b + threadIdx.x;                                                     \
        for (int it = 0; it < ITER; ++it) {                                                        \
            asm volatile(INSN : "+v"(r0) : "v"(x), "v"(u)); asm volatile(INSN : "+v"(r1) : "v"(x), "v"(u)); \
            asm volatile(INSN : "+v"(r2) : "v"(x), "v"(u)); asm volatile(INSN : "+v"(r3) : "v"(x), "v"(u)); \
            asm volatile(INSN : "+v"(r4) : "v"(x), "v"(u)); asm volatile(INSN : "+v"(r5) : "v"(x), "v"(u)); \
            asm volatile(INSN : "+v"(r6) : "v"(x), "v"(u)); asm volatile(INSN : "+v"(r7) : "v"(x), "v"(u)); \
        }                                                                                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;        \
    }
DEF_F64(k_add_f64, "v_add_f64 %0, %0, %1", threadIdx.x)
DEF_F64(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %2", threadIdx.x)
DEF_F64(k_fma_f64, "v_fma_f64 %0, %1, %1, %0", threadIdx.x)

template <class F> static double run(const char *name, F launch, double insts_per_wave, int waves_per_simd_note) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = 2.4;
    double *out; hipMalloc(&out, sizeof(double) * 1024 * 4096);
    // blocks of 256 threads = 4 waves = 1 wave per SIMD; WPS blocks per CU -> WPS waves per SIMD
    auto report = [&](const char *name, float ms, double wave_insts_per_wave, int wps) {
        double cycles = ms * 1e-3 * ghz * 1e9;
        printf("%-44s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (%d waves/SIMD)\n", name, ms,
               cycles / (wave_insts_per_wave * wps), wps);
    };
    for (int wps : {1, 2, 4}) {
        int blocks = cus * wps;
        float ms;
#define RUN(KERNEL, ...) ({ hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); \
        hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(256), 0, 0, __VA_ARGS__); hipDeviceSynchronize(); \
        hipEventRecord(e0); hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(256), 0, 0, __VA_ARGS__); hipEventRecord(e1); \
        hipEventSynchronize(e1); float m; hipEventElapsedTime(&m, e0, e1); m; })
        ms = RUN(k_mfma, out, 1.0, 2.0); report("v_mfma_f64_16x16x4_f64", ms, 8.0 * ITER, wps);
        ms = RUN(k_add_f64, out, 1.0, 2.0); report("v_add_f64", ms, 8.0 * ITER, wps);
        ms = RUN(k_cvt_f64_u32, out, 1.0, 2.0); report("v_cvt_f64_u32", ms, 8.0 * ITER, wps);
        ms = RUN(k_fma_f64, out, 1.0, 2.0); report("v_fma_f64", ms, 8.0 * ITER, wps);
        ms = RUN((k_mix<0, 16>), out, 1.0, 2.0, 3u, 5u); report("16 x v_mad_u64_u32 (per instr)", ms, 16.0 * ITER, wps);
        ms = RUN((k_mix<1, 0>), out, 1.0, 2.0, 3u, 5u); report("1 x mfma, dependent chain (per mfma)", ms, 1.0 * ITER, wps);
        ms = RUN((k_mix<1, 16>), out, 1.0, 2.0, 3u, 5u); report("1 mfma + 16 mad  (cycles per iteration)", ms, 1.0 * ITER, wps);
        ms = RUN((k_mix<2, 32>), out, 1.0, 2.0, 3u, 5u); report("2 mfma + 32 mad  (cycles per iteration)", ms, 1.0 * ITER, wps);
        ms = RUN((k_mix<1, 32>), out, 1.0, 2.0, 3u, 5u); report("1 mfma + 32 mad  (cycles per iteration)", ms, 1.0 * ITER, wps);
    }
    return 0;
}
