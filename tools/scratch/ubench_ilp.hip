// How many independent v_mad_u64_u32 chains does ONE wave per SIMD need to reach the issue rate?
// (leaf hashing of <= 2^16 rows runs one wave per SIMD; an MDS row is two interleaved accumulator chains)
//   hipcc --offload-arch=gfx950 -O3 -o tools/scratch/ubench_ilp tools/scratch/ubench_ilp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;

template <int CH>
__global__ void __launch_bounds__(256) chains(u64 *out, u32 x, int iters) {
    u64 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = threadIdx.x + c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 24 / CH; ++k) {
#pragma unroll
            for (int c = 0; c < CH; ++c) asm volatile("v_mad_u64_u32 %0, vcc, %1, 17, %0" : "+v"(acc[c]) : "v"(x) : "vcc");
        }
    }
    u64 s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH>
static void run(u64 *d, int waves_per_simd) {
    const int iters = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    dim3 grid(256 * waves_per_simd);
    chains<CH><<<grid, 256>>>(d, 3, 10);
    hipEventRecord(a);
    chains<CH><<<grid, 256>>>(d, 3, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 24 * waves_per_simd);
    printf("chains %d, waves/SIMD %d: %.2f cycles per mad per wave-slot\n", CH, waves_per_simd, cyc);
}

int main() {
    u64 *d; hipMalloc(&d, 8 * 256 * 256 * 8);
    for (int w : {1, 2, 4}) { run<1>(d, w); run<2>(d, w); run<3>(d, w); run<4>(d, w); run<6>(d, w); run<12>(d, w); }
    return 0;
}
