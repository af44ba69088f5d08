// tools/kbench.hip -- kernel micro-benchmark / regression harness for the commit kernels (NTT, leaf hashing, tree).
// Links the library's own host code (csrc/ntt_host.inc, csrc/merkle_host.inc) and kernels without the AIR / FRI /
// segment code, so a kernel experiment rebuilds in seconds instead of minutes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/kbench tools/kbench.hip
//   tools/kbench [cols=116] [log_n=20] [reps=5]
// Prints per-stage ms (HIP events) and FNV-1a checksums of coefficients / LDE / digests for a fixed splitmix64 input:
// the checksums of a change must equal those of the committed code (bit-exactness is the first gate).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../zk_evm_amd/csrc/ctx.hpp"
#include "../zk_evm_amd/csrc/merkle.cuh"
#include "../zk_evm_amd/csrc/ntt.cuh"
#include "../zk_evm_amd/csrc/ntt_host.inc"
#include "../zk_evm_amd/csrc/merkle_host.inc"

extern "C" size_t zk_merkle_num_digests(unsigned log_leaves, unsigned cap_height) {
    size_t tot = 0;
    for (unsigned l = log_leaves + 1; l-- > cap_height;) tot += (size_t)1 << l;
    return tot;
}

__global__ void fill_kernel(u64 *out, size_t n, u64 seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    out[i] = z ^ (z >> 31);
}
__global__ void fnv_kernel(const u64 *in, size_t n, size_t per, u64 *out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t s = t * per;
    if (s >= n) return;
    size_t e = s + per < n ? s + per : n;
    u64 h = 0xCBF29CE484222325ULL;
    for (size_t i = s; i < e; ++i) { h ^= in[i]; h *= 0x100000001B3ULL; }
    out[t] = h;
}
static u64 checksum(zk_ctx *ctx, const u64 *d, size_t n) {
    const size_t per = 4096, parts = (n + per - 1) / per;
    u64 *dp = nullptr;
    hipMalloc(&dp, parts * 8);
    fnv_kernel<<<(unsigned)((parts + 255) / 256), 256, 0, ctx->stream>>>(d, n, per, dp);
    std::vector<u64> h(parts);
    hipMemcpyAsync(h.data(), dp, parts * 8, hipMemcpyDeviceToHost, ctx->stream);
    hipStreamSynchronize(ctx->stream);
    hipFree(dp);
    u64 r = 0xCBF29CE484222325ULL;
    for (u64 x : h) { r ^= x; r *= 0x100000001B3ULL; }
    return r;
}

int main(int argc, char **argv) {
    const size_t cols = argc > 1 ? (size_t)atol(argv[1]) : 116;
    const int log_n = argc > 2 ? atoi(argv[2]) : 20;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    const int rate_bits = 1, cap_height = 4;
    const size_t n = (size_t)1 << log_n, N = n << rate_bits;
    const int log_N = log_n + rate_bits;
    zk_ctx ctxs;
    zk_ctx *ctx = &ctxs;
    hipSetDevice(0);
    hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    ctx->stream = ctx->own_stream;
    ctx->plans = initial_plans();          // ZK_NTT_SWAP_PLANS / the table compiled in; ZK_NTT_SWAP=0|1 forces one form
    hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_pass_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_pass_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_strided_swap_kernel<true, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_strided_swap_kernel<false, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    u64 *vals, *coeffs, *lde, *dig;
    const size_t nd = zk_merkle_num_digests(log_N, cap_height);
    hipMalloc(&vals, cols * n * 8); hipMalloc(&coeffs, cols * n * 8); hipMalloc(&lde, cols * N * 8); hipMalloc(&dig, nd * 32);
    fill_kernel<<<(unsigned)((cols * n + 255) / 256), 256, 0, ctx->stream>>>(vals, cols * n, 0x6FEB51B7EC230F25ULL);
    const u64 *coset = nullptr;
    if (get_coset_table(ctx, log_n, GL_GENERATOR, false, &coset) != ZK_OK) { printf("coset: %s\n", ctx->err.c_str()); return 1; }
    hipEvent_t ev[5];
    for (auto &e : ev) hipEventCreate(&e);
    double tot[4] = {0, 0, 0, 0};
    for (int it = 0; it < reps + 1; ++it) {
        hipEventRecord(ev[0], ctx->stream);
        // the commitment's own plan; ev[1] = between the two transforms
        int rc = ntt_values_to_coeffs_to_lde(ctx, vals, n, coeffs, n, lde, N, cols, log_n, rate_bits, coset, ev[1]);
        hipEventRecord(ev[2], ctx->stream);
        if (rc == ZK_OK) rc = hash_rows(ctx, ZK_HASH_POSEIDON, lde, N, cols, N, log_N, 1, dig);
        hipEventRecord(ev[3], ctx->stream);
        if (rc == ZK_OK) rc = merkle_levels(ctx, ZK_HASH_POSEIDON, dig, log_N, cap_height);
        hipEventRecord(ev[4], ctx->stream);
        hipStreamSynchronize(ctx->stream);
        if (rc != ZK_OK || hipGetLastError() != hipSuccess) { printf("error: %s\n", ctx->err.c_str()); return 1; }
        if (it) for (int i = 0; i < 4; ++i) { float ms; hipEventElapsedTime(&ms, ev[i], ev[i + 1]); tot[i] += ms / reps; }
    }
    printf("cols %zu log_n %d : ifft %.3f ms  lde %.3f ms  leaf_hash %.3f ms  tree %.3f ms | ntt %.1f GB/s\n", cols, log_n,
           tot[0], tot[1], tot[2], tot[3], 40.0 * cols * n / ((tot[0] + tot[1]) * 1e-3) / 1e9);
    printf("fnv coeffs %016llx lde %016llx digests %016llx\n", (unsigned long long)checksum(ctx, coeffs, cols * n),
           (unsigned long long)checksum(ctx, lde, cols * N), (unsigned long long)checksum(ctx, dig, nd * 4));
    return 0;
}
