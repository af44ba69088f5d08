// The launch of one table AIR's quotient kernel (starky `compute_quotient_polys`, the per-point part): shared by the AIR
// translation units zk_airs_*.hip, which instantiate it for their group of AIRs.
#pragma once
#include "internal.hpp"
#include "merkle.cuh"     // keccak_f1600 (fri.cuh's PoW kernel)
#include "quotient.cuh"
#include "airs.cuh"

// `shape_key` identifies everything the number of yielded constraints depends on (AIR, challenges, entry counts of
// every lookup and CTL z-data); K is measured once per shape by running the kernel body for one point in counting
// mode and cached in the ctx.
template <class Air>
static int launch_quotient(zk_ctx *ctx, QuotientArgs A, u32 size, const std::vector<u64> &shape_key, DevBuf &scratch,
                           bool heavy = false) {
    hipStream_t st = ctx->stream;
    u32 K = 0;
    auto it = ctx->constraint_counts.find(shape_key);
    if (it != ctx->constraint_counts.end()) {
        K = it->second;
    } else {
        u32 *d_count = nullptr;
        ZK_TRY(scratch.alloc(&d_count, 1));
        QuotientArgs C = A;
        C.count_out = d_count;
        quotient_count_kernel<Air><<<1, 64, 0, st>>>(C);
        ZK_TRY(check_launch(ctx, "quotient_count_kernel"));
        HIP_TRY(ctx, hipMemcpyAsync(&K, d_count, sizeof(u32), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (K > ZK_QUOTIENT_MAX_CONSTRAINTS)
            return set_err(ctx, ZK_ERR_UNSUPPORTED, "AIR yields %u constraints (supported: up to %u)", K, ZK_QUOTIENT_MAX_CONSTRAINTS);
        ctx->constraint_counts[shape_key] = K;
    }
    u64 *d_pow = nullptr;
    int *d_err = nullptr;
    ZK_TRY(scratch.alloc(&d_pow, (size_t)2 * K + 2));
    ZK_TRY(scratch.alloc(&d_err, 1));
    HIP_TRY(ctx, hipMemsetAsync(d_err, 0, sizeof(int), st));
    if (K) {
        alpha_power_table_kernel<<<(K + 255) / 256, 256, 0, st>>>(d_pow, K, K, A.alphas[0], A.alphas[A.n_challenges > 1 ? 1 : 0]);
        ZK_TRY(check_launch(ctx, "alpha_power_table_kernel"));
    }
    A.alpha_pow[0] = d_pow;
    A.alpha_pow[1] = d_pow + K;
    A.n_constraints = K;
    A.err_flag = d_err;
    if (heavy) quotient_kernel_heavy<Air><<<(size + 255) / 256, 256, 0, st>>>(A);
    else quotient_kernel<Air><<<(size + 255) / 256, 256, 0, st>>>(A);
    ZK_TRY(check_launch(ctx, "quotient_kernel"));
    int err = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&err, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));   // (also: host program buffers may go away after return)
    if (err) return set_err(ctx, ZK_ERR_HIP, "constraint count mismatch in the quotient kernel (expected %u)", K);
    return ZK_OK;
}


#define ZK_AIR_CASE(ID, AIR, HEAVY)                                                                                       \
    case ID:                                                                                                            \
        if (AIR::COLUMNS && n_trace_cols != AIR::COLUMNS)                                                               \
            return set_err(ctx, ZK_ERR_BAD_ARG, "AIR %u expects %u trace columns, got %zu", air_id, (unsigned)AIR::COLUMNS, n_trace_cols); \
        return launch_quotient<AIR>(ctx, A, size, shape_key, scratch, HEAVY);
#define ZK_AIR_CASE_CPU(ID, AIR)                                                                                          \
    case ID:                                                                                                            \
        if (n_trace_cols != AIR::COLUMNS)                                                                               \
            return set_err(ctx, ZK_ERR_BAD_ARG, "AIR %u expects %u trace columns, got %zu", air_id, (unsigned)AIR::COLUMNS, n_trace_cols); \
        if (n_air_consts != 4) return set_err(ctx, ZK_ERR_BAD_ARG, "the CPU AIR needs 4 kernel-label constants");       \
        return launch_quotient<AIR>(ctx, A, size, shape_key, scratch, false);
