// The AIR half of the quotient (starky `compute_quotient_polys`, the table's own `eval_packed_generic` per point): shared by
// the AIR translation units zk_airs_*.hip, which instantiate it for their group of AIRs.  The driver (quotient_host.inc, core
// unit) supplies the alpha powers, the total constraint count and the error flag, and runs the lookup / CTL checks kernel.
#pragma once
#define ZK_DEVICE_FUNCS_ONLY          // the device headers' own kernels (hashing, NTT, FRI, column generators) are not needed here
#include "internal.hpp"
#include "quotient.cuh"
#include "airs.cuh"

// count != nullptr: measure K_air (constraints the AIR yields per point) into *count; otherwise launch the AIR kernel.
template <class Air, bool HEAVY = false>
static int launch_quotient_air(zk_ctx *ctx, const QuotientArgs &A, u32 size, DevBuf &scratch, u32 *count) {
    hipStream_t st = ctx->stream;
    if (count) {
        u32 *d_count = nullptr;
        ZK_TRY(scratch.alloc(&d_count, 1));
        QuotientArgs C = A;
        C.count_out = d_count;
        quotient_count_kernel<Air><<<1, 64, 0, st>>>(C);
        ZK_TRY(check_launch(ctx, "quotient_count_kernel"));
        HIP_TRY(ctx, hipMemcpyAsync(count, d_count, sizeof(u32), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        return ZK_OK;
    }
    if constexpr (HEAVY) quotient_kernel_heavy<Air><<<(size + 255) / 256, 256, 0, st>>>(A);
    else quotient_kernel<Air><<<(size + 255) / 256, 256, 0, st>>>(A);
    return check_launch(ctx, "quotient_kernel");
}

#define ZK_AIR_CASE(ID, AIR, HEAVY)                                                                                       \
    case ID:                                                                                                            \
        if (AIR::COLUMNS && n_trace_cols != AIR::COLUMNS)                                                               \
            return set_err(ctx, ZK_ERR_BAD_ARG, "AIR %u expects %u trace columns, got %zu", air_id, (unsigned)AIR::COLUMNS, n_trace_cols); \
        return launch_quotient_air<AIR, HEAVY>(ctx, A, size, scratch, count);
#define ZK_AIR_CASE_CPU(ID, AIR)                                                                                          \
    case ID:                                                                                                            \
        if (n_trace_cols != AIR::COLUMNS)                                                                               \
            return set_err(ctx, ZK_ERR_BAD_ARG, "AIR %u expects %u trace columns, got %zu", air_id, (unsigned)AIR::COLUMNS, n_trace_cols); \
        if (n_air_consts != 4) return set_err(ctx, ZK_ERR_BAD_ARG, "the CPU AIR needs 4 kernel-label constants");       \
        return launch_quotient_air<AIR, false>(ctx, A, size, scratch, count);
