// Translation unit of libzkstark_hip.so: the quotient kernels of one group of table AIRs (airs.cuh), see internal.hpp.
#include "quotient_launch.hpp"
#include "arith_quotient.cuh"
#include <cstdlib>
// ZK_ARITH_TILED=0 (tuning / A-B only): the one-lane-per-point Arithmetic AIR kernel instead of the LDS-tiled one
static const bool kArithTiled = !(getenv("ZK_ARITH_TILED") && getenv("ZK_ARITH_TILED")[0] == 0x30);
int zki_quotient_airs_a(zk_ctx *ctx, uint32_t air_id, const QuotientArgs &A, u32 size, DevBuf &scratch, size_t n_trace_cols,
                        size_t n_air_consts, u32 *count) {
    (void)n_air_consts;
    switch (air_id) {
        ZK_AIR_CASE(ZK_AIR_NONE, AirNone, false)
        ZK_AIR_CASE(ZK_AIR_MEM_CONTINUATION, AirMemContinuation, false)
        case ZK_AIR_ARITHMETIC: {
            if (n_trace_cols != AirArithmetic::COLUMNS)
                return set_err(ctx, ZK_ERR_BAD_ARG, "AIR %u expects %u trace columns, got %zu", air_id, (unsigned)AirArithmetic::COLUMNS, n_trace_cols);
            if (count || !kArithTiled || A.sharded)   // the constraint count, ZK_ARITH_TILED=0, or a row shard (its rows are not
                                                      // consecutive points): the one-lane-per-point form
                return launch_quotient_air<AirArithmetic, true>(ctx, A, size, scratch, count);
            const size_t lds = (size_t)ZK_ARITH_LDS_WORDS * sizeof(u64);
            if (!(ctx->func_attrs & 2u)) {    // once per ctx = on every device a process drives
                HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&quotient_arith_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                ctx->func_attrs |= 2u;
            }
            quotient_arith_kernel<<<(size + ZK_ARITH_POINTS - 1) / ZK_ARITH_POINTS, 64 * ZK_ARITH_WAVES, lds, ctx->stream>>>(A);
            return check_launch(ctx, "quotient_arith_kernel");
        }
        default: return ZK_AIR_NOT_MINE;
    }
}
