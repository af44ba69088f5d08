// Translation unit of libzkstark_hip.so: the quotient kernels of one group of table AIRs (airs.cuh), see internal.hpp.
#include "quotient_launch.hpp"
#include <cstdlib>
// ZK_ARITH_HEAVY=0 (tuning only): launch the Arithmetic quotient without the __launch_bounds__(256, 4) cap
static const bool kArithHeavy = !(getenv("ZK_ARITH_HEAVY") && getenv("ZK_ARITH_HEAVY")[0] == 0x30);
int zki_quotient_airs_a(zk_ctx *ctx, uint32_t air_id, const QuotientArgs &A, u32 size, DevBuf &scratch, size_t n_trace_cols,
                        size_t n_air_consts, u32 *count) {
    (void)n_air_consts;
    switch (air_id) {
        ZK_AIR_CASE(ZK_AIR_NONE, AirNone, false)
        ZK_AIR_CASE(ZK_AIR_MEM_CONTINUATION, AirMemContinuation, false)
        case ZK_AIR_ARITHMETIC:
            if (n_trace_cols != AirArithmetic::COLUMNS)
                return set_err(ctx, ZK_ERR_BAD_ARG, "AIR %u expects %u trace columns, got %zu", air_id, (unsigned)AirArithmetic::COLUMNS, n_trace_cols);
            return kArithHeavy ? launch_quotient_air<AirArithmetic, true>(ctx, A, size, scratch, count)
                               : launch_quotient_air<AirArithmetic, false>(ctx, A, size, scratch, count);
        default: return ZK_AIR_NOT_MINE;
    }
}
