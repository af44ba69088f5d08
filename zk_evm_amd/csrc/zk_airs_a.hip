// Translation unit of libzkstark_hip.so: the quotient kernels of one group of table AIRs (airs.cuh), see internal.hpp.
#include "quotient_launch.hpp"
#include <cstdlib>
// ZK_ARITH_HEAVY=0 (tuning only): launch the Arithmetic quotient without the __launch_bounds__(256, 4) cap
static const bool kArithHeavy = !(getenv("ZK_ARITH_HEAVY") && getenv("ZK_ARITH_HEAVY")[0] == 0x30);
int zki_quotient_airs_a(zk_ctx *ctx, uint32_t air_id, const QuotientArgs &A, u32 size, const std::vector<u64> &shape_key,
                        DevBuf &scratch, size_t n_trace_cols, size_t n_air_consts) {
    (void)n_air_consts;
    switch (air_id) {
        ZK_AIR_CASE(ZK_AIR_NONE, AirNone, false)
        ZK_AIR_CASE(ZK_AIR_MEM_CONTINUATION, AirMemContinuation, false)
        ZK_AIR_CASE(ZK_AIR_ARITHMETIC, AirArithmetic, kArithHeavy)
        default: return ZK_AIR_NOT_MINE;
    }
}
