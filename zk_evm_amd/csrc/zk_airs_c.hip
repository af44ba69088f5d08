// Translation unit of libzkstark_hip.so: the quotient kernels of one group of table AIRs (airs.cuh), see internal.hpp.
#include "quotient_launch.hpp"

int zki_quotient_airs_c(zk_ctx *ctx, uint32_t air_id, const QuotientArgs &A, u32 size, DevBuf &scratch, size_t n_trace_cols,
                        size_t n_air_consts, u32 *count) {
    (void)n_air_consts;
    switch (air_id) {
        ZK_AIR_CASE(ZK_AIR_KECCAK, AirKeccak, false)
        ZK_AIR_CASE(ZK_AIR_KECCAK_SPONGE, AirKeccakSponge, false)
        ZK_AIR_CASE(ZK_AIR_LOGIC, AirLogic, false)
        default: return ZK_AIR_NOT_MINE;
    }
}
