// Translation unit of libzkstark_hip.so: the quotient kernels of one group of table AIRs (airs.cuh), see internal.hpp.
#include "quotient_launch.hpp"

int zki_quotient_airs_d(zk_ctx *ctx, uint32_t air_id, const QuotientArgs &A, u32 size, DevBuf &scratch, size_t n_trace_cols,
                        size_t n_air_consts, u32 *count) {
    (void)n_air_consts;
    switch (air_id) {
        ZK_AIR_CASE(ZK_AIR_BYTE_PACKING, AirBytePacking, true)
        ZK_AIR_CASE(ZK_AIR_MEMORY, AirMemory, false)
        ZK_AIR_CASE(ZK_AIR_POSEIDON, AirPoseidon, false)
        default: return ZK_AIR_NOT_MINE;
    }
}
