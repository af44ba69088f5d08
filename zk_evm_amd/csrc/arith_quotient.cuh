// The ArithmeticStark AIR half of the quotient as an LDS-tiled kernel (replaces quotient_kernel_heavy<AirArithmetic>).
//
// Why: the AIR (arithmetic_stark.rs:203-252 and its operation modules) is a dozen 16/32-limb polynomial products over the
// same ~100 limb columns of rows i and i + 2.  One lane per point can neither keep them in registers (463 VGPRs) nor
// re-read them cheaply: the two LDE rows of a wave are 119 KB, past L1 and -- with every wave of the chip streaming its
// own -- past L2, so the one-lane form fetched 15x the table's bytes (r03a: 49 GB per launch, 171 spilled VGPRs under the
// 128-register cap, 13.8 ms).
//
// How: a workgroup of 8 waves owns 64 consecutive coset points.  It stages the 116 columns of the 64 + 2 rows those points
// touch into LDS ONCE (61 KB; row i + 2 of lane l is row i of lane l + 2), then each wave evaluates a different group of
// constraint families for all 64 points, reading its operands from the tile.  The alpha-weighted sum over constraints is
// additive, so every wave runs the dot-product consumer from the first position of its families (AirArithmetic::POS_*)
// and the eight partial sums are added through LDS.  Same field values at the same alpha powers as the one-lane form:
// bit-identical quotient.  HBM traffic: each column 66/64 times.
#pragma once
#include "quotient.cuh"
#include "airs.cuh"

#define ZK_ARITH_POINTS 64
#define ZK_ARITH_ROWS (ZK_ARITH_POINTS + 2)
#define ZK_ARITH_WAVES 8
#define ZK_ARITH_LDS_WORDS (AirArithmetic::COLUMNS * ZK_ARITH_ROWS + ZK_ARITH_WAVES * 2 * ZK_ARITH_POINTS + 4 * ZK_ARITH_POINTS)

// column c of the tile row `r0` (the point's own row, or the next row = r0 + 2^qd_bits)
struct ArithTileRow {
    const u64 *t;
    u32 r0;
    __device__ __forceinline__ Fe operator[](u32 col) const { return Fe(t[col * ZK_ARITH_ROWS + r0]); }
};

static __global__ void __launch_bounds__(64 * ZK_ARITH_WAVES, 4) quotient_arith_kernel(QuotientArgs A) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    u64 *tile = lds;                                                        // [116][66]
    u64 *part = lds + AirArithmetic::COLUMNS * ZK_ARITH_ROWS;               // [8][2][64]
    u64 *sel = part + ZK_ARITH_WAVES * 2 * ZK_ARITH_POINTS;                 // z_last, lagrange_first, lagrange_last, inv_zh
    const u32 size_log = A.log_n + A.qd_bits, size = 1u << size_log;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 i0 = blockIdx.x * ZK_ARITH_POINTS, i = i0 + lane;
    const u32 next = 1u << A.qd_bits;
    // ---- stage: tile[c][k] = column c at point (i0 + k) mod size, k < 64 + next ----
    const u32 rows = ZK_ARITH_POINTS + next;
    for (u32 e = tid; e < AirArithmetic::COLUMNS * rows; e += 64 * ZK_ARITH_WAVES) {
        const u32 c = e / rows, k = e - c * rows;
        const u32 row = ((i0 + k) & (size - 1)) << A.step_log;
        tile[c * ZK_ARITH_ROWS + k] = A.trace[(size_t)c * A.trace_stride + row];
    }
    if (wave == 0) {                                    // selectors of the 64 points, shared by the eight waves
        PointSetup P;
        point_setup(A, i & (size - 1), 0, P);
        sel[lane] = P.cons.z_last.v;
        sel[64 + lane] = P.cons.lagrange_first.v;
        sel[128 + lane] = P.cons.lagrange_last.v;
        sel[192 + lane] = P.inv_zh.v;
    }
    __syncthreads();
    const ArithTileRow lv{tile, lane}, nv{tile, lane + next};
    DotConsumer cons;
    dot_acc_init(cons.d0); dot_acc_init(cons.d1);
    cons.z_last = Fe(sel[lane]); cons.lagrange_first = Fe(sel[64 + lane]); cons.lagrange_last = Fe(sel[128 + lane]);
    auto seek = [&](u32 pos) { cons.ap0 = A.alpha_pow[0] + (A.n_constraints - pos); cons.ap1 = A.alpha_pow[1] + (A.n_constraints - pos); };
    typedef AirArithmetic R;
    switch (wave) {                                      // wave-uniform: each wave runs its own families
        case 0: seek(R::POS_HEAD); R::part_head(lv, nv, cons); seek(R::POS_BYTE_SHL); R::part_byte_shl(lv, nv, cons); break;
        case 1: seek(R::POS_DIV); R::divmod_helper(lv, nv, cons, lv[R::IS_DIV], R::IN0, R::IN1, R::OUT, R::AUX0); break;
        case 2: seek(R::POS_MOD); R::divmod_helper(lv, nv, cons, lv[R::IS_MOD], R::IN0, R::IN1, R::AUX0, R::OUT); break;
        case 3: seek(R::POS_SHR); R::divmod_helper(lv, nv, cons, lv[R::IS_SHR], R::IN1, R::IN2, R::OUT, R::AUX0); break;
        case 4: seek(R::POS_MODULAR_A); R::part_modular_a(lv, nv, cons); seek(R::POS_MODULAR_D); R::part_modular_d2(lv, nv, cons); break;
        case 5: seek(R::POS_MODULAR_B); R::part_modular_b(lv, nv, cons); break;
        case 6: seek(R::POS_MODULAR_C); R::part_modular_c(lv, nv, cons); break;
        default:
            seek(R::POS_MODULAR_D); R::part_modular_d1(lv, nv, cons);
            // one position check per launch: this family must end where BYTE begins
            if (i == 0 && cons.ap0 != A.alpha_pow[0] + (A.n_constraints - R::POS_BYTE_SHL)) atomicExch(A.err_flag, 3);
            break;
    }
    part[(wave * 2 + 0) * ZK_ARITH_POINTS + lane] = dot_acc_reduce(cons.d0);
    part[(wave * 2 + 1) * ZK_ARITH_POINTS + lane] = dot_acc_reduce(cons.d1);
    __syncthreads();
    if (wave == 0 && i < size) {
        u64 r0 = 0, r1 = 0;
#pragma unroll
        for (u32 w = 0; w < ZK_ARITH_WAVES; ++w) {
            r0 = gl_add(r0, part[(w * 2 + 0) * ZK_ARITH_POINTS + lane]);
            r1 = gl_add(r1, part[(w * 2 + 1) * ZK_ARITH_POINTS + lane]);
        }
        if (A.n_air_constraints == A.n_constraints) {   // no lookup / CTL checks follow: finish here
            const u64 inv_zh = sel[192 + lane];
            r0 = gl_canon(gl_mul(r0, inv_zh));
            r1 = gl_canon(gl_mul(r1, inv_zh));
        }
        A.out[i] = r0;
        if (A.n_challenges > 1) A.out[A.out_stride + i] = r1;
    }
}
