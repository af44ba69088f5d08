// Poseidon-12 over Goldilocks for gfx950: one sponge state per lane, all 12 words in VGPRs.
//
// Parameters are plonky2 1.0.0's ([EXT] plonky2/src/hash/poseidon.rs, poseidon_goldilocks.rs):
// 4 full + 22 partial + 4 full rounds, x^7 S-box, circulant MDS [17,15,41,16,2,28,13,13,39,18,34,20]
// + diag(8,0,..).  Reached in the reference through MerkleTree::new inside
// PolynomialBatch::from_values (evm_arithmetization/src/prover.rs:100) and through Challenger
// (prover.rs:118).
//
// MI355X mapping.  The permutation is integer-ALU bound, and on gfx950 every useful integer op
// (v_mad_u64_u32, carry adds, 64-bit shifts, cndmask) issues at ~4.3 cycles per wave64 per SIMD
// (profiles/archive/r01_ubench_valu_issue_rates.txt), so the design minimises *instruction count*:
//   * MDS layer: each state word is split into 32-bit halves; row r accumulates
//     sum_i C[i]*half[(i+r)%12] in one 64-bit register with 12 v_mad_u64_u32 (inline constants,
//     no per-term reduction: sums stay < 2^43).  The NEXT round's constant is the accumulator's
//     initial value (an SGPR pair from constant memory), so constant addition costs nothing.
//   * the two 43-bit sums are folded to one lazy u64 with 2 instructions on the common path (one mad by 2^32-1, one
//     carry add; the conditional +EPS fires for 2^-19 of the lanes and lives in an unlikely block) -- see pos_fold().
//   * the 23 linear layers between the two groups of full S-box layers (the last full round of the first half and the 22
//     partial rounds) are taken three at a time as ONE 12 x 12 product with the entries of MDS^3 (pos_block3, seven
//     blocks) and a final pair with MDS^2 (pos_partial_pair).
//   * an MFMA formulation was evaluated and rejected: the i8 matrix pipe could do the 12x12
//     byte-limb products, but re-laying 64-bit lane-private words out as MFMA operands and
//     recombining 8 i32 partial sums per word costs more VALU work than the 288 mads it replaces
//     ("Why not MFMA").
#pragma once
#include "gl.cuh"
#include "../../include/poseidon_constants.h"

// Round constants split into zero-extended 32-bit halves: ZK_RCS[round*12+i] = {lo32, hi32} as two
// u64, directly usable as the 64-bit addend of v_mad_u64_u32.
struct RcSplit { u64 lo, hi; };
static __constant__ u64 ZK_RC[ZK_POSEIDON_ROUNDS * ZK_POSEIDON_WIDTH] = ZK_POSEIDON_RC_INIT;
static __constant__ RcSplit ZK_RCS[ZK_POSEIDON_ROUNDS * ZK_POSEIDON_WIDTH] = ZK_POSEIDON_RCS_INIT;
// Two partial rounds as one linear step (see pos_partial_pair): M^2 = circ(ZK_M2C) away from row 0 / column 0, constants
// K_p = M rc' + rc'' of pair p split like ZK_RCS.
static __constant__ u32 ZK_M2C[12] = ZK_POSEIDON_M2_CIRC_INIT;
static __constant__ u32 ZK_M2ROW0[12] = ZK_POSEIDON_M2_ROW0_INIT;
static __constant__ u32 ZK_M2COL0[12] = ZK_POSEIDON_M2_COL0_INIT;
static __constant__ RcSplit ZK_RCS2[ZK_POSEIDON_PARTIAL_PAIRS * ZK_POSEIDON_WIDTH] = ZK_POSEIDON_RCS2_INIT;
// Three linear layers as one step (see pos_block3): M^3 (dense, entries < 2^21), K of block b at ZK_RCS3[12 b ..], KZ at ZK_RCS3Z[b].
static __constant__ u32 ZK_M3[144] = ZK_POSEIDON_M3_INIT;
static __constant__ RcSplit ZK_RCS3[ZK_POSEIDON_BLOCK3_COUNT * ZK_POSEIDON_WIDTH] = ZK_POSEIDON_RCS3_INIT;
static __constant__ RcSplit ZK_RCS3Z[ZK_POSEIDON_BLOCK3_COUNT] = ZK_POSEIDON_RCS3Z_INIT;

__device__ __forceinline__ u64 pos_sbox(u64 x) {
    u64 x2 = gl_mul_fast(x, x);
    u64 x4 = gl_mul_fast(x2, x2);
    u64 x3 = gl_mul_fast(x, x2);
    return gl_mul_fast(x3, x4);
}

// acc += x * C  (one v_mad_u64_u32)
template <u32 C>
__device__ __forceinline__ void pos_mac(u64 &acc, u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "n"(C) : "vcc");
#else        // the same integers in portable C++: the host pass of hipcc (parsed, never run) and the CPU emulation (tests/emu/)
    acc += (u64)x * C;
#endif
}

// One MDS row for both 32-bit halves as ONE asm statement (24 v_mad_u64_u32 with inline
// constants; hipcc would otherwise strength-reduce x*16, x*2 .. into shift/zero-extend/add chains
// and pad every single-instruction asm with s_nop).  x[i] / y[i] are the low / high halves of
// state word (i + r) % 12 -- the rotation is done by operand binding at the call site.
// Operands: %0 al, %1 ah, %2/%3 initial accumulators (SGPR pairs: next round's constant halves),
// %4..%15 x0..x11, %16..%27 y0..y11.
#define POS_MAC2_FIRST(XI, YI, C)                                   \
    "v_mad_u64_u32 %0, vcc, %" #XI ", " #C ", %2\n\t"               \
    "v_mad_u64_u32 %1, vcc, %" #YI ", " #C ", %3\n\t"
#define POS_MAC2_FIRST0(XI, YI, C)                                  \
    "v_mad_u64_u32 %0, vcc, %" #XI ", " #C ", 0\n\t"                \
    "v_mad_u64_u32 %1, vcc, %" #YI ", " #C ", 0\n\t"
#define POS_MAC2(XI, YI, C)                                         \
    "v_mad_u64_u32 %0, vcc, %" #XI ", " #C ", %0\n\t"               \
    "v_mad_u64_u32 %1, vcc, %" #YI ", " #C ", %1\n\t"
// MDS_MATRIX_CIRC = 17 15 41 16 2 28 13 13 39 18 34 20 (static_assert'ed below)
#define POS_ROW_TAIL                                                               \
    POS_MAC2(5, 17, 15) POS_MAC2(6, 18, 41) POS_MAC2(7, 19, 16) POS_MAC2(8, 20, 2) \
    POS_MAC2(9, 21, 28) POS_MAC2(10, 22, 13) POS_MAC2(11, 23, 13)                  \
    POS_MAC2(12, 24, 39) POS_MAC2(13, 25, 18) POS_MAC2(14, 26, 34) POS_MAC2(15, 27, 20)

template <bool HAS_RC>
__device__ __forceinline__ void pos_row(u64 &al, u64 &ah, u64 rcl, u64 rch, const u32 (&lo)[12],
                                        const u32 (&hi)[12], int r) {
#if !defined(__HIP_DEVICE_COMPILE__)
    constexpr u32 C_[12] = ZK_POSEIDON_MDS_CIRC_INIT;
    al = HAS_RC ? rcl : 0; ah = HAS_RC ? rch : 0;
    for (int i = 0; i < 12; ++i) { al += (u64)lo[(i + r) % 12] * C_[i]; ah += (u64)hi[(i + r) % 12] * C_[i]; }
#else
#define X(i) "v"(lo[((i) + r) % 12])
#define Y(i) "v"(hi[((i) + r) % 12])
    if (HAS_RC) {
        asm(POS_MAC2_FIRST(4, 16, 17) POS_ROW_TAIL
            : "=&v"(al), "=&v"(ah)
            : "s"(rcl), "s"(rch), X(0), X(1), X(2), X(3), X(4), X(5), X(6), X(7), X(8), X(9), X(10),
              X(11), Y(0), Y(1), Y(2), Y(3), Y(4), Y(5), Y(6), Y(7), Y(8), Y(9), Y(10), Y(11)
            : "vcc");
    } else {
        asm(POS_MAC2_FIRST0(4, 16, 17) POS_ROW_TAIL
            : "=&v"(al), "=&v"(ah)
            : "s"(rcl), "s"(rch), X(0), X(1), X(2), X(3), X(4), X(5), X(6), X(7), X(8), X(9), X(10),
              X(11), Y(0), Y(1), Y(2), Y(3), Y(4), Y(5), Y(6), Y(7), Y(8), Y(9), Y(10), Y(11)
            : "vcc");
    }
#undef X
#undef Y
#endif
}

// value = al + ah * 2^32  (al, ah < 2^50: < 2^44 from a plain MDS row, < 2^50 from pos_partial_pair's M^2 row plus the
// delta term)  ->  lazy u64 representative
__device__ __forceinline__ u64 pos_fold(u64 al, u64 ah) {
    u32 ah0 = (u32)ah, ah1 = (u32)(ah >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    // ah1 * 2^64 == ah1 * (2^32 - 1) with ah1 < 2^18; al + that < 2^51: no overflow
    asm("v_mad_u64_u32 %0, vcc, %1, -1, %0" : "+v"(al) : "v"(ah1) : "vcc");
    u32 l = (u32)al, h = (u32)(al >> 32), e;
    // h += ah0; on carry-out add 2^64 == EPS (cannot carry twice: the wrapped h is < 2^19).  h < 2^19 before the add, so
    // the carry needs ah0 > 2^32 - 2^19: probability <= 2^-13 per lane (2^-19 for the plain rows) -- the correction is an unlikely block entered when
    // some lane of the wave carried (test and branch on the scalar unit), and the fold is two VALU instructions.
    u64 cm;
    asm("v_add_co_u32 %0, %1, %0, %2" : "+v"(h), "=&s"(cm) : "v"(ah0));
    if (__builtin_expect(cm != 0, 0)) {
        asm("v_cndmask_b32_e64 %2, 0, -1, %3\n\t"
            "v_add_co_u32 %0, vcc, %0, %2\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "+v"(l), "+v"(h), "=&v"(e)
            : "s"(cm)
            : "vcc");
    }
    return ((u64)h << 32) | l;
#else        // portable: the same 64-bit words, lane by lane
    al += (u64)ah1 * 0xFFFFFFFFu;
    u64 t = al + ((u64)ah0 << 32);                     // h += ah0 ...
    if (t < al) t += 0xFFFFFFFFu;                      // ... a carry out is 2^64 == EPS
    return t;
#endif
}

// s <- MDS * s + rc   (rc = constants of the following round, or nothing when HAS_RC is false)
template <bool HAS_RC>
__device__ __forceinline__ void pos_mds(u64 (&s)[12], const RcSplit *rc) {
    constexpr u32 C[12] = ZK_POSEIDON_MDS_CIRC_INIT;
    static_assert(C[0] == 17 && C[1] == 15 && C[2] == 41 && C[3] == 16 && C[4] == 2 && C[5] == 28 &&
                  C[6] == 13 && C[7] == 13 && C[8] == 39 && C[9] == 18 && C[10] == 34 && C[11] == 20,
                  "POS_ROW asm hard-codes MDS_MATRIX_CIRC");
    u32 lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32); }
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        u64 al, ah;
        pos_row<HAS_RC>(al, ah, HAS_RC ? rc[r].lo : 0, HAS_RC ? rc[r].hi : 0, lo, hi, r);
        if (r == 0) {  // MDS_MATRIX_DIAG = (8, 0, .., 0)
            pos_mac<8>(al, lo[0]);
            pos_mac<8>(ah, hi[0]);
        }
        s[r] = pos_fold(al, ah);
    }
}

// ---- two partial rounds in one linear step -------------------------------------------------------------
// In a partial round only element 0 goes through the S-box, so with x = the state after that S-box, M = the MDS matrix and
// rc', rc'' the constants of the next two rounds,
//     w0    = (M x)_0 + rc'_0                       (the only entry of the intermediate state that is needed)
//     delta = sbox(w0) - w0
//     v''   = M^2 x + (M rc' + rc'') + delta M[:, 0]
// is the state two rounds later (constants of the round after included, as everywhere in this file): one MDS row, two
// S-boxes and ONE 12 x 12 product with the entries of M^2 (< 2^13: the 32-bit halves still sum to < 2^50 in a 64-bit
// accumulator) plus a 13th term per row, instead of two full MDS products: 24 + 312 mads per pair instead of 576, the
// same field values.  The 22 partial rounds are 11 such pairs; ~15 % of the permutation's instructions.
// kinit + d * CD + sum_j x[j] * k[j]   (kinit, k: wave-uniform constants in SGPRs, CD inline).  The delta term comes FIRST:
// its multiplier is an inline constant, so that mad can take the SGPR pair kinit as its addend (one constant-bus operand
// per instruction) and the accumulator needs no v_mov initialisation.
template <u32 CD>
__device__ __forceinline__ u64 pos_row2_half(u64 kinit, const u32 (&x)[12], const u32 (&k)[12], u32 d) {
    u64 acc;
#if !defined(__HIP_DEVICE_COMPILE__)
    acc = kinit + (u64)d * CD;
    for (int j = 0; j < 12; ++j) acc += (u64)x[j] * k[j];
#else
    asm("v_mad_u64_u32 %0, vcc, %25, %26, %27\n\t"
        "v_mad_u64_u32 %0, vcc, %1, %13, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %2, %14, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %15, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %4, %16, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %17, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %18, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %19, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %8, %20, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %21, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %22, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %11, %23, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %12, %24, %0"
        : "=&v"(acc)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]),
          "v"(x[11]), "s"(k[0]), "s"(k[1]), "s"(k[2]), "s"(k[3]), "s"(k[4]), "s"(k[5]), "s"(k[6]), "s"(k[7]), "s"(k[8]), "s"(k[9]),
          "s"(k[10]), "s"(k[11]), "v"(d), "n"(CD), "s"(kinit)
        : "vcc");
#endif
    return acc;
}
template <int R>
__device__ __forceinline__ u64 pos_m2_row(const u32 (&lo)[12], const u32 (&hi)[12], u32 dl, u32 dh, const RcSplit *kp) {
    constexpr u32 C[12] = ZK_POSEIDON_MDS_CIRC_INIT;
    constexpr u32 CD = C[(12 - R) % 12] + (R == 0 ? 8 : 0);            // M[R][0]
    u32 k[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) k[j] = R == 0 ? ZK_M2ROW0[j] : j == 0 ? ZK_M2COL0[R] : ZK_M2C[(j - R + 12) % 12];
    const u64 al = pos_row2_half<CD>(kp[R].lo, lo, k, dl);
    const u64 ah = pos_row2_half<CD>(kp[R].hi, hi, k, dh);
    return pos_fold(al, ah);
}
// s (constants of round `round` included, `round` a partial round; its S-box already applied unless LEAD) -> the state two
// rounds later; pair = (round - 4) / 2
template <bool LEAD>
__device__ __forceinline__ void pos_partial_pair(u64 (&s)[12], int round, int pair) {
    if (LEAD) s[0] = pos_sbox(s[0]);
    u32 lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32); }
    const RcSplit *rc1 = &ZK_RCS[(round + 1) * 12];
    u64 al, ah;
    pos_row<true>(al, ah, rc1[0].lo, rc1[0].hi, lo, hi, 0);
    pos_mac<8>(al, lo[0]);
    pos_mac<8>(ah, hi[0]);
    const u64 w0 = pos_fold(al, ah);
    const u64 delta = gl_sub(pos_sbox(w0), w0);
    const u32 dl = (u32)delta, dh = (u32)(delta >> 32);
    const RcSplit *kp = &ZK_RCS2[pair * 12];
    s[0] = pos_m2_row<0>(lo, hi, dl, dh, kp);
    s[1] = pos_m2_row<1>(lo, hi, dl, dh, kp);
    s[2] = pos_m2_row<2>(lo, hi, dl, dh, kp);
    s[3] = pos_m2_row<3>(lo, hi, dl, dh, kp);
    s[4] = pos_m2_row<4>(lo, hi, dl, dh, kp);
    s[5] = pos_m2_row<5>(lo, hi, dl, dh, kp);
    s[6] = pos_m2_row<6>(lo, hi, dl, dh, kp);
    s[7] = pos_m2_row<7>(lo, hi, dl, dh, kp);
    s[8] = pos_m2_row<8>(lo, hi, dl, dh, kp);
    s[9] = pos_m2_row<9>(lo, hi, dl, dh, kp);
    s[10] = pos_m2_row<10>(lo, hi, dl, dh, kp);
    s[11] = pos_m2_row<11>(lo, hi, dl, dh, kp);
}

// ---- three linear layers in one step ---------------------------------------------------------------------
// The same idea one layer further.  With x = the state after the S-box layer of round r (a full layer for r = 3, element 0
// only afterwards), rounds r + 1 and r + 2 partial:
//     w0 = (M x)_0 + rc[r+1]_0,                  d1 = sbox(w0) - w0
//     z0 = (M^2 x)_0 + KZ + M[0][0] d1,          d2 = sbox(z0) - z0
//     out = M^3 x + K + d1 M^2[:, 0] + d2 M[:, 0]
// (KZ, K: tools/gen_poseidon_constants.py, which also checks this schedule against the plain rounds).  The entries of M^3
// are < 2^21, so a row's two half sums stay < 2^57 in their 64-bit accumulators: 26 + 26 + 12 * 28 = 388 mads for three
// layers where pairs need 507 and plain MDS products 870.  The wider sums make the fold's carry common (2^-6 per lane), so
// these rows use the branch-free four-instruction pos_fold_wide.
__device__ __forceinline__ u64 pos_fold_wide(u64 al, u64 ah) {
    const u32 ah0 = (u32)ah, ah1 = (u32)(ah >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    // ah1 * 2^64 == ah1 * (2^32 - 1), ah1 < 2^25: al + that < 2^58
    asm("v_mad_u64_u32 %0, vcc, %1, -1, %0" : "+v"(al) : "v"(ah1) : "vcc");
    u32 l = (u32)al, h = (u32)(al >> 32), c;
    // h += ah0; a carry out is 2^64 == 2^32 - 1, added by one more mad (the wrapped h is < 2^26: no second carry)
    asm("v_add_co_u32 %0, vcc, %0, %2\n\t"
        "v_cndmask_b32_e64 %1, 0, 1, vcc"
        : "+v"(h), "=&v"(c)
        : "v"(ah0)
        : "vcc");
    u64 t = ((u64)h << 32) | l;
    asm("v_mad_u64_u32 %0, vcc, %1, -1, %0" : "+v"(t) : "v"(c) : "vcc");
    return t;
#else
    al += (u64)ah1 * 0xFFFFFFFFu;
    u64 t = al + ((u64)ah0 << 32);
    if (t < al) t += 0xFFFFFFFFu;
    return t;
#endif
}
// kinit + d2 * CD + d1 * kc + sum_j x[j] * k[j]   (kinit, k, kc: wave-uniform constants in SGPRs, CD inline; d2 * CD first,
// see pos_row2_half)
template <u32 CD>
__device__ __forceinline__ u64 pos_row3_half(u64 kinit, const u32 (&x)[12], const u32 (&k)[12], u32 d1, u32 kc, u32 d2) {
    u64 acc;
#if !defined(__HIP_DEVICE_COMPILE__)
    acc = kinit + (u64)d2 * CD + (u64)d1 * kc;
    for (int j = 0; j < 12; ++j) acc += (u64)x[j] * k[j];
#else
    asm("v_mad_u64_u32 %0, vcc, %27, %28, %29\n\t"
        "v_mad_u64_u32 %0, vcc, %25, %26, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %1, %13, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %2, %14, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %15, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %4, %16, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %17, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %18, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %19, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %8, %20, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %21, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %22, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %11, %23, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %12, %24, %0"
        : "=&v"(acc)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]),
          "v"(x[11]), "s"(k[0]), "s"(k[1]), "s"(k[2]), "s"(k[3]), "s"(k[4]), "s"(k[5]), "s"(k[6]), "s"(k[7]), "s"(k[8]), "s"(k[9]),
          "s"(k[10]), "s"(k[11]), "v"(d1), "s"(kc), "v"(d2), "n"(CD), "s"(kinit)
        : "vcc");
#endif
    return acc;
}
template <int R>
__device__ __forceinline__ u64 pos_m3_row(const u32 (&lo)[12], const u32 (&hi)[12], u32 d1l, u32 d1h, u32 d2l, u32 d2h,
                                          const RcSplit *kp) {
    constexpr u32 C[12] = ZK_POSEIDON_MDS_CIRC_INIT;
    constexpr u32 CD = C[(12 - R) % 12] + (R == 0 ? 8 : 0);            // M[R][0]
    u32 k[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) k[j] = ZK_M3[R * 12 + j];
    const u32 kc = ZK_M2COL0[R];                                       // M^2[R][0]
    const u64 al = pos_row3_half<CD>(kp[R].lo, lo, k, d1l, kc, d2l);
    const u64 ah = pos_row3_half<CD>(kp[R].hi, hi, k, d1h, kc, d2h);
    return pos_fold_wide(al, ah);
}
// s = the state after the S-box layer of round r = 3 + 3 blk  ->  the state entering round r + 3 (constants included)
__device__ __forceinline__ void pos_block3(u64 (&s)[12], int blk) {
    constexpr u32 C[12] = ZK_POSEIDON_MDS_CIRC_INIT;
    const int r = 3 + 3 * blk;
    u32 lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32); }
    const RcSplit *rc1 = &ZK_RCS[(r + 1) * 12];
    u64 al, ah;
    pos_row<true>(al, ah, rc1[0].lo, rc1[0].hi, lo, hi, 0);
    pos_mac<8>(al, lo[0]);
    pos_mac<8>(ah, hi[0]);
    const u64 w0 = pos_fold(al, ah);
    const u64 d1 = gl_sub(pos_sbox(w0), w0);
    const u32 d1l = (u32)d1, d1h = (u32)(d1 >> 32);
    u32 k0[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) k0[j] = ZK_M2ROW0[j];
    al = pos_row2_half<C[0] + 8>(ZK_RCS3Z[blk].lo, lo, k0, d1l);       // sums < 2^49: the narrow fold applies
    ah = pos_row2_half<C[0] + 8>(ZK_RCS3Z[blk].hi, hi, k0, d1h);
    const u64 z0 = pos_fold(al, ah);
    const u64 d2 = gl_sub(pos_sbox(z0), z0);
    const u32 d2l = (u32)d2, d2h = (u32)(d2 >> 32);
    const RcSplit *kp = &ZK_RCS3[blk * 12];
    s[0] = pos_m3_row<0>(lo, hi, d1l, d1h, d2l, d2h, kp);
    s[1] = pos_m3_row<1>(lo, hi, d1l, d1h, d2l, d2h, kp);
    s[2] = pos_m3_row<2>(lo, hi, d1l, d1h, d2l, d2h, kp);
    s[3] = pos_m3_row<3>(lo, hi, d1l, d1h, d2l, d2h, kp);
    s[4] = pos_m3_row<4>(lo, hi, d1l, d1h, d2l, d2h, kp);
    s[5] = pos_m3_row<5>(lo, hi, d1l, d1h, d2l, d2h, kp);
    s[6] = pos_m3_row<6>(lo, hi, d1l, d1h, d2l, d2h, kp);
    s[7] = pos_m3_row<7>(lo, hi, d1l, d1h, d2l, d2h, kp);
    s[8] = pos_m3_row<8>(lo, hi, d1l, d1h, d2l, d2h, kp);
    s[9] = pos_m3_row<9>(lo, hi, d1l, d1h, d2l, d2h, kp);
    s[10] = pos_m3_row<10>(lo, hi, d1l, d1h, d2l, d2h, kp);
    s[11] = pos_m3_row<11>(lo, hi, d1l, d1h, d2l, d2h, kp);
}

// In/out: arbitrary u64 representatives; callers canonicalise what they emit.
__device__ __forceinline__ void poseidon_permute(u64 (&s)[12]) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = gl_add_canon(s[i], ZK_RC[i]);
    int round = 0;
#pragma unroll 1
    for (int k = 0; k < ZK_POSEIDON_HALF_FULL_ROUNDS - 1; ++k) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s[i] = pos_sbox(s[i]);
        ++round;
        pos_mds<true>(s, &ZK_RCS[round * 12]);
    }
    // linear layers of rounds 3 .. 23 in seven blocks of three, the S-box of rounds 6, 9, .., 24 between them
    static_assert(ZK_POSEIDON_HALF_FULL_ROUNDS - 1 + 3 * ZK_POSEIDON_BLOCK3_COUNT + 2 ==
                  ZK_POSEIDON_HALF_FULL_ROUNDS + ZK_POSEIDON_PARTIAL_ROUNDS, "seven blocks of three layers and one pair");
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = pos_sbox(s[i]);
#pragma unroll 1
    for (int b = 0; b < ZK_POSEIDON_BLOCK3_COUNT; ++b) {
        pos_block3(s, b);
        s[0] = pos_sbox(s[0]);
    }
    round = ZK_POSEIDON_HALF_FULL_ROUNDS - 1 + 3 * ZK_POSEIDON_BLOCK3_COUNT;           // 24
    pos_partial_pair<false>(s, round, (round - ZK_POSEIDON_HALF_FULL_ROUNDS) / 2);
    round += 2;
#pragma unroll 1
    for (int k = 0; k < ZK_POSEIDON_HALF_FULL_ROUNDS - 1; ++k) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s[i] = pos_sbox(s[i]);
        ++round;
        pos_mds<true>(s, &ZK_RCS[round * 12]);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = pos_sbox(s[i]);
    pos_mds<false>(s, nullptr);
}


// ---- lane-cooperative permutation ---------------------------------------------------------------------
// For the top of a Merkle tree there are fewer states than lanes, and a level costs one full permutation latency
// (20.5 k dependent-issue instructions, ~37 us) however few nodes it has.  Here ONE state is spread over a group of 16
// lanes (element e = lane % 16, 12 active): every lane runs one S-box and one MDS row, the row's twelve inputs arrive
// through ds_bpermute ("wavefront shuffles for the round state"), ~140 instructions per round instead of ~700, so a
// level's latency drops ~5x.  Throughput per lane is 16x worse: only used where lanes would idle anyway.
__device__ __forceinline__ u64 poseidon_permute_coop(u64 s, u32 e, u32 lane) {
    constexpr u32 C[12] = ZK_POSEIDON_MDS_CIRC_INIT;
    const u32 gbase = lane & ~15u;
    const bool act = e < 12;
    const u32 ec = act ? e : 0;
    s = gl_add_canon(s, act ? ZK_RC[ec] : 0);
    int round = 0;
    auto mds = [&](bool has_rc) {
        const u32 lo = (u32)s, hi = (u32)(s >> 32);
        u64 al = has_rc ? ZK_RCS[round * 12 + ec].lo : 0, ah = has_rc ? ZK_RCS[round * 12 + ec].hi : 0;
#pragma unroll
        for (u32 j = 0; j < 12; ++j) {
            const u32 src = gbase + (ec + j >= 12 ? ec + j - 12 : ec + j);
            const u32 xl = (u32)__shfl((int)lo, (int)src, 64), xh = (u32)__shfl((int)hi, (int)src, 64);
            al += (u64)xl * C[j];
            ah += (u64)xh * C[j];
        }
        if (e == 0) { al += (u64)lo * 8; ah += (u64)hi * 8; }      // MDS_MATRIX_DIAG = (8, 0, .., 0)
        s = pos_fold(al, ah);
    };
    for (int k = 0; k < ZK_POSEIDON_HALF_FULL_ROUNDS; ++k) { s = pos_sbox(s); ++round; mds(true); }
    for (int k = 0; k < ZK_POSEIDON_PARTIAL_ROUNDS; ++k) { if (e == 0) s = pos_sbox(s); ++round; mds(true); }
    for (int k = 0; k < ZK_POSEIDON_HALF_FULL_ROUNDS - 1; ++k) { s = pos_sbox(s); ++round; mds(true); }
    s = pos_sbox(s);
    mds(false);
    return s;
}
