// Poseidon-12 over Goldilocks for gfx950: one sponge state per lane, all 12 words in VGPRs.
//
// Parameters are plonky2 1.0.0's ([EXT] plonky2/src/hash/poseidon.rs, poseidon_goldilocks.rs):
// 4 full + 22 partial + 4 full rounds, x^7 S-box, circulant MDS [17,15,41,16,2,28,13,13,39,18,34,20]
// + diag(8,0,..).  Reached in the reference through MerkleTree::new inside
// PolynomialBatch::from_values (evm_arithmetization/src/prover.rs:100) and through Challenger
// (prover.rs:118).
//
// MI355X mapping: the permutation is integer-ALU bound (no HBM traffic beyond the absorbed
// words).  The MDS layer exploits the <= 6-bit matrix entries: each state word is split into two
// 32-bit halves, each half-row is a chain of 12 v_mad_u64_u32 with an inline constant (no 128-bit
// products, no modular reduction inside the sum), and the two 41-bit sums are folded once.
#pragma once
#include "gl.cuh"
#include "../../include/poseidon_constants.h"

__constant__ u64 ZK_RC[ZK_POSEIDON_ROUNDS * ZK_POSEIDON_WIDTH] = ZK_POSEIDON_RC_INIT;

__device__ __forceinline__ u64 pos_sbox(u64 x) {
    u64 x2 = gl_sqr(x);
    u64 x4 = gl_sqr(x2);
    u64 x3 = gl_mul(x, x2);
    return gl_mul(x3, x4);
}

__device__ __forceinline__ void pos_mds(u64 (&s)[12]) {
    constexpr u32 C[12] = ZK_POSEIDON_MDS_CIRC_INIT;
    u32 lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32); }
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        u64 al = 0, ah = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            al += (u64)lo[(i + r) % 12] * C[i];
            ah += (u64)hi[(i + r) % 12] * C[i];
        }
        if (r == 0) { al += (u64)lo[0] * 8u; ah += (u64)hi[0] * 8u; }  // MDS_MATRIX_DIAG
        // value = al + ah * 2^32, al, ah < 2^41
        u64 t = al + (ah << 32);             // low 64 bits of al + (ah_lo32 << 32)
        u64 carry = t < al ? 1 : 0;
        u32 top = (u32)(ah >> 32) + (u32)carry;   // coefficient of 2^64 (< 2^10)
        s[r] = gl_reduce96(top, t);
    }
}

template <bool FULL>
__device__ __forceinline__ void pos_round(u64 (&s)[12], int round) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = gl_add_canon(s[i], ZK_RC[round * 12 + i]);
    if (FULL) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s[i] = pos_sbox(s[i]);
    } else {
        s[0] = pos_sbox(s[0]);
    }
    pos_mds(s);
}

// In/out: arbitrary u64 representatives; callers canonicalise what they emit.
__device__ __forceinline__ void poseidon_permute(u64 (&s)[12]) {
    int round = 0;
#pragma unroll 1
    for (int k = 0; k < ZK_POSEIDON_HALF_FULL_ROUNDS; ++k) pos_round<true>(s, round++);
#pragma unroll 1
    for (int k = 0; k < ZK_POSEIDON_PARTIAL_ROUNDS; ++k) pos_round<false>(s, round++);
#pragma unroll 1
    for (int k = 0; k < ZK_POSEIDON_HALF_FULL_ROUNDS; ++k) pos_round<true>(s, round++);
}
