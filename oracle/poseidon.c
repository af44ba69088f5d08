/*
 * oracle/poseidon.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * Restatement of plonky2 1.0.0 Poseidon over Goldilocks ([EXT] plonky2/src/hash/poseidon.rs:
 * `Poseidon::poseidon` = full_rounds(4), partial_rounds(22), full_rounds(4); `constant_layer`,
 * `sbox_layer` (x^7), `mds_layer` via `mds_row_shf`; plonky2/src/hash/hashing.rs:
 * `hash_n_to_m_no_pad` (rate 8, OVERWRITE absorption), `compress` (two_to_one);
 * plonky2/src/hash/poseidon.rs `PoseidonHash::hash_or_noop` via hash_types `from_partial`).
 * The crate is not vendored; this follows the *naive* (non-"fast partial round") definition, which
 * upstream asserts to be equivalent (`partial_rounds_naive` test).
 *
 * Pinned by reference-tree KATs: smt_trie/src/keys.rs:10-15 (HASH_ZEROS),
 * evm_arithmetization/src/proof.rs:505-510 (EMPTY_CONSOLIDATED_BLOCKHASH),
 * smt_trie/src/code.rs:56-84 (hash_contract_bytecode) -- see tests/test_oracle_kat.py.
 */
#include "goldilocks.h"
#include "../include/poseidon_constants.h"
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

static const uint64_t RC[ZK_POSEIDON_ROUNDS * ZK_POSEIDON_WIDTH] = ZK_POSEIDON_RC_INIT;
static const uint64_t MDS_CIRC[12] = ZK_POSEIDON_MDS_CIRC_INIT;
static const uint64_t MDS_DIAG[12] = ZK_POSEIDON_MDS_DIAG_INIT;

static inline uint64_t sbox7(uint64_t x) {
    uint64_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2);
    return gl_mul(x3, x4);
}

/* [EXT] poseidon.rs `mds_row_shf`: res = sum_i v[(i+r)%12]*CIRC[i] + v[r]*DIAG[r].
 * The constants are < 2^6, so the two 32-bit halves of the state words are accumulated separately in 64-bit integers
 * (12 * 41 * 2^32 < 2^42: no overflow) and recombined once: value = al + ah * 2^32, reduced with 2^64 = 2^32 - 1.
 * (The direct u128 form, `acc += (u128)st * C; out = acc % p`, made a permutation 3x slower and with it the bench's
 * CPU baseline; tests/test_oracle_kat.py pins this one to the reference's known answers all the same.) */
static void mds_layer(uint64_t st[12]) {
    uint64_t lo[24], hi[24], out[12];
    for (int i = 0; i < 12; ++i) {
        lo[i] = lo[i + 12] = st[i] & 0xFFFFFFFFULL;
        hi[i] = hi[i + 12] = st[i] >> 32;
    }
    for (int r = 0; r < 12; ++r) {
        uint64_t al = lo[r] * MDS_DIAG[r], ah = hi[r] * MDS_DIAG[r];
        for (int i = 0; i < 12; ++i) {
            al += lo[r + i] * MDS_CIRC[i];
            ah += hi[r + i] * MDS_CIRC[i];
        }
        u128 v = (u128)al + ((u128)ah << 32);
        out[r] = gl_reduce128(v);
    }
    memcpy(st, out, sizeof out);
}

void orc_poseidon_permute(uint64_t st[12]) {
    for (int i = 0; i < 12; ++i) st[i] = gl_canon(st[i]);
    int round = 0;
    for (int phase = 0; phase < 3; ++phase) {
        int n = phase == 1 ? ZK_POSEIDON_PARTIAL_ROUNDS : ZK_POSEIDON_HALF_FULL_ROUNDS;
        for (int k = 0; k < n; ++k, ++round) {
            for (int i = 0; i < 12; ++i) st[i] = gl_add(st[i], RC[round * 12 + i]);
            if (phase == 1) st[0] = sbox7(st[0]);
            else for (int i = 0; i < 12; ++i) st[i] = sbox7(st[i]);
            mds_layer(st);
        }
    }
}

/* Which evaluation of the permutation the sponge / compression functions below use: the blocked one of
 * poseidon_fast.c (default; it checks itself against `orc_poseidon_permute` at first use) or the plain one above
 * (ORACLE_PLAIN_POSEIDON=1 in the environment, or orc_poseidon_use_fast(0)).  Same outputs either way. */
static int use_fast = -1;
void orc_poseidon_use_fast(int on) { use_fast = on ? 1 : 0; }
static inline void permute(uint64_t st[12]) {
    if (use_fast < 0) { const char *e = getenv("ORACLE_PLAIN_POSEIDON"); use_fast = !(e && e[0] == '1'); }
    if (use_fast) orc_poseidon_permute_fast(st);
    else orc_poseidon_permute(st);
}

/* the permutation as the hashes above evaluate it (fri.c: the proof-of-work grind) */
void orc_poseidon_permute_auto(uint64_t st[12]) { permute(st); }

/* [EXT] hashing.rs `hash_n_to_m_no_pad` with m = 4. */
void orc_poseidon_hash_no_pad(const uint64_t *in, size_t n, uint64_t out[4]) {
    uint64_t st[12] = {0};
    for (size_t off = 0; off < n; off += 8) {
        size_t len = n - off < 8 ? n - off : 8;
        for (size_t i = 0; i < len; ++i) st[i] = gl_canon(in[off + i]);
        permute(st);
    }
    /* n == 0: no permutation at all, output = zeros (matches upstream loop structure). */
    memcpy(out, st, 4 * sizeof(uint64_t));
}

/* [EXT] `hash_or_noop`: <= 4 elements are copied (zero padded), not hashed. */
void orc_poseidon_hash_or_noop(const uint64_t *in, size_t n, uint64_t out[4]) {
    if (n <= 4) {
        for (size_t i = 0; i < 4; ++i) out[i] = i < n ? gl_canon(in[i]) : 0;
    } else {
        orc_poseidon_hash_no_pad(in, n, out);
    }
}

/* [EXT] hashing.rs `compress`. */
void orc_poseidon_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
    uint64_t st[12] = {0};
    memcpy(st, l, 32);
    memcpy(st + 4, r, 32);
    permute(st);
    memcpy(out, st, 32);
}
