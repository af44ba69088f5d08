/*
 * oracle/challenger.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * Restatement of plonky2 1.0.0 `Challenger<F, H>` ([EXT] plonky2/src/iop/challenger.rs): duplex
 * sponge, rate 8 / width 12, OVERWRITE absorption; observing clears the output buffer; a duplex
 * runs when 8 inputs are buffered or on demand; `get_challenge` pops from the END of the output
 * buffer; `compact()` flushes pending inputs, clears outputs and returns the state.
 * Permutations: Poseidon (PoseidonGoldilocksConfig) or `KeccakPermutation` ([EXT]
 * plonky2/src/hash/keccak.rs: state -> field elements parsed from the hash onion
 * H(s) || H(H(s)) || ..., rejecting words >= p).
 * Hash -> elements for `observe_hash`: Poseidon HashOut = its 4 elements; BytesHash<25> = 7,7,7,4
 * byte little-endian chunks ([EXT] hash_types.rs `GenericHashOut::to_vec`).
 * Reference call sites: evm_arithmetization/src/prover.rs:118-127 (observe caps),
 * get_challenges.rs:11-227 (public values), prover.rs:320 (compact).
 * No reference KAT pins the transcript ("parity unpinned").
 */
#include "goldilocks.h"
#include "oracle.h"
#include <string.h>

static void keccak_permutation(uint64_t st[12]) {
    uint8_t bytes[96], h[32];
    for (int i = 0; i < 12; ++i) { uint64_t w = gl_canon(st[i]); memcpy(bytes + 8 * i, &w, 8); }
    size_t len = 96;
    int got = 0;
    uint8_t cur[96];
    memcpy(cur, bytes, 96);
    while (got < 12) {
        orc_keccak256(cur, len, h);
        memcpy(cur, h, 32);
        len = 32;
        for (int k = 0; k < 4 && got < 12; ++k) {
            uint64_t w;
            memcpy(&w, h + 8 * k, 8);
            if (w < GL_P) st[got++] = w;
        }
    }
}

static void permute(orc_challenger *c) {
    if (c->hasher == ORC_HASH_POSEIDON) orc_poseidon_permute(c->state);
    else keccak_permutation(c->state);
}

void orc_challenger_init(orc_challenger *c, int hasher) {
    memset(c, 0, sizeof *c);
    c->hasher = hasher;
}

static void duplexing(orc_challenger *c) {
    for (int i = 0; i < c->n_in; ++i) c->state[i] = c->in[i];
    c->n_in = 0;
    permute(c);
    memcpy(c->out, c->state, 8 * sizeof(uint64_t));
    c->n_out = 8;
}

void orc_challenger_observe(orc_challenger *c, const uint64_t *e, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        c->n_out = 0;
        c->in[c->n_in++] = gl_canon(e[i]);
        if (c->n_in == 8) duplexing(c);
    }
}

void orc_hash_to_elements(int hasher, const uint64_t *slot, uint64_t out[4]) {
    if (hasher == ORC_HASH_POSEIDON) { memcpy(out, slot, 32); return; }
    const uint8_t *b = (const uint8_t *)slot;
    for (int k = 0; k < 4; ++k) {
        uint64_t w = 0;
        int len = k < 3 ? 7 : 4;
        memcpy(&w, b + 7 * k, len);
        out[k] = w;
    }
}

void orc_challenger_observe_cap(orc_challenger *c, const uint64_t *slots, size_t n_digests) {
    for (size_t i = 0; i < n_digests; ++i) {
        uint64_t e[4];
        orc_hash_to_elements(c->hasher, slots + 4 * i, e);
        orc_challenger_observe(c, e, 4);
    }
}

uint64_t orc_challenger_get(orc_challenger *c) {
    if (c->n_in != 0 || c->n_out == 0) duplexing(c);
    return c->out[--c->n_out];
}

void orc_challenger_get_ext(orc_challenger *c, uint64_t out[2]) {
    out[0] = orc_challenger_get(c);
    out[1] = orc_challenger_get(c);
}

void orc_challenger_compact(orc_challenger *c, uint64_t out_state[12]) {
    if (c->n_in != 0) duplexing(c);
    c->n_out = 0;
    memcpy(out_state, c->state, 12 * sizeof(uint64_t));
}
