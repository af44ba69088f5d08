/*
 * oracle/keccak.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * Keccak-f[1600] / Keccak-256 (original Keccak padding 0x01, as used by Ethereum and by
 * plonky2's `KeccakHash<N>`; [EXT] plonky2/src/hash/keccak.rs: `hash_no_pad` = keccak256 of the
 * little-endian canonical u64 encoding of the elements, truncated to N bytes; `two_to_one` =
 * keccak256(left || right) truncated; `hash_or_noop`: raw bytes if 8*len <= N).
 * The reference uses it with N = 25 (`KeccakGoldilocksConfig`, reference
 * evm_arithmetization/tests/simple_transfer.rs:30).
 *
 * Pinned by reference KATs common/src/lib.rs:5-15 (keccak256("") and keccak256(0x80)).
 */
#include "goldilocks.h"
#include "oracle.h"
#include <string.h>
#include <stdlib.h>

static const uint64_t KRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43,
                             25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};

static inline uint64_t rol(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

void orc_keccak_f1600(uint64_t a[25]) {
    for (int rnd = 0; rnd < 24; ++rnd) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) a[i] ^= d[i % 5];
        /* rho + pi: B[y, 2x+3y] = rot(A[x,y]) with lane index x + 5y */
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y)
                b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(a[x + 5 * y], KROT[x + 5 * y]);
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x)
                a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= KRC[rnd];
    }
}

void orc_keccak256(const uint8_t *in, size_t len, uint8_t out[32]) {
    uint64_t st[25] = {0};
    const size_t rate = 136;
    uint8_t blk[136];
    while (len >= rate) {
        for (size_t i = 0; i < rate / 8; ++i) { uint64_t w; memcpy(&w, in + 8 * i, 8); st[i] ^= w; }
        orc_keccak_f1600(st);
        in += rate; len -= rate;
    }
    memset(blk, 0, rate);
    memcpy(blk, in, len);
    blk[len] ^= 0x01;
    blk[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate / 8; ++i) { uint64_t w; memcpy(&w, blk + 8 * i, 8); st[i] ^= w; }
    orc_keccak_f1600(st);
    memcpy(out, st, 32);
}

/* KeccakHash<25>: a hash is 25 bytes; we carry it in 32-byte slots (bytes 25..31 zero). */
void orc_keccak25_hash_no_pad(const uint64_t *in, size_t n, uint8_t out[32]) {
    uint8_t *buf = (uint8_t *)malloc(n * 8 + 1);
    for (size_t i = 0; i < n; ++i) { uint64_t w = gl_canon(in[i]); memcpy(buf + 8 * i, &w, 8); }
    uint8_t h[32];
    orc_keccak256(buf, n * 8, h);
    free(buf);
    memset(out, 0, 32);
    memcpy(out, h, 25);
}

void orc_keccak25_hash_or_noop(const uint64_t *in, size_t n, uint8_t out[32]) {
    if (n * 8 <= 25) {
        memset(out, 0, 32);
        for (size_t i = 0; i < n; ++i) { uint64_t w = gl_canon(in[i]); memcpy(out + 8 * i, &w, 8); }
    } else {
        orc_keccak25_hash_no_pad(in, n, out);
    }
}

void orc_keccak25_two_to_one(const uint8_t l[32], const uint8_t r[32], uint8_t out[32]) {
    uint8_t buf[50], h[32];
    memcpy(buf, l, 25);
    memcpy(buf + 25, r, 25);
    orc_keccak256(buf, 50, h);
    memset(out, 0, 32);
    memcpy(out, h, 25);
}
