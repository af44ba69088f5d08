/*
 * oracle/goldilocks.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * CPU restatement of plonky2_field 1.0.0 `GoldilocksField` and its quadratic extension
 * ([EXT] field/src/goldilocks_field.rs, field/src/goldilocks_extensions.rs -- crates.io dependency
 * pinned at Cargo.lock:3727-3730 of the reference, NOT vendored under /root/reference).  The in-tree
 * description of the reduction is reference book/src/framework/field.md:5-19.
 *
 * Written for clarity with unsigned __int128; the HIP library (zk_evm_amd/csrc) uses a different
 * 32-bit-limb formulation, so agreement between the two is a real check.
 */
#ifndef ORACLE_GOLDILOCKS_H
#define ORACLE_GOLDILOCKS_H

#include <stdint.h>
#include <stddef.h>

typedef unsigned __int128 u128;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL /* 2^64 mod p */
/* [EXT] goldilocks_field.rs: MULTIPLICATIVE_GROUP_GENERATOR (also `coset_shift()`),
 * POWER_OF_TWO_GENERATOR (order 2^32).  Orders re-derived in tests/test_oracle_field.py. */
#define GL_GENERATOR 14293326489335486720ULL
#define GL_POW2_GENERATOR 7277203076849721926ULL
#define GL_TWO_ADICITY 32
/* Quadratic extension F[X]/(X^2 - 7) ([EXT] goldilocks_extensions.rs: `W = 7`). */
#define GL_EXT_W 7ULL

static inline uint64_t gl_canon(uint64_t a) { return a >= GL_P ? a - GL_P : a; }

/* The defining statement of the reduction; kept for tests (orc_gl_reduce128_check compares the two). */
static inline uint64_t gl_reduce128_slow(u128 x) { return (uint64_t)(x % GL_P); }
/* Same value with the identities of field.md:5-19 (2^64 = 2^32 - 1, 2^96 = -1 mod p): no 128-bit division, which
 * dominated the oracle's Poseidon (360 reductions per permutation) and with it the bench's CPU baseline. */
static inline uint64_t gl_reduce128(u128 x) {
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    uint64_t t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= GL_EPS;
    uint64_t t1 = hi_lo * GL_EPS;
    uint64_t r = t0 + t1;
    if (r < t1) r += GL_EPS;
    return gl_canon(r);
}

/* All oracle values are kept canonical (< p); inputs are canonicalised defensively. */
static inline uint64_t gl_add(uint64_t a, uint64_t b) {
    a = gl_canon(a); b = gl_canon(b);
    uint64_t s = a + b;
    if (s < a || s >= GL_P) s -= GL_P;
    return s;
}
static inline uint64_t gl_sub(uint64_t a, uint64_t b) {
    a = gl_canon(a); b = gl_canon(b);
    return a >= b ? a - b : a + (GL_P - b);
}
static inline uint64_t gl_neg(uint64_t a) { return gl_sub(0, a); }
static inline uint64_t gl_mul(uint64_t a, uint64_t b) {
    /* Fast reduction of the 128-bit product (field.md:5-19): 2^64 = 2^32-1, 2^96 = -1 (mod p). */
    u128 x = (u128)a * b;
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    uint64_t t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= GL_EPS; /* borrow: +p == -EPS mod 2^64 */
    uint64_t t1 = hi_lo * GL_EPS;
    uint64_t r = t0 + t1;
    if (r < t1) r += GL_EPS; /* carry: 2^64 == EPS */
    return gl_canon(r);
}
static inline uint64_t gl_sqr(uint64_t a) { return gl_mul(a, a); }

static inline uint64_t gl_pow(uint64_t b, uint64_t e) {
    uint64_t r = 1;
    b = gl_canon(b);
    while (e) {
        if (e & 1) r = gl_mul(r, b);
        b = gl_mul(b, b);
        e >>= 1;
    }
    return r;
}
static inline uint64_t gl_inv(uint64_t a) { return gl_pow(a, GL_P - 2); }

/* [EXT] Field::primitive_root_of_unity(n_log) = POWER_OF_TWO_GENERATOR^(2^(32-n_log)). */
static inline uint64_t gl_root_of_unity(unsigned n_log) {
    uint64_t r = GL_POW2_GENERATOR;
    for (unsigned i = n_log; i < GL_TWO_ADICITY; ++i) r = gl_sqr(r);
    return r;
}

/* ---- quadratic extension: a = a0 + a1*X, X^2 = 7 ---- */
typedef struct { uint64_t c[2]; } gl2_t;

static inline gl2_t gl2_from(uint64_t a) { gl2_t r = {{gl_canon(a), 0}}; return r; }
static inline gl2_t gl2_add(gl2_t a, gl2_t b) {
    gl2_t r = {{gl_add(a.c[0], b.c[0]), gl_add(a.c[1], b.c[1])}}; return r;
}
static inline gl2_t gl2_sub(gl2_t a, gl2_t b) {
    gl2_t r = {{gl_sub(a.c[0], b.c[0]), gl_sub(a.c[1], b.c[1])}}; return r;
}
static inline gl2_t gl2_mul(gl2_t a, gl2_t b) {
    gl2_t r;
    r.c[0] = gl_add(gl_mul(a.c[0], b.c[0]), gl_mul(GL_EXT_W, gl_mul(a.c[1], b.c[1])));
    r.c[1] = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
    return r;
}
static inline gl2_t gl2_scale(gl2_t a, uint64_t s) {
    gl2_t r = {{gl_mul(a.c[0], s), gl_mul(a.c[1], s)}}; return r;
}
static inline gl2_t gl2_inv(gl2_t a) {
    /* 1/(a0 + a1 X) = (a0 - a1 X) / (a0^2 - 7 a1^2) */
    uint64_t nrm = gl_sub(gl_sqr(a.c[0]), gl_mul(GL_EXT_W, gl_sqr(a.c[1])));
    uint64_t ni = gl_inv(nrm);
    gl2_t r = {{gl_mul(a.c[0], ni), gl_mul(gl_neg(a.c[1]), ni)}};
    return r;
}
static inline gl2_t gl2_pow(gl2_t b, uint64_t e) {
    gl2_t r = gl2_from(1);
    while (e) {
        if (e & 1) r = gl2_mul(r, b);
        b = gl2_mul(b, b);
        e >>= 1;
    }
    return r;
}

static inline size_t bitrev(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

#endif
