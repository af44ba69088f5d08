/*
 * oracle/poseidon_fast.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * The SAME permutation as oracle/poseidon.c (`orc_poseidon_permute`, the plain 4 + 22 + 4 round definition that the
 * reference-tree KATs pin), evaluated the way a tuned CPU prover evaluates it, so that `cpu_baseline` (bench.py) times
 * plonky2-class code instead of a clarity-first restatement (r03 verdict, weak 7: the plain form costs ~5 us per
 * permutation per core, plonky2's own AVX2 / "fast partial rounds" code ~1 us):
 *   * lazy arithmetic: any u64 represents its residue; one branch-light reduction per product, none per addition;
 *   * the MDS product on 32-bit halves with the (< 2^6) constants as plain integers, written so that gcc vectorises it;
 *   * the 23 linear layers between the full S-box layers of rounds 3 and 26 -- each followed by ONE S-box -- in blocks
 *     of three: with x the state after an S-box, `M^3 x + K + d1 M^2 e0 + d2 M e0` is the state three layers later, where
 *     d_k = sbox(w_k) - w_k are the two single S-boxes in between, w1 = (M x)_0 + rc, w2 = (M^2 x)_0 + c + d1 M_00
 *     (168 multiply-adds for three layers instead of 432).  M^2 and M^3 are exact integer matrices (entries < 2^21).
 * Every constant of the blocked schedule (M^2, M^3, K, c) is DERIVED HERE AT FIRST USE from the plain round constants and
 * MDS entries, and the whole fast permutation is checked against `orc_poseidon_permute` on pseudo-random states before its
 * first result is handed out (abort on any difference); tests/test_oracle_kat.py and tests/test_oracle_fast_poseidon.py
 * compare the two again.  Nothing here comes from the product's generated constants.
 */
#include "goldilocks.h"
#include "../include/poseidon_constants.h"
#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

static const uint64_t RC[ZK_POSEIDON_ROUNDS * ZK_POSEIDON_WIDTH] = ZK_POSEIDON_RC_INIT;
static const uint64_t MDS_CIRC[12] = ZK_POSEIDON_MDS_CIRC_INIT;
static const uint64_t MDS_DIAG[12] = ZK_POSEIDON_MDS_DIAG_INIT;

/* ---- lazy field arithmetic: results are any u64 congruent to the value ---- */
static inline uint64_t red128(u128 x) {
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    uint64_t t0 = lo - hi_hi;
    if (__builtin_expect(lo < hi_hi, 0)) t0 -= GL_EPS;
    uint64_t t1 = hi_lo * GL_EPS;
    uint64_t r = t0 + t1;
    r += (0 - (uint64_t)(r < t1)) & GL_EPS;          /* the carry is a coin flip: no branch */
    return r;
}
static inline uint64_t mul_l(uint64_t a, uint64_t b) { return red128((u128)a * b); }
static inline uint64_t sbox7_l(uint64_t x) {
    uint64_t x2 = mul_l(x, x), x4 = mul_l(x2, x2), x3 = mul_l(x, x2);
    return mul_l(x3, x4);
}
/* a + c for a canonical constant c < p: a second wrap is impossible */
static inline uint64_t add_c(uint64_t a, uint64_t c) {
    uint64_t s = a + c;
    s += (0 - (uint64_t)(s < c)) & GL_EPS;
    return s;
}

#define FIRST_BLOCK_ROUND 3          /* the blocks start after the S-box layer of round 3 ... */
#define N_BLOCKS 8                   /* ... seven of three layers and one of two: 23 layers, up to the S-boxes of round 26 */
static uint64_t M1[12][12], M2[12][12], M3[12][12];     /* M, M^2, M^3 over the integers */
static struct { int k; uint64_t c1, c2, K[12]; } BLK[N_BLOCKS];
/* the same matrices TRANSPOSED for the vector product (four output rows per AVX2 register, one input broadcast per step):
 * TM[j][i] = M[i][j]; T3 / T2 carry, as inputs 12 and 13, the columns that multiply d1 and d2 */
static uint64_t TM[12][12] __attribute__((aligned(32))), T3[14][12] __attribute__((aligned(32))), T2[13][12] __attribute__((aligned(32)));
static int have_avx2 = 0;
static volatile int ready = 0;

static void permute_fast_unchecked(uint64_t st[12]);

static void mat_mul(uint64_t out[12][12], uint64_t a[12][12], uint64_t b[12][12]) {
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 12; ++j) {
            uint64_t s = 0;
            for (int t = 0; t < 12; ++t) s += a[i][t] * b[t][j];
            out[i][j] = s;
        }
}
static void mat_vec_mod(uint64_t out[12], uint64_t m[12][12], const uint64_t v[12]) {
    for (int i = 0; i < 12; ++i) {
        uint64_t s = 0;
        for (int j = 0; j < 12; ++j) s = gl_add(s, gl_mul(m[i][j] % GL_P, v[j]));
        out[i] = s;
    }
}
static void init_once(void) {
#pragma omp critical(orc_poseidon_fast_init)
    {
        if (!ready) {
            /* out[r] = sum_i v[(i + r) % 12] CIRC[i] + v[r] DIAG[r]  =>  M[r][(i + r) % 12] += CIRC[i], M[r][r] += DIAG[r] */
            memset(M1, 0, sizeof M1);
            for (int r = 0; r < 12; ++r) {
                for (int i = 0; i < 12; ++i) M1[r][(i + r) % 12] += MDS_CIRC[i];
                M1[r][r] += MDS_DIAG[r];
            }
            mat_mul(M2, M1, M1);
            mat_mul(M3, M2, M1);
            for (int i = 0; i < 12; ++i) {
                for (int j = 0; j < 12; ++j) { TM[j][i] = M1[i][j]; T3[j][i] = M3[i][j]; T2[j][i] = M2[i][j]; }
                T3[12][i] = M2[i][0]; T3[13][i] = M1[i][0];
                T2[12][i] = M1[i][0];
            }
#if defined(__x86_64__)
            have_avx2 = __builtin_cpu_supports("avx2");
#endif
            int r = FIRST_BLOCK_ROUND;
            for (int b = 0; b < N_BLOCKS; ++b) {
                const int k = b + 1 < N_BLOCKS ? 3 : 2;
                const uint64_t *r1 = RC + 12 * (r + 1), *r2 = RC + 12 * (r + 2), *r3 = RC + 12 * (r + 3);
                uint64_t a[12], c[12];
                BLK[b].k = k;
                BLK[b].c1 = r1[0];
                mat_vec_mod(a, M1, r1);                                   /* M rc' */
                if (k == 3) {
                    BLK[b].c2 = gl_add(a[0], r2[0]);
                    mat_vec_mod(a, M2, r1);                               /* M^2 rc' + M rc'' + rc''' */
                    mat_vec_mod(c, M1, r2);
                    for (int i = 0; i < 12; ++i) BLK[b].K[i] = gl_add(gl_add(a[i], c[i]), r3[i]);
                } else {
                    BLK[b].c2 = 0;
                    for (int i = 0; i < 12; ++i) BLK[b].K[i] = gl_add(a[i], r2[i]);   /* M rc' + rc'' */
                }
                r += k;
            }
            if (r != ZK_POSEIDON_HALF_FULL_ROUNDS + ZK_POSEIDON_PARTIAL_ROUNDS) { fprintf(stderr, "oracle: bad Poseidon block schedule\n"); abort(); }
            /* the blocked form against the plain definition, before anyone sees a result */
            uint64_t z = 0x9E3779B97F4A7C15ULL;
            for (int t = 0; t < 256; ++t) {
                uint64_t a[12], b[12];
                for (int i = 0; i < 12; ++i) {
                    z = z * 6364136223846793005ULL + 1442695040888963407ULL;
                    a[i] = b[i] = t < 4 ? (t & 1 ? GL_P - 1 - (uint64_t)i : (uint64_t)i * (t >> 1)) : (z ^ (z >> 29));
                }
                orc_poseidon_permute(a);
                permute_fast_unchecked(b);
                for (int i = 0; i < 12; ++i)
                    if (a[i] != gl_canon(b[i])) { fprintf(stderr, "oracle: fast Poseidon differs from the plain definition\n"); abort(); }
            }
            ready = 1;
        }
    }
}

static inline void split(const uint64_t st[12], uint64_t lo[12], uint64_t hi[12]) {
    for (int i = 0; i < 12; ++i) { lo[i] = st[i] & 0xFFFFFFFFULL; hi[i] = st[i] >> 32; }
}
/* sum_j m[j] x_j as al + ah 2^32 (no reduction; m[j] < 2^21: al, ah < 2^57) */
static inline u128 dot_row(const uint64_t m[12], const uint64_t lo[12], const uint64_t hi[12]) {
    uint64_t al = 0, ah = 0;
    for (int j = 0; j < 12; ++j) { al += lo[j] * m[j]; ah += hi[j] * m[j]; }
    return (u128)al + ((u128)ah << 32);
}
#if defined(__x86_64__)
/* al[i] = sum_j T[j][i] lo[j], ah[i] = sum_j T[j][i] hi[j] for n_in inputs: vpmuludq on (< 2^32) x (< 2^21) */
__attribute__((target("avx2"))) static void matvec_avx2(uint64_t al[12], uint64_t ah[12], const uint64_t (*T)[12], int n_in,
                                                        const uint64_t *lo, const uint64_t *hi) {
    __m256i l0 = _mm256_setzero_si256(), l1 = l0, l2 = l0, h0 = l0, h1 = l0, h2 = l0;
    for (int j = 0; j < n_in; ++j) {
        const __m256i c0 = _mm256_load_si256((const __m256i *)(T[j])), c1 = _mm256_load_si256((const __m256i *)(T[j] + 4)),
                      c2 = _mm256_load_si256((const __m256i *)(T[j] + 8));
        const __m256i bl = _mm256_set1_epi64x((long long)lo[j]), bh = _mm256_set1_epi64x((long long)hi[j]);
        l0 = _mm256_add_epi64(l0, _mm256_mul_epu32(bl, c0)); h0 = _mm256_add_epi64(h0, _mm256_mul_epu32(bh, c0));
        l1 = _mm256_add_epi64(l1, _mm256_mul_epu32(bl, c1)); h1 = _mm256_add_epi64(h1, _mm256_mul_epu32(bh, c1));
        l2 = _mm256_add_epi64(l2, _mm256_mul_epu32(bl, c2)); h2 = _mm256_add_epi64(h2, _mm256_mul_epu32(bh, c2));
    }
    _mm256_storeu_si256((__m256i *)al, l0); _mm256_storeu_si256((__m256i *)(al + 4), l1); _mm256_storeu_si256((__m256i *)(al + 8), l2);
    _mm256_storeu_si256((__m256i *)ah, h0); _mm256_storeu_si256((__m256i *)(ah + 4), h1); _mm256_storeu_si256((__m256i *)(ah + 8), h2);
}
#endif
/* out[i] = sum_j T[j][i] x[j] (+ add[i]), inputs as 32-bit halves; n_in <= 14 */
static inline void matvec(uint64_t out[12], const uint64_t (*T)[12], int n_in, const uint64_t *lo, const uint64_t *hi,
                          const uint64_t *add) {
    uint64_t al[12], ah[12];
#if defined(__x86_64__)
    if (have_avx2) matvec_avx2(al, ah, T, n_in, lo, hi);
    else
#endif
    for (int i = 0; i < 12; ++i) {
        uint64_t a = 0, b = 0;
        for (int j = 0; j < n_in; ++j) { a += lo[j] * T[j][i]; b += hi[j] * T[j][i]; }
        al[i] = a; ah[i] = b;
    }
    for (int i = 0; i < 12; ++i) out[i] = red128((u128)al[i] + ((u128)ah[i] << 32) + (add ? add[i] : 0));
}
static inline void full_round(uint64_t st[12], const uint64_t *rc) {
    uint64_t lo[12], hi[12];
    for (int i = 0; i < 12; ++i) st[i] = sbox7_l(add_c(st[i], rc[i]));
    split(st, lo, hi);
    matvec(st, TM, 12, lo, hi, NULL);
}

static void permute_fast_unchecked(uint64_t st[12]) {
    int round = 0;
    for (; round < FIRST_BLOCK_ROUND; ++round) full_round(st, RC + 12 * round);
    for (int i = 0; i < 12; ++i) st[i] = sbox7_l(add_c(st[i], RC[12 * round + i]));      /* round 3 up to its S-boxes */
    for (int b = 0; b < N_BLOCKS; ++b) {
        uint64_t lo[14], hi[14], out[12];
        split(st, lo, hi);
        const uint64_t w1 = red128(dot_row(M1[0], lo, hi) + BLK[b].c1);
        const uint64_t d1 = gl_sub(gl_canon(sbox7_l(w1)), gl_canon(w1));
        if (BLK[b].k == 3) {
            const uint64_t w2 = red128(dot_row(M2[0], lo, hi) + BLK[b].c2 + (u128)d1 * M1[0][0]);
            const uint64_t d2 = gl_sub(gl_canon(sbox7_l(w2)), gl_canon(w2));
            lo[12] = d1 & 0xFFFFFFFFULL; hi[12] = d1 >> 32;
            lo[13] = d2 & 0xFFFFFFFFULL; hi[13] = d2 >> 32;
            matvec(out, T3, 14, lo, hi, BLK[b].K);
        } else {
            lo[12] = d1 & 0xFFFFFFFFULL; hi[12] = d1 >> 32;
            matvec(out, T2, 13, lo, hi, BLK[b].K);
        }
        round += BLK[b].k;                                  /* `out` = the state of round `round` before its S-box layer */
        if (b + 1 < N_BLOCKS) { memcpy(st, out, sizeof out); st[0] = sbox7_l(out[0]); }
        else for (int i = 0; i < 12; ++i) st[i] = sbox7_l(out[i]);       /* round 26: a full layer */
    }
    {   /* round 26's MDS layer, then the last three full rounds */
        uint64_t lo[12], hi[12];
        split(st, lo, hi);
        matvec(st, TM, 12, lo, hi, NULL);
        for (++round; round < ZK_POSEIDON_ROUNDS; ++round) full_round(st, RC + 12 * round);
    }
}

void orc_poseidon_permute_fast(uint64_t st[12]) {
    if (!ready) init_once();
    permute_fast_unchecked(st);
    for (int i = 0; i < 12; ++i) st[i] = gl_canon(st[i]);
}
