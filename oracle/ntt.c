/*
 * oracle/ntt.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * Restatement of plonky2_field 1.0.0 FFT semantics ([EXT] field/src/fft.rs, polynomial/mod.rs):
 *   PolynomialValues.values[i] = f(w^i) (natural order), w = primitive_root_of_unity(log n);
 *   `ifft`  : values -> natural-order coefficients (scaled by 1/n);
 *   `fft`   : coefficients -> natural-order values;
 *   `coset_fft(shift)` : coeff[i] *= shift^i then fft;
 *   `lde(rate_bits)`   : zero-pad coefficients to n << rate_bits.
 * Reference call sites: evm_arithmetization/src/prover.rs:100 (via PolynomialBatch::from_values).
 *
 * Textbook iterative radix-2 (bit-reverse, then decimation-in-time), deliberately a different
 * schedule from the HIP kernels (decimation-in-frequency, LDS-tiled multi-pass).
 */
#include "goldilocks.h"
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

/* Lazy arithmetic for the butterflies: any u64 represents its residue, one reduction per product, wrap-around corrections
 * without data-dependent branches; orc_fft / orc_ifft canonicalise once at the end.  Same values as the textbook loop this
 * replaced (r03 verdict, weak 7: the CPU baseline should be plonky2-class code, not a clarity-first restatement). */
static inline uint64_t mul_l(uint64_t a, uint64_t b) {
    u128 x = (u128)a * b;
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    uint64_t t0 = lo - hi_hi;
    if (__builtin_expect(lo < hi_hi, 0)) t0 -= GL_EPS;
    uint64_t t1 = hi_lo * GL_EPS;
    uint64_t r = t0 + t1;
    r += (0 - (uint64_t)(r < t1)) & GL_EPS;
    return r;
}
static inline uint64_t add_l(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    uint64_t c = (0 - (uint64_t)(s < a)) & GL_EPS;
    uint64_t r = s + c;
    r += (0 - (uint64_t)(r < c)) & GL_EPS;
    return r;
}
static inline uint64_t sub_l(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    uint64_t c = (0 - (uint64_t)(a < b)) & GL_EPS;
    uint64_t r = d - c;
    r -= (0 - (uint64_t)(d < c)) & GL_EPS;
    return r;
}

/* root^k, k < n/2, per (log_n, direction): built once, shared by every column and thread */
static uint64_t *tw_cache[2][33];
static const uint64_t *twiddles(unsigned log_n, int inverse, uint64_t root) {
    uint64_t *t = __atomic_load_n(&tw_cache[inverse][log_n], __ATOMIC_ACQUIRE);      /* (published with a release store below) */
    if (t) return t;
#pragma omp critical(orc_ntt_twiddles)
    {
        t = __atomic_load_n(&tw_cache[inverse][log_n], __ATOMIC_ACQUIRE);
        if (!t) {
            size_t half = (size_t)1 << (log_n - 1);
            uint64_t *n = (uint64_t *)malloc(sizeof(uint64_t) * half);
            n[0] = 1;
            for (size_t k = 1; k < half; ++k) n[k] = gl_mul(n[k - 1], root);
            __atomic_store_n(&tw_cache[inverse][log_n], n, __ATOMIC_RELEASE);
            t = n;
        }
    }
    return t;
}

/* stages [s0, s1] (pair distance 2^(s-1)) over a[lo, lo + len): twiddle of pair j in a block = tw[j << (log_n - s)] */
static inline void stages(uint64_t *a, size_t lo, size_t len, unsigned s0, unsigned s1, unsigned log_n, const uint64_t *tw) {
    for (unsigned s = s0; s <= s1; ++s) {
        const size_t m = (size_t)1 << s, half = m >> 1;
        const unsigned sh = log_n - s;
        for (size_t k = lo; k < lo + len; k += m)
            for (size_t j = 0; j < half; ++j) {
                const uint64_t u = a[k + j], t = mul_l(a[k + j + half], tw[j << sh]);
                a[k + j] = add_l(u, t);
                a[k + j + half] = sub_l(u, t);
            }
    }
}

/* in-place, natural in -> natural out (lazy representatives); root must have order exactly 2^log_n.
 * Bit-reversal, then decimation in time: the first BLOCK_LOG stages block by block (a block stays in the cache for all of
 * them: one sweep over the array instead of one per stage), the remaining ones stage by stage. */
#define ORC_NTT_BLOCK_LOG 12
static void ntt_core(uint64_t *a, unsigned log_n, uint64_t root, int inverse) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; ++i) {
        size_t j = bitrev(i, log_n);
        if (i < j) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    const uint64_t *tw = twiddles(log_n, inverse, root);
    const unsigned bl = log_n < ORC_NTT_BLOCK_LOG ? log_n : ORC_NTT_BLOCK_LOG;
    for (size_t b = 0; b < n; b += (size_t)1 << bl) stages(a, b, (size_t)1 << bl, 1, bl, log_n, tw);
    if (bl < log_n) stages(a, 0, n, bl + 1, log_n, log_n, tw);
}

void orc_fft(uint64_t *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; ++i) a[i] = gl_canon(a[i]);
    if (log_n == 0) return;
    ntt_core(a, log_n, gl_root_of_unity(log_n), 0);
    for (size_t i = 0; i < n; ++i) a[i] = gl_canon(a[i]);
}

void orc_ifft(uint64_t *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; ++i) a[i] = gl_canon(a[i]);
    if (log_n == 0) return;
    ntt_core(a, log_n, gl_inv(gl_root_of_unity(log_n)), 1);
    uint64_t ninv = gl_inv((uint64_t)n);
    for (size_t i = 0; i < n; ++i) a[i] = gl_canon(mul_l(a[i], ninv));
}

void orc_coset_fft(uint64_t *a, unsigned log_n, uint64_t shift) {
    size_t n = (size_t)1 << log_n;
    uint64_t s = 1;
    shift = gl_canon(shift);
    for (size_t i = 0; i < n; ++i) { a[i] = gl_mul(gl_canon(a[i]), s); s = gl_mul(s, shift); }
    orc_fft(a, log_n);
}

void orc_coset_ifft(uint64_t *a, unsigned log_n, uint64_t shift) {
    size_t n = (size_t)1 << log_n;
    orc_ifft(a, log_n);
    uint64_t sinv = gl_inv(shift), s = 1;
    for (size_t i = 0; i < n; ++i) { a[i] = gl_mul(a[i], s); s = gl_mul(s, sinv); }
}

/* coeffs (n) -> LDE values on the coset g*<w_N> (N = n << rate_bits), natural order.
 * [EXT] fri/oracle.rs `PolynomialBatch::lde_values`: p.lde(rate_bits).coset_fft(F::coset_shift()). */
void orc_lde(const uint64_t *coeffs, unsigned log_n, unsigned rate_bits, uint64_t *out) {
    size_t n = (size_t)1 << log_n, N = n << rate_bits;
    memcpy(out, coeffs, n * sizeof(uint64_t));
    memset(out + n, 0, (N - n) * sizeof(uint64_t));
    orc_coset_fft(out, log_n + rate_bits, GL_GENERATOR);
}

/* direct O(n) evaluation helpers used by property tests */
uint64_t orc_eval_poly(const uint64_t *coeffs, size_t n, uint64_t x) {
    uint64_t acc = 0;
    for (size_t i = n; i-- > 0;) acc = gl_add(gl_mul(acc, x), gl_canon(coeffs[i]));
    return acc;
}

void orc_eval_poly_ext(const uint64_t *coeffs, size_t n, const uint64_t x[2], uint64_t out[2]) {
    gl2_t acc = gl2_from(0), xx = {{gl_canon(x[0]), gl_canon(x[1])}};
    for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, xx), gl2_from(coeffs[i]));
    out[0] = acc.c[0]; out[1] = acc.c[1];
}
