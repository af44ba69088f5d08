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

/* in-place, natural in -> natural out; root must have order exactly 2^log_n */
static void ntt_core(uint64_t *a, unsigned log_n, uint64_t root) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; ++i) {
        size_t j = bitrev(i, log_n);
        if (i < j) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    /* per-stage twiddle tables to keep the oracle usable as a CPU baseline */
    uint64_t *tw = (uint64_t *)malloc(sizeof(uint64_t) * (n / 2 + 1));
    for (unsigned s = 1; s <= log_n; ++s) {
        size_t m = (size_t)1 << s, half = m >> 1;
        uint64_t wm = root;
        for (unsigned k = s; k < log_n; ++k) wm = gl_sqr(wm);
        tw[0] = 1;
        for (size_t j = 1; j < half; ++j) tw[j] = gl_mul(tw[j - 1], wm);
        for (size_t k = 0; k < n; k += m) {
            for (size_t j = 0; j < half; ++j) {
                uint64_t u = a[k + j], t = gl_mul(a[k + j + half], tw[j]);
                a[k + j] = gl_add(u, t);
                a[k + j + half] = gl_sub(u, t);
            }
        }
    }
    free(tw);
}

void orc_fft(uint64_t *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; ++i) a[i] = gl_canon(a[i]);
    if (log_n == 0) return;
    ntt_core(a, log_n, gl_root_of_unity(log_n));
}

void orc_ifft(uint64_t *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; ++i) a[i] = gl_canon(a[i]);
    if (log_n == 0) return;
    ntt_core(a, log_n, gl_inv(gl_root_of_unity(log_n)));
    uint64_t ninv = gl_inv((uint64_t)n);
    for (size_t i = 0; i < n; ++i) a[i] = gl_mul(a[i], ninv);
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
