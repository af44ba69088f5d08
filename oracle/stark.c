/*
 * oracle/stark.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * The row / coset-point loops of starky 1.0.0 ([EXT] starky/src/{prover.rs `compute_quotient_polys`, lookup.rs
 * `lookup_helper_columns` / `get_helper_cols`, cross_table_lookup.rs `partial_sums`}; reference call sites
 * evm_arithmetization/src/prover.rs:137,322) in C with OpenMP over rows -- the axis rayon uses there -- so that the
 * oracle reaches 2^12..2^20 rows and can serve as the timed CPU baseline of a whole table proof.
 *
 * What is evaluated per row is NOT restated here: it is a "tape", a straight-line field program obtained by running
 * the Python restatements (oracle/airs.py: each table's `eval_packed_generic`; oracle/stark.py: `Column`, `Filter`,
 * `eval_packed_lookups`, `eval_cross_table_lookup_checks`) once on symbols (oracle/tape.py).  This file only knows
 * ADD / SUB / MUL, the constraint-consumer recurrence and the selector formulas.
 */
#include "goldilocks.h"
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* nodes: [0, n_in) inputs (already stored in v), [n_in, n_in + n_consts) constants, then one per op */
static inline void tape_run(uint64_t *v, const uint32_t *ops, size_t n_ops, size_t first) {
    uint64_t *dst = v + first;
    for (size_t k = 0; k < n_ops; ++k) {
        const uint32_t *o = ops + 3 * k;
        const uint64_t a = v[o[1]], b = v[o[2]];
        dst[k] = o[0] == 2 ? gl_mul(a, b) : o[0] == 0 ? gl_add(a, b) : gl_sub(a, b);
    }
}

/* Evaluate the tape at every row r < n.  Input node j reads in_cols[j][r + in_off[j]]; a read past the last row
 * wraps when `wrap`, and is 0 otherwise (`Column::eval_table`: the next-row part is dropped at the last row).
 * out_cols[k][r] = node out_nodes[k]. */
int orc_tape_rows(const uint32_t *ops, size_t n_ops, const uint64_t *consts, size_t n_consts, size_t n_in,
                  const uint64_t *const *in_cols, const int64_t *in_off, int wrap, size_t n,
                  const uint32_t *out_nodes, size_t n_out, uint64_t *const *out_cols) {
    const size_t n_nodes = n_in + n_consts + n_ops;
    int fail = 0;
#pragma omp parallel
    {
        uint64_t *v = (uint64_t *)malloc(sizeof(uint64_t) * (n_nodes ? n_nodes : 1));
        if (!v) {
#pragma omp atomic write
            fail = 1;
        } else {
            memcpy(v + n_in, consts, sizeof(uint64_t) * n_consts);
#pragma omp for schedule(static)
            for (size_t r = 0; r < n; ++r) {
                for (size_t j = 0; j < n_in; ++j) {
                    size_t rr = r + (size_t)in_off[j];
                    if (rr >= n) { if (wrap) rr -= n; else { v[j] = 0; continue; } }
                    v[j] = gl_canon(in_cols[j][rr]);
                }
                tape_run(v, ops, n_ops, n_in + n_consts);
                for (size_t k = 0; k < n_out; ++k) out_cols[k][r] = v[out_nodes[k]];
            }
            free(v);
        }
    }
    return fail ? -1 : 0;
}

/* acc[d] += f[d] == 1 ? 1 / v[d] : 0;  a filter value outside {0, 1} is starky's "Non-binary filter?" panic: -1.
 * Montgomery batch inversion per block (plonky2 `batch_multiplicative_inverse`). */
int orc_masked_inverse_accumulate(const uint64_t *f, const uint64_t *v, size_t n, uint64_t *acc) {
    enum { B = 1024 };
    int bad = 0;
#pragma omp parallel for schedule(static)
    for (size_t s = 0; s < n; s += B) {
        const size_t e = s + B < n ? s + B : n;
        uint64_t pre[B];
        uint64_t run = 1;
        for (size_t i = s; i < e; ++i) {
            if (f[i] > 1) {
#pragma omp atomic write
                bad = 1;
            }
            pre[i - s] = run;
            if (f[i] == 1) run = gl_mul(run, v[i]);
        }
        uint64_t inv = gl_inv(run);      /* a zero denominator gives 0, as Field::try_inverse would panic: not reached */
        for (size_t i = e; i-- > s;) {
            if (f[i] != 1) continue;
            acc[i] = gl_add(acc[i], gl_mul(inv, pre[i - s]));
            inv = gl_mul(inv, v[i]);
        }
    }
    return bad ? -1 : 0;
}

/* logUp Z of one lookup: z[0] = 0, z[i+1] = z[i] + sum_h helpers[h][i] - freq[i] / table_den[i]
 * ([EXT] lookup.rs `lookup_helper_columns`; table_den = table + challenge) */
void orc_lookup_z(const uint64_t *const *helpers, size_t n_h, const uint64_t *freq, const uint64_t *table_den, size_t n,
                  uint64_t *z) {
    uint64_t *x = (uint64_t *)calloc(n, sizeof(uint64_t));
    uint64_t *one = (uint64_t *)calloc(n ? n : 1, sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i) one[i] = 1;
    orc_masked_inverse_accumulate(one, table_den, n, x);           /* x = 1 / table_den */
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        uint64_t s = 0;
        for (size_t h = 0; h < n_h; ++h) s = gl_add(s, helpers[h][i]);
        x[i] = gl_sub(s, gl_mul(gl_canon(freq[i]), x[i]));
    }
    z[0] = 0;
    for (size_t i = 0; i + 1 < n; ++i) z[i + 1] = gl_add(z[i], x[i]);
    free(x);
    free(one);
}

/* CTL Z: z[i] = sum_{j >= i} sum_h helpers[h][j]  ([EXT] cross_table_lookup.rs `partial_sums`) */
void orc_ctl_z(const uint64_t *const *helpers, size_t n_h, size_t n, uint64_t *z) {
    uint64_t run = 0;
    for (size_t i = n; i-- > 0;) {
        for (size_t h = 0; h < n_h; ++h) run = gl_add(run, helpers[h][i]);
        z[i] = run;
    }
}

/* [EXT] starky prover.rs `compute_quotient_polys` up to (excluding) the coset_ifft.
 * Tape inputs: lv[C], nv[C], aux_lv[A], aux_nv[A]; outputs = the constraints in yield order; kinds[k]: 0 plain,
 * 1 transition (x (x - g^-1)), 2 first row (x L_first), 3 last row (x L_last).
 * leaves: committed leaves [N][cols] row-major, rows bit-reversed (orc_commit_values); point i of the coset of size
 * n << qdb reads LDE row i * 2^(rate_bits - qdb), the "next" row is point (i + 2^qdb) mod size.
 * out[c][i] = (sum_k alpha_c^(K-1-k) constraint_k(i)) / Z_H(x_i),  x_i = g * w^i. */
int orc_quotient_values(const uint32_t *ops, size_t n_ops, const uint64_t *consts, size_t n_consts,
                        const uint32_t *out_nodes, const uint32_t *kinds, size_t n_constraints,
                        const uint64_t *trace_leaves, size_t C, const uint64_t *aux_leaves, size_t A,
                        unsigned degree_bits, unsigned rate_bits, unsigned qdb, const uint64_t *alphas, size_t n_alphas,
                        uint64_t *const *out) {
    const size_t n = (size_t)1 << degree_bits, size = n << qdb;
    const unsigned log_lde = degree_bits + rate_bits;
    const size_t step = (size_t)1 << (rate_bits - qdb), next_step = (size_t)1 << qdb;
    const size_t n_in = 2 * C + 2 * A, n_nodes = n_in + n_consts + n_ops;
    const uint64_t w = gl_root_of_unity(degree_bits + qdb);
    const uint64_t last = gl_inv(gl_root_of_unity(degree_bits));
    const uint64_t n_inv = gl_inv((uint64_t)n);
    int fail = 0;
#pragma omp parallel
    {
        uint64_t *v = (uint64_t *)malloc(sizeof(uint64_t) * n_nodes);
        uint64_t acc[16];
        if (!v || n_alphas > 16) {
#pragma omp atomic write
            fail = 1;
        } else {
            memcpy(v + n_in, consts, sizeof(uint64_t) * n_consts);
#pragma omp for schedule(static)
            for (size_t i = 0; i < size; ++i) {
                const uint64_t x = gl_mul(GL_GENERATOR, gl_pow(w, i));
                const uint64_t zh = gl_sub(gl_pow(x, n), 1);
                const uint64_t z_last = gl_sub(x, last);
                const uint64_t zn = gl_mul(zh, n_inv);
                const uint64_t lf = gl_mul(zn, gl_inv(gl_sub(x, 1)));
                const uint64_t ll = gl_mul(gl_mul(zn, last), gl_inv(z_last));
                const size_t i_next = (i + next_step) % size;
                const uint64_t *r0 = trace_leaves + bitrev(i * step, log_lde) * C;
                const uint64_t *r1 = trace_leaves + bitrev(i_next * step, log_lde) * C;
                memcpy(v, r0, 8 * C);
                memcpy(v + C, r1, 8 * C);
                if (A) {
                    memcpy(v + 2 * C, aux_leaves + bitrev(i * step, log_lde) * A, 8 * A);
                    memcpy(v + 2 * C + A, aux_leaves + bitrev(i_next * step, log_lde) * A, 8 * A);
                }
                tape_run(v, ops, n_ops, n_in + n_consts);
                for (size_t c = 0; c < n_alphas; ++c) acc[c] = 0;
                for (size_t k = 0; k < n_constraints; ++k) {
                    uint64_t cv = v[out_nodes[k]];
                    switch (kinds[k]) {
                        case 1: cv = gl_mul(cv, z_last); break;
                        case 2: cv = gl_mul(cv, lf); break;
                        case 3: cv = gl_mul(cv, ll); break;
                        default: break;
                    }
                    for (size_t c = 0; c < n_alphas; ++c) acc[c] = gl_add(gl_mul(acc[c], alphas[c]), cv);
                }
                const uint64_t zinv = gl_inv(zh);
                for (size_t c = 0; c < n_alphas; ++c) out[c][i] = gl_mul(acc[c], zinv);
            }
        }
        free(v);
    }
    return fail ? -1 : 0;
}

/* ---- plonky2 PLONK prover row loops ([EXT] plonky2 1.0.0 plonk/prover.rs; reference call sites
 * evm_arithmetization/src/fixed_recursive_verifier.rs:2146 `root.circuit.prove`, :3167-3179 `shrink`) ------------- */

/* `wires_permutation_partial_products_and_zs` for one (beta, gamma): per row i the quotients
 * (w_j + beta k_j x_i + gamma) / (w_j + beta sigma_j(x_i) + gamma), j < routed, multiplied in chunks of `chunk`;
 * running products across the row starting from Z(x_i); the row's last running product is Z(g x_i) and is swapped with
 * Z(x_i).  wires / sigmas: column-major [cols][n].  out: (num_chunks) columns [num_chunks][n] laid out as plonky2 returns
 * them: partial products 0 .. num_chunks-2, then Z. */
int orc_plonk_partial_products(const uint64_t *wires, const uint64_t *sigmas, const uint64_t *k_is, size_t routed,
                               size_t chunk, unsigned degree_bits, uint64_t beta, uint64_t gamma, uint64_t *out) {
    const size_t n = (size_t)1 << degree_bits, nch = (routed + chunk - 1) / chunk;
    const uint64_t w = gl_root_of_unity(degree_bits);
    uint64_t *q = (uint64_t *)malloc(sizeof(uint64_t) * n * nch);   /* chunk products, row-major */
    if (!q) return -1;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        const uint64_t x = gl_pow(w, i);
        for (size_t c = 0; c < nch; ++c) {
            uint64_t num = 1, den = 1;
            for (size_t j = c * chunk; j < routed && j < (c + 1) * chunk; ++j) {
                const uint64_t wv = gl_canon(wires[j * n + i]);
                num = gl_mul(num, gl_add(gl_add(wv, gl_mul(beta, gl_mul(k_is[j], x))), gamma));
                den = gl_mul(den, gl_add(gl_add(wv, gl_mul(beta, gl_canon(sigmas[j * n + i]))), gamma));
            }
            q[i * nch + c] = gl_mul(num, gl_inv(den));      /* product of quotients == quotient of products */
        }
    }
    uint64_t z = 1;
    for (size_t i = 0; i < n; ++i) {
        uint64_t acc = z;
        for (size_t c = 0; c < nch; ++c) {
            acc = gl_mul(acc, q[i * nch + c]);
            if (c + 1 < nch) out[c * n + i] = acc;
        }
        out[(nch - 1) * n + i] = z;   /* swap(z_x, last): the column holds Z(x_i) */
        z = acc;
    }
    free(q);
    return 0;
}

/* [EXT] plonk/prover.rs `compute_quotient_polys` up to (excluding) the coset_ifft.  One tape evaluates every
 * vanishing-polynomial term of `eval_vanishing_poly_base_batch` at one point; its inputs are
 *   constants_sigmas row (C0) | wires row (C1) | zs/partial-products row (C2) | the same oracle's row at i_next (C2) |
 *   x (the shifted point) | L_0(x)
 * leaves: [N][cols] row-major, rows bit-reversed (orc_commit_values).  out[c][i] = (sum_k alpha_c^k term_k) / Z_H(x_i)
 * (`reduce_with_powers_multi`: the FIRST term carries alpha^0). */
int orc_plonk_quotient_values(const uint32_t *ops, size_t n_ops, const uint64_t *consts, size_t n_consts,
                              const uint32_t *out_nodes, size_t n_terms, const uint64_t *l0, size_t C0,
                              const uint64_t *l1, size_t C1, const uint64_t *l2, size_t C2, unsigned degree_bits,
                              unsigned rate_bits, unsigned qdb, const uint64_t *alphas, size_t n_alphas,
                              uint64_t *const *out) {
    const size_t n = (size_t)1 << degree_bits, size = n << qdb;
    const unsigned log_lde = degree_bits + rate_bits;
    const size_t step = (size_t)1 << (rate_bits - qdb), next_step = (size_t)1 << qdb;
    const size_t n_in = C0 + C1 + 2 * C2 + 2, n_nodes = n_in + n_consts + n_ops;
    const uint64_t w = gl_root_of_unity(degree_bits + qdb);
    const uint64_t n_f = gl_canon((uint64_t)n);
    int fail = 0;
#pragma omp parallel
    {
        uint64_t *v = (uint64_t *)malloc(sizeof(uint64_t) * n_nodes);
        uint64_t acc[16];
        if (!v || n_alphas > 16) {
#pragma omp atomic write
            fail = 1;
        } else {
            memcpy(v + n_in, consts, sizeof(uint64_t) * n_consts);
#pragma omp for schedule(static)
            for (size_t i = 0; i < size; ++i) {
                const uint64_t x = gl_mul(GL_GENERATOR, gl_pow(w, i));            /* shifted_x */
                const uint64_t zh = gl_sub(gl_pow(x, n), 1);                      /* ZeroPolyOnCoset::eval(i) */
                const uint64_t l0x = gl_mul(zh, gl_inv(gl_mul(n_f, gl_sub(x, 1)))); /* eval_l_0(i, x) */
                const size_t i_next = (i + next_step) % size;
                const size_t r = bitrev(i * step, log_lde), rn = bitrev(i_next * step, log_lde);
                uint64_t *p = v;
                memcpy(p, l0 + r * C0, 8 * C0); p += C0;
                memcpy(p, l1 + r * C1, 8 * C1); p += C1;
                memcpy(p, l2 + r * C2, 8 * C2); p += C2;
                memcpy(p, l2 + rn * C2, 8 * C2); p += C2;
                p[0] = x; p[1] = l0x;
                tape_run(v, ops, n_ops, n_in + n_consts);
                for (size_t c = 0; c < n_alphas; ++c) acc[c] = 0;
                for (size_t k = n_terms; k-- > 0;) {
                    const uint64_t t = v[out_nodes[k]];
                    for (size_t c = 0; c < n_alphas; ++c) acc[c] = gl_add(gl_mul(acc[c], alphas[c]), t);
                }
                const uint64_t zinv = gl_inv(zh);
                for (size_t c = 0; c < n_alphas; ++c) out[c][i] = gl_mul(acc[c], zinv);
            }
        }
        free(v);
    }
    return fail ? -1 : 0;
}
