/*
 * oracle/fri.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * Restatement of plonky2 1.0.0 FRI as starky drives it (reached from the reference at
 * evm_arithmetization/src/prover.rs:322 -> starky `prove_with_commitment` ->
 * `PolynomialBatch::prove_openings`), following [EXT]:
 *   fri/reduction_strategies.rs  ConstantArityBits(arity_bits, final_poly_bits)
 *   fri/oracle.rs                prove_openings: alpha; per batch reduce_polys_base, divide_by_linear,
 *                                shift_poly; lde; coset_fft(g)
 *   fri/prover.rs                fri_committed_trees, fri_proof_of_work, fri_prover_query_rounds
 *   fri/verifier.rs              verify_fri_proof, fri_combine_initial, compute_evaluation
 *   plonk/get_challenges / fri/challenges.rs  get_fri_challenges (order of observations)
 * Everything is done the reference's way (coefficient domain, sequential synthetic division) --
 * the HIP implementation works in the value domain instead, so agreement is a real check.
 * PoW: the reference searches with rayon `find_any` (non-deterministic witness); the
 * oracle and the product both return the SMALLEST valid witness.
 * No reference golden vector pins these orders ("parity unpinned"); orc_fri_verify restates the
 * verifier so that prover/verifier consistency at least is checked.
 */
#include "goldilocks.h"
#include "oracle.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

size_t orc_fri_reduction_arity_bits(unsigned degree_bits, const orc_cfg *cfg, uint32_t *out, size_t max) {
    size_t k = 0;
    unsigned d = degree_bits;
    while (d > cfg->final_poly_bits && d + cfg->rate_bits >= cfg->cap_height + cfg->arity_bits) {
        if (out && k < max) out[k] = cfg->arity_bits;
        ++k;
        d -= cfg->arity_bits;
    }
    return k;
}

/* ---- flat proof layout (documented in include/zkstark.h) ---------------------------------- */
typedef struct {
    size_t R, cap_len, Q, K, F;
    unsigned log_N, cap_height;
    uint32_t arity[32];
    size_t cols[64];
    size_t off_caps, off_final, off_pow, off_queries, query_words, total;
} fri_layout;

static void layout_init(fri_layout *L, const orc_cfg *cfg, unsigned degree_bits, const size_t *oracle_cols,
                        size_t n_oracles) {
    memset(L, 0, sizeof *L);
    L->R = orc_fri_reduction_arity_bits(degree_bits, cfg, L->arity, 32);
    L->cap_height = cfg->cap_height;
    L->cap_len = (size_t)1 << cfg->cap_height;
    L->Q = cfg->num_query_rounds;
    L->K = n_oracles;
    L->log_N = degree_bits + cfg->rate_bits;
    unsigned d = degree_bits;
    for (size_t r = 0; r < L->R; ++r) d -= L->arity[r];
    L->F = (size_t)1 << d;
    for (size_t k = 0; k < n_oracles; ++k) L->cols[k] = oracle_cols[k];
    size_t o = 6 + L->R + L->K;
    L->off_caps = o; o += L->R * L->cap_len * 4;
    L->off_final = o; o += L->F * 2;
    L->off_pow = o; o += 1;
    L->off_queries = o;
    size_t qw = 0;
    for (size_t k = 0; k < L->K; ++k) qw += L->cols[k] + 4 * (size_t)(L->log_N - L->cap_height);
    unsigned lg = L->log_N;
    for (size_t r = 0; r < L->R; ++r) {
        lg -= L->arity[r];
        qw += 2 * ((size_t)1 << L->arity[r]) + 4 * (size_t)(lg - L->cap_height);
    }
    L->query_words = qw;
    L->total = o + qw * L->Q;
}

size_t orc_fri_proof_words(const orc_cfg *cfg, unsigned degree_bits, const size_t *oracle_cols, size_t n_oracles) {
    fri_layout L;
    layout_init(&L, cfg, degree_bits, oracle_cols, n_oracles);
    return L.total;
}

static void write_header(const fri_layout *L, uint64_t *p) {
    p[0] = L->R; p[1] = L->cap_len; p[2] = L->Q; p[3] = L->K; p[4] = L->F; p[5] = L->log_N;
    for (size_t r = 0; r < L->R; ++r) p[6 + r] = L->arity[r];
    for (size_t k = 0; k < L->K; ++k) p[6 + L->R + k] = L->cols[k];
}

/* ---- openings ------------------------------------------------------------------------------ */
/* One Horner evaluation per opened polynomial, independent of one another: the polynomials of a batch are dealt over the
 * threads (plonky2 evaluates them under rayon the same way); every value is what the serial loop computes. */
void orc_fri_openings(const orc_batch *oracles, const orc_fri_batch *batches, size_t n_batches, uint64_t *out) {
    for (size_t b = 0; b < n_batches; ++b) {
        const long np = (long)batches[b].n_polys;
#pragma omp parallel for schedule(dynamic, 8)
        for (long k = 0; k < np; ++k) {
            const orc_batch *o = &oracles[batches[b].oracle_idx[k]];
            size_t n = (size_t)1 << o->log_n;
            orc_eval_poly_ext(o->coeffs + (size_t)batches[b].poly_idx[k] * n, n, batches[b].point, out + 2 * k);
        }
        out += 2 * (size_t)np;
    }
}

/* ---- helpers ------------------------------------------------------------------------------- */
static void ext_coset_fft(gl2_t *a, unsigned log_n, uint64_t shift) {
    size_t n = (size_t)1 << log_n;
    uint64_t *t0 = (uint64_t *)malloc(8 * n), *t1 = (uint64_t *)malloc(8 * n);
    for (size_t i = 0; i < n; ++i) { t0[i] = a[i].c[0]; t1[i] = a[i].c[1]; }
    orc_coset_fft(t0, log_n, shift);
    orc_coset_fft(t1, log_n, shift);
    for (size_t i = 0; i < n; ++i) { a[i].c[0] = t0[i]; a[i].c[1] = t1[i]; }
    free(t0); free(t1);
}

static void observe_ext(orc_challenger *ch, gl2_t e) { orc_challenger_observe(ch, e.c, 2); }
static gl2_t get_ext(orc_challenger *ch) { gl2_t r; orc_challenger_get_ext(ch, r.c); return r; }

/* leaves of a FRI commit-phase tree: bit-reversed values, chunks of `arity`, flattened */
static uint64_t *fri_tree_leaves(const gl2_t *values, unsigned log_n, unsigned arity_bits) {
    size_t n = (size_t)1 << log_n;
    uint64_t *leaves = (uint64_t *)malloc(16 * n);
    for (size_t i = 0; i < n; ++i) {
        gl2_t v = values[bitrev(i, log_n)];
        leaves[2 * i] = v.c[0];
        leaves[2 * i + 1] = v.c[1];
    }
    (void)arity_bits;
    return leaves;  /* leaf k = words [2*arity*k, 2*arity*(k+1)) */
}

/* ---- prove_openings ------------------------------------------------------------------------ */
void orc_fri_prove_openings(const orc_cfg *cfg, unsigned degree_bits, const orc_batch *oracles,
                            size_t n_oracles, const orc_fri_batch *batches, size_t n_batches,
                            orc_challenger *ch, uint64_t *proof) {
    size_t cols[64];
    for (size_t k = 0; k < n_oracles; ++k) cols[k] = oracles[k].n_cols;
    fri_layout L;
    layout_init(&L, cfg, degree_bits, cols, n_oracles);
    write_header(&L, proof);
    const size_t n = (size_t)1 << degree_bits;
    const unsigned log_N = L.log_N;
    const size_t N = (size_t)1 << log_N;

    /* [EXT] oracle.rs prove_openings */
    gl2_t alpha = get_ext(ch);
    gl2_t *final_poly = (gl2_t *)calloc(N, sizeof(gl2_t));
    gl2_t *comp = (gl2_t *)malloc(n * sizeof(gl2_t));
    for (size_t b = 0; b < n_batches; ++b) {
        /* composition = sum_k alpha^k f_k: the same sum in the same order of k for every coefficient i, the coefficients in
         * blocks over the threads (a block of `comp` stays in the cache while the columns stream past once) */
        const size_t np = batches[b].n_polys;
        gl2_t *apows = (gl2_t *)malloc((np + 1) * sizeof(gl2_t));
        apows[0] = gl2_from(1);
        for (size_t k = 0; k < np; ++k) apows[k + 1] = gl2_mul(apows[k], alpha);
        const gl2_t apow = apows[np];
        const size_t blk = 2048;
#pragma omp parallel for schedule(static)
        for (long i0 = 0; i0 < (long)n; i0 += (long)blk) {
            const size_t i1 = (size_t)i0 + blk < n ? (size_t)i0 + blk : n;
            for (size_t i = (size_t)i0; i < i1; ++i) comp[i] = gl2_from(0);
            for (size_t k = 0; k < np; ++k) {
                const orc_batch *o = &oracles[batches[b].oracle_idx[k]];
                const uint64_t *c = o->coeffs + (size_t)batches[b].poly_idx[k] * n;
                const gl2_t a = apows[k];
                for (size_t i = (size_t)i0; i < i1; ++i) comp[i] = gl2_add(comp[i], gl2_scale(a, c[i]));
            }
        }
        free(apows);
        /* divide_by_linear(point): Horner scan from the top; drop the remainder; pad with 0 */
        gl2_t z = {{gl_canon(batches[b].point[0]), gl_canon(batches[b].point[1])}};
        gl2_t acc = gl2_from(0);
        gl2_t *bs = (gl2_t *)malloc(n * sizeof(gl2_t));
        for (size_t i = n; i-- > 0;) { acc = gl2_add(gl2_mul(acc, z), comp[i]); bs[i] = acc; }
        /* bs[i] = sum_{j>=i} c_j z^(j-i); quotient coeff i = bs[i+1]; bs[0] is the remainder */
        /* shift_poly: final *= alpha^count, count = polys of THIS batch; then += quotient */
        for (size_t i = 0; i < n; ++i) {
            gl2_t q = i + 1 < n ? bs[i + 1] : gl2_from(0);
            final_poly[i] = gl2_add(gl2_mul(final_poly[i], apow), q);
        }
        free(bs);
    }
    free(comp);
    /* lde(rate_bits) + coset_fft(g): coeffs stay in final_poly, values computed separately */
    gl2_t *coeffs = (gl2_t *)malloc(N * sizeof(gl2_t));
    gl2_t *values = (gl2_t *)malloc(N * sizeof(gl2_t));
    memcpy(coeffs, final_poly, N * sizeof(gl2_t));
    memcpy(values, final_poly, N * sizeof(gl2_t));
    free(final_poly);
    ext_coset_fft(values, log_N, GL_GENERATOR);

    /* [EXT] prover.rs fri_committed_trees */
    uint64_t **tree_leaves = (uint64_t **)calloc(L.R + 1, sizeof(void *));
    uint64_t **tree_digests = (uint64_t **)calloc(L.R + 1, sizeof(void *));
    unsigned lg = log_N;
    uint64_t shift = GL_GENERATOR;
    for (size_t r = 0; r < L.R; ++r) {
        unsigned ab = L.arity[r];
        size_t arity = (size_t)1 << ab, cur = (size_t)1 << lg;
        tree_leaves[r] = fri_tree_leaves(values, lg, ab);
        unsigned log_leaves = lg - ab;
        size_t nd = orc_merkle_num_digests(log_leaves, cfg->cap_height);
        tree_digests[r] = (uint64_t *)malloc(32 * nd);
        orc_merkle_build(tree_leaves[r], log_leaves, 2 * arity, cfg->cap_height, cfg->hasher, tree_digests[r]);
        const uint64_t *cap = tree_digests[r] + 4 * (nd - L.cap_len);
        memcpy(proof + L.off_caps + r * L.cap_len * 4, cap, 32 * L.cap_len);
        orc_challenger_observe_cap(ch, cap, L.cap_len);
        gl2_t beta = get_ext(ch);
        /* coeffs'[k] = sum_i beta^i coeffs[arity*k + i] */
        size_t nxt = cur >> ab;
        /* (in place: coeffs[k] is written from coeffs[arity k ..], which for k >= 1 lie above it -- not a loop for several threads) */
        for (size_t k = 0; k < nxt; ++k) {
            gl2_t acc = gl2_from(0);
            for (size_t i = arity; i-- > 0;) acc = gl2_add(gl2_mul(acc, beta), coeffs[arity * k + i]);
            coeffs[k] = acc;
        }
        for (unsigned i = 0; i < ab; ++i) shift = gl_sqr(shift);
        lg -= ab;
        memcpy(values, coeffs, nxt * sizeof(gl2_t));
        ext_coset_fft(values, lg, shift);
    }
    /* truncate to len >> rate_bits, observe */
    size_t flen = ((size_t)1 << lg) >> cfg->rate_bits;
    for (size_t i = 0; i < flen; ++i) {
        proof[L.off_final + 2 * i] = coeffs[i].c[0];
        proof[L.off_final + 2 * i + 1] = coeffs[i].c[1];
        observe_ext(ch, coeffs[i]);
    }
    free(coeffs); free(values);

    /* [EXT] prover.rs fri_proof_of_work (smallest witness) */
    {
        uint64_t inter[12];
        memcpy(inter, ch->state, sizeof inter);
        for (int i = 0; i < ch->n_in; ++i) inter[i] = ch->in[i];
        int pos = ch->n_in;
        /* candidates in blocks over the threads (plonky2 grinds under rayon); the SMALLEST valid witness of the first block
         * that holds one is the smallest overall */
        uint64_t w = 0;
        for (uint64_t base = 0;; base += 4096) {
            uint64_t best = UINT64_MAX;
#pragma omp parallel for schedule(static) reduction(min : best)
            for (long d = 0; d < 4096; ++d) {
                const uint64_t cand = base + (uint64_t)d;
                orc_challenger tmp = *ch;
                memcpy(tmp.state, inter, sizeof inter);
                tmp.state[pos] = cand;
                tmp.n_in = 0;
                if (tmp.hasher == ORC_HASH_POSEIDON) orc_poseidon_permute_auto(tmp.state);
                else { /* reuse the challenger's own permutation through a duplex */
                    orc_challenger t2 = *ch;
                    t2.n_out = 0;
                    orc_challenger_observe(&t2, &cand, 1);
                    (void)orc_challenger_get(&t2);
                    memcpy(tmp.state, t2.state, sizeof inter);
                }
                uint64_t resp = tmp.state[7];
                int lz = resp ? __builtin_clzll(resp) : 64;
                if ((unsigned)lz >= cfg->proof_of_work_bits && cand < best) best = cand;
            }
            if (best != UINT64_MAX) { w = best; break; }
        }
        proof[L.off_pow] = w;
        orc_challenger_observe(ch, &w, 1);
        (void)orc_challenger_get(ch); /* pow response */
    }

    /* [EXT] prover.rs fri_prover_query_rounds */
    uint64_t *rands = (uint64_t *)malloc(8 * L.Q);
    for (size_t q = 0; q < L.Q; ++q) rands[q] = orc_challenger_get(ch);
    for (size_t q = 0; q < L.Q; ++q) {
        size_t x = (size_t)(rands[q] % N);
        uint64_t *w = proof + L.off_queries + q * L.query_words;
        for (size_t k = 0; k < n_oracles; ++k) {
            memcpy(w, oracles[k].leaves + x * oracles[k].n_cols, 8 * oracles[k].n_cols);
            w += oracles[k].n_cols;
            orc_merkle_prove(oracles[k].digests, log_N, cfg->cap_height, x, w);
            w += 4 * (size_t)(log_N - cfg->cap_height);
        }
        unsigned lgr = log_N;
        for (size_t r = 0; r < L.R; ++r) {
            unsigned ab = L.arity[r];
            size_t arity = (size_t)1 << ab;
            size_t leaf = x >> ab;
            memcpy(w, tree_leaves[r] + 2 * arity * leaf, 16 * arity);
            w += 2 * arity;
            lgr -= ab;
            orc_merkle_prove(tree_digests[r], lgr, cfg->cap_height, leaf, w);
            w += 4 * (size_t)(lgr - cfg->cap_height);
            x = leaf;
        }
    }
    free(rands);
    for (size_t r = 0; r < L.R; ++r) { free(tree_leaves[r]); free(tree_digests[r]); }
    free(tree_leaves); free(tree_digests);
}

/* ---- verifier ------------------------------------------------------------------------------ */
static gl2_t ext_pow_small(gl2_t b, size_t e) { return gl2_pow(b, (uint64_t)e); }

/* [EXT] verifier.rs compute_evaluation: interpolate {(x*g^i, evals_rev[i])} and evaluate at beta */
static gl2_t compute_evaluation(uint64_t x, size_t idx_in_coset, unsigned arity_bits, const gl2_t *evals_in, gl2_t beta) {
    size_t arity = (size_t)1 << arity_bits;
    gl2_t evals[64];
    for (size_t i = 0; i < arity; ++i) evals[i] = evals_in[bitrev(i, arity_bits)];
    size_t rev = bitrev(idx_in_coset, arity_bits);
    uint64_t g = gl_root_of_unity(arity_bits);
    uint64_t coset_start = gl_mul(x, gl_pow(g, arity - rev));
    uint64_t pts[64];
    uint64_t y = 1;
    for (size_t i = 0; i < arity; ++i) { pts[i] = gl_mul(coset_start, y); y = gl_mul(y, g); }
    /* Lagrange: sum_i evals[i] * prod_{j!=i} (beta - p_j)/(p_i - p_j) */
    gl2_t sum = gl2_from(0);
    for (size_t i = 0; i < arity; ++i) {
        gl2_t num = gl2_from(1);
        uint64_t den = 1;
        for (size_t j = 0; j < arity; ++j) {
            if (j == i) continue;
            num = gl2_mul(num, gl2_sub(beta, gl2_from(pts[j])));
            den = gl_mul(den, gl_sub(pts[i], pts[j]));
        }
        sum = gl2_add(sum, gl2_mul(evals[i], gl2_scale(num, gl_inv(den))));
    }
    return sum;
}

int orc_fri_verify(const orc_cfg *cfg, unsigned degree_bits, const size_t *oracle_cols, size_t n_oracles,
                   const uint64_t *const *caps, const orc_fri_batch *batches, size_t n_batches,
                   const uint64_t *openings, orc_challenger *ch, const uint64_t *proof, int *why) {
    int dummy;
    if (!why) why = &dummy;
    fri_layout L;
    layout_init(&L, cfg, degree_bits, oracle_cols, n_oracles);
    *why = 1;
    if (proof[0] != L.R || proof[1] != L.cap_len || proof[2] != L.Q || proof[3] != L.K || proof[4] != L.F ||
        proof[5] != L.log_N)
        return 0;
    const unsigned log_N = L.log_N;
    const size_t N = (size_t)1 << log_N;
    /* get_fri_challenges */
    gl2_t alpha = get_ext(ch);
    gl2_t betas[32];
    for (size_t r = 0; r < L.R; ++r) {
        orc_challenger_observe_cap(ch, proof + L.off_caps + r * L.cap_len * 4, L.cap_len);
        betas[r] = get_ext(ch);
    }
    orc_challenger_observe(ch, proof + L.off_final, 2 * L.F);
    orc_challenger_observe(ch, proof + L.off_pow, 1);
    uint64_t pow_resp = orc_challenger_get(ch);
    *why = 2;
    if ((unsigned)(pow_resp ? __builtin_clzll(pow_resp) : 64) < cfg->proof_of_work_bits) return 0;
    uint64_t *xs = (uint64_t *)malloc(8 * L.Q);
    for (size_t q = 0; q < L.Q; ++q) xs[q] = orc_challenger_get(ch) % N;

    /* PrecomputedReducedOpenings: per batch sum_k alpha^k y_k */
    gl2_t reduced[16];
    {
        const uint64_t *o = openings;
        for (size_t b = 0; b < n_batches; ++b) {
            gl2_t acc = gl2_from(0);
            for (size_t k = batches[b].n_polys; k-- > 0;) {
                gl2_t y = {{o[2 * k], o[2 * k + 1]}};
                acc = gl2_add(gl2_mul(acc, alpha), y);
            }
            reduced[b] = acc;
            o += 2 * batches[b].n_polys;
        }
    }
    int ok = 1;
    for (size_t q = 0; q < L.Q && ok; ++q) {
        size_t x = (size_t)xs[q];
        const uint64_t *w = proof + L.off_queries + q * L.query_words;
        const uint64_t *leafs[64];
        for (size_t k = 0; k < n_oracles; ++k) {
            leafs[k] = w;
            const uint64_t *sib = w + L.cols[k];
            *why = 3;
            if (!orc_merkle_verify(w, L.cols[k], x, sib, log_N - cfg->cap_height, caps[k], cfg->hasher)) { ok = 0; break; }
            w = sib + 4 * (size_t)(log_N - cfg->cap_height);
        }
        if (!ok) break;
        uint64_t sub_x = gl_mul(GL_GENERATOR, gl_pow(gl_root_of_unity(log_N), bitrev(x, log_N)));
        /* fri_combine_initial */
        gl2_t sum = gl2_from(0);
        for (size_t b = 0; b < n_batches; ++b) {
            gl2_t acc = gl2_from(0);
            for (size_t k = batches[b].n_polys; k-- > 0;) {
                uint64_t e = leafs[batches[b].oracle_idx[k]][batches[b].poly_idx[k]];
                acc = gl2_add(gl2_mul(acc, alpha), gl2_from(e));
            }
            gl2_t numer = gl2_sub(acc, reduced[b]);
            gl2_t pt = {{gl_canon(batches[b].point[0]), gl_canon(batches[b].point[1])}};
            gl2_t denom = gl2_sub(gl2_from(sub_x), pt);
            sum = gl2_mul(sum, ext_pow_small(alpha, batches[b].n_polys));
            sum = gl2_add(sum, gl2_mul(numer, gl2_inv(denom)));
        }
        gl2_t old_eval = sum;
        unsigned lg = log_N;
        for (size_t r = 0; r < L.R; ++r) {
            unsigned ab = L.arity[r];
            size_t arity = (size_t)1 << ab;
            gl2_t evals[64];
            for (size_t i = 0; i < arity; ++i) { evals[i].c[0] = w[2 * i]; evals[i].c[1] = w[2 * i + 1]; }
            size_t coset_index = x >> ab, within = x & (arity - 1);
            *why = 4;
            if (evals[within].c[0] != old_eval.c[0] || evals[within].c[1] != old_eval.c[1]) { ok = 0; break; }
            old_eval = compute_evaluation(sub_x, within, ab, evals, betas[r]);
            lg -= ab;
            *why = 5;
            if (!orc_merkle_verify(w, 2 * arity, coset_index, w + 2 * arity, lg - cfg->cap_height,
                                   proof + L.off_caps + r * L.cap_len * 4, cfg->hasher)) { ok = 0; break; }
            w += 2 * arity + 4 * (size_t)(lg - cfg->cap_height);
            for (unsigned i = 0; i < ab; ++i) sub_x = gl_sqr(sub_x);
            x = coset_index;
        }
        if (!ok) break;
        /* final_poly.eval(subgroup_x) == old_eval */
        gl2_t acc = gl2_from(0);
        for (size_t i = L.F; i-- > 0;) {
            gl2_t c = {{proof[L.off_final + 2 * i], proof[L.off_final + 2 * i + 1]}};
            acc = gl2_add(gl2_scale(acc, sub_x), c);
        }
        *why = 6;
        if (acc.c[0] != old_eval.c[0] || acc.c[1] != old_eval.c[1]) ok = 0;
    }
    free(xs);
    if (ok) *why = 0;
    return ok;
}
