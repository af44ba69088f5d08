/*
 * oracle/merkle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * Restatement of plonky2 1.0.0 `MerkleTree::new(leaves, cap_height)`, `MerkleTree::prove` and
 * `verify_merkle_proof_to_cap` ([EXT] plonky2/src/hash/merkle_tree.rs, merkle_proofs.rs) and of
 * `PolynomialBatch::from_values / from_coeffs` ([EXT] plonky2/src/fri/oracle.rs), the function the
 * reference calls at evm_arithmetization/src/prover.rs:100-107 and verifier.rs:68-77:
 *   coeffs = ifft(values); lde = coset_fft(g, lde(coeffs, rate_bits)); leaves = transpose(lde);
 *   reverse_index_bits_in_place(leaves); tree = MerkleTree::new(leaves, cap_height).
 * Leaf digest = H::hash_or_noop(leaf); node = H::two_to_one(l, r); cap = the 2^cap_height subtree
 * roots in index order.  (Upstream stores digests in a subtree-interleaved array; that layout is
 * internal -- what a proof exposes is the cap and the bottom-up sibling list, which is what this
 * file reproduces.)
 */
#include "goldilocks.h"
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#include <stdio.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static void hash_leaf(int hasher, const uint64_t *leaf, size_t len, uint64_t out[4]) {
    if (hasher == ORC_HASH_POSEIDON) orc_poseidon_hash_or_noop(leaf, len, out);
    else orc_keccak25_hash_or_noop(leaf, len, (uint8_t *)out);
}
static void hash_node(int hasher, const uint64_t *l, const uint64_t *r, uint64_t out[4]) {
    if (hasher == ORC_HASH_POSEIDON) orc_poseidon_two_to_one(l, r, out);
    else orc_keccak25_two_to_one((const uint8_t *)l, (const uint8_t *)r, (uint8_t *)out);
}

size_t orc_merkle_num_digests(unsigned log_leaves, unsigned cap_height) {
    size_t tot = 0;
    for (unsigned l = log_leaves + 1; l-- > cap_height;) tot += (size_t)1 << l;
    return tot;
}

void orc_merkle_build(const uint64_t *leaves, unsigned log_leaves, size_t leaf_len,
                      unsigned cap_height, int hasher, uint64_t *digests) {
    size_t N = (size_t)1 << log_leaves;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < N; ++i) hash_leaf(hasher, leaves + i * leaf_len, leaf_len, digests + 4 * i);
    uint64_t *prev = digests;
    for (unsigned l = log_leaves; l-- > cap_height;) {
        size_t cnt = (size_t)1 << l;
        uint64_t *cur = prev + 4 * (cnt * 2);
#pragma omp parallel for schedule(static) if (cnt > 256)
        for (size_t i = 0; i < cnt; ++i) hash_node(hasher, prev + 8 * i, prev + 8 * i + 4, cur + 4 * i);
        prev = cur;
    }
}

void orc_merkle_prove(const uint64_t *digests, unsigned log_leaves, unsigned cap_height,
                      size_t leaf_index, uint64_t *siblings) {
    const uint64_t *lvl = digests;
    size_t idx = leaf_index;
    for (unsigned l = log_leaves; l > cap_height; --l) {
        memcpy(siblings, lvl + 4 * (idx ^ 1), 32);
        siblings += 4;
        lvl += 4 * ((size_t)1 << l);
        idx >>= 1;
    }
}

int orc_merkle_verify(const uint64_t *leaf, size_t leaf_len, size_t leaf_index,
                      const uint64_t *siblings, unsigned n_siblings, const uint64_t *cap,
                      int hasher) {
    uint64_t cur[4], nxt[4];
    hash_leaf(hasher, leaf, leaf_len, cur);
    size_t idx = leaf_index;
    for (unsigned k = 0; k < n_siblings; ++k) {
        if (idx & 1) hash_node(hasher, siblings + 4 * k, cur, nxt);
        else hash_node(hasher, cur, siblings + 4 * k, nxt);
        memcpy(cur, nxt, 32);
        idx >>= 1;
    }
    return memcmp(cur, cap + 4 * idx, 32) == 0;
}

void orc_commit_coeffs(const uint64_t *coeffs, size_t n_cols, unsigned log_n, unsigned rate_bits,
                       unsigned cap_height, int hasher, uint64_t *leaves_out,
                       uint64_t *digests_out, uint64_t *cap_out) {
    size_t n = (size_t)1 << log_n, N = n << rate_bits;
    unsigned log_N = log_n + rate_bits;
    uint64_t *leaves = leaves_out ? leaves_out : (uint64_t *)malloc(sizeof(uint64_t) * N * n_cols);
    size_t nd = orc_merkle_num_digests(log_N, cap_height);
    uint64_t *digests = digests_out ? digests_out : (uint64_t *)malloc(32 * nd);
    /* columns in groups of ORC_COL_GROUP: a leaf row's entries of one group are written together (one cache line touched
     * per leaf and group instead of one per leaf and column: the scattered 8-byte stores were half of this function) */
#define ORC_COL_GROUP 4
    const size_t n_groups = (n_cols + ORC_COL_GROUP - 1) / ORC_COL_GROUP;
    uint32_t *rev = (uint32_t *)malloc(sizeof(uint32_t) * N);
    for (size_t j = 0; j < N; ++j) rev[j] = (uint32_t)bitrev(j, log_N);
#pragma omp parallel
    {
        uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * N * ORC_COL_GROUP);
#pragma omp for schedule(dynamic)
        for (size_t g = 0; g < n_groups; ++g) {
            const size_t c0 = g * ORC_COL_GROUP, nc = n_cols - c0 < ORC_COL_GROUP ? n_cols - c0 : ORC_COL_GROUP;
            for (size_t k = 0; k < nc; ++k) orc_lde(coeffs + (c0 + k) * n, log_n, rate_bits, tmp + k * N);
            /* transpose + reverse_index_bits: leaf bitrev(j) holds the value at natural index j */
            for (size_t j = 0; j < N; ++j) {
                uint64_t *dst = leaves + (size_t)rev[j] * n_cols + c0;
                for (size_t k = 0; k < nc; ++k) dst[k] = tmp[k * N + j];
            }
        }
        free(tmp);
    }
    free(rev);
    {
        const char *tm = getenv("ORC_COMMIT_TIMING");
        static double t_last;
        if (tm && tm[0] == '1') t_last = omp_get_wtime();
        orc_merkle_build(leaves, log_N, n_cols, cap_height, hasher, digests);
        if (tm && tm[0] == '1') fprintf(stderr, "    of which orc_merkle_build %.3f s\n", omp_get_wtime() - t_last);
    }
    if (0) orc_merkle_build(leaves, log_N, n_cols, cap_height, hasher, digests);
    if (cap_out) memcpy(cap_out, digests + 4 * (nd - ((size_t)1 << cap_height)), 32 << cap_height);
    if (!leaves_out) free(leaves);
    if (!digests_out) free(digests);
}

void orc_commit_values(const uint64_t *values, size_t n_cols, unsigned log_n, unsigned rate_bits,
                       unsigned cap_height, int hasher, uint64_t *coeffs_out,
                       uint64_t *leaves_out, uint64_t *digests_out, uint64_t *cap_out) {
    size_t n = (size_t)1 << log_n;
    uint64_t *coeffs = coeffs_out ? coeffs_out : (uint64_t *)malloc(sizeof(uint64_t) * n * n_cols);
    const char *tm = getenv("ORC_COMMIT_TIMING");
    double t0 = omp_get_wtime();
    memcpy(coeffs, values, sizeof(uint64_t) * n * n_cols);
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < n_cols; ++c) orc_ifft(coeffs + c * n, log_n);
    double t1 = omp_get_wtime();
    orc_commit_coeffs(coeffs, n_cols, log_n, rate_bits, cap_height, hasher, leaves_out, digests_out,
                      cap_out);
    if (tm && tm[0] == '1') fprintf(stderr, "orc_commit_values %zu x 2^%u: copy + ifft %.3f s, lde + transpose + tree %.3f s\n", n_cols, log_n, t1 - t0, omp_get_wtime() - t1);
    if (!coeffs_out) free(coeffs);
}

/* the fast 128-bit reduction against its defining statement x % p; returns the number of mismatches */
size_t orc_gl_reduce128_check(const uint64_t *lo, const uint64_t *hi, size_t n) {
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) {
        u128 x = ((u128)hi[i] << 64) | lo[i];
        bad += gl_reduce128(x) != gl_reduce128_slow(x);
    }
    return bad;
}

/* single-core permutations per second of the evaluation the hashes use (fast != 0) or of the plain definition: bench.py
 * prints it next to cpu_baseline so a reader can place this oracle against a tuned CPU prover (~10^6 / s / core) */
double orc_poseidon_perms_per_second(int fast, size_t n) {
    uint64_t st[12] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12};
#ifdef _OPENMP
    double t0 = omp_get_wtime();
#else
    double t0 = 0;
#endif
    for (size_t i = 0; i < n; ++i) { if (fast) orc_poseidon_permute_fast(st); else orc_poseidon_permute(st); }
#ifdef _OPENMP
    double t1 = omp_get_wtime();
#else
    double t1 = 1;
#endif
    return st[0] == 0xFFFFFFFFFFFFFFFFULL ? 0.0 : (double)n / (t1 - t0);
}

/* thin exports of the inline field ops for python tests */
uint64_t orc_gl_add(uint64_t a, uint64_t b) { return gl_add(a, b); }
uint64_t orc_gl_sub(uint64_t a, uint64_t b) { return gl_sub(a, b); }
uint64_t orc_gl_mul(uint64_t a, uint64_t b) { return gl_mul(gl_canon(a), gl_canon(b)); }
uint64_t orc_gl_inv(uint64_t a) { return gl_inv(a); }
uint64_t orc_gl_pow(uint64_t a, uint64_t e) { return gl_pow(a, e); }
uint64_t orc_gl_root_of_unity(unsigned log_n) { return gl_root_of_unity(log_n); }
void orc_gl2_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {
    gl2_t x = {{gl_canon(a[0]), gl_canon(a[1])}}, y = {{gl_canon(b[0]), gl_canon(b[1])}};
    gl2_t r = gl2_mul(x, y); out[0] = r.c[0]; out[1] = r.c[1];
}
void orc_gl2_inv(const uint64_t a[2], uint64_t out[2]) {
    gl2_t x = {{gl_canon(a[0]), gl_canon(a[1])}};
    gl2_t r = gl2_inv(x); out[0] = r.c[0]; out[1] = r.c[1];
}
