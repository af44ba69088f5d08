/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 * CPU restatement of the reference hot path (plonky2/starky 1.0.0 behaviour behind
 * evm_arithmetization/src/prover.rs:100,137,322).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product (zk_evm_amd/) never does.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_HASH_POSEIDON = 0, ORC_HASH_KECCAK25 = 1 };

/* field */
uint64_t orc_gl_add(uint64_t a, uint64_t b);
uint64_t orc_gl_sub(uint64_t a, uint64_t b);
uint64_t orc_gl_mul(uint64_t a, uint64_t b);
uint64_t orc_gl_inv(uint64_t a);
uint64_t orc_gl_pow(uint64_t a, uint64_t e);
uint64_t orc_gl_root_of_unity(unsigned log_n);
void orc_gl2_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);
void orc_gl2_inv(const uint64_t a[2], uint64_t out[2]);

/* hashes */
void orc_poseidon_permute(uint64_t st[12]);
void orc_poseidon_permute_auto(uint64_t st[12]);   /* the fast or the plain evaluation, whichever the hashes use */
/* the same permutation, blocked schedule + lazy arithmetic (poseidon_fast.c); orc_poseidon_use_fast: which one the hashes use */
void orc_poseidon_permute_fast(uint64_t st[12]);
void orc_poseidon_use_fast(int on);
double orc_poseidon_perms_per_second(int fast, size_t n);
void orc_poseidon_hash_no_pad(const uint64_t *in, size_t n, uint64_t out[4]);
void orc_poseidon_hash_or_noop(const uint64_t *in, size_t n, uint64_t out[4]);
void orc_poseidon_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]);
void orc_keccak_f1600(uint64_t a[25]);
void orc_keccak256(const uint8_t *in, size_t len, uint8_t out[32]);
void orc_keccak25_hash_no_pad(const uint64_t *in, size_t n, uint8_t out[32]);
void orc_keccak25_hash_or_noop(const uint64_t *in, size_t n, uint8_t out[32]);
void orc_keccak25_two_to_one(const uint8_t l[32], const uint8_t r[32], uint8_t out[32]);

/* NTT */
void orc_fft(uint64_t *a, unsigned log_n);
void orc_ifft(uint64_t *a, unsigned log_n);
void orc_coset_fft(uint64_t *a, unsigned log_n, uint64_t shift);
void orc_coset_ifft(uint64_t *a, unsigned log_n, uint64_t shift);
void orc_lde(const uint64_t *coeffs, unsigned log_n, unsigned rate_bits, uint64_t *out);
uint64_t orc_eval_poly(const uint64_t *coeffs, size_t n, uint64_t x);
void orc_eval_poly_ext(const uint64_t *coeffs, size_t n, const uint64_t x[2], uint64_t out[2]);

/* Merkle tree over N = 2^log_leaves leaves of `leaf_len` elements (row-major).
 * digests: all levels concatenated, 32-byte slots, level 0 (N leaf digests) first, up to and
 * including the cap level (2^cap_height entries).  Total slots = orc_merkle_num_digests(). */
size_t orc_merkle_num_digests(unsigned log_leaves, unsigned cap_height);
void orc_merkle_build(const uint64_t *leaves, unsigned log_leaves, size_t leaf_len,
                      unsigned cap_height, int hasher, uint64_t *digests);
/* siblings bottom-up, (log_leaves - cap_height) 32-byte slots ([EXT] MerkleTree::prove) */
void orc_merkle_prove(const uint64_t *digests, unsigned log_leaves, unsigned cap_height,
                      size_t leaf_index, uint64_t *siblings);
/* [EXT] merkle_proofs.rs `verify_merkle_proof_to_cap`; returns 1 if ok */
int orc_merkle_verify(const uint64_t *leaf, size_t leaf_len, size_t leaf_index,
                      const uint64_t *siblings, unsigned n_siblings, const uint64_t *cap,
                      int hasher);

/* PolynomialBatch::from_values (blinding = false).
 * values: n_cols columns, column c at values + c*n (n = 2^log_n), any u64 representatives.
 * coeffs_out: [n_cols][n]; leaves_out: [N][n_cols] row-major, bit-reversed row order
 * (N = n << rate_bits); digests_out: orc_merkle_num_digests(log_n+rate_bits, cap_height) slots;
 * the cap is the last 2^cap_height slots of digests_out. Any *_out may be NULL. */
void orc_commit_values(const uint64_t *values, size_t n_cols, unsigned log_n, unsigned rate_bits,
                       unsigned cap_height, int hasher, uint64_t *coeffs_out,
                       uint64_t *leaves_out, uint64_t *digests_out, uint64_t *cap_out);
/* PolynomialBatch::from_coeffs */
void orc_commit_coeffs(const uint64_t *coeffs, size_t n_cols, unsigned log_n, unsigned rate_bits,
                       unsigned cap_height, int hasher, uint64_t *leaves_out,
                       uint64_t *digests_out, uint64_t *cap_out);

/* ---- Challenger (plonky2 iop/challenger.rs) ------------------------------------------------ */
typedef struct {
    int hasher;
    uint64_t state[12];
    uint64_t in[8];
    int n_in;
    uint64_t out[8];
    int n_out;
} orc_challenger;
void orc_challenger_init(orc_challenger *c, int hasher);
void orc_challenger_observe(orc_challenger *c, const uint64_t *e, size_t n);
void orc_challenger_observe_cap(orc_challenger *c, const uint64_t *slots, size_t n_digests);
uint64_t orc_challenger_get(orc_challenger *c);
void orc_challenger_get_ext(orc_challenger *c, uint64_t out[2]);
void orc_challenger_compact(orc_challenger *c, uint64_t out_state[12]);
void orc_hash_to_elements(int hasher, const uint64_t *slot, uint64_t out[4]);

/* ---- FRI (plonky2 fri/{oracle,prover,verifier,reduction_strategies}.rs) -------------------- */
typedef struct {
    uint32_t rate_bits, cap_height, hasher, num_challenges, proof_of_work_bits, num_query_rounds,
        arity_bits, final_poly_bits;
} orc_cfg; /* same layout as zk_cfg */

/* a committed PolynomialBatch as produced by orc_commit_values/orc_commit_coeffs */
typedef struct {
    size_t n_cols;
    unsigned log_n;
    const uint64_t *coeffs;  /* [n_cols][n] natural order */
    const uint64_t *leaves;  /* [N][n_cols] row-major, bit-reversed rows */
    const uint64_t *digests; /* level-concatenated */
} orc_batch;

/* FriInstanceInfo: batches of (point, [(oracle, poly)]) */
typedef struct {
    uint64_t point[2];
    size_t n_polys;
    const uint32_t *oracle_idx;
    const uint32_t *poly_idx;
} orc_fri_batch;

/* [EXT] FriReductionStrategy::ConstantArityBits -> reduction_arity_bits; returns count */
size_t orc_fri_reduction_arity_bits(unsigned degree_bits, const orc_cfg *cfg, uint32_t *out, size_t max);
/* number of u64 words of the flat FriProof layout documented in include/zkstark.h */
size_t orc_fri_proof_words(const orc_cfg *cfg, unsigned degree_bits, const size_t *oracle_cols, size_t n_oracles);
/* evaluate every polynomial of every batch at the batch point: out = concatenated ext values */
void orc_fri_openings(const orc_batch *oracles, const orc_fri_batch *batches, size_t n_batches, uint64_t *out);
/* PolynomialBatch::prove_openings: advances the challenger, writes the flat proof. */
void orc_fri_prove_openings(const orc_cfg *cfg, unsigned degree_bits, const orc_batch *oracles,
                            size_t n_oracles, const orc_fri_batch *batches, size_t n_batches,
                            orc_challenger *ch, uint64_t *proof_out);
/* verify_fri_proof restatement.  `ch` must be in the state right after the openings were observed
 * (the function re-derives alpha, betas, pow response and query indices).  caps[k] = cap slots of
 * oracle k; openings = orc_fri_openings layout.  Returns 1 if the proof verifies, 0 and a reason
 * code in *why otherwise. */
int orc_fri_verify(const orc_cfg *cfg, unsigned degree_bits, const size_t *oracle_cols, size_t n_oracles,
                   const uint64_t *const *caps, const orc_fri_batch *batches, size_t n_batches,
                   const uint64_t *openings, orc_challenger *ch, const uint64_t *proof, int *why);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
