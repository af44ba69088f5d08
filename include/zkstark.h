/*
 * zkstark.h -- C ABI of libzkstark_hip.so, the MI355X (gfx950) STARK commitment / proving backend.
 *
 * Drop-in boundary: the reference has no runtime plugin API for this
 * path; the seam is the crate boundary to plonky2/starky.  Each entry point below names the
 * reference call site it replaces (paths relative to the zk_evm checkout).  All pointers are plain
 * host or device pointers, no C++/torch types.  Every function returns 0 on success or a negative
 * zk_status; zk_last_error(ctx) gives a human-readable message for the last failure on that ctx.
 *
 * Conventions (parity-critical):
 *   - field elements are u64 Goldilocks representatives; inputs may be non-canonical (any u64),
 *     outputs are always canonical (< p = 2^64 - 2^32 + 1);
 *   - a "column" is a contiguous array of n = 2^log_n elements, values[i] = f(w^i) (natural order),
 *     exactly plonky2 `PolynomialValues<F>`;
 *   - a digest occupies a 32-byte slot: Poseidon = 4 x u64; Keccak-25 = 25 bytes + 7 zero bytes;
 *   - leaf index i of a committed batch is the LDE row at natural index bitrev(i) (plonky2's
 *     `reverse_index_bits_in_place` on the transposed LDE).
 */
#ifndef ZKSTARK_H
#define ZKSTARK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ZK_OK = 0,
    ZK_ERR_BAD_ARG = -1,
    ZK_ERR_OOM = -2,
    ZK_ERR_HIP = -3,
    ZK_ERR_ABORTED = -4,
    ZK_ERR_UNSUPPORTED = -5,
    ZK_ERR_COMM = -6,            /* a collective failed, timed out, or another rank of the job reported an error */
} zk_status;

typedef enum { ZK_HASH_POSEIDON = 0, ZK_HASH_KECCAK25 = 1 } zk_hasher;

/* Mirrors the fields of plonky2 `FriConfig` / starky `StarkConfig` the path consumes
 * (reference: StarkConfig::standard_fast_config() at zero/src/prover_state/mod.rs:283,
 * TEST_STARK_CONFIG at evm_arithmetization/src/testing_utils.rs:41-51). */
typedef struct {
    uint32_t rate_bits;          /* LDE blow-up = 2^rate_bits (1 in production).  degree_bits + rate_bits <= 28 in every commit /
                                  * quotient / shard entry point (ZK_ERR_UNSUPPORTED above: the NTT passes address their twiddle
                                  * tables through 32-bit buffer offsets); the reference's own ceiling is the field's
                                  * two-adicity, 32 -- its largest table, Memory, is capped at 2^22 + 1 (prove_stdio.rs:89-101) */
    uint32_t cap_height;         /* Merkle cap has 2^cap_height digests (4); any value up to the tree height */
    uint32_t hasher;             /* zk_hasher */
    uint32_t num_challenges;     /* 2 */
    uint32_t proof_of_work_bits; /* 16 */
    uint32_t num_query_rounds;   /* 84 */
    uint32_t arity_bits;         /* FRI ConstantArityBits(4, 5): arity_bits = 4 */
    uint32_t final_poly_bits;    /* ... final_poly_bits = 5 */
} zk_cfg;

typedef struct zk_ctx zk_ctx;     /* one per GPU / stream.  The library is re-entrant ACROSS contexts (no global state);
                                   * one zk_ctx must not be used by two threads at once: it owns the HBM arena free
                                   * list, the stream, the cached tables and the error string */
typedef struct zk_batch zk_batch; /* device-resident PolynomialBatch */

/* ---- context ---------------------------------------------------------------------------- */
int zk_ctx_create(int device, zk_ctx **out);
/* also frees every zk_batch of this ctx that is still alive (their handles become invalid) */
void zk_ctx_destroy(zk_ctx *ctx);
/* Run all work of this ctx on an existing hipStream_t (e.g. torch's current stream); NULL selects
 * the HIP default (null) stream.  A fresh ctx runs on its own non-blocking stream. */
int zk_ctx_set_stream(zk_ctx *ctx, void *hip_stream);
int zk_ctx_synchronize(zk_ctx *ctx);
/* Device memory.  All batch and scratch HBM of a ctx comes from a grow-only arena (csrc/arena.hpp): blocks
 * freed by zk_batch_free are handed out again in stream order instead of going back to the driver, because
 * multi-GB hipMalloc / hipMallocAsync calls cost 0.2-2.7 s each on this platform.  zk_ctx_set_stream drains the
 * old stream first.  reserve: make sure one free block of `bytes` exists (pre-size before a latency-critical
 * proof); trim: synchronise and hipFree every idle slab; stats: bytes held / handed out / high-water mark. */
int zk_ctx_mem_reserve(zk_ctx *ctx, size_t bytes);
int zk_ctx_mem_trim(zk_ctx *ctx, size_t *released);
int zk_ctx_mem_stats(const zk_ctx *ctx, size_t *reserved, size_t *in_use, size_t *peak_in_use);
/* Device buffers from the ctx arena for callers without a device-memory library of their own (the Rust shim; the Python
 * mirror uses torch tensors, tests/cabi/segment.c the HIP runtime directly).  zk_dev_upload_columns copies host columns
 * (`Vec<PolynomialValues<F>>`: cols[c] -> n elements) into a column-major device matrix d_out[c * col_stride + i] on the
 * ctx stream and returns once the host memory may be reused. */
int zk_dev_alloc(zk_ctx *ctx, size_t bytes, void **d_out);
int zk_dev_free(zk_ctx *ctx, void *d_ptr);
int zk_dev_upload_columns(zk_ctx *ctx, const uint64_t *const *cols, size_t n_cols, size_t n, uint64_t *d_out,
                          size_t col_stride);
const char *zk_last_error(const zk_ctx *ctx);
/* Cooperative cancellation, polled between kernels: mirrors `abort_signal` /
 * `check_abort_signal` (evm_arithmetization/src/prover.rs:56,346-354). NULL disables. */
int zk_ctx_set_abort_flag(zk_ctx *ctx, volatile const int *abort_flag);
/* The same signal as ONE BYTE, which is what the reference owns: `abort_signal: Option<Arc<AtomicBool>>`
 * (prover.rs:56) -- the Rust side passes `AtomicBool::as_ptr()` and a store from any thread (`zero`'s abort handler,
 * polled per table at fixed_recursive_verifier.rs:2123) is seen mid-proof.  Either flag aborts; NULL disables this one. */
int zk_ctx_set_abort_flag_u8(zk_ctx *ctx, volatile const uint8_t *abort_flag);
/* The plan table.  Three internal decisions have two implementations that produce the same words (the NTT passes of a
 * transform shape: LDS tile kernels | lane-swap kernels; from_values over all columns at once | in column batches; the small
 * Merkle levels of a segment's trace trees per tree | batched).  Which one serves a shape is DATA held by the ctx -- a string
 * of items "v20f0=2;d21f1=2;b20r1=96x1;T=1;" (csrc/ntt_host.inc) -- never a run-time experiment: a ctx starts with the
 * process's ZK_NTT_SWAP_PLANS (read once, when the library is loaded) or else the table compiled into the library, and this
 * call replaces it (NULL: back to that initial table; "": the first implementation everywhere).  The library starts no
 * process, runs no timing trial and never writes the environment; the offline tool `zk_ntt_tune` prints such a string for a
 * device.  No reference counterpart (the CPU prover has one FFT); results do not depend on the table.
 * zk_ctx_get_plans copies the current string (NUL-terminated, truncated to max) and returns its full length. */
int zk_ctx_set_plans(zk_ctx *ctx, const char *plans);
size_t zk_ctx_get_plans(const zk_ctx *ctx, char *out, size_t max);
/* Per-stage device timings of the last commit on this ctx, in ms, keyed like the reference's
 * TimingTree scopes (prover.rs:92,99): [0]=ifft [1]=lde/coset-fft [2]=leaf hash [3]=tree. */
int zk_ctx_last_timings(const zk_ctx *ctx, float out_ms[4]);
/* Running totals of the same four HIP-event stage timings over every commit (from_values / from_coeffs) since the
 * last reset, with the algorithmic work they cover: leaf-hash bytes (8*C*N read + 32*N written per commit),
 * Poseidon permutations of the leaf hashing, NTT bytes (40*C*n per from_values, 24*C*n per from_coeffs).
 * bench.py derives the leaf-hash kernel's roofline inside a timed multi-table proof from these. */
int zk_ctx_commit_totals(zk_ctx *ctx, double out_ms[4], uint64_t *n_commits, double *leaf_hash_bytes,
                         double *leaf_hash_perms, double *ntt_bytes, int reset);
/* The same totals for the commitments zk_prove_segment runs on the ctx's SIDE lane (auxiliary commitments, small
 * tables: a second, low-priority stream that overlaps the main one).  Their event-to-event times include the sharing of
 * the chip with main-lane kernels, so they are kept out of zk_ctx_commit_totals and of any roofline derived from it. */
int zk_ctx_side_commit_totals(zk_ctx *ctx, double out_ms[4], uint64_t *n_commits, double *leaf_hash_bytes,
                              double *ntt_bytes, int reset);

/* ---- PolynomialBatch::from_values / from_coeffs ------------------------------------------
 * Replaces plonky2 `PolynomialBatch::from_values(values, rate_bits, blinding=false, cap_height,
 * timing, None)` as called at evm_arithmetization/src/prover.rs:100-107, verifier.rs:68-77 and
 * keccak/keccak_stark.rs:718; `from_coeffs` is the form used for the quotient chunks
 * (starky prover, reached from prover.rs:322). */
/* host columns: cols[c] points at n = 2^log_n u64 (exactly Vec<PolynomialValues<F>>) */
int zk_commit_columns(zk_ctx *ctx, const zk_cfg *cfg, const uint64_t *const *cols, size_t n_cols,
                      unsigned log_n, zk_batch **out);
/* device-resident values: column c starts at d_values + c * col_stride (elements) */
int zk_commit_columns_device(zk_ctx *ctx, const zk_cfg *cfg, const uint64_t *d_values,
                             size_t col_stride, size_t n_cols, unsigned log_n, zk_batch **out);
/* device-resident natural-order coefficients */
int zk_commit_coeffs_device(zk_ctx *ctx, const zk_cfg *cfg, const uint64_t *d_coeffs,
                            size_t col_stride, size_t n_cols, unsigned log_n, zk_batch **out);
void zk_batch_free(zk_batch *b);

/* shape */
size_t zk_batch_num_cols(const zk_batch *b);
unsigned zk_batch_log_n(const zk_batch *b);
unsigned zk_batch_log_lde(const zk_batch *b);
/* `merkle_tree.cap` : 2^cap_height 32-byte slots -> host */
int zk_batch_cap(const zk_batch *b, uint64_t *out);
/* natural-order coefficients of column `col` (plonky2 `polynomials[col].coeffs`) -> host */
int zk_batch_coeffs(const zk_batch *b, size_t col, uint64_t *out);
/* `merkle_tree.leaves[leaf_index]` (n_cols elements) -> host */
int zk_batch_leaf(const zk_batch *b, size_t leaf_index, uint64_t *out);
/* `merkle_tree.prove(leaf_index).siblings`: (log_lde - cap_height) 32-byte slots, bottom-up */
int zk_batch_merkle_path(const zk_batch *b, size_t leaf_index, uint64_t *out);
/* `get_lde_values(index, step)` = leaves[bitrev(index * step)] -> host */
int zk_batch_lde_values(const zk_batch *b, size_t index, size_t step, uint64_t *out);
/* device views (valid until zk_batch_free):
 *   lde: [n_cols][N] natural row order, column-major;  digests: level-concatenated 32-B slots,
 *   level 0 = N leaf digests in leaf order, last 2^cap_height slots = cap. */
const uint64_t *zk_batch_lde_device(const zk_batch *b);
const uint64_t *zk_batch_digests_device(const zk_batch *b);

/* ---- primitive entry points (used by the parity tests and by later stages) ---------------
 * All operate on device memory, in place, on the ctx stream.  Orders are plonky2's:
 * zk_ifft / zk_fft   = `PolynomialValues::ifft` / `PolynomialCoeffs::fft` (natural <-> natural);
 * zk_coset_fft       = `coset_fft(shift)`; zk_coset_ifft = `coset_ifft(shift)`;
 * zk_lde             = `p.lde(rate_bits).coset_fft(F::coset_shift())`. */
int zk_ifft(zk_ctx *ctx, uint64_t *d_data, size_t col_stride, size_t n_cols, unsigned log_n);
int zk_fft(zk_ctx *ctx, uint64_t *d_data, size_t col_stride, size_t n_cols, unsigned log_n);
int zk_coset_fft(zk_ctx *ctx, uint64_t *d_data, size_t col_stride, size_t n_cols, unsigned log_n,
                 uint64_t shift);
int zk_coset_ifft(zk_ctx *ctx, uint64_t *d_data, size_t col_stride, size_t n_cols, unsigned log_n,
                  uint64_t shift);
int zk_lde(zk_ctx *ctx, const uint64_t *d_coeffs, size_t in_stride, uint64_t *d_out,
           size_t out_stride, size_t n_cols, unsigned log_n, unsigned rate_bits);
/* Element-wise Goldilocks vector op on device arrays: out[i] = a[i] (op) b[i], canonical output.
 * op: 0 = add, 1 = sub, 2 = mul, 3 = square (b ignored), 4 = inverse (b ignored; 0 -> 0);
 *     5 = the NTT's canonical-output multiply, stored WITHOUT a final canonicalisation (must be < p on its own),
 *     6 / 7 = a + b^2 / a - b^2 through the butterfly's one-correction add / sub.
 *     8 = the multiply of the Poseidon S-box (its first fold correction lives in an unlikely block).
 * (plonky2_field `Field` ops; exists so every field primitive is testable through the ABI.) */
int zk_gl_vec_op(zk_ctx *ctx, uint32_t op, const uint64_t *d_a, const uint64_t *d_b,
                 uint64_t *d_out, size_t n);
/* Poseidon permutation applied independently to n_states 12-element states (row-major). */
int zk_poseidon_permute(zk_ctx *ctx, uint64_t *d_states, size_t n_states);
/* Keccak-f[1600] applied independently to n_states 25-lane states (row-major). */
int zk_keccak_f1600(zk_ctx *ctx, uint64_t *d_states, size_t n_states);
/* `H::hash_or_noop` of every row of a column-major matrix [n_cols][n_rows] (row r =
 * (col_0[r], .., col_{C-1}[r])); digest of row r is written to slot r (no bit reversal). */
int zk_hash_rows(zk_ctx *ctx, uint32_t hasher, const uint64_t *d_cols, size_t col_stride,
                 size_t n_cols, size_t n_rows, uint64_t *d_digests);
/* `MerkleTree::new` on precomputed leaf digests: d_digests holds 2^log_leaves leaf slots followed
 * by room for all upper levels (zk_merkle_num_digests slots in total). */
size_t zk_merkle_num_digests(unsigned log_leaves, unsigned cap_height);
int zk_merkle_build(zk_ctx *ctx, uint32_t hasher, uint64_t *d_digests, unsigned log_leaves,
                    unsigned cap_height);

/* ---- Challenger -------------------------------------------------------------------------------
 * plonky2 `Challenger<F, H>` (duplex sponge, overwrite mode; reference use:
 * evm_arithmetization/src/prover.rs:118-127 `observe_cap`, get_challenges.rs:11-227
 * `observe_elements`, prover.rs:320 `compact`).  Host-side object: the transcript is tiny. */
typedef struct zk_challenger zk_challenger;
int zk_challenger_create(uint32_t hasher, zk_challenger **out);
void zk_challenger_free(zk_challenger *ch);
int zk_challenger_clone(const zk_challenger *ch, zk_challenger **out);
int zk_challenger_observe_elements(zk_challenger *ch, const uint64_t *elements, size_t n);
/* `observe_cap`: n_digests 32-byte slots (Keccak-25 digests are observed as 7,7,7,4-byte elements) */
int zk_challenger_observe_cap(zk_challenger *ch, const uint64_t *slots, size_t n_digests);
uint64_t zk_challenger_get_challenge(zk_challenger *ch);
int zk_challenger_get_extension_challenge(zk_challenger *ch, uint64_t out[2]);
/* `compact()`: flush pending inputs, drop buffered outputs, return the 12-element sponge state
 * (stored as `init_challenger_state` in StarkProofWithMetadata, prover.rs:335-338) */
int zk_challenger_compact(zk_challenger *ch, uint64_t state_out[12]);
/* The whole transcript state as 31 words -- sponge state[12], input buffer[8], its length, output buffer[8], its length,
 * hasher -- so that a proof whose tables live on different GPUs can hand the Fiat-Shamir chain from one process to the
 * next (table-parallel segment proofs, SURVEY 8(e) level 2; zk_evm_amd/sharding.py).  import returns ZK_ERR_BAD_ARG on
 * lengths above 8 or an unknown hasher. */
#define ZK_CHALLENGER_STATE_WORDS 31
int zk_challenger_export(const zk_challenger *ch, uint64_t out[31]);
int zk_challenger_import(zk_challenger *ch, const uint64_t in[31]);

/* ---- openings + FRI ---------------------------------------------------------------------------
 * `FriInstanceInfo` in flat form: a batch is an opening point in F_{p^2} and the list of
 * (oracle index, polynomial index) opened there (starky `Stark::fri_instance`: batches at zeta,
 * g*zeta and, for tables with CTLs, 1). */
typedef struct {
    uint64_t point[2];
    size_t n_polys;
    const uint32_t *oracle_idx;
    const uint32_t *poly_idx;
} zk_fri_batch;

/* [EXT] FriReductionStrategy::ConstantArityBits -> reduction_arity_bits; returns the count, or (size_t)-1 for a
 * configuration plonky2 itself panics on (`assert!(degree_bits >= arity_bits)` inside the reduction loop) */
size_t zk_fri_reduction_arity_bits(const zk_cfg *cfg, unsigned degree_bits, uint32_t *out, size_t max);
/* Evaluate every polynomial of every batch at the batch point (the evaluation work of starky
 * `StarkOpeningSet::new`): out = 2 u64 per (batch, poly), batch-major, host memory. */
int zk_fri_openings(zk_ctx *ctx, const zk_batch *const *oracles, size_t n_oracles,
                    const zk_fri_batch *batches, size_t n_batches, uint64_t *out);
/* Size in u64 words of the flat FriProof below (0 on bad arguments). */
size_t zk_fri_proof_words(const zk_cfg *cfg, unsigned degree_bits, const size_t *oracle_cols,
                          size_t n_oracles);
/* `PolynomialBatch::prove_openings(instance, oracles, challenger, fri_params)` -> FriProof.
 * `openings` = the zk_fri_openings output (already observed by the caller, as starky does);
 * the challenger is advanced exactly as plonky2 advances it (alpha, per-round cap/beta, final
 * polynomial, PoW witness + response, query indices).  The proof-of-work witness is the SMALLEST
 * valid one (the reference's rayon `find_any` is not deterministic).  Flat layout (u64 words):
 *   [0]=R rounds [1]=cap_len [2]=Q queries [3]=K oracles [4]=F final-poly length [5]=log2(LDE size)
 *   arity_bits[R]; n_cols[K];
 *   commit_phase_merkle_caps: R x cap_len x 4;  final_poly: F x 2;  pow_witness: 1;
 *   per query: per oracle { leaf[n_cols], siblings[(log_lde - cap_height) x 4] },
 *              per round  { evals[arity x 2], siblings[(log2(#leaves of that tree) - cap_height) x 4] } */
int zk_fri_prove_openings(zk_ctx *ctx, const zk_cfg *cfg, const zk_batch *const *oracles,
                          size_t n_oracles, const zk_fri_batch *batches, size_t n_batches,
                          const uint64_t *openings, zk_challenger *challenger, uint64_t *proof_out);

/* ---- logUp / cross-table-lookup auxiliary columns --------------------------------
 * The AIR side conditions are data.  Flat "program" encoding (u64 words, host memory):
 *   Column  := n_local, n_next, constant, (col_idx, coef) x n_local, (col_idx, coef) x n_next
 *              = sum coef*local[idx] + sum coef*next[idx] + constant   (starky `Column<F>`)
 *   Filter  := n_products, n_constants, (Column, Column) x n_products, Column x n_constants
 *              = sum a*b + sum c   (starky `Filter<F>`; the default filter is the constant 1)
 *   Entry   := n_columns, Column x n_columns, Filter      (one `(columns, filter)` looking entry)
 *   program := n_entries, entry_offset[n_entries], table_col_offset, freq_col_offset, payload
 *              (offsets are word indices into the program; the last two are 0 for CTL programs)
 * Outputs are column-major on the device: ceil(n_entries/(constraint_degree-1)) helper columns,
 * then Z. */
/* starky `lookup_helper_columns(lookup, trace, challenge, constraint_degree)`: single-column
 * entries, denominator = column + challenge; Z(first) = 0, Z(next) = Z + sum h - freq/(table+challenge).
 * (reference lookups: arithmetic_stark.rs:320-327, byte_packing_stark.rs:426-437,
 * keccak_sponge_stark.rs:946-953, memory_stark.rs:858-885) */
int zk_lookup_helper_columns(zk_ctx *ctx, const uint64_t *d_trace, size_t col_stride,
                             size_t n_trace_cols, unsigned log_n, const uint64_t *program,
                             size_t program_words, uint64_t challenge, unsigned constraint_degree,
                             uint64_t *d_out, size_t out_stride, size_t *n_out_cols);
/* starky `cross_table_lookup::partial_sums(trace, columns_filters, (beta, gamma), degree)`:
 * denominator = sum_j beta^j col_j + gamma; Z[i] = sum_{j >= i} sum_h h[j].  With a single entry
 * only Z is produced (n_out_cols = 1).  (reference CTL registry: all_stark.rs:153-417) */
int zk_ctl_partial_sums(zk_ctx *ctx, const uint64_t *d_trace, size_t col_stride, size_t n_trace_cols,
                        unsigned log_n, const uint64_t *program, size_t program_words, uint64_t beta,
                        uint64_t gamma, unsigned constraint_degree, uint64_t *d_out,
                        size_t out_stride, size_t *n_out_cols);

/* ---- quotient polynomials -----------------------------------------------------
 * Table AIRs restated from the reference (`Stark::eval_packed_generic` of each table). */
typedef enum {
    ZK_AIR_NONE = 0,             /* no table constraints: lookup / CTL checks only (tests) */
    ZK_AIR_MEM_CONTINUATION = 1, /* MemBefore / MemAfter: memory_continuation_stark.rs:110-122 */
    ZK_AIR_LOGIC = 2,            /* logic.rs:249-303 */
    ZK_AIR_MEMORY = 3,           /* memory/memory_stark.rs:474-626 */
    ZK_AIR_BYTE_PACKING = 4,     /* byte_packing/byte_packing_stark.rs:296-352 */
    ZK_AIR_CPU = 8,              /* cpu/cpu_stark.rs:594-626 (18 modules); air_consts = {halt_final pc, init pc,
                                    syscall_jumptable, exception_jumptable} (cpu/control_flow.rs:37-47,
                                    cpu/syscalls_exceptions.rs:68,73) */
    ZK_AIR_KECCAK_SPONGE = 7,    /* keccak_sponge/keccak_sponge_stark.rs:546-715 */
    ZK_AIR_KECCAK = 6,           /* keccak/keccak_stark.rs:266-426 + keccak/round_flags.rs:14-60 */
    ZK_AIR_CPU_ERIGON = 10,      /* the Cpu table of a `cdk_erigon` build: 86 columns (`poseidon` flag after
                                  * jumpdest_keccak_general), no JUMPDEST-bit read; air_consts as for ZK_AIR_CPU */
    ZK_AIR_POSEIDON = 9,         /* `cdk_erigon` only: poseidon/poseidon_stark.rs:445-690 (322 columns) */
    ZK_AIR_ARITHMETIC = 5,       /* arithmetic/arithmetic_stark.rs:203-252 + mul/addcy/divmod/modular/byte/shift */
} zk_air;
/* starky `compute_quotient_polys` + chunk split + `PolynomialBatch::from_coeffs`:
 * evaluates, on the coset of size n * quotient_degree_factor, the alpha-combination of
 *   (1) the table constraints, (2) the logUp checks (`eval_packed_lookups_generic`),
 *   (3) the CTL checks (`eval_cross_table_lookup_checks`), divides by Z_H, interpolates, splits
 * into degree-n chunks and commits them.  Returns the quotient batch
 * (num_challenges * quotient_degree_factor polynomials).
 *   lookup_program := n_lookups, offset[n_lookups], payload  (each lookup = a
 *                     zk_lookup_helper_columns program); lookup_challenges as starky: the CTL betas.
 *   ctl_program    := n_zdata, offset[n_zdata], payload; each z-data := beta, gamma, n_helpers,
 *                     then a zk_ctl_partial_sums program.  The auxiliary batch must hold, in
 *                     order: all lookup columns, all CTL helper columns, all CTL Z columns. */
int zk_quotient_polys(zk_ctx *ctx, const zk_cfg *cfg, uint32_t air_id, const uint64_t *air_consts,
                      size_t n_air_consts, const zk_batch *trace, const zk_batch *aux,
                      const uint64_t *alphas, const uint64_t *lookup_program, size_t lookup_words,
                      const uint64_t *lookup_challenges, size_t n_lookup_challenges,
                      const uint64_t *ctl_program, size_t ctl_words, unsigned constraint_degree,
                      zk_batch **quotient_out);

/* ---- one-call provers (compiled host driver, csrc/segment_host.inc) ---------------------------
 * zk_prove_table replaces the reference's `prove_single_table` call (evm_arithmetization/src/prover.rs:301-341 ->
 * starky `prove_with_commitment`, called with public_inputs = &[] and ctl data / challenges = Some): compact the
 * challenger (-> init_challenger_state), logUp helper columns, auxiliary commitment, alphas, quotient polynomials
 * and their commitment, zeta, openings, FRI proof.
 *   lookup_program : as zk_quotient_polys (the table's `Stark::lookups()`), NULL if none
 *   ctl_zdata      : as zk_quotient_polys `ctl_program` (this table's `CtlData::zs_columns`), NULL if none
 *   d_ctl_cols     : device, column stride ctl_col_stride: for z-data i its n_helpers_i helper columns then its Z,
 *                    z-data after z-data (the concatenated zk_ctl_partial_sums outputs)
 *   ctl_challenges : beta, gamma per challenge (lookup challenges = the betas); NULL = draw lookup challenges
 *   challenger     : advanced exactly as the reference advances it
 * The result is host memory owned by the library until zk_table_proof_free. */
typedef struct zk_table_proof zk_table_proof;
typedef struct {
    unsigned degree_bits;
    size_t n_trace_cols, n_aux_cols, n_quotient_cols, n_ctl_zs;
    size_t cap_digests;              /* 2^cap_height 32-byte digests per cap */
    const uint64_t *trace_cap;       /* StarkProof.trace_cap */
    const uint64_t *aux_cap;         /* StarkProof.auxiliary_polys_cap (NULL when the table has no auxiliary polys) */
    const uint64_t *quotient_cap;    /* StarkProof.quotient_polys_cap */
    const uint64_t *openings;        /* StarkOpeningSet, 2 u64 per value: at zeta (trace, aux, quotient chunks), at
                                        g*zeta (trace, aux), at 1 (ctl_zs_first; present iff the table has CTL Zs) */
    size_t n_openings;
    const uint64_t *opening_proof;   /* flat FriProof (layout under zk_fri_prove_openings) */
    size_t proof_words;
    uint64_t init_challenger_state[12];   /* StarkProofWithMetadata.init_challenger_state (prover.rs:335-338) */
} zk_table_proof_view;
int zk_prove_table(zk_ctx *ctx, const zk_cfg *cfg, uint32_t air_id, const uint64_t *air_consts, size_t n_air_consts,
                   const uint64_t *d_trace, size_t col_stride, const zk_batch *trace_commitment,
                   const uint64_t *lookup_program, size_t lookup_words, const uint64_t *ctl_zdata, size_t ctl_words,
                   const uint64_t *d_ctl_cols, size_t ctl_col_stride, const uint64_t *ctl_challenges,
                   unsigned constraint_degree, int requires_ctls, zk_challenger *challenger, zk_table_proof **out);
/* The same split at the one point the transcript allows (SURVEY 8(e) level 2; prover.rs:134-144,328): a table's auxiliary
 * polynomials -- logUp helper columns under the CTL betas, CTL helper / Z columns -- and their commitment depend on the
 * CTL challenges only, so zk_table_aux_commit may run for every table at once (different GPUs, or one GPU's side lane)
 * BEFORE the serial chain of per-table proofs; zk_prove_table_with_aux then takes that commitment instead of rebuilding
 * it (aux_commitment == NULL: identical to zk_prove_table).  `ctl_challenges` (beta, gamma per challenge) is required by
 * zk_table_aux_commit and by a zk_prove_table_with_aux call that is given a commitment.  The caller frees the batch. */
int zk_table_aux_commit(zk_ctx *ctx, const zk_cfg *cfg, const uint64_t *d_trace, size_t col_stride, size_t n_trace_cols,
                        unsigned log_n, const uint64_t *lookup_program, size_t lookup_words, const uint64_t *ctl_zdata,
                        size_t ctl_words, const uint64_t *d_ctl_cols, size_t ctl_col_stride, const uint64_t *ctl_challenges,
                        unsigned constraint_degree, zk_batch **aux_out);
int zk_prove_table_with_aux(zk_ctx *ctx, const zk_cfg *cfg, uint32_t air_id, const uint64_t *air_consts, size_t n_air_consts,
                            const uint64_t *d_trace, size_t col_stride, const zk_batch *trace_commitment,
                            const uint64_t *lookup_program, size_t lookup_words, const uint64_t *ctl_zdata, size_t ctl_words,
                            const uint64_t *d_ctl_cols, size_t ctl_col_stride, const uint64_t *ctl_challenges,
                            unsigned constraint_degree, int requires_ctls, const zk_batch *aux_commitment,
                            zk_challenger *challenger, zk_table_proof **out);
int zk_table_proof_get(const zk_table_proof *proof, zk_table_proof_view *view);
void zk_table_proof_free(zk_table_proof *proof);

/* zk_prove_segment replaces the reference's `prove_with_traces` (prover.rs:72-194; SURVEY 8(b) granularity B): one
 * call per segment.  Tables are given in `Table::all()` order (all_stark.rs:133-146); every table's trace is
 * committed and its cap observed (a zero cap for an optional table that is not in use, prover.rs:118-127), then the
 * public-value elements (get_challenges.rs:195-218 order; the caller flattens `PublicValues`), then starky
 * `get_ctl_data` and the per-table proofs in table order (prover.rs:251-259).
 *   ctl_wiring := n_ctls, offset[n_ctls], per CTL: n_looking, (table, Entry) of the LOOKED table, then
 *                 (table, Entry) x n_looking in `looking_tables` order   (`all_stark.cross_table_lookups`)
 *   mem_before_table / mem_after_table: indices whose trace caps become PublicValues.mem_before / mem_after
 *                 (prover.rs:261-271; mem_after is zeroed when that table is not in use), or -1.
 * Cancellation: zk_ctx_set_abort_flag (polled between kernels and before every table -> ZK_ERR_ABORTED). */
typedef struct {
    const uint64_t *d_trace;         /* device, column-major: column c at d_trace + c*col_stride, 2^log_n rows */
    size_t col_stride, n_cols;
    unsigned log_n;
    uint32_t air_id;                 /* zk_air */
    const uint64_t *air_consts;
    size_t n_air_consts;
    const uint64_t *lookup_program;  /* the table's lookups (zk_quotient_polys lookup_program), NULL if none */
    size_t lookup_words;
    int in_use;                      /* `table_in_use` */
    int optional;                    /* member of OPTIONAL_TABLE_INDICES (all_stark.rs:124-131) */
} zk_table_in;
/* Debug mode = the reference's `#[cfg(debug_assertions)] check_ctls` (prover.rs:164-184, [EXT] starky
 * cross_table_lookup::debug_utils): with the switch on, zk_prove_segment verifies right after `get_ctl_data`, for every
 * CTL and challenge, that the looking tables' running sums (plus the CTL's extra looking rows) equal the looked table's
 * -- the logUp form of the multiset equality starky checks row by row -- and fails with ZK_ERR_BAD_ARG naming the CTL
 * when a witness is inconsistent, instead of emitting a proof the verifier will reject.  Extra looking rows (the Memory
 * CTL's public-value writes, `get_memory_extra_looking_values`, verifier.rs:547-...) are handed over per CTL index as
 * n_rows x width field elements, each row contributing 1 / (sum_j beta^j row[j] + gamma); they are consumed by the next
 * zk_prove_segment on this ctx.  Costs one 8-byte read-back per Z column. */
int zk_ctx_set_check_ctls(zk_ctx *ctx, int on);
int zk_ctx_set_ctl_extra_looking(zk_ctx *ctx, size_t ctl_index, const uint64_t *rows, size_t n_rows, size_t width);
typedef struct zk_segment_proof zk_segment_proof;
int zk_prove_segment(zk_ctx *ctx, const zk_cfg *cfg, const zk_table_in *tables, size_t n_tables,
                     const uint64_t *ctl_wiring, size_t wiring_words, const uint64_t *public_value_elements,
                     size_t n_public_values, unsigned constraint_degree, int mem_before_table, int mem_after_table,
                     zk_segment_proof **out);
size_t zk_segment_proof_num_tables(const zk_segment_proof *proof);
/* NULL for a table that was not in use; owned by the segment proof */
const zk_table_proof *zk_segment_proof_table(const zk_segment_proof *proof, size_t table);
/* (beta, gamma) x num_challenges; returns the number of words available */
size_t zk_segment_proof_ctl_challenges(const zk_segment_proof *proof, uint64_t *out, size_t max_words);
/* the two memory caps as `MemCap::from_merkle_cap` builds them (proof.rs:606-621): cap_digests x 4 field elements
 * each, a hash h contributing `h.to_vec()` -- its four words for Poseidon, the 7,7,7,4-byte little-endian chunks of
 * the 25-byte digest for Keccak-25; returns the words per cap */
size_t zk_segment_proof_mem_caps(const zk_segment_proof *proof, uint64_t *mem_before, uint64_t *mem_after,
                                 size_t max_words);
/* host wall-clock per stage in ms, the reference's TimingTree scopes: [0] "compute all trace commitments",
 * [1] "compute CTL data", [2 + t] "prove <table t> STARK"; returns the number of entries */
size_t zk_segment_proof_stage_ms(const zk_segment_proof *proof, double *out, size_t max);
void zk_segment_proof_free(zk_segment_proof *proof);

/* ---- PLONK prover for the recursion layer -----------------------------------------
 * plonky2 1.0.0 `prove` ([EXT] plonky2/src/plonk/prover.rs `prove_with_partition_witness`) as the reference runs it after
 * every segment STARK: `StarkWrapperCircuit::prove` / `shrink` and `root.circuit.prove`
 * (evm_arithmetization/src/fixed_recursive_verifier.rs:2146, 3167-3179; `CircuitConfig::standard_recursion_config()`:
 * 135 wires, 80 routed, 2 challenges, quotient degree factor 8, FRI rate_bits 3, cap 4, 28 queries, 16 PoW bits).
 * This slice covers the whole protocol skeleton -- wires commitment, transcript, permutation argument (partial products
 * and Zs), selector-filtered gate constraints, quotient, openings, FRI over the four oracles -- for circuits built from the
 * gate kinds below; the remaining gate types of the recursion circuits are listed in DESIGN.md (PLONK plan).  Witness
 * generation (running the generators) stays with the caller. */
typedef enum {
    ZK_PLONK_GATE_NOOP = 0,          /* gates/noop.rs */
    ZK_PLONK_GATE_CONSTANT = 1,      /* gates/constant.rs `ConstantGate { num_consts = param }` */
    ZK_PLONK_GATE_PUBLIC_INPUT = 2,  /* gates/public_input.rs */
    ZK_PLONK_GATE_ARITHMETIC = 3,    /* gates/arithmetic_base.rs `ArithmeticGate { num_ops = param }` */
    ZK_PLONK_GATE_ARITHMETIC_EXTENSION = 4, /* gates/arithmetic_extension.rs `{ num_ops = param }` (D = 2) */
    ZK_PLONK_GATE_MUL_EXTENSION = 5, /* gates/multiplication_extension.rs `{ num_ops = param }` */
    ZK_PLONK_GATE_BASE_SUM_2 = 6,    /* gates/base_sum.rs `BaseSumGate<2> { num_limbs = param }` */
    ZK_PLONK_GATE_REDUCING = 7,      /* gates/reducing.rs `{ num_coeffs = param }` */
    ZK_PLONK_GATE_REDUCING_EXTENSION = 8, /* gates/reducing_extension.rs `{ num_coeffs = param }` */
    ZK_PLONK_GATE_EXPONENTIATION = 9,/* gates/exponentiation.rs `{ num_power_bits = param }` */
    ZK_PLONK_GATE_POSEIDON = 10,     /* gates/poseidon.rs (135 wires, 123 constraints) */
    ZK_PLONK_GATE_RANDOM_ACCESS = 11,/* gates/random_access.rs; param = bits | num_copies << 8 | num_extra_constants << 16 */
    ZK_PLONK_GATE_POSEIDON_MDS = 12, /* gates/poseidon_mds.rs */
    ZK_PLONK_GATE_COSET_INTERPOLATION = 13, /* gates/coset_interpolation.rs; param = subgroup_bits | degree << 8 (the
                                        barycentric weights are recomputed from subgroup_bits) */
} zk_plonk_gate_kind;
/* one entry of `common_data.gates` (sorted by (degree, id) as the builder sorts them) with its selector:
 * `selectors_info.selector_indices[gate]` and `selectors_info.groups[selector_index]` = [group_start, group_end) */
typedef struct {
    uint32_t kind, param, selector_index, group_start, group_end;
} zk_plonk_gate;
typedef struct {
    uint32_t degree_bits;
    uint32_t num_wires, num_routed_wires;
    uint32_t num_constants;          /* columns of the constants part: selectors first, then the gates' constants */
    uint32_t num_selectors;
    uint32_t quotient_degree_factor; /* 8; the partial-product chunk size */
    uint32_t num_gate_constraints;   /* max over the gates */
    zk_cfg fri;                      /* rate_bits 3, cap_height 4, num_challenges 2, ... (hasher: Poseidon) */
} zk_plonk_common;
typedef struct zk_plonk_circuit zk_plonk_circuit;
typedef struct zk_plonk_proof zk_plonk_proof;
/* d_constants_sigmas: device, column-major, (num_constants + num_routed_wires) columns of 2^degree_bits VALUES:
 * the selector / constant polynomials, then the sigma polynomials (`sigma_j(w^i)` = k_{j'} w^{i'} of the cell the
 * permutation maps (i, j) to).  Commits them once (`constants_sigmas_commitment`).  k_is: `get_unique_coset_shifts`. */
int zk_plonk_circuit_create(zk_ctx *ctx, const zk_plonk_common *common, const zk_plonk_gate *gates, size_t n_gates,
                            const uint64_t *d_constants_sigmas, size_t col_stride, const uint64_t *k_is,
                            const uint64_t circuit_digest[4], zk_plonk_circuit **out);
void zk_plonk_circuit_free(zk_plonk_circuit *circuit);
/* `constants_sigmas_cap` (verifier data): 2^cap_height 32-byte slots */
int zk_plonk_circuit_cap(const zk_plonk_circuit *circuit, uint64_t *out);
/* d_wires: device, column-major witness `wire_values[wire][row]` (num_wires columns).  The proof is host memory owned by
 * the library until zk_plonk_proof_free. */
int zk_plonk_prove(zk_plonk_circuit *circuit, const uint64_t *d_wires, size_t col_stride, const uint64_t *public_inputs,
                   size_t n_public_inputs, zk_plonk_proof **out);
/* n_proofs proofs of ONE circuit from ONE call: d_wires[k] / public_inputs[k] as for zk_plonk_prove, out[k] receives proof k.
 * The reference proves the same few circuits for every segment (one shrink chain per table + the root,
 * fixed_recursive_verifier.rs:2053-2160, 3167-3179) and a proof at 2^12-2^14 rows cannot fill the GPU, so the library keeps
 * `in_flight` (0 = 4, at most 16) worker contexts per circuit -- stream + arena each, created on first use -- and runs the
 * witnesses through them concurrently; the calling thread is one of the workers.  Every proof equals what zk_plonk_prove
 * returns for the same witness.  On failure: the first error, no proofs. */
int zk_plonk_prove_batch(zk_plonk_circuit *circuit, const uint64_t *const *d_wires, size_t col_stride,
                         const uint64_t *const *public_inputs, size_t n_public_inputs, size_t n_proofs, unsigned in_flight,
                         zk_plonk_proof **out);
typedef struct {
    size_t cap_digests;
    const uint64_t *wires_cap;                       /* Proof.wires_cap */
    const uint64_t *plonk_zs_partial_products_cap;   /* Proof.plonk_zs_partial_products_cap */
    const uint64_t *quotient_polys_cap;              /* Proof.quotient_polys_cap */
    const uint64_t *openings;                        /* OpeningSet, 2 u64 per value, in `to_fri_openings` order: constants,
                                                        plonk_sigmas, wires, plonk_zs, partial_products, quotient_polys (at
                                                        zeta), then plonk_zs_next (at g * zeta) */
    size_t n_openings;
    const uint64_t *opening_proof;                   /* flat FriProof (layout under zk_fri_prove_openings), 4 oracles */
    size_t proof_words;
    uint64_t public_inputs_hash[4];
    double stage_ms[6];   /* [0] wires commitment [1] partial products + commitment [2] quotient + commitment [3] openings [4] FRI */
} zk_plonk_proof_view;
int zk_plonk_proof_get(const zk_plonk_proof *proof, zk_plonk_proof_view *view);
void zk_plonk_proof_free(zk_plonk_proof *proof);

/* ---- trace finalisation on the device -------------------------------------
 * Keccak table: replaces `KeccakStark::generate_trace_rows` (evm_arithmetization/src/keccak/keccak_stark.rs:65-234):
 * 24 rows x 2431 columns per permutation from its 25-word input (reference order input[y*5 + x]) and the timestamp
 * that links it to the KeccakSponge table, zero rows up to 2^log_n.  inputs / timestamps are host memory; the
 * trace is written column-major on the device (column c at d_out + c*col_stride), ready for zk_prove_segment. */
int zk_keccak_generate_trace(zk_ctx *ctx, const uint64_t *inputs, const uint64_t *timestamps, size_t n_perms,
                             unsigned log_n, uint64_t *d_out, size_t col_stride);

/* Logic table: replaces `LogicStark::generate_trace_rows` (evm_arithmetization/src/logic.rs:165-240).
 * ops (host): n_ops x 9 words = operator (0 AND, 1 OR, 2 XOR), input0 as four 64-bit little-endian limbs, input1
 * likewise; 523 columns are written column-major on the device, zero rows from n_ops to 2^log_n. */
int zk_logic_generate_trace(zk_ctx *ctx, const uint64_t *ops, size_t n_ops, unsigned log_n, uint64_t *d_out,
                            size_t col_stride);
/* MemBefore / MemAfter table: `mem_before_values_to_rows` + `MemoryContinuationStark::generate_trace`
 * (memory_continuation/memory_continuation_stark.rs:53-98).  entries (host): n x 7 words = context, segment, virt,
 * value as four 64-bit little-endian limbs; 12 columns on the device, zero rows from n to 2^log_n (the reference pads
 * to max(128, next_power_of_two(n)) -- the caller picks log_n accordingly). */
int zk_memory_continuation_generate_trace(zk_ctx *ctx, const uint64_t *entries, size_t n, unsigned log_n,
                                          uint64_t *d_out, size_t col_stride);
/* `initial_memory_merkle_cap(rate_bits, cap_height)` (evm_arithmetization/src/verifier.rs:14-78):
 * the cap of the MemBefore-shaped commitment of the kernel code bytes (segment Code) followed by the 256-entry shift
 * table (segment ShiftTable), padded to the next power of two.  cap_out: 2^cap_height x 4 words. */
int zk_initial_memory_merkle_cap(zk_ctx *ctx, const zk_cfg *cfg, const uint8_t *kernel_code, size_t code_len,
                                 uint64_t *cap_out);
/* BytePacking table: replaces `BytePackingStark::generate_trace` (byte_packing/byte_packing_stark.rs:174-283) including
 * its range-check columns.  ops (host): n_ops x 10 words = is_read, context, segment, virt, timestamp, length (1..32;
 * the reference drops empty operations), the bytes as four 64-bit words (byte k at bits 8*(k%8) of word k/8).
 * 71 columns, 2^log_n >= max(n_ops, 256) rows. */
int zk_byte_packing_generate_trace(zk_ctx *ctx, const uint64_t *ops, size_t n_ops, unsigned log_n, uint64_t *d_out,
                                   size_t col_stride);
/* KeccakSponge table: replaces `KeccakSpongeStark::generate_trace` (keccak_sponge/keccak_sponge_stark.rs:252-533)
 * including its range-check columns.  ops (host): n_ops x 5 words = context, segment, virt, timestamp, input length;
 * inputs: the inputs concatenated.  Each operation yields length/136 + 1 rows; 438 columns, 2^log_n >= 256 rows. */
int zk_keccak_sponge_generate_trace(zk_ctx *ctx, const uint64_t *ops, size_t n_ops, const uint8_t *inputs,
                                    size_t input_bytes, unsigned log_n, uint64_t *d_out, size_t col_stride);
/* Range-check finalisation, in place on a device trace: `generate_range_checks` of the Arithmetic, BytePacking and
 * KeccakSponge tables (arithmetic_stark.rs:130-156, byte_packing_stark.rs:254-283, keccak_sponge_stark.rs:503-533):
 * counter_col[i] = min(i, range_max - 1); freq_col[x] = number of cells of columns [first_col, first_col + n_cols)
 * equal to x (the column is overwritten).  A cell >= range_max is an error, as the reference asserts. */
int zk_range_check_columns(zk_ctx *ctx, uint64_t *d_trace, size_t col_stride, size_t n_trace_cols, unsigned log_n,
                           size_t first_col, size_t n_cols, size_t counter_col, size_t freq_col, uint64_t range_max);

/* Arithmetic table: replaces `ArithmeticStark::generate_trace` (arithmetic/arithmetic_stark.rs:158-190) and
 * `Operation::to_rows` (arithmetic/mod.rs:253-359: addcy.rs, mul.rs, modular.rs, divmod.rs, shift.rs, byte.rs).
 * ops (host): n_ops x 18 words = code, opcode, input0, input1, input2, result (four 64-bit little-endian limbs each).
 * code is the operation's flag column: 0 ADD, 1 MUL, 2 SUB, 3 DIV, 4 MOD, 5 ADDMOD, 6 MULMOD, 7 ADDFP254, 8 MULFP254,
 * 9 SUBFP254, 10 SUBMOD, 11 LT, 12 GT, 13 BYTE (input0 = index), 14 SHL, 15 SHR (input0 = shift, input1 = value),
 * 16 = RangeCheckOperation (opcode and result are only read for this one; every other result is recomputed).
 * DIV, MOD, SHR and the modular operations take two rows.  116 columns column-major on the device, zero rows up to
 * 2^log_n >= max(rows, 2^16), range-check columns included; *n_rows_out = rows used by operations. */
int zk_arithmetic_generate_trace(zk_ctx *ctx, const uint64_t *ops, size_t n_ops, unsigned log_n, uint64_t *d_out,
                                 size_t col_stride, size_t *n_rows_out);

/* Poseidon table (`cdk_erigon`): replaces `PoseidonStark::generate_trace` (poseidon/poseidon_stark.rs:183-425).
 * ops (host): n_ops x 13 words.  word 0 = kind: 0 = PoseidonSimpleOp, words 1..12 = the permutation input (field
 * elements); 1 = PoseidonGeneralOp, words 1..6 = context, segment, virt, timestamp, len, number of input bytes -- the
 * (already padded) input is the next `bytes` bytes of `inputs`, a multiple of 56 = FELT_MAX_BYTES * SPONGE_RATE, one
 * row per 56-byte block; the reference indexes `is_final_input_len` with len % 56, so len % 56 < 8 is required.
 * 322 columns column-major on the device; rows after the operations hold the permutation of the zero state. */
int zk_poseidon_generate_trace(zk_ctx *ctx, const uint64_t *ops, size_t n_ops, const uint8_t *inputs, size_t input_bytes,
                               unsigned log_n, uint64_t *d_out, size_t col_stride);

/* Memory table: replaces `MemoryStark::generate_trace` (evm_arithmetization/src/memory/memory_stark.rs:405-455 and
 * everything it calls, :104-403) -- the sort by (context, segment, virt, timestamp), `fill_gaps`, `pad_memory_ops`,
 * `into_row`, the first-change flags / range_check / frequencies / stale-context columns and the extraction of the
 * final memory (the next segment's mem_before) -- on the device.  Two steps, because the height of the table is only
 * known after fill_gaps:
 *   zk_memory_trace_begin   ops (host): n_ops x 9 words = flags (bit 0 is_read, bit 1 filter), timestamp, context,
 *                           segment (unscaled), virt, value as four 64-bit little-endian limbs; mem_before (host):
 *                           n_before x 7 words = context, segment, virt, value limbs (each becomes a filtered write at
 *                           timestamp 0).  timestamp / context / segment / virt < 2^32.  Operations with equal keys
 *                           keep their input order (ops first, then mem_before), as the reference's stable sort does.
 *   zk_memory_gen_log_n / _unpadded_length   the padded height 2^log_n and the reference's third return value
 *   zk_memory_trace_finish  writes the 30 columns column-major on the device (column c at d_out + c*col_stride) and
 *                           reports the number of final-memory entries; stale_contexts (host) must be unique and
 *                           smaller than the height.  An operation log the reference would panic on ("Range check
 *                           ... is too large", a context beyond the height) is ZK_ERR_BAD_ARG.
 *   zk_memory_gen_final_values     the reference's `final_values` in row order, n x 7 words like mem_before (host)
 *   zk_memory_gen_mem_after_trace  the MemAfter table (12 columns, zero rows up to 2^log_n) straight from the device
 *                                  copy of those entries */
typedef struct zk_memory_gen zk_memory_gen;
int zk_memory_trace_begin(zk_ctx *ctx, const uint64_t *ops, size_t n_ops, const uint64_t *mem_before, size_t n_before,
                          zk_memory_gen **out);
size_t zk_memory_gen_unpadded_length(const zk_memory_gen *gen);
unsigned zk_memory_gen_log_n(const zk_memory_gen *gen);
int zk_memory_trace_finish(zk_ctx *ctx, zk_memory_gen *gen, const uint64_t *stale_contexts, size_t n_stale,
                           uint64_t *d_out, size_t col_stride, size_t *n_mem_after);
size_t zk_memory_gen_num_mem_after(const zk_memory_gen *gen);
int zk_memory_gen_final_values(zk_ctx *ctx, const zk_memory_gen *gen, uint64_t *entries_out);
int zk_memory_gen_mem_after_trace(zk_ctx *ctx, const zk_memory_gen *gen, unsigned log_n, uint64_t *d_out,
                                  size_t col_stride);
void zk_memory_gen_free(zk_memory_gen *gen);

/* ---- sharding INSIDE one table (SURVEY 8(e) level 3; orchestration: zk_evm_amd/sharding.py over torch.distributed) -------
 * The reference commits and proves a table on one machine (`prover.rs:90-111`, `prove_single_table` `prover.rs:301-341`); these
 * entry points are the per-rank pieces when ONE table's commitment and proof are spread over W = 2^shard_log_w GPUs:
 * columns sharded for the NTTs, an all-to-all to ROW shards (rank q = the leaves [q N/W, (q+1) N/W) of the Merkle tree, i.e.
 * the natural rows j with j mod W = bitrev_W(q), in leaf order), leaf hashing + one subtree group per rank, all-gather of the
 * sub-roots = the cap; quotient values, FRI batch combination and the initial-tree query openings on the row shards; the
 * (two-column) FRI layers and the (four-column) quotient chunks replicated after an all-gather.
 *
 * zk_shard_pack_leaf_rows: d_lde = this rank's LDE columns [n_cols][2^log_lde] (natural order) -> d_out [W][n_cols][Nl],
 *   block q = the rows of rank q in leaf order: the send buffers of the all-to-all, written in one pass.
 * zk_batch_from_parts: a zk_batch VIEW over caller-owned device memory (zk_batch_free releases the handle only):
 *   column shard: d_coeffs = [n_cols][n] bit-reversed coefficients of this rank's columns (zk_fri_openings on them);
 *   row shard (shard_log_w > 0): d_rows = [n_cols][Nl] leaf-ordered rows, d_digests = the levels of this rank's subtrees
 *   (zk_hash_rows + zk_merkle_build with cap_height - shard_log_w), cap = the whole tree's cap (host, 2^cap_height x 4).
 * zk_gl_add_scalar_columns: column c += add[c]: the carry of a running sum (CTL Z column) computed per row block. */
/* zk_shard_values_to_lde: the NTT half of `from_values` on this rank's columns: d_values [n_cols][n] -> d_coeffs [n_cols][n]
 *   (bit-reversed coefficient order, as every zk_batch keeps them) and d_lde [n_cols][n << rate_bits] (natural order). */
int zk_shard_values_to_lde(zk_ctx *ctx, const uint64_t *d_values, size_t n_cols, unsigned log_n, unsigned rate_bits,
                           uint64_t *d_coeffs, uint64_t *d_lde);
int zk_shard_pack_leaf_rows(zk_ctx *ctx, const uint64_t *d_lde, size_t col_stride, size_t n_cols, unsigned log_lde,
                            unsigned shard_log_w, uint64_t *d_out);
int zk_batch_from_parts(zk_ctx *ctx, const zk_cfg *cfg, size_t n_cols, unsigned log_n, const uint64_t *d_coeffs,
                        const uint64_t *d_rows, const uint64_t *d_digests, const uint64_t *cap, unsigned shard_log_w,
                        unsigned shard_rank, zk_batch **out);
int zk_gl_add_scalar_columns(zk_ctx *ctx, uint64_t *d_cols, size_t col_stride, size_t n_cols, size_t n_rows,
                             const uint64_t *add);
/* The quotient VALUES at this rank's rows (zk_quotient_polys, first half): d_*_rows = the row shards of the trace / auxiliary
 * LDE, d_*_next = the shard of the rank that holds the NEXT rows (natural row + 2: one rank for the whole shard; the same
 * pointers when W <= 2).  d_out: [num_challenges][Nl].  zk_quotient_commit_values = the second half on the all-gathered
 * values (d_values: [num_challenges][2n], NATURAL order): coset iNTT, chunk split, `from_coeffs` commitment. */
int zk_quotient_values_sharded(zk_ctx *ctx, const zk_cfg *cfg, uint32_t air_id, const uint64_t *air_consts,
                               size_t n_air_consts, const uint64_t *d_trace_rows, const uint64_t *d_trace_next,
                               size_t n_trace_cols, const uint64_t *d_aux_rows, const uint64_t *d_aux_next,
                               size_t n_aux_cols, unsigned log_n, unsigned shard_log_w, unsigned shard_rank,
                               const uint64_t *alphas, const uint64_t *lookup_program, size_t lookup_words,
                               const uint64_t *lookup_challenges, size_t n_lookup_challenges,
                               const uint64_t *ctl_program, size_t ctl_words, unsigned constraint_degree,
                               uint64_t *d_out);
int zk_quotient_commit_values(zk_ctx *ctx, const zk_cfg *cfg, const uint64_t *d_values, unsigned log_n,
                              unsigned constraint_degree, zk_batch **quotient_out);
/* `prove_openings` in two steps.  zk_fri_combine_sharded: the batch combination sum_b alpha^.. (G_b(x) - y_b) / (x - z_b) at
 * this rank's rows (every oracle a row-shard view of the same sharding), alpha drawn by the caller from the replicated
 * transcript; d_out [2][Nl] (extension components), leaf order.  zk_fri_prove_from_values: everything after it from the
 * all-gathered values (d_vals: [2][N], NATURAL order, overwritten) -- commit-phase trees, final polynomial, proof of work,
 * query rounds -- on every rank alike; the challenger must be in the state right after alpha.  The initial-tree opening of
 * query x is written for whole oracles and for a row shard that owns leaf x, and left zero otherwise: the caller takes those
 * words from the owner's proof.  xs_out (optional): the num_query_rounds query indices. */
int zk_fri_combine_sharded(zk_ctx *ctx, const zk_cfg *cfg, const zk_batch *const *oracles, size_t n_oracles,
                           const zk_fri_batch *batches, size_t n_batches, const uint64_t *openings,
                           const uint64_t alpha[2], uint64_t *d_out);
int zk_fri_prove_from_values(zk_ctx *ctx, const zk_cfg *cfg, const zk_batch *const *oracles, size_t n_oracles,
                             const zk_fri_batch *batches, size_t n_batches, uint64_t *d_vals, zk_challenger *chal,
                             uint64_t *proof, uint64_t *xs_out);

/* The FRI commit phase itself over row shards -- every layer stays on the rank that owns its leaves, one sub-root all-gather per
 * round -- as an alternative to zk_fri_prove_from_values (which replicates the two-column layers after ONE all-gather); same proof.
 *   zk_fri_commit_round_sharded : this rank's part of a layer (d_vals [2][len_local], LEAF order) -> its leaves (2^arity_bits
 *                                 consecutive values), local subtrees down to 2^(cap_height - shard_log_w) roots in d_digests
 *   zk_fri_fold_values_sharded  : the fold on values: f'(x^arity) = sum_j (beta / x)^j u_j with u = the leaf's inverse DFT;
 *                                 `shift` = the coset shift of THIS layer; d_out [2][len_local >> arity_bits], leaf order
 *   zk_fri_proof_of_work        : the grind on the caller's transcript (smallest witness; observed; response checked)
 *   zk_fri_initial_openings     : leaf rows + Merkle paths of the queries xs in the given oracles (zeros where another rank owns
 *                                 the leaf); out [n_queries][sum_k (n_cols_k + 4 (log_lde - cap_height))] */
int zk_fri_commit_round_sharded(zk_ctx *ctx, const zk_cfg *cfg, const uint64_t *d_vals, unsigned log_layer,
                                unsigned shard_log_w, uint64_t *d_digests);
int zk_fri_fold_values_sharded(zk_ctx *ctx, const zk_cfg *cfg, const uint64_t *d_vals, unsigned log_layer,
                               unsigned shard_log_w, unsigned shard_rank, uint64_t shift, const uint64_t beta[2],
                               uint64_t *d_out);
int zk_fri_proof_of_work(zk_ctx *ctx, const zk_cfg *cfg, zk_challenger *chal, uint64_t *witness_out);
int zk_fri_initial_openings(zk_ctx *ctx, const zk_cfg *cfg, const zk_batch *const *oracles, size_t n_oracles,
                            const uint64_t *xs, size_t n_queries, uint64_t *out);

/* ---- the multi-GPU provers behind this ABI (SURVEY 8(e) levels 2 and 3; csrc/comm_host.inc, shard_prove_host.inc) ------------
 * A zk_comm is the set of ranks of one job -- one process (or thread) per GPU, one zk_ctx each.  The reference runs ONE prover
 * per machine (`zero/src/ops.rs:24-67` -> `prove`, `evm_arithmetization/src/prover.rs:72-194`); these calls replace that seam
 * when a segment's tables (level 2) or one table's rows (level 3) are spread over the GPUs of a node.
 *   zk_comm_unique_id + zk_comm_create : RCCL (the C API of <rccl/rccl.h>, loaded at run time; one rank per GPU, xGMI).  Rank 0
 *       makes the id and hands it to the others by whatever the caller has (a pipe, MPI, a file); every rank then calls
 *       zk_comm_create -- collectively, like ncclCommInitRank.  Sends / receives are cut into pieces of at most 256 MiB
 *       (ZK_COMM_PIECE_MB; RCCL 2.26 corrupted larger ones: tools/rccl_repro.py).
 *   zk_comm_create_host : host-staged transport over POSIX shared memory `name` (unique per job; no '/'): ranks may share a GPU
 *       (tests on a one-GPU box; a fallback where RCCL cannot come up).  slot_bytes = outbox per rank (0 = 32 MiB,
 *       ZK_COMM_SLOT_MB overrides); ctx may be NULL for a communicator that only moves host payloads (zk_comm_*_host).
 *       A rank that waits ZK_COMM_TIMEOUT_S (300) for a peer gives up with ZK_ERR_COMM -- and so do all the others.
 * Failures.  The ranks run the same sequence of collectives, and every collective starts with an exchange of one status word per
 * rank.  A rank whose LOCAL step fails (out of memory, a refused launch, a bad argument only it can see) returns its own error
 * from the multi-rank call it is in; the others learn of it inside the collective they were entering and return ZK_ERR_COMM from
 * the same call -- at once, no time limit involved, on RCCL as on the host transport -- and the communicator is still in step:
 * the next multi-rank call may use it.  Only a failure INSIDE a data exchange (a copy that fails, an RCCL error, a peer that is
 * gone: the host transport's time limit) kills the communicator: every later call on it returns ZK_ERR_COMM immediately; free it
 * on every rank.  world must be a power of two. */
#define ZK_COMM_ID_BYTES 128
typedef struct zk_comm zk_comm;
int zk_comm_unique_id(uint8_t out[ZK_COMM_ID_BYTES]);
int zk_comm_create(zk_ctx *ctx, const uint8_t id[ZK_COMM_ID_BYTES], unsigned rank, unsigned world, zk_comm **out);
int zk_comm_create_host(zk_ctx *ctx, const char *name, unsigned rank, unsigned world, size_t slot_bytes, zk_comm **out);
void zk_comm_free(zk_comm *comm);
unsigned zk_comm_rank(const zk_comm *comm);
unsigned zk_comm_world(const zk_comm *comm);
const char *zk_comm_transport(const zk_comm *comm);      /* "rccl" | "host" */
const char *zk_comm_last_error(const zk_comm *comm);
/* out[0] bytes sent to other ranks, [1] bytes received from them, [2] collectives issued -- since creation */
int zk_comm_stats(const zk_comm *comm, uint64_t out[3]);
int zk_comm_barrier(zk_comm *comm);
/* The collectives the provers are built from, for callers that move their own data through the job's communicator (and for the
 * tests): all-gather / broadcast of host bytes (either transport), all-to-all with per-pair byte counts of host buffers (host
 * transport) and of device buffers (either; on ctx's stream; RCCL: whole 8-byte words), all-gather of device buffers. */
int zk_comm_all_gather_host(zk_comm *comm, const void *send, size_t bytes, void *recv);
int zk_comm_broadcast_host(zk_comm *comm, void *buf, size_t bytes, unsigned root);
int zk_comm_all_to_all_host(zk_comm *comm, const void *const *send, const size_t *send_bytes, void *const *recv,
                            const size_t *recv_bytes);
int zk_comm_all_to_all_device(zk_comm *comm, const void *const *d_send, const size_t *send_bytes, void *const *d_recv,
                              const size_t *recv_bytes);
int zk_comm_all_gather_device(zk_comm *comm, const void *d_send, size_t bytes, void *d_recv);
/* host wall clock per stage of the sharded calls, accumulated: [0] all-to-all #1 + iNTT + LDE + pack, [1] all-to-all #2,
 * [2] leaf hashing + subtrees + cap all-gather, [3] auxiliary columns + carries, [4] quotient, [5] openings, [6] FRI;
 * returns the number of stages */
size_t zk_comm_last_timing(zk_comm *comm, double *out_ms, size_t max, int reset);

/* Level 3: ONE table over all ranks of `comm` (W = 2^k <= 2^cap_height GPUs).  Rank q passes the contiguous ROW BLOCK q of the
 * trace (d_row_block: column c at + c * col_stride, 2^log_n / W rows; log_n = the WHOLE table's).  Collective: every rank calls
 * with the same table description and a transcript in the same state; the proof -- equal to zk_prove_table's on one GPU, word
 * for word -- and the advanced transcript come back on EVERY rank.
 *   zk_commit_rows_sharded : `PolynomialBatch::from_values` (prover.rs:90-111): column-sharded NTTs between two all-to-alls, leaf
 *       hashing on row shards, the sub-roots all-gathered into the cap (zk_sharded_batch_cap; the same on every rank).
 *   zk_prove_table_sharded : `prove_single_table` (prover.rs:301-341).  lookup_program / ctl_zdata / ctl_challenges /
 *       requires_ctls as for zk_prove_table -- but the CTL helper / Z columns are built HERE, on the row blocks, with the
 *       cross-block carries (no d_ctl_cols).  trace_commitment: the table's zk_commit_rows_sharded result when the caller made it
 *       earlier (a segment commits every trace before the transcript starts), else NULL; the caller still frees it.
 *       fri_mode 0: the two-column FRI layers replicated after ONE all-gather; 1: every layer on the rank that owns its leaves,
 *       one sub-root all-gather per round.  Same proof either way. */
typedef struct zk_sharded_batch zk_sharded_batch;
int zk_commit_rows_sharded(zk_ctx *ctx, zk_comm *comm, const zk_cfg *cfg, const uint64_t *d_row_block, size_t col_stride,
                           size_t n_cols, unsigned log_n, zk_sharded_batch **out);
int zk_sharded_batch_cap(const zk_sharded_batch *batch, uint64_t *cap_out);
const zk_batch *zk_sharded_batch_rows(const zk_sharded_batch *batch);       /* the row-shard view (zk_batch_from_parts form) */
const zk_batch *zk_sharded_batch_columns(const zk_sharded_batch *batch);    /* the column-shard view; NULL if this rank owns none */
void zk_sharded_batch_free(zk_sharded_batch *batch);
int zk_prove_table_sharded(zk_ctx *ctx, zk_comm *comm, const zk_cfg *cfg, uint32_t air_id, const uint64_t *air_consts,
                           size_t n_air_consts, const uint64_t *d_row_block, size_t col_stride, size_t n_trace_cols,
                           unsigned log_n, zk_sharded_batch *trace_commitment, const uint64_t *lookup_program,
                           size_t lookup_words, const uint64_t *ctl_zdata, size_t ctl_words, const uint64_t *ctl_challenges,
                           unsigned constraint_degree, int requires_ctls, unsigned fri_mode, zk_challenger *challenger,
                           zk_table_proof **out);

/* Level 2: the tables of ONE segment over the ranks of `comm` -- `prove_with_traces` (prover.rs:72-194) as a collective call.
 * What shards (prover.rs:90-111): every table's trace commitment is independent of the transcript, and a table's CTL / logUp
 * columns and its whole `prove_single_table` only read that table's own trace -- so table t lives on exactly one rank and no
 * bulk data moves: ONE all-gather of the caps, then the 31-word challenger state owner -> all after each table of the (serial,
 * prover.rs:251-259) chain.  zk_assign_tables is the assignment (largest cost first onto the least loaded rank; deterministic):
 * the caller puts table t's trace on rank owner[t] and passes d_trace = NULL elsewhere.  row_sharded[t] != 0 (NULL = none): table t
 * is spread over ALL ranks instead (level 3: d_trace = this rank's ROW BLOCK, col_stride its stride, log_n the whole table's).
 * Every rank passes the same table descriptions (n_cols, log_n, air, programs, in_use, optional), wiring and public values;
 * EVERY rank gets the whole segment proof, bit-identical to zk_prove_segment's on one GPU.  Latency, not throughput: independent
 * segments on independent GPUs (one zk_prove_segment per rank, no collective) remain the throughput path. */
int zk_assign_tables(const size_t *n_cols, const unsigned *log_n, size_t n_tables, unsigned world, const uint8_t *row_sharded,
                     uint32_t *owner_out);
int zk_prove_segment_table_parallel(zk_ctx *ctx, zk_comm *comm, const zk_cfg *cfg, const zk_table_in *tables, size_t n_tables,
                                    const uint8_t *row_sharded, const uint64_t *ctl_wiring, size_t wiring_words,
                                    const uint64_t *public_value_elements, size_t n_public_values, unsigned constraint_degree,
                                    int mem_before_table, int mem_after_table, unsigned fri_mode, zk_segment_proof **out);

/* library / device info */
const char *zk_version(void);
int zk_device_info(int device, char *name_out, size_t name_len, int *cu_count, size_t *hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* ZKSTARK_H */
