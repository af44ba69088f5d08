// Host-side interface BETWEEN the translation units of libzkstark_hip.so (not part of the C ABI: hidden visibility).
// The library is built from several .hip files compiled in parallel (zk_evm_amd/build.py): the core (context, NTT,
// Merkle, commit, FRI, STARK columns, segment driver), the table AIR quotient kernels in groups, the PLONK prover, and
// the witness-table generators.  Kernels never cross a TU: each is launched from the TU that defines it.
#pragma once
#include <vector>

#include "ctx.hpp"
#include "host_hash.hpp"

#define ZK_INTERNAL __attribute__((visibility("hidden")))

struct zk_challenger { zkhost::Challenger c; };

static inline const uint64_t *cu(const u64 *p) { return reinterpret_cast<const uint64_t *>(p); }
static inline uint64_t *mu(u64 *p) { return reinterpret_cast<uint64_t *>(p); }

struct BatchGuard {   // frees batches / arena blocks on every exit path
    zk_ctx *ctx;
    std::vector<zk_batch *> batches;
    std::vector<void *> blocks;
    explicit BatchGuard(zk_ctx *c) : ctx(c) {}
    ~BatchGuard() { for (auto *b : batches) zk_batch_free(b); for (void *p : blocks) ctx->arena.free(p); }
    void drop(zk_batch *b) { for (auto &x : batches) if (x == b) { zk_batch_free(b); x = nullptr; } }
};

// mode: values (from_values), natural-order coefficients (from_coeffs), or coefficients already in
// the device's bit-reversed order (internal: quotient chunks).
enum CommitMode { COMMIT_VALUES = 0, COMMIT_COEFFS = 1, COMMIT_COEFFS_BITREV = 2 };
ZK_INTERNAL int zki_commit(zk_ctx *ctx, const zk_cfg *cfg, const u64 *d_in, size_t in_stride, size_t n_cols, unsigned log_n,
                           CommitMode mode, zk_batch **out);
ZK_INTERNAL int zki_get_twiddles(zk_ctx *ctx, int log_size, bool inverse, const u64 **out);
ZK_INTERNAL int zki_get_coset_table(zk_ctx *ctx, int log_n, u64 shift, bool inverse, const u64 **out);
ZK_INTERNAL int zki_ntt_values_to_coeffs(zk_ctx *ctx, const u64 *src, size_t src_stride, u64 *dst, size_t dst_stride,
                                         size_t n_cols, int log_n, const u64 *out_scale);

// quotient kernels of the table AIRs, one function per AIR group / TU (zk_airs_*.hip): launch the AIR kernel, or with
// `count` measure the number of constraints the AIR yields.  Returns ZK_AIR_NOT_MINE when `air_id` belongs to another group.
struct QuotientArgs;
#define ZK_AIR_NOT_MINE 1
#define ZK_AIR_GROUP_DECL(name)                                                                                          \
    ZK_INTERNAL int name(zk_ctx *ctx, uint32_t air_id, const QuotientArgs &A, u32 size, DevBuf &scratch, size_t n_trace_cols, \
                         size_t n_air_consts, u32 *count)
ZK_AIR_GROUP_DECL(zki_quotient_airs_a);
ZK_AIR_GROUP_DECL(zki_quotient_airs_b);
ZK_AIR_GROUP_DECL(zki_quotient_airs_c);
ZK_AIR_GROUP_DECL(zki_quotient_airs_d);
