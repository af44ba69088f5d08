// libzkstark_hip.so -- host side of the C ABI declared in include/zkstark.h.
// Core translation unit (context, NTT, Merkle, commit, FRI, STARK columns, quotient driver, segment driver); the table
// AIR kernels, the PLONK prover and the witness-table generators are separate TUs (internal.hpp, zk_evm_amd/build.py).
//
// Host responsibilities: per-context twiddle / coset tables, pass planning for the multi-pass
// NTT, stage sequencing on one HIP stream, device-memory ownership behind zk_batch handles.
// There is NO CPU fallback: every entry point either runs the HIP kernels or returns an error.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/zkstark.h"
#include "arena.hpp"
#include "gl.cuh"
#include "merkle.cuh"
#include "ntt.cuh"
#include "fri.cuh"
#include "stark.cuh"
#include "quotient.cuh"
#include "host_hash.hpp"

#include "ctx.hpp"
#include "internal.hpp"

// ------------------------------------------------------------------------------------------
extern "C" const char *zk_version(void) { return "zkstark-hip 0.1 (gfx950)"; }

extern "C" int zk_device_info(int device, char *name_out, size_t name_len, int *cu_count,
                              size_t *hbm_bytes) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ZK_ERR_HIP;
    if (name_out && name_len) {
        snprintf(name_out, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return ZK_OK;
}

static std::string initial_plans();
extern "C" int zk_ctx_create(int device, zk_ctx **out) {
    if (!out) return ZK_ERR_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
        return ZK_ERR_HIP;  // no GPU: fail loudly, there is no CPU path
    zk_ctx *ctx = new zk_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return ZK_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cu_count = prop.multiProcessorCount;
    // allow the NTT kernels their full LDS tile (default dynamic limit is 64 KiB)
    hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_pass_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_pass_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // (per DEVICE, hence here and not behind a once-per-process flag: a process may drive several GPUs, one ctx each)
    hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_strided_swap_kernel<true, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_strided_swap_kernel<false, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    ctx->plans = initial_plans();              // (ntt_host.inc: ZK_NTT_SWAP_PLANS as read at load, else the table compiled in)
    *out = ctx;
    return ZK_OK;
}

extern "C" void zk_batch_free(zk_batch *b);

extern "C" void zk_ctx_destroy(zk_ctx *ctx) {
    if (!ctx) return;
    HostProf::dump();
    hipSetDevice(ctx->device);
    while (!ctx->live_batches.empty()) zk_batch_free(*ctx->live_batches.begin());
    hipStreamSynchronize(ctx->stream);
    if (ctx->side_stream) hipStreamSynchronize(ctx->side_stream);
    if (ctx->tail_stream) hipStreamSynchronize(ctx->tail_stream);
    ctx->arena.destroy();
    for (auto &kv : ctx->tw_fwd) hipFree(kv.second);
    for (auto &kv : ctx->tw_inv) hipFree(kv.second);
    for (auto &kv : ctx->tw_inv_br) hipFree(kv.second);
    for (auto &kv : ctx->coset_tabs) hipFree(kv.second);
    for (auto &kv : ctx->coset_inv_tabs) hipFree(kv.second);
    for (auto &kv : ctx->wave_coset2_tabs) hipFree(kv.second);
    if (ctx->ntt_batch_stream) hipStreamDestroy(ctx->ntt_batch_stream);
    for (auto &e : ctx->ntt_batch_ev) if (e) hipEventDestroy(e);
    for (auto &e : ctx->ev_pool) if (e) hipEventDestroy(e);
    if (ctx->h_caps) hipHostFree(ctx->h_caps);
    if (ctx->h_big) hipHostFree(ctx->h_big);
    if (ctx->h_seq) hipHostFree((void *)ctx->h_seq);
    if (ctx->h_flags) hipHostFree(ctx->h_flags);
    for (auto &kv : ctx->dev_consts) if (kv.second.d) hipFree(kv.second.d);
    for (char *c : ctx->stage_chunks) hipHostFree(c);
    if (ctx->side_stream) hipStreamDestroy(ctx->side_stream);
    if (ctx->tail_stream) hipStreamDestroy(ctx->tail_stream);
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" int zk_ctx_set_stream(zk_ctx *ctx, void *hip_stream) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    if ((hipStream_t)hip_stream != ctx->stream) {
        // the arena recycles blocks in stream order: drain the old stream before work moves to another one
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->stream = (hipStream_t)hip_stream;  // NULL is the HIP default (null) stream
    return ZK_OK;
}
extern "C" int zk_ctx_mem_reserve(zk_ctx *ctx, size_t bytes) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    hipError_t e = ctx->arena.reserve(bytes);
    if (e != hipSuccess) return set_err(ctx, ZK_ERR_OOM, "cannot reserve %zu bytes of HBM: %s", bytes, hipGetErrorString(e));
    return ZK_OK;
}
extern "C" int zk_ctx_mem_trim(zk_ctx *ctx, size_t *released) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    size_t r = ctx->arena.trim();
    if (released) *released = r;
    return ZK_OK;
}
extern "C" int zk_ctx_mem_stats(const zk_ctx *ctx, size_t *reserved, size_t *in_use, size_t *peak_in_use) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    if (reserved) *reserved = ctx->arena.reserved();
    if (in_use) *in_use = ctx->arena.in_use();
    if (peak_in_use) *peak_in_use = ctx->arena.peak_in_use();
    return ZK_OK;
}
extern "C" int zk_dev_alloc(zk_ctx *ctx, size_t bytes, void **d_out) {
    if (!ctx || !d_out) return ZK_ERR_BAD_ARG;
    *d_out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = ctx->arena.alloc(d_out, bytes ? bytes : 8);
    if (e != hipSuccess) return set_err(ctx, e == hipErrorOutOfMemory ? ZK_ERR_OOM : ZK_ERR_HIP, "device arena: %s", hipGetErrorString(e));
    return ZK_OK;
}
extern "C" int zk_dev_free(zk_ctx *ctx, void *d_ptr) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    if (d_ptr) ctx->arena.free(d_ptr);
    return ZK_OK;
}
extern "C" int zk_dev_upload_columns(zk_ctx *ctx, const uint64_t *const *cols, size_t n_cols, size_t n, uint64_t *d_out,
                                     size_t col_stride) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    if (!cols || !d_out || col_stride < n) return set_err(ctx, ZK_ERR_BAD_ARG, "bad upload arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (size_t c = 0; c < n_cols; ++c) {
        if (!cols[c]) return set_err(ctx, ZK_ERR_BAD_ARG, "null column %zu", c);
        HIP_TRY(ctx, hipMemcpyAsync(d_out + c * col_stride, cols[c], n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}
extern "C" int zk_ctx_synchronize(zk_ctx *ctx) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}
extern "C" const char *zk_last_error(const zk_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }
extern "C" int zk_ctx_set_abort_flag(zk_ctx *ctx, volatile const int *abort_flag) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    ctx->abort_flag = abort_flag;
    return ZK_OK;
}
extern "C" int zk_ctx_set_abort_flag_u8(zk_ctx *ctx, volatile const uint8_t *abort_flag) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    ctx->abort_flag_u8 = abort_flag;
    return ZK_OK;
}
extern "C" int zk_ctx_commit_totals(zk_ctx *ctx, double out_ms[4], uint64_t *n_commits, double *leaf_hash_bytes,
                                    double *leaf_hash_perms, double *ntt_bytes, int reset) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    if (out_ms) for (int i = 0; i < 4; ++i) out_ms[i] = ctx->total_ms[i];
    if (n_commits) *n_commits = ctx->total_commits;
    if (leaf_hash_bytes) *leaf_hash_bytes = ctx->total_leaf_bytes;
    if (leaf_hash_perms) *leaf_hash_perms = ctx->total_leaf_perms;
    if (ntt_bytes) *ntt_bytes = ctx->total_ntt_bytes;
    if (reset) {
        for (double &v : ctx->total_ms) v = 0;
        ctx->total_leaf_bytes = ctx->total_leaf_perms = ctx->total_ntt_bytes = 0;
        ctx->total_commits = 0;
    }
    return ZK_OK;
}
extern "C" int zk_ctx_side_commit_totals(zk_ctx *ctx, double out_ms[4], uint64_t *n_commits, double *leaf_hash_bytes,
                                         double *ntt_bytes, int reset) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    if (out_ms) for (int i = 0; i < 4; ++i) out_ms[i] = ctx->side_ms[i];
    if (n_commits) *n_commits = ctx->side_commits;
    if (leaf_hash_bytes) *leaf_hash_bytes = ctx->side_leaf_bytes;
    if (ntt_bytes) *ntt_bytes = ctx->side_ntt_bytes;
    if (reset) {
        for (double &v : ctx->side_ms) v = 0;
        ctx->side_leaf_bytes = ctx->side_ntt_bytes = 0;
        ctx->side_commits = 0;
    }
    return ZK_OK;
}
extern "C" int zk_ctx_last_timings(const zk_ctx *ctx, float out_ms[4]) {
    if (!ctx || !out_ms) return ZK_ERR_BAD_ARG;
    memcpy(out_ms, ctx->timings, sizeof ctx->timings);
    return ZK_OK;
}

#include "ntt_host.inc"
#include "merkle_host.inc"

static int check_ntt_args(zk_ctx *ctx, const void *d, size_t stride, size_t n_cols, unsigned log_n) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    if (!d && n_cols) return set_err(ctx, ZK_ERR_BAD_ARG, "null data pointer");
    if (log_n > 31) return set_err(ctx, ZK_ERR_BAD_ARG, "log_n %u exceeds the field's 2-adicity budget", log_n);
    if (n_cols > 1 && stride < ((size_t)1 << log_n)) return set_err(ctx, ZK_ERR_BAD_ARG, "col_stride < n");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ZK_OK;
}

extern "C" int zk_ifft(zk_ctx *ctx, uint64_t *d, size_t stride, size_t n_cols, unsigned log_n) {
    ZK_TRY(check_ntt_args(ctx, d, stride, n_cols, log_n));
    if (!n_cols) return ZK_OK;
    ZK_TRY(ntt_values_to_coeffs(ctx, (u64 *)d, stride, (u64 *)d, stride, n_cols, log_n, nullptr));
    return bitrev_columns(ctx, (u64 *)d, stride, n_cols, log_n);
}
extern "C" int zk_coset_ifft(zk_ctx *ctx, uint64_t *d, size_t stride, size_t n_cols, unsigned log_n,
                             uint64_t shift) {
    ZK_TRY(check_ntt_args(ctx, d, stride, n_cols, log_n));
    if (!n_cols) return ZK_OK;
    if (gl_canon(shift) == 0) return set_err(ctx, ZK_ERR_BAD_ARG, "coset shift must be non-zero");
    const u64 *tab = nullptr;
    ZK_TRY(get_coset_table(ctx, log_n, shift, true, &tab));
    ZK_TRY(ntt_values_to_coeffs(ctx, (u64 *)d, stride, (u64 *)d, stride, n_cols, log_n, tab));
    return bitrev_columns(ctx, (u64 *)d, stride, n_cols, log_n);
}
extern "C" int zk_fft(zk_ctx *ctx, uint64_t *d, size_t stride, size_t n_cols, unsigned log_n) {
    ZK_TRY(check_ntt_args(ctx, d, stride, n_cols, log_n));
    if (!n_cols) return ZK_OK;
    ZK_TRY(bitrev_columns(ctx, (u64 *)d, stride, n_cols, log_n));
    return ntt_coeffs_to_values(ctx, (u64 *)d, stride, (u64 *)d, stride, n_cols, log_n, 0, nullptr);
}
extern "C" int zk_coset_fft(zk_ctx *ctx, uint64_t *d, size_t stride, size_t n_cols, unsigned log_n,
                            uint64_t shift) {
    ZK_TRY(check_ntt_args(ctx, d, stride, n_cols, log_n));
    if (!n_cols) return ZK_OK;
    const u64 *tab = nullptr;
    ZK_TRY(get_coset_table(ctx, log_n, shift, false, &tab));
    ZK_TRY(bitrev_columns(ctx, (u64 *)d, stride, n_cols, log_n));
    return ntt_coeffs_to_values(ctx, (u64 *)d, stride, (u64 *)d, stride, n_cols, log_n, 0, tab);
}

// natural-order coefficients in, natural-order LDE values out (API form; the batch path below
// keeps coefficients bit-reversed and skips the permutation)
extern "C" int zk_lde(zk_ctx *ctx, const uint64_t *d_coeffs, size_t in_stride, uint64_t *d_out,
                      size_t out_stride, size_t n_cols, unsigned log_n, unsigned rate_bits) {
    ZK_TRY(check_ntt_args(ctx, d_coeffs, in_stride, n_cols, log_n));
    if (log_n + rate_bits > 31 || !d_out) return set_err(ctx, ZK_ERR_BAD_ARG, "bad lde size/output");
    if (!n_cols) return ZK_OK;
    size_t n = (size_t)1 << log_n;
    u64 *tmp = nullptr;
    DevBuf scratch(ctx);
    ZK_TRY(scratch.alloc(&tmp, n_cols * n));
    HIP_TRY(ctx, hipMemcpy2DAsync(tmp, n * 8, d_coeffs, in_stride * 8, n * 8, n_cols,
                                  hipMemcpyDeviceToDevice, ctx->stream));
    int rc = bitrev_columns(ctx, tmp, n, n_cols, log_n);
    const u64 *tab = nullptr;
    if (rc == ZK_OK) rc = get_coset_table(ctx, log_n, GL_GENERATOR, false, &tab);
    if (rc == ZK_OK)
        rc = ntt_coeffs_to_values(ctx, tmp, n, (u64 *)d_out, out_stride, n_cols, log_n, rate_bits, tab);
    return rc;
}

extern "C" int zk_gl_vec_op(zk_ctx *ctx, uint32_t op, const uint64_t *d_a, const uint64_t *d_b,
                            uint64_t *d_out, size_t n) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    if (op > 8) return set_err(ctx, ZK_ERR_BAD_ARG, "unknown field op %u", op);
    if (!n) return ZK_OK;
    if (!d_a || !d_out || (!d_b && op != 3 && op != 4)) return set_err(ctx, ZK_ERR_BAD_ARG, "null pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    gl_vec_op_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(
        op, (const u64 *)d_a, (const u64 *)d_b, (u64 *)d_out, n);
    return check_launch(ctx, "gl_vec_op_kernel");
}

// ------------------------------------------------------------------------------------------
// hashing
extern "C" int zk_poseidon_permute(zk_ctx *ctx, uint64_t *d_states, size_t n_states) {
    if (!ctx || (!d_states && n_states)) return ZK_ERR_BAD_ARG;
    if (!n_states) return ZK_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    poseidon_permute_states_kernel<<<(unsigned)((n_states + 255) / 256), 256, 0, ctx->stream>>>(
        (u64 *)d_states, n_states);
    return check_launch(ctx, "poseidon_permute_states_kernel");
}
extern "C" int zk_keccak_f1600(zk_ctx *ctx, uint64_t *d_states, size_t n_states) {
    if (!ctx || (!d_states && n_states)) return ZK_ERR_BAD_ARG;
    if (!n_states) return ZK_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    keccak_f1600_states_kernel<<<(unsigned)((n_states + 255) / 256), 256, 0, ctx->stream>>>(
        (u64 *)d_states, n_states);
    return check_launch(ctx, "keccak_f1600_states_kernel");
}

extern "C" int zk_hash_rows(zk_ctx *ctx, uint32_t hasher, const uint64_t *d_cols, size_t col_stride,
                            size_t n_cols, size_t n_rows, uint64_t *d_digests) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    if (!d_digests || (!d_cols && n_cols)) return set_err(ctx, ZK_ERR_BAD_ARG, "null pointer");
    if (!n_rows) return ZK_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return hash_rows(ctx, hasher, (const u64 *)d_cols, col_stride, n_cols, n_rows, 0, 0, (u64 *)d_digests);
}

extern "C" size_t zk_merkle_num_digests(unsigned log_leaves, unsigned cap_height) {
    size_t tot = 0;
    if (cap_height > log_leaves) return 0;
    for (unsigned l = log_leaves + 1; l-- > cap_height;) tot += (size_t)1 << l;
    return tot;
}

extern "C" int zk_merkle_build(zk_ctx *ctx, uint32_t hasher, uint64_t *d_digests, unsigned log_leaves,
                               unsigned cap_height) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    if (!d_digests || cap_height > log_leaves || log_leaves > 31)
        return set_err(ctx, ZK_ERR_BAD_ARG, "bad merkle arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return merkle_levels(ctx, hasher, (u64 *)d_digests, log_leaves, cap_height);
}

// ------------------------------------------------------------------------------------------
// PolynomialBatch
extern "C" void zk_batch_free(zk_batch *b) {
    if (!b) return;
    hipSetDevice(b->ctx->device);
    b->ctx->live_batches.erase(b);
    zk_ctx *ctx = b->ctx;   // blocks return to the ctx arena (reused in stream order)
    if (!b->borrowed) {
        ctx->arena.free(b->d_coeffs);
        ctx->arena.free(b->d_lde);
        ctx->arena.free(b->d_digests);
    }
    delete b;
}

static int check_cfg(zk_ctx *ctx, const zk_cfg *cfg, size_t n_cols, unsigned log_n) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    if (!cfg) return set_err(ctx, ZK_ERR_BAD_ARG, "null cfg");
    if (n_cols == 0) return set_err(ctx, ZK_ERR_BAD_ARG, "empty batch (n_cols == 0)");
    if (cfg->hasher > ZK_HASH_KECCAK25) return set_err(ctx, ZK_ERR_BAD_ARG, "unknown hasher");
    // the NTT passes read their twiddle tables through buffer loads with 32-bit byte offsets (ntt.cuh): 2^28 points = 2 GiB
    if (log_n + cfg->rate_bits > 28)
        return set_err(ctx, ZK_ERR_UNSUPPORTED, "LDE size 2^%u exceeds the supported 2^28 points", log_n + cfg->rate_bits);
    if (cfg->cap_height > log_n + cfg->rate_bits)
        return set_err(ctx, ZK_ERR_BAD_ARG, "cap_height %u exceeds tree height %u (plonky2 asserts the same)",
                       cfg->cap_height, log_n + cfg->rate_bits);
    return ZK_OK;
}

// ---- commits in flight -----------------------------------------------------------------------------------------
// A commitment is enqueued (iNTT, LDE, leaf hashing, tree, cap read-back into a pinned slot) without any host
// synchronisation and finished later (cap copied into the batch, stage times added to the ctx totals): the segment
// driver keeps several in flight, on the main stream and on the side lane.
#define ZK_CAP_SLOTS 64
struct PendingCommit {
    zk_batch *b = nullptr;
    hipEvent_t ev[5] = {};
    hipEvent_t tail_ev = nullptr;       // main stream -> tail stream hand-over inside merkle_levels
    hipStream_t stream = nullptr;
    const u64 *h_cap = nullptr;
    u64 *h_cap_own = nullptr;           // caps above 64 words (cap_height > 4) land in their own pinned buffer
    CommitMode mode = COMMIT_VALUES;
    bool side = false;
    bool deferred_tree = false;         // the small levels and the cap read-back are commit_tree_batch_flush's
};
static hipEvent_t ev_get(zk_ctx *ctx) {
    if (!ctx->ev_pool.empty()) { hipEvent_t e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}
static void ev_put(zk_ctx *ctx, hipEvent_t e) { if (e) ctx->ev_pool.push_back(e); }
static void pending_release(zk_ctx *ctx, PendingCommit &pc) {
    for (auto &e : pc.ev) { ev_put(ctx, e); e = nullptr; }
    ev_put(ctx, pc.tail_ev); pc.tail_ev = nullptr;
    if (pc.h_cap_own) { (void)hipHostFree(pc.h_cap_own); pc.h_cap_own = nullptr; }
}

static int commit_enqueue(zk_ctx *ctx, const zk_cfg *cfg, const u64 *d_in, size_t in_stride,
                          size_t n_cols, unsigned log_n, CommitMode mode, PendingCommit *pc) {
    HostProf hp("commit_enqueue");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ZK_TRY(check_abort(ctx));
    const size_t n = (size_t)1 << log_n;
    const unsigned log_N = log_n + cfg->rate_bits;
    const size_t N = (size_t)1 << log_N;
    if (!ctx->h_caps) HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_caps, (size_t)ZK_CAP_SLOTS * 64 * sizeof(u64), hipHostMallocDefault));
    zk_batch *b = new zk_batch();
    ctx->live_batches.insert(b);
    b->ctx = ctx; b->n_cols = n_cols; b->log_n = log_n; b->rate_bits = cfg->rate_bits;
    b->cap_height = cfg->cap_height; b->hasher = cfg->hasher;
    b->n_digests = zk_merkle_num_digests(log_N, cfg->cap_height);
    int rc = ZK_OK;
    auto fail = [&](int code) { pending_release(ctx, *pc); zk_batch_free(b); return code; };
#define B_HIP(expr)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(set_err(ctx, e_ == hipErrorOutOfMemory ? ZK_ERR_OOM : ZK_ERR_HIP,    \
                                "%s failed: %s", #expr, hipGetErrorString(e_)));             \
    } while (0)
    // arena blocks of the current lane: repeated commits reuse the same HBM without hipMalloc cost
    B_HIP(ctx->arena.alloc((void **)&b->d_coeffs, n_cols * n * sizeof(u64)));
    B_HIP(ctx->arena.alloc((void **)&b->d_lde, n_cols * N * sizeof(u64)));
    B_HIP(ctx->arena.alloc((void **)&b->d_digests, b->n_digests * 32));

    const u64 *coset = nullptr;
    if ((rc = get_coset_table(ctx, log_n, GL_GENERATOR, false, &coset)) != ZK_OK) return fail(rc);
    pc->b = b; pc->stream = ctx->stream; pc->mode = mode; pc->side = ctx->arena.cur != 0;
    for (auto &e : pc->ev) e = ev_get(ctx);
    // make sure table construction is not billed to a stage
    hipEventRecord(pc->ev[0], ctx->stream);
    if (mode == COMMIT_VALUES) {
        // inverse transform and extension as one plan (the last pass of the one and the first of the other fused on their
        // common tiles, ntt_host.inc); ev[1] is recorded after the fused pass
        rc = ntt_values_to_coeffs_to_lde(ctx, d_in, in_stride, b->d_coeffs, n, b->d_lde, N, n_cols, log_n, cfg->rate_bits,
                                         coset, pc->ev[1]);
        if (rc != ZK_OK) return fail(rc);
    } else {
        B_HIP(hipMemcpy2DAsync(b->d_coeffs, n * 8, d_in, in_stride * 8, n * 8, n_cols,
                               hipMemcpyDeviceToDevice, ctx->stream));
        if (mode == COMMIT_COEFFS) rc = bitrev_columns(ctx, b->d_coeffs, n, n_cols, log_n);
        if (rc != ZK_OK) return fail(rc);
        hipEventRecord(pc->ev[1], ctx->stream);
        if ((rc = check_abort(ctx)) != ZK_OK) return fail(rc);
        rc = ntt_coeffs_to_values(ctx, b->d_coeffs, n, b->d_lde, N, n_cols, log_n, cfg->rate_bits, coset);
        if (rc != ZK_OK) return fail(rc);
    }
    hipEventRecord(pc->ev[2], ctx->stream);
    if ((rc = check_abort(ctx)) != ZK_OK) return fail(rc);
    rc = hash_rows(ctx, cfg->hasher, b->d_lde, N, n_cols, N, (int)log_N, 1, b->d_digests);
    if (rc != ZK_OK) return fail(rc);
    hipEventRecord(pc->ev[3], ctx->stream);
    // The levels of <= 2^kTreeTailLog nodes are latency-bound (one or two waves per SIMD at best, then the cooperative
    // kernels): when the caller enqueues several commitments back to back on the main lane (the trace commitments of a
    // segment) they and the cap read-back move to the tail stream, and the next commitment's NTT starts under them.
    if (ctx->tree_batch && cfg->hasher == ZK_HASH_POSEIDON && b->cap.empty() && ((size_t)4 << cfg->cap_height) <= 64) {
        // r05: the small levels of all the trees of a phase are built together, one launch per level (commit_tree_batch_flush);
        // this lane only builds the levels that fill the chip and says when they are done
        rc = merkle_levels(ctx, cfg->hasher, b->d_digests, log_N, cfg->cap_height, nullptr, nullptr, kTreeBatchTopLog);
        if (rc != ZK_OK) return fail(rc);
        pc->tail_ev = ev_get(ctx);
        B_HIP(hipEventRecord(pc->tail_ev, ctx->stream));
        pc->deferred_tree = true;
        ctx->tree_batch->push_back(pc);
        return ZK_OK;
    }
    hipStream_t tail = ctx->commit_tail && !pc->side ? ctx->commit_tail : nullptr;
    if (tail) pc->tail_ev = ev_get(ctx);
    rc = merkle_levels(ctx, cfg->hasher, b->d_digests, log_N, cfg->cap_height, tail, pc->tail_ev);
    if (rc != ZK_OK) return fail(rc);
    hipStream_t const main_stream = ctx->stream;
    if (tail) ctx->stream = tail;                       // (the tree ended there: merkle_levels hands over at its first small level)
    hipEventRecord(pc->ev[4], ctx->stream);
    b->cap.resize((size_t)4 << cfg->cap_height);
    u64 *slot = nullptr;
    if (b->cap.size() <= 64) slot = ctx->h_caps + (ctx->cap_slot_next++ % ZK_CAP_SLOTS) * 64;
    else if (hipHostMalloc((void **)&pc->h_cap_own, b->cap.size() * 8, hipHostMallocDefault) == hipSuccess) slot = pc->h_cap_own;
    else { ctx->stream = main_stream; return fail(set_err(ctx, ZK_ERR_OOM, "pinned buffer for a cap of 2^%u digests", cfg->cap_height)); }
    pc->h_cap = slot;
    hipError_t cap_rc = copy_to_pinned(ctx, slot, b->d_digests + 4 * (b->n_digests - ((size_t)1 << cfg->cap_height)), b->cap.size() * 8);
    ctx->stream = main_stream;
    B_HIP(cap_rc);
#undef B_HIP
    return ZK_OK;
}

// The commitments collected in ctx->tree_batch: wait (on `st`) for each one's large levels, build the remaining levels of all
// the trees level by level in ONE launch each, read the caps back.  The caller synchronises `st` before commit_finish.
// ZK_TREE_BATCH: 0 = per tree (r01-r04), 1 = batched, 2 (default) = what the ctx's plan table says ("T=1;"; no item: per tree).
static const int kTreeBatch = env_int("ZK_TREE_BATCH", 2, 0, 2);
static bool tree_batch_on(const zk_ctx *ctx) {
    if (ctx->tree_batch_force >= 0) return ctx->tree_batch_force == 1;      // (the tuner's own runs, tune_host.inc)
    if (kTreeBatch != 2) return kTreeBatch == 1;
    return plans_tree_tops(ctx->plans) == 1;
}
static int commit_tree_batch_flush(zk_ctx *ctx, const zk_cfg *cfg, hipStream_t st) {
    std::vector<PendingCommit *> items;
    if (ctx->tree_batch) items.swap(*ctx->tree_batch);
    if (items.empty()) return ZK_OK;
    std::vector<u64 *> dig;
    std::vector<unsigned> logs;
    for (PendingCommit *pc : items) {
        HIP_TRY(ctx, hipStreamWaitEvent(st, pc->tail_ev, 0));
        dig.push_back(pc->b->d_digests);
        logs.push_back(pc->b->log_n + pc->b->rate_bits);
    }
    ZK_TRY(merkle_levels_batched(ctx, dig.data(), logs.data(), items.size(), cfg->cap_height, (unsigned)kTreeBatchTopLog, st));
    hipStream_t const keep = ctx->stream;
    ctx->stream = st;
    hipError_t e = hipSuccess;
    for (PendingCommit *pc : items) {
        zk_batch *b = pc->b;
        hipEventRecord(pc->ev[4], st);
        b->cap.resize((size_t)4 << cfg->cap_height);
        u64 *slot = ctx->h_caps + (ctx->cap_slot_next++ % ZK_CAP_SLOTS) * 64;
        pc->h_cap = slot;
        pc->stream = st;
        if (e == hipSuccess) e = copy_to_pinned(ctx, slot, b->d_digests + 4 * (b->n_digests - ((size_t)1 << cfg->cap_height)), b->cap.size() * 8);
    }
    ctx->stream = keep;
    if (e != hipSuccess) return set_err(ctx, ZK_ERR_HIP, "cap read-back: %s", hipGetErrorString(e));
    return ZK_OK;
}

// The commit's stream must have been drained past its cap read-back (stream or event synchronisation by the caller).
static void commit_finish(zk_ctx *ctx, PendingCommit *pc) {
    zk_batch *b = pc->b;
    memcpy(b->cap.data(), pc->h_cap, b->cap.size() * 8);
    const size_t n = (size_t)1 << b->log_n, N = n << b->rate_bits;
    double *tot = pc->side ? ctx->side_ms : ctx->total_ms;
    for (int i = 0; i < 4; ++i) {
        float ms = 0;
        hipEventElapsedTime(&ms, pc->ev[i], pc->ev[i + 1]);
        if (!pc->side) ctx->timings[i] = ms;
        tot[i] += ms;
    }
    // algorithmic bytes of the leaf-hash launch: read the LDE once, write N digests
    const double leaf_bytes = 8.0 * (double)b->n_cols * (double)N + 32.0 * (double)N;
    const double ntt_bytes = (pc->mode == COMMIT_VALUES ? 40.0 : 24.0) * (double)b->n_cols * (double)n;
    if (pc->side) {
        ctx->side_leaf_bytes += leaf_bytes; ctx->side_ntt_bytes += ntt_bytes; ctx->side_commits += 1;
    } else {
        ctx->total_leaf_bytes += leaf_bytes;
        ctx->total_leaf_perms += b->n_cols > 4 ? (double)N * (double)((b->n_cols + 7) / 8) : 0.0;
        ctx->total_ntt_bytes += ntt_bytes;
        ctx->total_commits += 1;
    }
    pending_release(ctx, *pc);
}

static int commit_impl(zk_ctx *ctx, const zk_cfg *cfg, const u64 *d_in, size_t in_stride,
                       size_t n_cols, unsigned log_n, CommitMode mode, zk_batch **out) {
    PendingCommit pc;
    ZK_TRY(commit_enqueue(ctx, cfg, d_in, in_stride, n_cols, log_n, mode, &pc));
    hipError_t e = zk_stream_wait(ctx, pc.stream);
    if (e != hipSuccess) {
        pending_release(ctx, pc);
        zk_batch_free(pc.b);
        return set_err(ctx, ZK_ERR_HIP, "commit: %s", hipGetErrorString(e));
    }
    commit_finish(ctx, &pc);
    *out = pc.b;
    return ZK_OK;
}

extern "C" int zk_commit_columns_device(zk_ctx *ctx, const zk_cfg *cfg, const uint64_t *d_values,
                                        size_t col_stride, size_t n_cols, unsigned log_n, zk_batch **out) {
    if (!out) return ZK_ERR_BAD_ARG;
    *out = nullptr;
    ZK_TRY(check_cfg(ctx, cfg, n_cols, log_n));
    if (!d_values) return set_err(ctx, ZK_ERR_BAD_ARG, "null values");
    if (n_cols > 1 && col_stride < ((size_t)1 << log_n)) return set_err(ctx, ZK_ERR_BAD_ARG, "col_stride < n");
    return commit_impl(ctx, cfg, (const u64 *)d_values, col_stride, n_cols, log_n, COMMIT_VALUES, out);
}

extern "C" int zk_commit_coeffs_device(zk_ctx *ctx, const zk_cfg *cfg, const uint64_t *d_coeffs,
                                       size_t col_stride, size_t n_cols, unsigned log_n, zk_batch **out) {
    if (!out) return ZK_ERR_BAD_ARG;
    *out = nullptr;
    ZK_TRY(check_cfg(ctx, cfg, n_cols, log_n));
    if (!d_coeffs) return set_err(ctx, ZK_ERR_BAD_ARG, "null coeffs");
    if (n_cols > 1 && col_stride < ((size_t)1 << log_n)) return set_err(ctx, ZK_ERR_BAD_ARG, "col_stride < n");
    return commit_impl(ctx, cfg, (const u64 *)d_coeffs, col_stride, n_cols, log_n, COMMIT_COEFFS, out);
}

extern "C" int zk_commit_columns(zk_ctx *ctx, const zk_cfg *cfg, const uint64_t *const *cols,
                                 size_t n_cols, unsigned log_n, zk_batch **out) {
    if (!out) return ZK_ERR_BAD_ARG;
    *out = nullptr;
    ZK_TRY(check_cfg(ctx, cfg, n_cols, log_n));
    if (!cols) return set_err(ctx, ZK_ERR_BAD_ARG, "null column array");
    for (size_t c = 0; c < n_cols; ++c)
        if (!cols[c]) return set_err(ctx, ZK_ERR_BAD_ARG, "null column %zu", c);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << log_n;
    u64 *d_vals = nullptr;
    HIP_TRY(ctx, hipMalloc(&d_vals, n_cols * n * sizeof(u64)));
    for (size_t c = 0; c < n_cols; ++c) {
        hipError_t e = hipMemcpyAsync(d_vals + c * n, cols[c], n * sizeof(u64), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { hipFree(d_vals); return set_err(ctx, ZK_ERR_HIP, "H2D copy failed: %s", hipGetErrorString(e)); }
    }
    int rc = commit_impl(ctx, cfg, d_vals, n, n_cols, log_n, COMMIT_VALUES, out);
    hipStreamSynchronize(ctx->stream);
    hipFree(d_vals);
    return rc;
}

extern "C" size_t zk_batch_num_cols(const zk_batch *b) { return b ? b->n_cols : 0; }
extern "C" unsigned zk_batch_log_n(const zk_batch *b) { return b ? b->log_n : 0; }
extern "C" unsigned zk_batch_log_lde(const zk_batch *b) { return b ? b->log_n + b->rate_bits : 0; }
extern "C" const uint64_t *zk_batch_lde_device(const zk_batch *b) { return b ? (const uint64_t *)b->d_lde : nullptr; }
extern "C" const uint64_t *zk_batch_digests_device(const zk_batch *b) { return b ? (const uint64_t *)b->d_digests : nullptr; }

extern "C" int zk_batch_cap(const zk_batch *b, uint64_t *out) {
    if (!b || !out) return ZK_ERR_BAD_ARG;
    memcpy(out, b->cap.data(), b->cap.size() * 8);
    return ZK_OK;
}

extern "C" int zk_batch_coeffs(const zk_batch *b, size_t col, uint64_t *out) {
    if (!b || !out) return ZK_ERR_BAD_ARG;
    zk_ctx *ctx = b->ctx;
    if (col >= b->n_cols) return set_err(ctx, ZK_ERR_BAD_ARG, "column %zu out of range", col);
    if (!b->d_coeffs) return set_err(ctx, ZK_ERR_BAD_ARG, "this batch view holds no coefficients (zk_batch_from_parts row shard)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t n = (size_t)1 << b->log_n;
    std::vector<u64> tmp(n);
    HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), b->d_coeffs + col * n, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < n; ++i) out[bitrev32((u32)i, b->log_n)] = gl_canon(tmp[i]);
    return ZK_OK;
}

static int gather_row(const zk_batch *b, size_t natural_row, uint64_t *out) {
    zk_ctx *ctx = b->ctx;
    // a view made by zk_batch_from_parts may hold one shard's leaf-ordered rows, or no values at all: not indexable as a
    // whole natural-order batch of stride N
    if (b->row_shard || !b->d_lde) return set_err(ctx, ZK_ERR_BAD_ARG, "this batch view holds no whole natural-order LDE (row shard / column-only view)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t N = (size_t)1 << (b->log_n + b->rate_bits);
    HIP_TRY(ctx, hipMemcpy2DAsync(out, 8, b->d_lde + natural_row, N * 8, 8, b->n_cols,
                                  hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

extern "C" int zk_batch_leaf(const zk_batch *b, size_t leaf_index, uint64_t *out) {
    if (!b || !out) return ZK_ERR_BAD_ARG;
    unsigned log_N = b->log_n + b->rate_bits;
    if (leaf_index >> log_N) return set_err(b->ctx, ZK_ERR_BAD_ARG, "leaf index out of range");
    return gather_row(b, bitrev32((u32)leaf_index, log_N), out);
}

extern "C" int zk_batch_lde_values(const zk_batch *b, size_t index, size_t step, uint64_t *out) {
    if (!b || !out) return ZK_ERR_BAD_ARG;
    unsigned log_N = b->log_n + b->rate_bits;
    if (step != 0 && index > SIZE_MAX / step) return set_err(b->ctx, ZK_ERR_BAD_ARG, "lde index out of range");
    size_t nat = index * step;
    if (nat >> log_N) return set_err(b->ctx, ZK_ERR_BAD_ARG, "lde index out of range");
    return gather_row(b, nat, out);  // leaves[bitrev(index*step)] == natural row index*step
}

extern "C" int zk_batch_merkle_path(const zk_batch *b, size_t leaf_index, uint64_t *out) {
    if (!b || !out) return ZK_ERR_BAD_ARG;
    zk_ctx *ctx = b->ctx;
    unsigned log_N = b->log_n + b->rate_bits;
    if (leaf_index >> log_N) return set_err(ctx, ZK_ERR_BAD_ARG, "leaf index out of range");
    if (b->row_shard || !b->d_digests) return set_err(ctx, ZK_ERR_BAD_ARG, "this batch view holds no whole Merkle tree (row shard / column-only view)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const u64 *lvl = b->d_digests;
    size_t idx = leaf_index;
    for (unsigned l = log_N; l > b->cap_height; --l) {
        HIP_TRY(ctx, hipMemcpyAsync(out, lvl + 4 * (idx ^ 1), 32, hipMemcpyDeviceToHost, ctx->stream));
        out += 4;
        lvl += 4 * ((size_t)1 << l);
        idx >>= 1;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

#include "fri_host.inc"
#include "stark_host.inc"
#include "quotient_host.inc"
#include "segment_host.inc"
#include "tune_host.inc"
#include "shard_host.inc"
#include "comm_host.inc"
#include "shard_prove_host.inc"

// ---- the cross-TU interface (internal.hpp) ----------------------------------------------------------------------
int zki_commit(zk_ctx *ctx, const zk_cfg *cfg, const u64 *d_in, size_t in_stride, size_t n_cols, unsigned log_n,
               CommitMode mode, zk_batch **out) {
    return commit_impl(ctx, cfg, d_in, in_stride, n_cols, log_n, mode, out);
}
int zki_get_twiddles(zk_ctx *ctx, int log_size, bool inverse, const u64 **out) { return get_twiddles(ctx, log_size, inverse, out); }
int zki_get_coset_table(zk_ctx *ctx, int log_n, u64 shift, bool inverse, const u64 **out) {
    return get_coset_table(ctx, log_n, shift, inverse, out);
}
// ---- the plan table (ntt_host.inc) ----------------------------------------------------------------------------------
extern "C" int zk_ctx_set_plans(zk_ctx *ctx, const char *plans) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    if (!plans) { ctx->plans = initial_plans(); return ZK_OK; }
    if (!plans_well_formed(plans) || strlen(plans) > 4096) return set_err(ctx, ZK_ERR_BAD_ARG, "zk_ctx_set_plans: not a plan string (items \"v20f0=2;b20r1=96x1;T=1;\")");
    ctx->plans = plans;
    return ZK_OK;
}
extern "C" size_t zk_ctx_get_plans(const zk_ctx *ctx, char *out, size_t max) {
    if (!ctx) return 0;
    if (out && max) { const size_t n = ctx->plans.size() < max - 1 ? ctx->plans.size() : max - 1; memcpy(out, ctx->plans.data(), n); out[n] = 0; }
    return ctx->plans.size();
}
// (internal; the offline tuner csrc/ntt_tune_main.c) what the trials on this ctx said, one line each
extern "C" size_t zki_ntt_tune_report(const zk_ctx *ctx, char *out, size_t max) {
    if (!ctx) return 0;
    if (out && max) { const size_t n = ctx->tune_report.size() < max - 1 ? ctx->tune_report.size() : max - 1; memcpy(out, ctx->tune_report.data(), n); out[n] = 0; }
    return ctx->tune_report.size();
}
// (internal; the offline tuner, and at small sizes the CPU emulation tests) the trials of the transform shapes 2^min_log .. 2^max_log
// and of the from_values shapes 2^batch_min_log .. 2^batch_max_log rows (batch_mb MiB batches): see zki_ntt_tune_all
extern "C" int zki_ntt_tune_range(zk_ctx *ctx, int min_log, int max_log, int batch_min_log, int batch_max_log, int batch_mb, int *n_differ) {
    if (!ctx) return ZK_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int differ = 0;
    for (int dit = 0; dit < 2; ++dit)
        for (int free_stages = 0; free_stages <= dit; ++free_stages)
            for (int L = std::max(min_log, ZK_NTT_WAVE_BITS + free_stages); L <= std::min(max_log, 22); ++L) {
                if (!ntt_swap_has_plan(dit != 0, L, free_stages)) continue;
                bool d = false;
                ZK_TRY(ntt_swap_trial(ctx, dit != 0, L, free_stages, &d));
                differ += d;
            }
    // with the plans decided: the column batches of the shapes that have a trial (rate_bits = 1: the STARK tables)
    for (int log_n = batch_min_log; log_n <= batch_max_log && log_n + 1 <= 22; ++log_n) {
        bool d = false;
        ZK_TRY(ntt_batch_trial(ctx, log_n, 1, &d, batch_mb));
        differ += d;
    }
    if (n_differ) *n_differ = differ;
    return ZK_OK;
}
// (internal; the offline tuner) both plans of every transform shape that has a lane-swap plan and the column-batch forms of every
// from_values shape that has a trial, run on the device and compared; the verdicts go into THIS ctx's plan table (zk_ctx_get_plans)
// and report.  *n_differ = the number of trials in which the second form produced different words: a parity failure of shipped code.
extern "C" int zki_ntt_tune_all(zk_ctx *ctx, int *n_differ) { return zki_ntt_tune_range(ctx, ZK_NTT_WAVE_BITS, 22, 17, 21, 96, n_differ); }
// (internal, for tests: no device involved) what a plan string says about a shape: 0 = no item, 1 = tile, 2 = lane-swap, -1 = the
// shape has no second plan; the column-batch item as MiB * 4 + streams (-1 = none); the tree tops (-1 = none)
extern "C" int zki_plans_ntt(const char *plans, int dit, int L, int free_stages) {
    if (!ntt_swap_has_plan(dit != 0, L, free_stages)) return -1;
    return plans_ntt(plans ? plans : "", dit != 0, L, free_stages);
}
extern "C" int zki_plans_batch(const char *plans, int log_n, int rate_bits) {
    int mb = 0, st = 1;
    return plans_batch(plans ? plans : "", log_n, rate_bits, &mb, &st) ? mb * 4 + st : -1;
}
extern "C" int zki_plans_tree_tops(const char *plans) { return plans_tree_tops(plans ? plans : ""); }
extern "C" const char *zki_builtin_plans(void) { return kBuiltinPlans; }
// (internal, for tests/test_ntt_plan_cpu.py: no device involved) the passes ntt_host.inc plans for a 2^L-point transform whose
// contiguous pass gets `free_stages` stages by replication: out[2 k] = log_d, out[2 k + 1] = r of pass k, largest distance first
extern "C" int zki_ntt_plan(int L, int free_stages, int dit, int *out, int max_passes) {
    if (L < 0 || L > 31 || free_stages < 0 || !out) return -1;
    const auto plan = plan_passes_for(L, free_stages, kNttSwap != 0, dit != 0);      // (the lane-swap plan wherever the switch allows one at all)
    int k = 0;
    for (const auto &ps : plan) { if (k >= max_passes) return -1; out[2 * k] = ps.log_d; out[2 * k + 1] = ps.r; ++k; }
    return k;
}
int zki_ntt_values_to_coeffs(zk_ctx *ctx, const u64 *src, size_t src_stride, u64 *dst, size_t dst_stride, size_t n_cols,
                             int log_n, const u64 *out_scale) {
    return ntt_values_to_coeffs(ctx, src, src_stride, dst, dst_stride, n_cols, log_n, out_scale);
}
