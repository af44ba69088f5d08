// zk_ctx and the error / scratch helpers shared by every host-side include of libzkstark_hip.so (and by the kernel
// micro-benchmark tools/kbench.hip, which links the same NTT / Merkle host code without the rest of the library).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/zkstark.h"
#include "arena.hpp"
#include "gl.cuh"

// Deferred host synchronisation (zk_prove_segment): while a sink is installed, the helper-column passes neither wait for
// their kernels nor read their error flag -- the host buffers they upload from are parked here, every pass takes an error
// slot, and the driver checks the slots once per join point.  That is what lets the auxiliary pipeline run ahead on the side
// lane while the host serves the Fiat-Shamir round trips of the per-table chain.
struct AsyncSink {
    std::deque<std::vector<u64>> keep64;
    std::deque<std::vector<u32>> keep32;
    int *d_errs = nullptr;      // device, `cap` slots, zeroed
    int *h_errs = nullptr;      // pinned host mirror
    u32 used = 0, cap = 0;
    std::vector<u64> &new64() { keep64.emplace_back(); return keep64.back(); }
    std::vector<u32> &new32() { keep32.emplace_back(); return keep32.back(); }
};

// ------------------------------------------------------------------------------------------
struct zk_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    volatile const int *abort_flag = nullptr;
    volatile const uint8_t *abort_flag_u8 = nullptr;   // the reference's AtomicBool (zk_ctx_set_abort_flag_u8)
    std::map<int, u64 *> tw_fwd, tw_inv, tw_inv_br;         // log size -> table (tw_inv_br: block-order levels)
    std::map<std::pair<int, u64>, u64 *> coset_tabs;        // (log_n, shift) -> s^bitrev(i)
    std::map<std::pair<int, u64>, u64 *> coset_inv_tabs;    // (log_n, shift) -> n^-1 s^-bitrev(i)
    hipStream_t side_stream = nullptr;  // lane 1: low priority, created on first use (segment_host.inc)
    std::vector<hipEvent_t> ev_pool;    // recycled timing / ordering events
    u64 *h_caps = nullptr;              // pinned host slots for cap read-backs of commits in flight (ZK_CAP_SLOTS x 64 words)
    uint64_t cap_slot_next = 0;
    float timings[4] = {0, 0, 0, 0};
    // running totals over all commits since the last reset (zk_ctx_commit_totals)
    double total_ms[4] = {0, 0, 0, 0};
    double total_leaf_bytes = 0, total_leaf_perms = 0, total_ntt_bytes = 0;
    uint64_t total_commits = 0;
    // the same for commits that ran on the side lane, i.e. concurrently with main-lane kernels: their event-to-event
    // times include the sharing of the chip and are kept out of the roofline figures
    double side_ms[4] = {0, 0, 0, 0};
    double side_leaf_bytes = 0, side_ntt_bytes = 0;
    uint64_t side_commits = 0;
    int cu_count = 0;
    std::set<zk_batch *> live_batches;  // freed by zk_ctx_destroy if the caller leaked them
    ArenaSet arena;                     // all batch + scratch HBM, one arena per lane (arena.hpp)
    // debug: starky `check_ctls` after get_ctl_data (zk_ctx_set_check_ctls); extra looking rows per CTL index
    bool check_ctls = false;
    std::map<size_t, std::pair<size_t, std::vector<u64>>> ctl_extra;   // ctl -> (row width, rows)
    AsyncSink *async = nullptr;         // installed by zk_prove_segment for the duration of the call
    // pinned staging for small host -> device uploads (stage_upload below)
    std::vector<char *> stage_chunks;
    size_t stage_cur = 0, stage_off = 0, stage_since_rewind = 0;
    std::map<std::vector<u64>, u32> constraint_counts;   // quotient: constraints yielded per (AIR, lookup/CTL shape)
};

struct zk_batch {
    zk_ctx *ctx = nullptr;
    size_t n_cols = 0;
    unsigned log_n = 0, rate_bits = 0, cap_height = 0;
    uint32_t hasher = 0;
    u64 *d_coeffs = nullptr;   // [n_cols][n], bit-reversed coefficient order
    u64 *d_lde = nullptr;      // [n_cols][N], natural order
    u64 *d_digests = nullptr;  // level-concatenated 32-byte slots
    size_t n_digests = 0;
    std::vector<u64> cap;      // host copy
};

static int set_err(zk_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        ctx->err = buf;
    }
    return code;
}

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return set_err(ctx, e_ == hipErrorOutOfMemory ? ZK_ERR_OOM : ZK_ERR_HIP,         \
                           "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,  \
                           __LINE__);                                                        \
    } while (0)

#define ZK_TRY(expr)               \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != ZK_OK) return rc_; \
    } while (0)

// Scoped device scratch from the ctx arena: every block is returned on scope exit, on error paths too.
struct DevBuf {
    zk_ctx *ctx;
    std::vector<void *> ptrs;
    explicit DevBuf(zk_ctx *c) : ctx(c) {}
    ~DevBuf() { for (void *p : ptrs) ctx->arena.free(p); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    template <class T> int alloc(T **out, size_t count) {
        void *p = nullptr;
        hipError_t e = ctx->arena.alloc(&p, count * sizeof(T) ? count * sizeof(T) : 8);
        if (e != hipSuccess) return set_err(ctx, e == hipErrorOutOfMemory ? ZK_ERR_OOM : ZK_ERR_HIP, "device arena: %s", hipGetErrorString(e));
        ptrs.push_back(p);
        *out = (T *)p;
        return ZK_OK;
    }
};

// Work enqueued inside the scope goes to the ctx's side stream and takes its memory from the side arena.
struct LaneScope {
    zk_ctx *ctx;
    hipStream_t prev_stream;
    int prev_lane;
    LaneScope(zk_ctx *c, hipStream_t s, int lane) : ctx(c), prev_stream(c->stream), prev_lane(c->arena.cur) {
        ctx->stream = s;
        ctx->arena.cur = lane;
    }
    ~LaneScope() { ctx->stream = prev_stream; ctx->arena.cur = prev_lane; }
    LaneScope(const LaneScope &) = delete;
    LaneScope &operator=(const LaneScope &) = delete;
};

// Small host -> device uploads (programs, compiled entries, descriptors, coefficient tables).  A hipMemcpyAsync from
// pageable memory stalls the calling thread on a staging copy and forced "host buffer goes out of scope" synchronisations;
// here the bytes are copied into pinned memory the ctx owns and the transfer is truly asynchronous on ctx->stream, so the
// caller's buffer is free again on return and no synchronisation is needed.  The pinned chunks are reused from the start
// whenever the streams are known idle (stage_rewind: end of a segment proof; or after STAGE_LIMIT bytes, with one sync).
static constexpr size_t ZK_STAGE_CHUNK = (size_t)1 << 20, ZK_STAGE_LIMIT = (size_t)16 << 20;
static inline void stage_rewind(zk_ctx *ctx) { ctx->stage_cur = 0; ctx->stage_off = 0; ctx->stage_since_rewind = 0; }
static hipError_t stage_upload(zk_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    if (!bytes) return hipSuccess;
    if (bytes > ZK_STAGE_CHUNK / 2) return hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (!ctx->async && ctx->stage_since_rewind + bytes > ZK_STAGE_LIMIT) {   // standalone calls: bound the footprint
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return e;
        if (ctx->side_stream && (e = hipStreamSynchronize(ctx->side_stream)) != hipSuccess) return e;
        stage_rewind(ctx);
    }
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (ctx->stage_chunks.empty() || ctx->stage_off + need > ZK_STAGE_CHUNK) {
        if (!ctx->stage_chunks.empty()) { ++ctx->stage_cur; ctx->stage_off = 0; }
        if (ctx->stage_cur >= ctx->stage_chunks.size()) {
            char *c = nullptr;
            hipError_t e = hipHostMalloc((void **)&c, ZK_STAGE_CHUNK, hipHostMallocDefault);
            if (e != hipSuccess) return e;
            ctx->stage_chunks.push_back(c);
        }
    }
    char *h = ctx->stage_chunks[ctx->stage_cur] + ctx->stage_off;
    memcpy(h, h_src, bytes);
    ctx->stage_off += need;
    ctx->stage_since_rewind += need;
    return hipMemcpyAsync(d_dst, h, bytes, hipMemcpyHostToDevice, ctx->stream);
}

static int check_abort(zk_ctx *ctx) {
    if ((ctx->abort_flag && *ctx->abort_flag) || (ctx->abort_flag_u8 && *ctx->abort_flag_u8))
        return set_err(ctx, ZK_ERR_ABORTED, "aborted");
    return ZK_OK;
}

static int check_launch(zk_ctx *ctx, const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return set_err(ctx, ZK_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return ZK_OK;
}

