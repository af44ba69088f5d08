// zk_ctx and the error / scratch helpers shared by every host-side include of libzkstark_hip.so (and by the kernel
// micro-benchmark tools/kbench.hip, which links the same NTT / Merkle host code without the rest of the library).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <memory>
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
    std::map<std::pair<int, const u64 *>, u64 *> wave_coset2_tabs;   // (log_n, load-factor table | null) -> the second coset's (ntt_host.inc)
    hipStream_t side_stream = nullptr;  // lane 1: low priority, created on first use (segment_host.inc)
    hipStream_t tail_stream = nullptr;  // tree tops of main-lane trace commitments (segment_host.inc), created on first use
    hipStream_t ntt_batch_stream = nullptr;   // every other column batch of a commitment's NTT (ntt_host.inc ZK_NTT_COL_BATCH_STREAMS), created on first use
    hipEvent_t ntt_batch_ev[2] = {nullptr, nullptr};
    hipStream_t commit_tail = nullptr;  // != nullptr: commit_enqueue sends the small Merkle levels + cap read-back of main-lane commits there
    // Which of two equivalent implementations serves a shape (ntt_host.inc "the plan table"): the ctx's own copy of the plan string,
    // set at creation (ZK_NTT_SWAP_PLANS or the table compiled in) and by zk_ctx_set_plans.  The *_force fields and the report belong to
    // the offline tuner (zki_ntt_tune_all, zki_tree_batch_trial), which runs both forms on this ctx.
    std::string plans, tune_report;
    int ntt_swap_force = -1, tree_batch_force = -1;
    std::vector<struct PendingCommit *> *tree_batch = nullptr;   // != nullptr: commit_enqueue leaves the small Merkle levels + cap read-back to commit_tree_batch_flush (zkstark.hip)
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
    uint32_t func_attrs = 0;            // kernels whose dynamic-LDS limit this ctx has raised on ITS device (a process may drive several GPUs: never once per process)
    std::set<zk_batch *> live_batches;  // freed by zk_ctx_destroy if the caller leaked them
    ArenaSet arena;                     // all batch + scratch HBM, one arena per lane (arena.hpp)
    // debug: starky `check_ctls` after get_ctl_data (zk_ctx_set_check_ctls); extra looking rows per CTL index
    bool check_ctls = false;
    std::map<size_t, std::pair<size_t, std::vector<u64>>> ctl_extra;   // ctl -> (row width, rows)
    AsyncSink *async = nullptr;         // installed by zk_prove_segment for the duration of the call
    // pinned staging for small host -> device uploads (stage_upload below)
    std::vector<char *> stage_chunks;
    size_t stage_cur = 0, stage_off = 0, stage_since_rewind = 0;
    struct StageReadback { const char *h_pinned; void *h_dst; size_t bytes; };
    std::vector<StageReadback> stage_pending;   // stage_download results not yet copied to their destinations
    char *h_big = nullptr;                      // pinned landing buffer for the large downloads (FRI query words)
    size_t h_big_bytes = 0;
    int *h_flags = nullptr;                     // pinned: the quotient kernels' deferred error flag (quotient_host.inc)
    volatile uint64_t *h_seq = nullptr;         // coherent pinned word the GPU bumps at a read-back point (zk_stream_wait)
    uint64_t seq_next = 0;
    std::map<std::pair<u64, u64>, std::shared_ptr<void>> entry_shapes;
    // device-resident copies of per-table-definition constants (programs, compiled shapes, AIR constants): const_upload
    struct DevConst { std::vector<u64> words; u64 *d = nullptr; };
    std::map<std::pair<u64, u64>, DevConst> dev_consts;
    std::map<std::pair<u64, u64>, std::shared_ptr<void>> dev_shapes;     // DevShape per table description: stark_host.inc
    size_t dev_const_bytes = 0;   // EntryShape per (program hash, words | slots): stark_host.inc
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
    // Views assembled by the caller from its own device memory (zk_batch_from_parts; SURVEY 8(e) level 3, sharding.py): the
    // blocks are not the arena's.  A COLUMN shard keeps d_coeffs of its columns only (openings); a ROW shard (shard_lw > 0)
    // keeps in d_lde the rows of ONE subtree group -- [n_cols][N >> shard_lw], row t = leaf shard_rank * (N >> shard_lw) + t,
    // i.e. LEAF order -- and in d_digests the levels of that subtree down to its 2^(cap_height - shard_lw) roots; `cap` is
    // the whole tree's cap (all-gathered by the caller).
    bool borrowed = false;
    bool row_shard = false;             // d_lde holds leaf-ordered rows of one shard (also with ONE rank: shard_lw = 0)
    unsigned shard_lw = 0, shard_rank = 0;
};

static int set_err(zk_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        ctx->err = buf;
        // An error return unwinds the frames that own the destinations of pending staged downloads (locals such as `err`,
        // `got`, scratch vectors): drop them, so that no later stage_collect on this ctx copies into a dead frame (r03
        // advisor).  The pinned chunks themselves stay valid.
        ctx->stage_pending.clear();
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

// ZK_HOST_PROFILE=1 (diagnostics): host time per named scope, printed when the ctx is destroyed
struct HostProf {
    static bool on() { static const bool v = getenv("ZK_HOST_PROFILE") && getenv("ZK_HOST_PROFILE")[0] == 0x31; return v; }
    static std::map<std::string, std::pair<double, u64>> &table() { static std::map<std::string, std::pair<double, u64>> t; return t; }
    static std::mutex &lock() { static std::mutex m; return m; }      // zk_plonk_prove_batch runs provers on several threads
    const char *name;
    std::chrono::steady_clock::time_point t0;
    explicit HostProf(const char *n) : name(n) { if (on()) t0 = std::chrono::steady_clock::now(); }
    ~HostProf() {
        if (!on()) return;
        std::lock_guard<std::mutex> g(lock());
        auto &e = table()[name];
        e.first += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        e.second += 1;
    }
    static void dump() {
        if (!on()) return;
        std::lock_guard<std::mutex> g(lock());
        for (auto &kv : table()) fprintf(stderr, "[zk host] %-36s %10.1f us total %8llu calls %8.1f us/call\n", kv.first.c_str(), kv.second.first, (unsigned long long)kv.second.second, kv.second.first / (double)kv.second.second);
    }
};

// Small host <-> device transfers (programs, compiled entries, descriptors, coefficient tables; caps, flags, opening values).
// On this ROCm a hipMemcpyAsync of a few hundred bytes costs the calling thread ~70 us -- pageable or pinned, either
// direction (kernel trace r03e: 242 back-to-back copies with 70 us gaps per two realistic-height proofs) -- where a kernel
// launch costs ~5.  So small transfers go through pinned memory the ctx owns and a copy KERNEL on ctx->stream:
//   stage_upload:   the bytes are memcpy'd into a pinned chunk, a kernel copies them to the device -- the caller's buffer is
//                   free again on return, no "host buffer goes out of scope" synchronisation;
//   stage_download: a kernel copies device words into a pinned chunk; after the caller has synchronised with the stream,
//                   stage_collect memcpy's them to their (pageable) destinations;
//   copy_to_pinned: the same for a destination that already is pinned memory.
// The pinned chunks are reused from the start whenever the streams are known idle (stage_rewind: end of a segment proof;
// or after ZK_STAGE_LIMIT bytes, with one synchronisation).
static constexpr size_t ZK_STAGE_CHUNK = (size_t)1 << 20, ZK_STAGE_LIMIT = (size_t)16 << 20;
static __global__ void zk_copy_words_kernel(u64 *__restrict__ dst, const u64 *__restrict__ src, size_t n_words) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) dst[i] = src[i];
}
static inline void stage_rewind(zk_ctx *ctx) { ctx->stage_cur = 0; ctx->stage_off = 0; ctx->stage_since_rewind = 0; }
// `bytes` of pinned memory (8-byte aligned, rounded up to 64), or nullptr
static char *stage_alloc(zk_ctx *ctx, size_t bytes, hipError_t *err) {
    *err = hipSuccess;
    if (!ctx->async && ctx->stage_pending.empty() && ctx->stage_since_rewind + bytes > ZK_STAGE_LIMIT) {
        if ((*err = hipStreamSynchronize(ctx->stream)) != hipSuccess) return nullptr;     // standalone calls: bound the footprint
        if (ctx->side_stream && (*err = hipStreamSynchronize(ctx->side_stream)) != hipSuccess) return nullptr;
        if (ctx->tail_stream && (*err = hipStreamSynchronize(ctx->tail_stream)) != hipSuccess) return nullptr;
        stage_rewind(ctx);
    }
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (ctx->stage_chunks.empty() || ctx->stage_off + need > ZK_STAGE_CHUNK) {
        if (!ctx->stage_chunks.empty()) { ++ctx->stage_cur; ctx->stage_off = 0; }
        if (ctx->stage_cur >= ctx->stage_chunks.size()) {
            char *c = nullptr;
            if ((*err = hipHostMalloc((void **)&c, ZK_STAGE_CHUNK, hipHostMallocDefault)) != hipSuccess) return nullptr;
            ctx->stage_chunks.push_back(c);
        }
    }
    char *h = ctx->stage_chunks[ctx->stage_cur] + ctx->stage_off;
    ctx->stage_off += need;
    ctx->stage_since_rewind += need;
    return h;
}
// device destinations come from the ctx arena (512-byte granules): rounding the copy up to whole words stays inside them
// ZK_STAGE_KERNEL=0 (A-B runs only): hipMemcpyAsync from / to the pinned chunk instead of the copy kernel
static const bool kStageKernel = !(getenv("ZK_STAGE_KERNEL") && getenv("ZK_STAGE_KERNEL")[0] == 0x30);
static hipError_t stage_upload(zk_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    HostProf hp("stage_upload");
    if (!bytes) return hipSuccess;
    if (bytes > ZK_STAGE_CHUNK / 2) return hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream);
    hipError_t e;
    char *h = stage_alloc(ctx, bytes, &e);
    if (!h) return e;
    memcpy(h, h_src, bytes);
    if (!kStageKernel) return hipMemcpyAsync(d_dst, h, bytes, hipMemcpyHostToDevice, ctx->stream);
    const size_t words = (bytes + 7) / 8;
    zk_copy_words_kernel<<<(unsigned)((words + 255) / 256), 256, 0, ctx->stream>>>((u64 *)d_dst, (const u64 *)h, words);
    return hipGetLastError();
}
// `h_pinned_dst` must be pinned (hipHostMalloc) memory with room for the byte count rounded up to whole words
static hipError_t copy_to_pinned(zk_ctx *ctx, void *h_pinned_dst, const void *d_src, size_t bytes) {
    if (!bytes) return hipSuccess;
    if (!kStageKernel) return hipMemcpyAsync(h_pinned_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream);
    const size_t words = (bytes + 7) / 8;
    zk_copy_words_kernel<<<(unsigned)((words + 255) / 256), 256, 0, ctx->stream>>>((u64 *)h_pinned_dst, (const u64 *)d_src, words);
    return hipGetLastError();
}
// device -> pageable host, small: lands in `h_dst` when stage_collect runs (after the caller's stream synchronisation).
// `d_src` must be 8-byte aligned with whole words readable (arena blocks are).
static hipError_t stage_download(zk_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    if (!bytes) return hipSuccess;
    if (bytes > ZK_STAGE_CHUNK / 2) {
        // Large and to pageable memory: the runtime would stage it through its own bounce buffers from the calling thread
        // (0.5 ms before the copy even starts, kernel trace r03f).  One pinned landing buffer per ctx, one such download
        // in flight at a time (the caller synchronises and collects before the next).
        for (auto &r : ctx->stage_pending)
            if (r.h_pinned == ctx->h_big) return hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (ctx->h_big_bytes < bytes) {
            if (ctx->h_big) (void)hipHostFree(ctx->h_big);
            ctx->h_big = nullptr; ctx->h_big_bytes = 0;
            const size_t want = (bytes + (bytes >> 2) + 4095) & ~(size_t)4095;
            hipError_t e = hipHostMalloc((void **)&ctx->h_big, want, hipHostMallocDefault);
            if (e != hipSuccess) { (void)hipGetLastError(); return hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream); }
            ctx->h_big_bytes = want;
        }
        ctx->stage_pending.push_back({ctx->h_big, h_dst, bytes});
        return hipMemcpyAsync(ctx->h_big, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream);
    }
    hipError_t e;
    char *h = stage_alloc(ctx, bytes, &e);
    if (!h) return e;
    ctx->stage_pending.push_back({h, h_dst, bytes});
    return copy_to_pinned(ctx, h, d_src, bytes);
}
// Wait for everything enqueued on `st` so far -- the Fiat-Shamir read-backs (a cap, an opening set, a PoW answer: ~100 per
// realistic-height segment).  hipStreamSynchronize costs the calling thread 20-40 us on this runtime beyond the GPU's own
// time; instead a one-lane kernel behind the copies writes the next sequence number into a coherent pinned word and the
// host spins on it (stream order: it runs after the copy kernels, whose stores went to the same kind of memory).  Falls back
// to hipStreamSynchronize when the word cannot be allocated, when ZK_SYNC_POLL=0, or after ~2 s without progress (then the
// synchronisation reports whatever went wrong).
static __global__ void zk_signal_kernel(volatile uint64_t *seq, uint64_t value) {
    __threadfence_system();
    *seq = value;
}
// Measured (r04e, realistic heights, A/B inside one gpurun call): 111.5 / 111.4 ms polled against 111.3 / 110.7 ms with
// hipStreamSynchronize -- no gain: the runtime's wait is not what the ~60 us around a read-back are made of.  Off by default
// (ZK_SYNC_POLL=1 turns it on).
static const bool kSyncPoll = getenv("ZK_SYNC_POLL") && getenv("ZK_SYNC_POLL")[0] == 0x31;
static hipError_t zk_stream_wait(zk_ctx *ctx, hipStream_t st) {
    if (!kSyncPoll) return hipStreamSynchronize(st);
    if (!ctx->h_seq) {
        void *p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return hipStreamSynchronize(st); }
        ctx->h_seq = (volatile uint64_t *)p;
        *ctx->h_seq = 0;
    }
    const uint64_t want = ++ctx->seq_next;
    zk_signal_kernel<<<1, 1, 0, st>>>(ctx->h_seq, want);
    if (hipGetLastError() != hipSuccess) return hipStreamSynchronize(st);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0; *ctx->h_seq < want; ++spins) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        if ((spins & 0xFFFF) == 0xFFFF && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2))
            return hipStreamSynchronize(st);              // a fault or a very long kernel: let the runtime say which
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return hipSuccess;
}
static inline void stage_collect(zk_ctx *ctx) {
    for (auto &r : ctx->stage_pending) memcpy(r.h_dst, r.h_pinned, r.bytes);
    ctx->stage_pending.clear();
}

// Constants of a table definition -- Column / Filter programs, compiled entry shapes, AIR constants -- are the same words in
// every proof.  The first use uploads them into device memory the ctx keeps (hipMalloc, synchronously: the copy is complete
// and visible to both lanes when the call returns); every later use is a hash lookup plus a word-for-word comparison, and NO
// transfer.  Why it matters: a kernel (or blit) that touches host memory costs the GPU ~60 us of pipeline time around it
// (kernel trace r03g: 269 such gaps per two realistic-height proofs), 150 small uploads per proof.  Returns nullptr when the
// cache is full (64 MB) -- the caller then uploads into scratch as before.
static inline u64 words_hash(const u64 *p, size_t words) {
    u64 h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < words; ++i) { h ^= p[i]; h *= 0x100000001b3ULL; h ^= h >> 29; }
    return h;
}
static const u64 *const_upload(zk_ctx *ctx, const u64 *words, size_t n_words) {
    if (!n_words) return nullptr;
    const std::pair<u64, u64> key{words_hash(words, n_words), (u64)n_words};
    auto it = ctx->dev_consts.find(key);
    if (it != ctx->dev_consts.end()) {
        if (it->second.words.size() == n_words && memcmp(it->second.words.data(), words, n_words * 8) == 0) return it->second.d;
        return nullptr;                                    // (a hash collision: do not cache the newcomer)
    }
    if (ctx->dev_const_bytes + n_words * 8 > ((size_t)64 << 20)) return nullptr;
    u64 *d = nullptr;
    if (hipMalloc((void **)&d, n_words * 8) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemcpyAsync(d, words, n_words * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d); return nullptr; }
    zk_ctx::DevConst &e = ctx->dev_consts[key];
    e.words.assign(words, words + n_words);
    e.d = d;
    ctx->dev_const_bytes += n_words * 8;
    return d;
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

