// zk_ctx and the error / scratch helpers shared by every host-side include of libzkstark_hip.so (and by the kernel
// micro-benchmark tools/kbench.hip, which links the same NTT / Merkle host code without the rest of the library).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/zkstark.h"
#include "arena.hpp"
#include "gl.cuh"

// ------------------------------------------------------------------------------------------
struct zk_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    volatile const int *abort_flag = nullptr;
    std::map<int, u64 *> tw_fwd, tw_inv, tw_inv_br;         // log size -> table (tw_inv_br: block-order levels)
    std::map<std::pair<int, u64>, u64 *> coset_tabs;        // (log_n, shift) -> s^bitrev(i)
    std::map<std::pair<int, u64>, u64 *> coset_inv_tabs;    // (log_n, shift) -> n^-1 s^-bitrev(i)
    hipEvent_t ev[5] = {};
    float timings[4] = {0, 0, 0, 0};
    // running totals over all commits since the last reset (zk_ctx_commit_totals)
    double total_ms[4] = {0, 0, 0, 0};
    double total_leaf_bytes = 0, total_leaf_perms = 0, total_ntt_bytes = 0;
    uint64_t total_commits = 0;
    int cu_count = 0;
    std::set<zk_batch *> live_batches;  // freed by zk_ctx_destroy if the caller leaked them
    DevArena arena;                     // all batch + scratch HBM (arena.hpp)
    // debug: starky `check_ctls` after get_ctl_data (zk_ctx_set_check_ctls); extra looking rows per CTL index
    bool check_ctls = false;
    std::map<size_t, std::pair<size_t, std::vector<u64>>> ctl_extra;   // ctl -> (row width, rows)
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

static int check_abort(zk_ctx *ctx) {
    if (ctx->abort_flag && *ctx->abort_flag) return set_err(ctx, ZK_ERR_ABORTED, "aborted");
    return ZK_OK;
}

static int check_launch(zk_ctx *ctx, const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return set_err(ctx, ZK_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return ZK_OK;
}

