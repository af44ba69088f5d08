// Memory table witness generation on the device: the data-parallel restatement of
// MemoryStark::generate_trace (evm_arithmetization/src/memory/memory_stark.rs:104-455).
//
//   reference (CPU, sequential)                         here (one kernel each)
//   sort_by_key (ctx, seg, virt, ts)   :208,215,221     two stable 64-bit radix-sort passes (rocPRIM) on packed keys
//   fill_gaps: while-loops pushing dummy reads :296-355 closed-form dummy count per adjacent pair + exclusive scan;
//                                                       every output row finds its pair by binary search -- the
//                                                       re-sorts of :215/:221 are not needed because the dummies
//                                                       of a pair sort strictly between its two operations
//   pad_memory_ops :357-383                             rows >= unpadded length are the padding operation
//   into_row, first-change flags, range_check :104-199  mem_rows_kernel (row i reads records i and i+1)
//   counter / frequencies / stale contexts :236-281     same kernel, u64 atomics on the frequency columns
//   mem_after extraction :437-446                       exclusive scan of mem_after_filter + scatter
#pragma once
#include "gl.cuh"

struct MemOpRec {          // 64 bytes
    u64 ts;
    u32 ctx, seg, virt, flags;   // flags: bit 0 is_read, bit 1 filter
    u32 val[8];
    u32 pad[2];
};

// memory/columns.rs:13-94
enum : u32 {
    MC_FILTER = 0, MC_TIMESTAMP = 1, MC_TIMESTAMP_INV = 2, MC_IS_READ = 3, MC_CTX = 4, MC_SEG = 5, MC_VIRT = 6, MC_VALUE = 7,
    MC_CTX_FIRST = 15, MC_SEG_FIRST = 16, MC_VIRT_FIRST = 17, MC_INIT_AUX = 18, MC_PREINIT = 19, MC_PREINIT_AUX = 20,
    MC_STALE_CONTEXTS = 21, MC_IS_PRUNED = 22, MC_STALE_FREQ = 23, MC_IS_STALE = 24, MC_MAYBE_AFTER = 25,
    MC_AFTER_FILTER = 26, MC_RANGE_CHECK = 27, MC_COUNTER = 28, MC_FREQUENCIES = 29, MC_NUM = 30
};

// ops: n_ops x 9 words {flags, timestamp, context, segment, virt, value as four 64-bit limbs};
// before: n_before x 7 words {context, segment, virt, value limbs}: a filtered write at timestamp 0 (:405-414)
static __global__ void mem_pack_kernel(const u64 *__restrict__ ops, u32 n_ops, const u64 *__restrict__ before, u32 n_before,
                                MemOpRec *__restrict__ recs, u64 *__restrict__ key_lo, u64 *__restrict__ key_hi,
                                u32 *__restrict__ idx) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ops + n_before) return;
    MemOpRec r;
    const u64 *v;
    if (i < n_ops) {
        const u64 *o = ops + (size_t)i * 9;
        r.flags = (u32)o[0] & 3u;
        r.ts = o[1];
        r.ctx = (u32)o[2]; r.seg = (u32)o[3]; r.virt = (u32)o[4];
        v = o + 5;
    } else {
        const u64 *o = before + (size_t)(i - n_ops) * 7;
        r.flags = 2u;
        r.ts = 0;
        r.ctx = (u32)o[0]; r.seg = (u32)o[1]; r.virt = (u32)o[2];
        v = o + 3;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { r.val[2 * k] = (u32)v[k]; r.val[2 * k + 1] = (u32)(v[k] >> 32); }
    r.pad[0] = r.pad[1] = 0;
    recs[i] = r;
    key_lo[i] = ((u64)r.virt << 32) | (u32)r.ts;
    key_hi[i] = ((u64)r.ctx << 32) | r.seg;
    idx[i] = i;
}

static __global__ void mem_gather_keys_kernel(const u64 *__restrict__ key_hi, const u32 *__restrict__ idx, u32 m, u64 *__restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[i] = key_hi[idx[i]];
}
static __global__ void mem_gather_recs_kernel(const MemOpRec *__restrict__ recs, const u32 *__restrict__ perm, u32 m, MemOpRec *__restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[i] = recs[perm[i]];
}

// the operation list fill_gaps iterates over: an optional dummy read of (0,0,0) at timestamp 1 in front (:299-316)
__device__ __forceinline__ MemOpRec mem_seq_at(const MemOpRec *sorted, u32 front, u32 j) {
    if (front && j == 0) {
        MemOpRec d;
        d.ts = 1; d.ctx = d.seg = d.virt = 0; d.flags = 1u;
#pragma unroll
        for (int k = 0; k < 8; ++k) d.val[k] = 0;
        d.pad[0] = d.pad[1] = 0;
        return d;
    }
    return sorted[j - front];
}

// number of dummy reads the reference's while-loops push for the pair (curr, next) (:318-354)
__device__ __forceinline__ u64 mem_gap_dummies(const MemOpRec &c, const MemOpRec &x, u64 max_rc) {
    if (c.ctx != x.ctx || c.seg != x.seg)
        return x.virt > max_rc ? ((u64)x.virt - max_rc + max_rc - 1) / max_rc : 0;
    if (c.virt != x.virt) {
        const u64 gap = (u64)x.virt - c.virt - 1;
        return gap > max_rc ? (gap - max_rc + max_rc) / (max_rc + 1) : 0;
    }
    const u64 dt = x.ts - c.ts;
    return dt > max_rc ? (dt - max_rc + max_rc - 1) / max_rc : 0;
}

// pos[j] = j + (dummies before element j); pos has big+1 entries, pos[big] = unpadded length
static __global__ void mem_gap_count_kernel(const MemOpRec *__restrict__ sorted, u32 front, u32 big, u64 max_rc, u64 *__restrict__ cnt,
                                     int *__restrict__ err) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > big) return;
    u64 c = 0;
    if (j + 1 < big) {
        const MemOpRec a = mem_seq_at(sorted, front, j), b = mem_seq_at(sorted, front, j + 1);
        c = mem_gap_dummies(a, b, max_rc);
        if (c > (1u << 28)) { atomicExch(err, 1); c = 0; }
    }
    cnt[j] = j < big ? c + 1 : 0;      // the element itself + its dummies; scanned exclusively -> pos[]
}

// rows[r] for r in [0, n): operation, gap dummy or padding, in final sorted order
static __global__ void mem_expand_kernel(const MemOpRec *__restrict__ sorted, u32 front, u32 big, u64 max_rc, const u64 *__restrict__ pos,
                                  u32 unpadded, u32 n, MemOpRec *__restrict__ rows) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    MemOpRec o;
    if (r >= unpadded) {                      // pad_memory_ops (:357-383): the last operation, one address further
        const MemOpRec last = mem_seq_at(sorted, front, big - 1);
        o = last;
        o.virt = last.virt + 1; o.ts = last.ts + 1; o.flags = 1u;
#pragma unroll
        for (int k = 0; k < 8; ++k) o.val[k] = 0;
        rows[r] = o;
        return;
    }
    u32 lo = 0, hi = big;                     // largest j with pos[j] <= r
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (pos[mid] <= r) lo = mid; else hi = mid;
    }
    const u32 j = lo, i = (u32)(r - pos[j]);
    const MemOpRec c = mem_seq_at(sorted, front, j);
    if (i == 0) { rows[r] = c; return; }
    const MemOpRec x = mem_seq_at(sorted, front, j + 1);
    const u32 cnt = (u32)(pos[j + 1] - pos[j] - 1);
    o.flags = 1u; o.pad[0] = o.pad[1] = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.val[k] = 0;
    if (c.ctx != x.ctx || c.seg != x.seg) {   // pushed in descending virt order; ascending after the re-sort
        o.ctx = x.ctx; o.seg = x.seg;
        o.virt = (u32)((u64)x.virt - (u64)(cnt - i + 1) * max_rc);
        o.ts = c.ts + 1;
    } else if (c.virt != x.virt) {
        o.ctx = c.ctx; o.seg = c.seg;
        o.virt = (u32)((u64)c.virt + (u64)i * (max_rc + 1));
        o.ts = c.ts + i;
    } else {
        o.ctx = c.ctx; o.seg = c.seg; o.virt = c.virt;
        o.ts = c.ts + (u64)i * max_rc;
#pragma unroll
        for (int k = 0; k < 8; ++k) o.val[k] = c.val[k];
    }
    rows[r] = o;
}

// insert_stale_contexts (:385-403): row index = the context number
static __global__ void mem_stale_kernel(const u64 *__restrict__ stale, u32 n_stale, u64 *__restrict__ out, size_t cs) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_stale) return;
    const u64 ctx = stale[i];
    out[MC_STALE_CONTEXTS * cs + ctx] = ctx + 1;
    out[MC_IS_PRUNED * cs + ctx] = 1;
}

// MemoryOp::into_row, generate_first_change_flags_and_rc, generate_trace_col_major (:104-199, :236-281).
// The frequency columns, stale_contexts and is_pruned must be zeroed / scattered before this kernel.
static __global__ void __launch_bounds__(256)
mem_rows_kernel(const MemOpRec *__restrict__ rows, u32 n, u64 *__restrict__ out, size_t cs, u32 *__restrict__ after_flag,
                int *__restrict__ err) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const MemOpRec c = rows[i], x = rows[i + 1 == n ? 0 : i + 1];
    const bool last = i + 1 == n;
    out[MC_FILTER * cs + i] = (c.flags >> 1) & 1u;
    out[MC_TIMESTAMP * cs + i] = c.ts;
    out[MC_TIMESTAMP_INV * cs + i] = c.ts ? gl_canon(gl_inv(c.ts)) : 0;
    out[MC_IS_READ * cs + i] = c.flags & 1u;
    out[MC_CTX * cs + i] = c.ctx;
    out[MC_SEG * cs + i] = c.seg;
    out[MC_VIRT * cs + i] = c.virt;
#pragma unroll
    for (int k = 0; k < 8; ++k) out[(MC_VALUE + k) * cs + i] = c.val[k];
    const bool cf = c.ctx != x.ctx;
    const bool sf = c.seg != x.seg && !cf;
    const bool vf = c.virt != x.virt && !sf && !cf;
    out[MC_CTX_FIRST * cs + i] = cf;
    out[MC_SEG_FIRST * cs + i] = sf;
    out[MC_VIRT_FIRST * cs + i] = vf;
    u64 rc;                                      // field subtraction: the wrap-around row compares with row 0
    if (last) rc = 0;
    else if (cf) rc = gl_canon(gl_sub(gl_sub(x.ctx, c.ctx), 1));
    else if (sf) rc = gl_canon(gl_sub(gl_sub(x.seg, c.seg), 1));
    else if (vf) rc = gl_canon(gl_sub(gl_sub(x.virt, c.virt), 1));
    else rc = gl_canon(gl_sub(x.ts, c.ts));
    if (rc >= n) { atomicExch(err, 2); rc = 0; }  // "Range check of {} is too large. Bug in fill_gaps?"
    out[MC_RANGE_CHECK * cs + i] = rc;
    const u64 ns = x.seg;
    const u64 aux = gl_mul(gl_sub(ns, 34), gl_sub(ns, 35));             // AccountsLinkedList, StorageLinkedList
    const u64 pre = gl_mul(gl_mul(ns, gl_sub(ns, 12)), aux);            // Code (0), TrieData (12)
    out[MC_PREINIT_AUX * cs + i] = gl_canon(aux);
    out[MC_PREINIT * cs + i] = gl_canon(pre);
    out[MC_INIT_AUX * cs + i] = ((cf || sf || vf) && (x.flags & 1u)) ? gl_canon(pre) : 0;
    out[MC_COUNTER * cs + i] = i;
    unsigned long long *freq = (unsigned long long *)(out + MC_FREQUENCIES * cs);
    atomicAdd(freq + rc, 1ull);
    if (cf || sf) atomicAdd(freq + (last ? 0u : x.virt), 1ull);
    u64 is_stale = 0, maybe = 0, after = 0;
    if (c.ctx >= n) { atomicExch(err, 3); return; }   // the reference indexes the stale_contexts column by context
    if (out[MC_STALE_CONTEXTS * cs + c.ctx] == (u64)c.ctx + 1) {
        is_stale = 1;
        atomicAdd((unsigned long long *)(out + MC_STALE_FREQ * cs) + c.ctx, 1ull);
    } else if (((c.flags >> 1) & 1u) && (cf || sf || vf)) {
        maybe = 1;
        bool nz = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) nz |= c.val[k] != 0;
        if (nz || c.seg == 0 || c.seg == 12 || c.seg == 34 || c.seg == 35) after = 1;   // PREINITIALIZED_SEGMENTS_INDICES
    }
    out[MC_IS_STALE * cs + i] = is_stale;
    out[MC_MAYBE_AFTER * cs + i] = maybe;
    out[MC_AFTER_FILTER * cs + i] = after;
    after_flag[i] = (u32)after;
}

// final memory, row order: entries k x 7 words {context, segment, virt, value as four 64-bit limbs}
static __global__ void mem_after_scatter_kernel(const MemOpRec *__restrict__ rows, const u32 *__restrict__ flag, const u32 *__restrict__ off,
                                         u32 n, u64 *__restrict__ entries) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const MemOpRec c = rows[i];
    u64 *e = entries + (size_t)off[i] * 7;
    e[0] = c.ctx; e[1] = c.seg; e[2] = c.virt;
#pragma unroll
    for (int k = 0; k < 4; ++k) e[3 + k] = (u64)c.val[2 * k] | ((u64)c.val[2 * k + 1] << 32);
}
