// STARK auxiliary-column kernels: logUp helper columns (starky `lookup_helper_columns`) and
// cross-table-lookup partial sums (starky `cross_table_lookup::partial_sums`), i.e. SURVEY K6/K7.
// Reference call sites: evm_arithmetization/src/prover.rs:137 (`get_ctl_data`) and prover.rs:322
// (`prove_with_commitment` -> lookup helper columns).  The column / filter *definitions* are data
// supplied by the caller in the flat "program" encoding documented in include/zkstark.h (the
// reference builds them in all_stark.rs:153-417 and each table's `lookups()`).
//
// MI355X mapping: one lane per trace row, columns read column-major (coalesced); the per-row
// reciprocals 1/(combine(row)) of up to 16 looking entries share ONE field inversion (Montgomery
// batch inversion held in registers) -- the reference batch-inverts along the rows of one column
// instead, which on a GPU would serialise a lane over strided memory.  Running sums are a
// three-kernel block scan over field addition.
#pragma once
#include "gl.cuh"

// ---- program interpreter ---------------------------------------------------------------------
// Column  := n_local, n_next, constant, (idx, coef) * n_local, (idx, coef) * n_next
// Filter  := n_products, n_constants, (Column, Column) * n_products, Column * n_constants
// Entry   := n_columns, Column * n_columns, Filter
struct TraceView {
    const u64 *base;
    size_t stride;  // elements between columns
    u32 n;          // rows
};

// Evaluates the column at `row` ("table" semantics: the next-row part is dropped on the last row)
// and advances pc past it.
__device__ __forceinline__ u64 prog_eval_column(const u64 *__restrict__ prog, u32 &pc, const TraceView &t, u32 row) {
    const u32 nl = (u32)prog[pc], nn = (u32)prog[pc + 1];
    u64 acc = prog[pc + 2];
    pc += 3;
    for (u32 i = 0; i < nl; ++i, pc += 2) {
        u64 v = t.base[(size_t)prog[pc] * t.stride + row];
        u64 c = prog[pc + 1];
        acc = gl_add(acc, c == 1 ? v : gl_mul(v, c));
    }
    const bool has_next = row + 1 < t.n;
    for (u32 i = 0; i < nn; ++i, pc += 2) {
        if (has_next) {
            u64 v = t.base[(size_t)prog[pc] * t.stride + row + 1];
            u64 c = prog[pc + 1];
            acc = gl_add(acc, c == 1 ? v : gl_mul(v, c));
        }
    }
    return acc;
}
__device__ __forceinline__ u64 prog_eval_filter(const u64 *__restrict__ prog, u32 &pc, const TraceView &t, u32 row) {
    const u32 np = (u32)prog[pc], nc = (u32)prog[pc + 1];
    pc += 2;
    u64 acc = 0;
    for (u32 i = 0; i < np; ++i) {
        u64 a = prog_eval_column(prog, pc, t, row);
        u64 b = prog_eval_column(prog, pc, t, row);
        acc = gl_add(acc, gl_mul(a, b));
    }
    for (u32 i = 0; i < nc; ++i) acc = gl_add(acc, prog_eval_column(prog, pc, t, row));
    return gl_canon(acc);
}

// ---- helper columns ------------------------------------------------------------------------------
// prog := n_entries, offset[n_entries], Entry...   helper h sums entries [h*chunk, (h+1)*chunk).
// helpers[h][row] = sum over its entries of filter ? 1/(sum_j beta^j col_j + gamma) : 0.
// extra_inv (optional): also writes 1/(gamma + table_col(row)) for the logUp table column, whose
// Column program sits at prog[extra_pc].
#define ZK_HELPER_BATCH 16
// denominator and filter of entry e at `row`
__device__ __forceinline__ void prog_eval_entry(const u64 *__restrict__ prog, u32 e, const TraceView &t, u32 row,
                                                u64 beta, u64 gamma, u64 &denom, u64 &filt) {
    u32 pc = (u32)prog[1 + e];
    const u32 ncols = (u32)prog[pc++];
    u64 acc = 0, bp = 1;  // sum_j beta^j col_j  (== reduce_with_powers(evals, beta))
    for (u32 j = 0; j < ncols; ++j) {
        u64 c = prog_eval_column(prog, pc, t, row);
        acc = gl_add(acc, j == 0 ? c : gl_mul(c, bp));
        bp = j == 0 ? beta : gl_mul(bp, beta);
    }
    denom = gl_canon(gl_add(acc, gamma));
    filt = prog_eval_filter(prog, pc, t, row);
}

__global__ void __launch_bounds__(256)
helper_cols_kernel(const u64 *__restrict__ prog, TraceView t, u64 beta, u64 gamma, u32 chunk,
                   u64 *__restrict__ helpers, size_t helper_stride, u32 extra_pc,
                   u64 *__restrict__ extra_inv, int *__restrict__ err_flag) {
    const u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= t.n) return;
    const u32 n_entries = (u32)prog[0];
    // ZK_HELPER_BATCH is a multiple of chunk (1 or 2), so batches never split a helper group
    for (u32 e0 = 0; e0 < n_entries; e0 += ZK_HELPER_BATCH) {
        u64 v[ZK_HELPER_BATCH], pre[ZK_HELPER_BATCH], flt[ZK_HELPER_BATCH];
        u64 run = 1;
#pragma unroll
        for (int i = 0; i < ZK_HELPER_BATCH; ++i) {
            v[i] = 1; flt[i] = 0; pre[i] = 1;
            if (e0 + i < n_entries) {
                u64 d, f;
                prog_eval_entry(prog, e0 + i, t, row, beta, gamma, d, f);
                if (f > 1) atomicExch(err_flag, 1);          // "Non-binary filter?" (plonky2 asserts)
                if (f == 1 && d == 0) { atomicExch(err_flag, 2); d = 1; }  // 1/0: plonky2 would panic
                flt[i] = f;
                v[i] = f == 1 ? d : 1;                       // dummy 1 where filtered out
                pre[i] = run;
                run = gl_mul(run, v[i]);
            }
        }
        u64 inv = gl_inv(run);                               // ONE inversion for the whole batch
#pragma unroll
        for (int i = ZK_HELPER_BATCH - 1; i >= 0; --i) {
            if (e0 + i < n_entries) {
                u64 vi = v[i];
                v[i] = flt[i] == 1 ? gl_mul(inv, pre[i]) : 0;
                inv = gl_mul(inv, vi);
            }
        }
        if (chunk == 1) {
#pragma unroll
            for (int i = 0; i < ZK_HELPER_BATCH; ++i)
                if (e0 + i < n_entries) helpers[(size_t)(e0 + i) * helper_stride + row] = gl_canon(v[i]);
        } else {
#pragma unroll
            for (int i = 0; i < ZK_HELPER_BATCH; i += 2)
                if (e0 + i < n_entries) {
                    u64 s = e0 + i + 1 < n_entries ? gl_add(v[i], v[i + 1]) : v[i];
                    helpers[(size_t)((e0 + i) >> 1) * helper_stride + row] = gl_canon(s);
                }
        }
    }
    if (extra_inv) {  // logUp table column: 1 / (gamma + table(row))
        u32 pc = extra_pc;
        u64 d = gl_canon(gl_add(prog_eval_column(prog, pc, t, row), gamma));
        if (d == 0) { atomicExch(err_flag, 2); d = 1; }
        extra_inv[row] = gl_canon(gl_inv(d));
    }
}

// x[row] = sum_h helpers[h][row]  ( - freq(row) * table_inv[row]  when freq_pc != 0 )
__global__ void helper_row_sums_kernel(const u64 *__restrict__ helpers, size_t helper_stride, u32 n_helpers,
                                       const u64 *__restrict__ prog, u32 freq_pc, TraceView t,
                                       const u64 *__restrict__ table_inv, u64 *__restrict__ x) {
    const u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= t.n) return;
    u64 s = 0;
    for (u32 h = 0; h < n_helpers; ++h) s = gl_add(s, helpers[(size_t)h * helper_stride + row]);
    if (table_inv) {
        u32 pc = freq_pc;
        u64 f = prog_eval_column(prog, pc, t, row);
        s = gl_sub(s, gl_mul(f, table_inv[row]));
    }
    x[row] = gl_canon(s);
}

// ---- field prefix sums ---------------------------------------------------------------------------
// mode 0: out[i] = sum_{j<i} x[j]  (exclusive prefix: logUp Z, Z(first) = 0)
// mode 1: out[i] = sum_{j>=i} x[j] (inclusive suffix: CTL Z, total at row 0)
#define ZK_SCAN_ITEMS 8
#define ZK_SCAN_THREADS 256
__device__ __forceinline__ u32 scan_src_index(u32 i, u32 n, int mode) { return mode == 0 ? i : n - 1 - i; }

__global__ void __launch_bounds__(ZK_SCAN_THREADS)
scan_block_kernel(const u64 *__restrict__ x, u32 n, int mode, u64 *__restrict__ incl, u64 *__restrict__ totals) {
    __shared__ u64 sh[ZK_SCAN_THREADS];
    const u32 base = (blockIdx.x * ZK_SCAN_THREADS + threadIdx.x) * ZK_SCAN_ITEMS;
    u64 loc[ZK_SCAN_ITEMS];
    u64 run = 0;
#pragma unroll
    for (int k = 0; k < ZK_SCAN_ITEMS; ++k) {
        u32 i = base + k;
        u64 v = i < n ? x[scan_src_index(i, n, mode)] : 0;
        run = gl_add(run, v);
        loc[k] = run;
    }
    sh[threadIdx.x] = run;
    __syncthreads();
    for (u32 off = 1; off < ZK_SCAN_THREADS; off <<= 1) {
        u64 add = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] = gl_add(sh[threadIdx.x], add);
        __syncthreads();
    }
    u64 excl = threadIdx.x ? sh[threadIdx.x - 1] : 0;
#pragma unroll
    for (int k = 0; k < ZK_SCAN_ITEMS; ++k) {
        u32 i = base + k;
        if (i < n) incl[i] = gl_add(loc[k], excl);
    }
    if (threadIdx.x == ZK_SCAN_THREADS - 1) totals[blockIdx.x] = sh[threadIdx.x];
}
// single block: exclusive scan of the block totals (n_blocks <= 2^31 / 2048, handled in a loop)
__global__ void scan_totals_kernel(u64 *totals, u32 n_blocks) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        u64 run = 0;
        for (u32 i = 0; i < n_blocks; ++i) { u64 v = totals[i]; totals[i] = run; run = gl_add(run, v); }
    }
}
__global__ void scan_finish_kernel(const u64 *__restrict__ incl, const u64 *__restrict__ totals, u32 n,
                                   int mode, u64 *__restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 v = gl_canon(gl_add(incl[i], totals[i / (ZK_SCAN_THREADS * ZK_SCAN_ITEMS)]));
    if (mode == 0) {
        if (i + 1 < n) out[i + 1] = v;
        if (i == 0) out[0] = 0;
    } else {
        out[n - 1 - i] = v;
    }
}
