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
#include "fri.cuh"   // DotAcc (delayed-reduction dot products)

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
//
// All `num_challenges` (beta, gamma) pairs are handled in ONE pass: the column values and the filter of an entry do
// not depend on the challenge, only the combination sum_j beta_k^j col_j + gamma_k does.  Linear combinations
// (`Column`) and the tuple combination are dot products with wave-uniform coefficients, accumulated unreduced
// (DotAcc: 8 VALU instructions per term instead of a field multiply + add); beta_k^j comes precomputed.
// Entries per Montgomery batch inversion.  helper_cols_kernel: 4 -- with 16 the (denominator, prefix product) arrays of the two
// challenges do not fit the registers next to the entry evaluator and live in scratch (528 bytes per lane; r03u: the kernel at
// 10.5 cycles per instruction, i.e. waiting, not issuing); with 8 or 4 the batch loop unrolls and they are registers (no scratch;
// 144 / 94 VGPRs).  The kernel has issue slots to spare, so four times the inversions (74 multiplies each since r03t) still come
// out ahead -- A/B in one call (profiles/archive/r03u_ab_helper_batch.log): CTL data 15.2 / 13.7 / 11.8 ms for 16 / 8 / 4 at 2^20
// (578 -> 575 ms per segment), 112.6 -> 110.2 ms per realistic-height segment; 2 is worse again (13.2 ms,
// profiles/archive/r03w_ab_batch_sizes.log).  lookup_singles_kernel is the opposite case -- no interpreter, issue-bound (4.1 cycles
// per instruction), its arrays are registers at any size -- so there FEWER inversions pay: 32 per batch (225 VGPRs, two waves
// per SIMD) against 16: the Arithmetic table's proof 40.0 -> 38.1 ms, the segment -1.2 ms; 8 is slower.
#ifndef ZK_HELPER_BATCH
#define ZK_HELPER_BATCH 4
#endif
#ifndef ZK_SINGLES_BATCH
#define ZK_SINGLES_BATCH 32
#endif
#ifndef ZK_HELPER_WAVES
#define ZK_HELPER_WAVES 1
#endif
#ifndef ZK_HELPER_UNROLL
#define ZK_HELPER_UNROLL 1
#endif
#define ZK_HELPER_MAX_CHALLENGES 2
#define ZK_HELPER_MAX_TUPLE 64
struct HelperChallenges {
    u64 gamma[ZK_HELPER_MAX_CHALLENGES];
    u64 bpow[ZK_HELPER_MAX_CHALLENGES][ZK_HELPER_MAX_TUPLE];   // beta_k^j
};

__device__ __forceinline__ void dot_acc_mac_u(DotAcc &d, u64 coef_uniform, u64 v) {
    dot_acc_mac(d, __builtin_amdgcn_readfirstlane((u32)coef_uniform), __builtin_amdgcn_readfirstlane((u32)(coef_uniform >> 32)), v);
}
// Column at `row` ("table" semantics: the next-row part is dropped on the last row); lazy u64.
__device__ __forceinline__ u64 prog_eval_column_dot(const u64 *__restrict__ prog, u32 &pc, const TraceView &t, u32 row) {
    const u32 nl = (u32)prog[pc], nn = (u32)prog[pc + 1];
    const u64 k = prog[pc + 2];
    pc += 3;
    const bool has_next = row + 1 < t.n;
    if (nl == 1 && nn == 0 && k == 0 && prog[pc + 1] == 1) {       // Column::single, by far the most common
        u64 v = t.base[(size_t)prog[pc] * t.stride + row];
        pc += 2;
        return v;
    }
    DotAcc a;
    dot_acc_init(a);
    for (u32 i = 0; i < nl; ++i, pc += 2) dot_acc_mac_u(a, prog[pc + 1], t.base[(size_t)prog[pc] * t.stride + row]);
    for (u32 i = 0; i < nn; ++i, pc += 2)
        if (has_next) dot_acc_mac_u(a, prog[pc + 1], t.base[(size_t)prog[pc] * t.stride + row + 1]);
    return gl_add(dot_acc_reduce(a), k);
}
__device__ __forceinline__ u64 prog_eval_filter_dot(const u64 *__restrict__ prog, u32 &pc, const TraceView &t, u32 row) {
    const u32 np = (u32)prog[pc], nc = (u32)prog[pc + 1];
    pc += 2;
    u64 acc = 0;
    for (u32 i = 0; i < np; ++i) {
        u64 a = prog_eval_column_dot(prog, pc, t, row);
        u64 b = prog_eval_column_dot(prog, pc, t, row);
        acc = gl_add(acc, gl_mul(a, b));
    }
    for (u32 i = 0; i < nc; ++i) acc = gl_add(acc, prog_eval_column_dot(prog, pc, t, row));
    return gl_canon(acc);
}
// ---- compiled entries ---------------------------------------------------------------------------------
// Interpreting an Entry word by word is a chain of dependent scalar loads (count -> index -> value) and was latency
// bound.  The host therefore flattens every entry once per call (stark_host.inc `compile_entries`): because the tuple
// combination is linear, sum_j beta^j (sum_i c_ji v_ji + k_j) + gamma becomes ONE dot product over (column, row
// offset) terms with coefficients beta^j c_ji (one array per challenge) plus a constant; filters become dot products
// too.  All loop bounds and term descriptors are wave-uniform and independent of each other.
//   blob := n_entries, lin_off, term_off, entry[n] {t0 | (base + 1) << 32, t1, fp0, fp1, fc0, fc1, K_0, K_1},
//           lin[] {t0, t1, constant}, term[] {col | next << 32, coef_0, coef_1}
// `base` (optional) names a lin whose terms -- with the coefficients of every challenge -- are the part of the tuple
// combination this entry shares with its neighbours (stark_host.inc, "BASES"): its value is computed once per row and
// challenge and remembered (memo_base), the entry adds its own terms and constant.
// A lin may be a DELTA lin: {t0 | 2^63, t1 | ref << 32, constant} = lin `ref` + its own few terms + constant.  The
// filters of consecutive CTL entries are often sums that differ by one column (KeccakSponge's 136 memory reads:
// is_full_input_block + sum_{j > i} is_final_input_len[j] -- 9316 terms as plain sums, 136 + 135 as deltas); the entry
// compiler (stark_host.inc) emits deltas against the previous entry's filter, and the evaluation remembers the last lin
// it computed (entries are visited in order, so the reference is always that one).
#define ZK_LIN_DELTA (1ULL << 63)
#define ZK_CBLOB_TWIN (~0ULL)        // index entry of a z-data whose blob is slot 1 of the previous one's (quotient_host.inc)
struct CBlob {
    const u64 *w;
    mutable u32 memo_lin;                          // last lin evaluated by clin_eval for this thread / row, and its value
    mutable u64 memo_val;
    // last base evaluated and its value per challenge slot.  ONE CBlob per (thread, row): the memos hold row values.
    mutable u32 memo_base, memo_base_ok;           // base lin + 1; bit k of _ok: slot k is valid
    mutable u64 memo_base_val[2];
    __device__ __forceinline__ explicit CBlob(const u64 *p)
        : w(p), memo_lin(0xFFFFFFFFu), memo_val(0), memo_base(0), memo_base_ok(0) { memo_base_val[0] = memo_base_val[1] = 0; }
    __device__ __forceinline__ u32 n_entries() const { return (u32)w[0]; }
    __device__ __forceinline__ const u64 *entry(u32 e) const { return w + 3 + 8 * (size_t)e; }
    __device__ __forceinline__ const u64 *lin(u32 l) const { return w + w[1] + 3 * (size_t)l; }
    __device__ __forceinline__ const u64 *term(u32 t) const { return w + w[2] + 3 * (size_t)t; }
};
// LD(col, next) -> value, or skip the term when `next` is not available (last row of a table)
template <int NCH, class LD>
__device__ __forceinline__ void cterms_dot(const CBlob &B, u32 t0, u32 t1, LD ld, DotAcc (&d)[NCH]) {
    u32 t = t0;
    for (; t + 4 <= t1; t += 4) {                    // 4 independent descriptor + value loads in flight
        u64 v[4];
        bool ok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const u64 w0 = B.term(t + i)[0]; ok[i] = ld((u32)w0, (u32)(w0 >> 32), v[i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ok[i]) {
#pragma unroll
                for (int k = 0; k < NCH; ++k) dot_acc_mac_u(d[k], B.term(t + i)[1 + k], v[i]);
            }
    }
    for (; t < t1; ++t) {
        const u64 w0 = B.term(t)[0];
        u64 v;
        if (ld((u32)w0, (u32)(w0 >> 32), v)) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) dot_acc_mac_u(d[k], B.term(t)[1 + k], v);
        }
    }
}
template <class LD>
__device__ __forceinline__ u64 clin_eval(const CBlob &B, u32 l, LD ld) {
    u64 acc = 0;
    bool have = false;
    for (u32 cur = l;;) {                            // a delta lin walks to its reference (normally one memo hit away)
        if (cur == B.memo_lin) { acc = have ? gl_add(acc, B.memo_val) : B.memo_val; break; }
        const u64 *L = B.lin(cur);
        const u32 t0 = (u32)L[0], t1 = (u32)L[1];
        u64 part = L[2];                             // t0 == t1: a constant (e.g. the filter of an unfiltered lookup)
        if (t0 != t1) {
            DotAcc d[1];
            dot_acc_init(d[0]);
            cterms_dot<1>(B, t0, t1, ld, d);
            part = gl_add(dot_acc_reduce(d[0]), L[2]);
        }
        acc = have ? gl_add(acc, part) : part;
        have = true;
        if (!(L[0] & ZK_LIN_DELTA)) break;
        cur = (u32)(L[1] >> 32);
    }
    B.memo_lin = l;
    B.memo_val = acc;
    return acc;
}
// the filter of entry E: sum of products of lins + sum of lins (lazy)
template <class LD>
__device__ __forceinline__ u64 centry_filter(const CBlob &B, const u64 *E, LD ld) {
    u64 acc = 0;
    for (u32 p = (u32)E[2]; p < (u32)E[3]; ++p) acc = gl_add(acc, gl_mul(clin_eval(B, 2 * p, ld), clin_eval(B, 2 * p + 1, ld)));
    for (u32 c = (u32)E[4]; c < (u32)E[5]; ++c) acc = gl_add(acc, clin_eval(B, c, ld));
    return acc;
}
// value of base lin `base1 - 1` for all NCH challenge slots (memoised: consecutive entries share it)
template <int NCH, class LD>
__device__ __forceinline__ void cbase_eval(const CBlob &B, u32 base1, LD ld, u64 (&bv)[NCH]) {
    if (B.memo_base != base1 || (B.memo_base_ok & ((1u << NCH) - 1)) != ((1u << NCH) - 1)) {     // wave-uniform
        const u64 *L = B.lin(base1 - 1);
        DotAcc d[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) dot_acc_init(d[k]);
        cterms_dot<NCH>(B, (u32)L[0], (u32)L[1], ld, d);
#pragma unroll
        for (int k = 0; k < NCH; ++k) B.memo_base_val[k] = dot_acc_reduce(d[k]);
        B.memo_base = base1;
        B.memo_base_ok = (1u << NCH) - 1;
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) bv[k] = B.memo_base_val[k];
}
// CANON: canonical denominators / filter (the column generators invert and compare them); false = lazy representatives
template <int NCH, bool CANON = true, class LD>
__device__ __forceinline__ void centry_eval(const CBlob &B, u32 e, LD ld, u64 (&denom)[NCH], u64 &filt) {
    const u64 *E = B.entry(e);
    const u32 t0 = (u32)E[0], t1 = (u32)E[1], base1 = (u32)(E[0] >> 32);
    u64 k[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) k[c] = E[6 + c];
    if (base1) {                                     // (wave-uniform: the blob is the same for every row)
        u64 bv[NCH];
        cbase_eval<NCH>(B, base1, ld, bv);
#pragma unroll
        for (int c = 0; c < NCH; ++c) k[c] = gl_add(bv[c], k[c]);
    }
    bool simple = t1 - t0 == 1;                      // `Column::single`: one term with coefficient 1 for every challenge
#pragma unroll
    for (int c = 0; c < NCH; ++c) simple = simple && B.term(t0)[1 + c] == 1;
    if (simple) {
        const u64 w0 = B.term(t0)[0];
        u64 v = 0;
        if (!ld((u32)w0, (u32)(w0 >> 32), v)) v = 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) denom[c] = CANON ? gl_canon(gl_add(v, k[c])) : gl_add(v, k[c]);
    } else if (t1 == t0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) denom[c] = CANON ? gl_canon(k[c]) : k[c];
    } else {
        DotAcc d[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) dot_acc_init(d[c]);
        cterms_dot<NCH>(B, t0, t1, ld, d);
#pragma unroll
        for (int c = 0; c < NCH; ++c) denom[c] = CANON ? gl_canon(gl_add(dot_acc_reduce(d[c]), k[c])) : gl_add(dot_acc_reduce(d[c]), k[c]);
    }
    filt = CANON ? gl_canon(centry_filter(B, E, ld)) : centry_filter(B, E, ld);
}

// one challenge, chosen at run time (wave-uniform slot 0 / 1): the quotient kernel walks lookups challenge by challenge
template <class LD>
__device__ __forceinline__ void cterms_dot_slot(const CBlob &B, u32 t0, u32 t1, u32 slot, LD ld, DotAcc &d) {
    u32 t = t0;
    for (; t + 4 <= t1; t += 4) {
        u64 v[4];
        bool ok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const u64 w0 = B.term(t + i)[0]; ok[i] = ld((u32)w0, (u32)(w0 >> 32), v[i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ok[i]) dot_acc_mac_u(d, B.term(t + i)[1 + slot], v[i]);
    }
    for (; t < t1; ++t) {
        const u64 w0 = B.term(t)[0];
        u64 v;
        if (ld((u32)w0, (u32)(w0 >> 32), v)) dot_acc_mac_u(d, B.term(t)[1 + slot], v);
    }
}
template <class LD>
__device__ __forceinline__ void centry_eval_slot(const CBlob &B, u32 e, u32 slot, LD ld, u64 &denom, u64 &filt) {
    const u64 *E = B.entry(e);
    const u32 t0 = (u32)E[0], t1 = (u32)E[1], base1 = (u32)(E[0] >> 32);
    u64 k = E[6 + slot];
    if (base1) {
        if (B.memo_base != base1 || !((B.memo_base_ok >> slot) & 1)) {
            const u64 *L = B.lin(base1 - 1);
            DotAcc d;
            dot_acc_init(d);
            cterms_dot_slot(B, (u32)L[0], (u32)L[1], slot, ld, d);
            if (B.memo_base != base1) B.memo_base_ok = 0;
            B.memo_base = base1;
            B.memo_base_ok |= 1u << slot;
            const u64 v = dot_acc_reduce(d);
            if (slot) B.memo_base_val[1] = v; else B.memo_base_val[0] = v;          // (no run-time register indexing)
        }
        k = gl_add(slot ? B.memo_base_val[1] : B.memo_base_val[0], k);
    }
    if (t1 - t0 == 1 && B.term(t0)[1 + slot] == 1) {    // `Column::single` (wave-uniform test)
        const u64 w0 = B.term(t0)[0];
        u64 v = 0;
        if (!ld((u32)w0, (u32)(w0 >> 32), v)) v = 0;
        denom = gl_add(v, k);
    } else if (t1 == t0) {
        denom = k;
    } else {
        DotAcc d;
        dot_acc_init(d);
        cterms_dot_slot(B, t0, t1, slot, ld, d);
        denom = gl_add(dot_acc_reduce(d), k);
    }
    filt = centry_filter(B, E, ld);
}

// denominators (one per challenge) and filter of entry e at `row`
template <int NCH>
__device__ __forceinline__ void prog_eval_entry(const u64 *__restrict__ prog, u32 e, const TraceView &t, u32 row,
                                                const HelperChallenges &H, u64 (&denom)[NCH], u64 &filt) {
    u32 pc = (u32)prog[1 + e];
    const u32 ncols = (u32)prog[pc++];
    DotAcc acc[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) dot_acc_init(acc[k]);
    for (u32 j = 0; j < ncols; ++j) {            // sum_j beta^j col_j  (== reduce_with_powers(evals, beta))
        const u64 c = prog_eval_column_dot(prog, pc, t, row);
#pragma unroll
        for (int k = 0; k < NCH; ++k) dot_acc_mac_u(acc[k], H.bpow[k][j], c);
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) denom[k] = gl_canon(gl_add(dot_acc_reduce(acc[k]), H.gamma[k]));
    filt = prog_eval_filter_dot(prog, pc, t, row);
}

#ifndef ZK_DEVICE_FUNCS_ONLY   // the column-generator kernels
// ---- compiled blobs made ON THE DEVICE (stark_host.inc MultiShape) --------------------------------------------------------
// A table's compiled entries are challenge-independent except for the coefficient words beta^j c and the constants gamma +
// sum k_j beta^j.  The serialised shape lives in device memory (const_upload); per proof ONE small kernel writes the blob:
// a template word >= p is a tag (p + 2 * patch + slot) and is replaced by its polynomial evaluated at that slot's challenge,
// every other word is copied.  The (beta, gamma) pairs of the proof travel as kernel arguments -- no host memory is touched.
//   ser := blob_words, n_patches, patch_off, mono_off, tmpl[blob_words], patch[n] {mono_start | n_mono << 32, pair | add_gamma << 8},
//          mono[] {c, j}
#define ZK_MAX_CH_PAIRS 48
struct ChTable { u64 beta[ZK_MAX_CH_PAIRS][2], gamma[ZK_MAX_CH_PAIRS][2]; };
static __global__ void entry_blobs_instantiate_kernel(const u64 *__restrict__ ser, u64 *__restrict__ blob, ChTable T) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (u32)ser[0]) return;
    u64 w = ser[4 + i];
    if (w >= GL_P && w != ZK_CBLOB_TWIN) {
        const u64 idx = w - GL_P;
        const u32 slot = (u32)idx & 1;
        const u64 *pa = ser + ser[2] + 2 * (idx >> 1);
        const u32 ms = (u32)pa[0], nm = (u32)(pa[0] >> 32), pair = (u32)pa[1] & 0xFF;
        u64 v = ((pa[1] >> 8) & 1) ? T.gamma[pair][slot] : 0;
        const u64 b = T.beta[pair][slot];
        const u64 *mo = ser + ser[3] + 2 * (size_t)ms;
        for (u32 m = 0; m < nm; ++m) v = gl_add(v, gl_mul(mo[2 * m], gl_pow(b, mo[2 * m + 1])));
        w = gl_canon(v);
    }
    blob[i] = w;
}

struct HelperOut {
    u64 *helpers[ZK_HELPER_MAX_CHALLENGES];     // challenge k: helper column h at helpers[k] + h * helper_stride
    u64 *extra_inv[ZK_HELPER_MAX_CHALLENGES];   // optional: 1 / (gamma_k + table column)
};
template <int NCH>
__global__ void __launch_bounds__(256, ZK_HELPER_WAVES)
helper_cols_kernel(const u64 *__restrict__ prog, const u64 *__restrict__ compiled, TraceView t, HelperChallenges H,
                   u32 chunk, HelperOut O, size_t helper_stride, u32 extra_pc, int *__restrict__ err_flag) {
    const u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= t.n) return;
    const CBlob B(compiled);
    const u32 n_entries = B.n_entries();
    const bool has_next = row + 1 < t.n;       // "table" semantics: next-row terms are dropped on the last row
    auto ld = [&](u32 col, u32 next, u64 &v) {
        if (next && !has_next) return false;
        v = t.base[(size_t)col * t.stride + row + next];
        return true;
    };
    // ZK_HELPER_BATCH is a multiple of chunk (1 or 2), so batches never split a helper group
    for (u32 e0 = 0; e0 < n_entries; e0 += ZK_HELPER_BATCH) {
        u64 v[NCH][ZK_HELPER_BATCH], pre[NCH][ZK_HELPER_BATCH], flt[ZK_HELPER_BATCH];
        u64 run[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) run[k] = 1;
#pragma unroll ZK_HELPER_UNROLL
        for (int i = 0; i < ZK_HELPER_BATCH; ++i) {
            flt[i] = 0;
#pragma unroll
            for (int k = 0; k < NCH; ++k) { v[k][i] = 1; pre[k][i] = 1; }
            if (e0 + i < n_entries) {
                u64 d[NCH], f;
                centry_eval<NCH>(B, e0 + i, ld, d, f);
                if (f > 1) atomicExch(err_flag, 1);          // "Non-binary filter?" (plonky2 asserts)
                flt[i] = f;
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    if (f == 1 && d[k] == 0) { atomicExch(err_flag, 2); d[k] = 1; }  // 1/0: plonky2 would panic
                    v[k][i] = f == 1 ? d[k] : 1;             // dummy 1 where filtered out
                    pre[k][i] = run[k];
                    run[k] = gl_mul(run[k], v[k][i]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            u64 inv = gl_inv(run[k]);                        // ONE inversion per challenge for the whole batch
#pragma unroll
            for (int i = ZK_HELPER_BATCH - 1; i >= 0; --i) {
                if (e0 + i < n_entries) {
                    u64 vi = v[k][i];
                    v[k][i] = flt[i] == 1 ? gl_mul(inv, pre[k][i]) : 0;
                    inv = gl_mul(inv, vi);
                }
            }
            u64 *out = O.helpers[k];
            if (chunk == 1) {
#pragma unroll
                for (int i = 0; i < ZK_HELPER_BATCH; ++i)
                    if (e0 + i < n_entries) out[(size_t)(e0 + i) * helper_stride + row] = gl_canon(v[k][i]);
            } else {
#pragma unroll
                for (int i = 0; i < ZK_HELPER_BATCH; i += 2)
                    if (e0 + i < n_entries) {
                        u64 s = e0 + i + 1 < n_entries ? gl_add(v[k][i], v[k][i + 1]) : v[k][i];
                        out[(size_t)((e0 + i) >> 1) * helper_stride + row] = gl_canon(s);
                    }
            }
        }
    }
    if (O.extra_inv[0]) {  // logUp table column: 1 / (gamma + table(row))
        u32 pc = extra_pc;
        const u64 tc = prog_eval_column_dot(prog, pc, t, row);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            u64 d = gl_canon(gl_add(tc, H.gamma[k]));
            if (d == 0) { atomicExch(err_flag, 2); d = 1; }
            O.extra_inv[k][row] = gl_canon(gl_inv(d));
        }
    }
}

// Fast path of helper_cols_kernel for logUp range checks whose entries are all `Column::single` with the default
// filter (every lookup of the reference except Memory's: arithmetic_stark.rs:320-327, byte_packing_stark.rs:426-437,
// keccak_sponge_stark.rs:946-953): no program interpretation, the 16 loads of a batch are independent.
// cols[e] = trace column of entry e; table_col = the (single) table column.
template <int NCH>
__global__ void __launch_bounds__(256)
lookup_singles_kernel(const u32 *__restrict__ cols, u32 n_entries, u32 table_col, TraceView t, HelperChallenges H, u32 chunk,
                      HelperOut O, size_t helper_stride, int *__restrict__ err_flag) {
    const u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= t.n) return;
    for (u32 e0 = 0; e0 < n_entries; e0 += ZK_SINGLES_BATCH) {
        u64 x[ZK_SINGLES_BATCH];
#pragma unroll
        for (int i = 0; i < ZK_SINGLES_BATCH; ++i) x[i] = e0 + i < n_entries ? t.base[(size_t)cols[e0 + i] * t.stride + row] : 0;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            u64 v[ZK_SINGLES_BATCH], pre[ZK_SINGLES_BATCH];
            u64 run = 1;
#pragma unroll
            for (int i = 0; i < ZK_SINGLES_BATCH; ++i) {
                u64 d = gl_canon(gl_add(x[i], H.gamma[k]));
                if (e0 + i >= n_entries) d = 1;
                else if (d == 0) { atomicExch(err_flag, 2); d = 1; }     // 1/0: plonky2 would panic
                v[i] = d;
                pre[i] = run;
                run = gl_mul(run, d);
            }
            u64 inv = gl_inv(run);
#pragma unroll
            for (int i = ZK_SINGLES_BATCH - 1; i >= 0; --i) {
                const u64 vi = v[i];
                v[i] = gl_mul(inv, pre[i]);
                inv = gl_mul(inv, vi);
            }
            u64 *out = O.helpers[k];
            if (chunk == 1) {
#pragma unroll
                for (int i = 0; i < ZK_SINGLES_BATCH; ++i)
                    if (e0 + i < n_entries) out[(size_t)(e0 + i) * helper_stride + row] = gl_canon(v[i]);
            } else {
#pragma unroll
                for (int i = 0; i < ZK_SINGLES_BATCH; i += 2)
                    if (e0 + i < n_entries) {
                        u64 s = e0 + i + 1 < n_entries ? gl_add(v[i], v[i + 1]) : v[i];
                        out[(size_t)((e0 + i) >> 1) * helper_stride + row] = gl_canon(s);
                    }
            }
        }
    }
    const u64 tc = t.base[(size_t)table_col * t.stride + row];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        u64 d = gl_canon(gl_add(tc, H.gamma[k]));
        if (d == 0) { atomicExch(err_flag, 2); d = 1; }
        O.extra_inv[k][row] = gl_canon(gl_inv(d));
    }
}

// x[row] = sum_h helpers[h][row]  ( - freq(row) * table_inv[row]  when freq_pc != 0 )
static __global__ void helper_row_sums_kernel(const u64 *__restrict__ helpers, size_t helper_stride, u32 n_helpers,
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

static __global__ void __launch_bounds__(ZK_SCAN_THREADS)
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
// single block: exclusive scan of the block totals, 256 lanes each owning a contiguous chunk
static __global__ void __launch_bounds__(256) scan_totals_kernel(u64 *totals, u32 n_blocks) {
    __shared__ u64 sh[256];
    const u32 per = (n_blocks + 255) / 256;
    const u32 lo = threadIdx.x * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
    u64 run = 0;
    for (u32 i = lo; i < hi; ++i) run = gl_add(run, totals[i]);
    sh[threadIdx.x] = run;
    __syncthreads();
    for (u32 off = 1; off < 256; off <<= 1) {
        u64 add = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] = gl_add(sh[threadIdx.x], add);
        __syncthreads();
    }
    run = threadIdx.x ? sh[threadIdx.x - 1] : 0;
    for (u32 i = lo; i < hi; ++i) { u64 v = totals[i]; totals[i] = run; run = gl_add(run, v); }
}
static __global__ void scan_finish_kernel(const u64 *__restrict__ incl, const u64 *__restrict__ totals, u32 n,
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
#endif  // ZK_DEVICE_FUNCS_ONLY
