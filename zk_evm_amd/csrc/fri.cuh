// Device kernels for polynomial openings and FRI (plonky2 1.0.0 `PolynomialBatch::prove_openings`,
// `fri_proof`; [EXT] plonky2/src/fri/{oracle,prover}.rs), reached from the reference through
// starky `prove_with_commitment` (evm_arithmetization/src/prover.rs:322).
//
// MI355X-first restructuring (results are identical field elements, the route differs):
//   * openings: coefficients live bit-reversed on the device, so f(z) = sum_p c[p] * z^bitrev(p) is
//     a dot product of each coefficient column with one weight table W[p] = z^bitrev(p) shared by
//     all columns (built from the log n squarings of z) -- one coalesced streaming pass.
//   * prove_openings: the reference combines COEFFICIENT vectors (reduce_polys_base), divides by
//     (X - z) with a sequential Horner scan, pads and runs a size-N FFT.  Here the same polynomial
//     is produced in the VALUE domain, point-parallel, straight from the LDE matrices that are
//     already resident in HBM:  V[j] = sum_b alpha^(k_b..) (sum_k alpha^k f_k(x_j) - y_b)/(x_j - z_b)
//     -- exactly what the verifier's `fri_combine_initial` evaluates -- followed by one coset iNTT
//     of two columns.  No scan, no (C+A+4)-column coefficient pass.
//   * commit-phase folding is done on bit-reversed coefficients, where the `arity` coefficients of
//     one output are a stride-M column: fully coalesced.
#pragma once
#include "gl.cuh"
#ifndef ZK_DEVICE_FUNCS_ONLY
#include "poseidon.cuh"
#endif

__device__ __forceinline__ gl2 gl2_mul_base(gl2 x, u64 s) { return gl2_make(gl_mul(x.a, s), gl_mul(x.b, s)); }
__device__ __forceinline__ gl2 gl2_inv_dev(gl2 x) {
    // 1/(a + bX) = (a - bX)/(a^2 - 7 b^2)
    u64 nrm = gl_sub(gl_sqr(x.a), gl_mul7(gl_sqr(x.b)));
    u64 ni = gl_inv(nrm);
    return gl2_make(gl_mul(x.a, ni), gl_mul(gl_neg(x.b), ni));
}

// ---- delayed-reduction dot products --------------------------------------------------------------
// sum of (scalar u64) * (vector u64) products, unreduced: value = s00 + s01 * 2^32 + s11 * 2^64
struct DotAcc {
    u64 s00, s01, s11;     // low 64 bits of the three partial-product columns
    u32 h00, h01, h11;     // their carries
};
__device__ __forceinline__ void dot_acc_init(DotAcc &d) { d.s00 = d.s01 = d.s11 = 0; d.h00 = d.h01 = d.h11 = 0; }
// c0, c1: SGPR halves of the coefficient; v: the lane's value
__device__ __forceinline__ void dot_acc_mac(DotAcc &d, u32 c0, u32 c1, u64 v) {
    const u32 v0 = (u32)v, v1 = (u32)(v >> 32);
#if !defined(__HIP_DEVICE_COMPILE__)      // portable: the same words (the host pass of hipcc -- parsed, never run -- and the CPU emulation, tests/emu/)
    u64 t;
    t = d.s00 + (u64)c0 * v0; d.h00 += t < d.s00; d.s00 = t;
    t = d.s01 + (u64)c0 * v1; d.h01 += t < d.s01; d.s01 = t;
    t = d.s01 + (u64)c1 * v0; d.h01 += t < d.s01; d.s01 = t;
    t = d.s11 + (u64)c1 * v1; d.h11 += t < d.s11; d.s11 = t;
#else
    asm("v_mad_u64_u32 %[s00], vcc, %[c0], %[v0], %[s00]\n\t"
        "v_addc_co_u32 %[h00], vcc, 0, %[h00], vcc\n\t"
        "v_mad_u64_u32 %[s01], vcc, %[c0], %[v1], %[s01]\n\t"
        "v_addc_co_u32 %[h01], vcc, 0, %[h01], vcc\n\t"
        "v_mad_u64_u32 %[s01], vcc, %[c1], %[v0], %[s01]\n\t"
        "v_addc_co_u32 %[h01], vcc, 0, %[h01], vcc\n\t"
        "v_mad_u64_u32 %[s11], vcc, %[c1], %[v1], %[s11]\n\t"
        "v_addc_co_u32 %[h11], vcc, 0, %[h11], vcc"
        : [s00] "+v"(d.s00), [s01] "+v"(d.s01), [s11] "+v"(d.s11), [h00] "+v"(d.h00), [h01] "+v"(d.h01),
          [h11] "+v"(d.h11)
        : [c0] "s"(c0), [c1] "s"(c1), [v0] "v"(v0), [v1] "v"(v1)
        : "vcc");
#endif
}
// (lo + hi * 2^64) mod p as a lazy u64; hi < 2^32:  2^64 = 2^32 - 1
__device__ __forceinline__ u64 fold96(u64 lo, u32 hi) { return gl_add(lo, ((u64)hi << 32) - hi); }
// The accumulated value W = s00 + s01 2^32 + (s11 + h00) 2^64 + h01 2^96 + h11 2^128 as a lazy u64.  Summed word by word
// (seven carry adds) it is [w4 : w3 : w2 : w1 : w0], and with 2^64 = 2^32 - 1, 2^96 = -1, 2^128 = -2^32 that is
// [w1 : w0] - [w4 : w3] + w2 (2^32 - 1): the fold of a 128-bit product (gl.cuh) with a 64-bit subtrahend -- 15 instructions.
// (r03s folded the three columns separately and recombined them with two field multiplies: 65 instructions, which was most
// of what a compiled lookup / CTL entry of one or two terms cost.)  Needs fewer than 2^31 accumulated products (w4 < 2^31:
// one borrow correction is enough); the term counts here are column counts.
__device__ __forceinline__ u64 dot_acc_reduce(const DotAcc &d) {
    const u32 a0 = (u32)d.s00, a1 = (u32)(d.s00 >> 32), b0 = (u32)d.s01, b1 = (u32)(d.s01 >> 32), c0 = (u32)d.s11,
              c1 = (u32)(d.s11 >> 32);
#if !defined(__HIP_DEVICE_COMPILE__)      // portable: the same twelve + three word operations
    u64 x = (u64)a1 + b0;                      const u32 pw1 = (u32)x;
    x = (u64)b1 + c0 + (x >> 32);              u32 pw2 = (u32)x;
    x = (u64)c1 + (x >> 32);                   u32 pw3 = (u32)x;
    u32 pw4 = d.h11 + (u32)(x >> 32);
    x = (u64)pw2 + d.h00;                      pw2 = (u32)x;
    x = (u64)pw3 + d.h01 + (x >> 32);          pw3 = (u32)x;
    pw4 += (u32)(x >> 32);
    const u64 m = ((u64)pw1 << 32) | a0, sub = ((u64)pw4 << 32) | pw3;
    u64 pr = m - sub;
    if (m < sub) pr -= 0xFFFFFFFFu;                                    /* borrow: -= EPS (== += p) */
    u64 q = pr + (u64)pw2 * 0xFFFFFFFFu;                               /* += w2 (2^32 - 1) */
    if (q < pr) q += 0xFFFFFFFFu;                                      /* a carry is 2^64 = 2^32 - 1 again */
    return q;
#else
    u32 w1, w2, w3, w4, lo, hi, e;
    asm("v_add_co_u32 %[w1], vcc, %[a1], %[b0]\n\t"
        "v_addc_co_u32 %[w2], vcc, %[b1], %[c0], vcc\n\t"
        "v_addc_co_u32 %[w3], vcc, 0, %[c1], vcc\n\t"
        "v_addc_co_u32 %[w4], vcc, 0, %[h11], vcc\n\t"
        "v_add_co_u32 %[w2], vcc, %[w2], %[h00]\n\t"
        "v_addc_co_u32 %[w3], vcc, %[w3], %[h01], vcc\n\t"
        "v_addc_co_u32 %[w4], vcc, 0, %[w4], vcc\n\t"
        "v_sub_co_u32 %[lo], vcc, %[a0], %[w3]\n\t"            /* [hi:lo] = [w1:w0] - [w4:w3] */
        "v_subb_co_u32 %[hi], vcc, %[w1], %[w4], vcc\n\t"
        "v_cndmask_b32_e64 %[e], 0, -1, vcc\n\t"               /* borrow: -= EPS (== += p) */
        "v_sub_co_u32 %[lo], vcc, %[lo], %[e]\n\t"
        "v_subbrev_co_u32 %[hi], vcc, 0, %[hi], vcc"
        : [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [w4] "=&v"(w4), [lo] "=&v"(lo), [hi] "=&v"(hi), [e] "=&v"(e)
        : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [c0] "v"(c0), [c1] "v"(c1), [h00] "v"(d.h00),
          [h01] "v"(d.h01), [h11] "v"(d.h11)
        : "vcc");
    u64 r = ((u64)hi << 32) | lo;
    asm("v_mad_u64_u32 %[r], vcc, %[t2], -1, %[r]\n\t"         /* += w2 (2^32 - 1); a carry is 2^64 = 2^32 - 1 again (gl.cuh, GL_ASM_TAIL) */
        "v_cndmask_b32_e64 %[c], 0, 1, vcc\n\t"
        "v_mad_u64_u32 %[r], vcc, %[c], -1, %[r]"
        : [r] "+v"(r), [c] "=&v"(e)
        : [t2] "v"(w2)
        : "vcc");
    return r;
#endif
}

// same with per-lane (VGPR) coefficients
__device__ __forceinline__ void dot_acc_mac_v(DotAcc &d, u64 c, u64 v) {
    const u32 c0 = (u32)c, c1 = (u32)(c >> 32), v0 = (u32)v, v1 = (u32)(v >> 32);
#if !defined(__HIP_DEVICE_COMPILE__)      // portable: the same words (the host pass of hipcc -- parsed, never run -- and the CPU emulation, tests/emu/)
    u64 t;
    t = d.s00 + (u64)c0 * v0; d.h00 += t < d.s00; d.s00 = t;
    t = d.s01 + (u64)c0 * v1; d.h01 += t < d.s01; d.s01 = t;
    t = d.s01 + (u64)c1 * v0; d.h01 += t < d.s01; d.s01 = t;
    t = d.s11 + (u64)c1 * v1; d.h11 += t < d.s11; d.s11 = t;
#else
    asm("v_mad_u64_u32 %[s00], vcc, %[c0], %[v0], %[s00]\n\t"
        "v_addc_co_u32 %[h00], vcc, 0, %[h00], vcc\n\t"
        "v_mad_u64_u32 %[s01], vcc, %[c0], %[v1], %[s01]\n\t"
        "v_addc_co_u32 %[h01], vcc, 0, %[h01], vcc\n\t"
        "v_mad_u64_u32 %[s01], vcc, %[c1], %[v0], %[s01]\n\t"
        "v_addc_co_u32 %[h01], vcc, 0, %[h01], vcc\n\t"
        "v_mad_u64_u32 %[s11], vcc, %[c1], %[v1], %[s11]\n\t"
        "v_addc_co_u32 %[h11], vcc, 0, %[h11], vcc"
        : [s00] "+v"(d.s00), [s01] "+v"(d.s01), [s11] "+v"(d.s11), [h00] "+v"(d.h00), [h01] "+v"(d.h01),
          [h11] "+v"(d.h11)
        : [c0] "v"(c0), [c1] "v"(c1), [v0] "v"(v0), [v1] "v"(v1)
        : "vcc");
#endif
}

// W[p] = z^bitrev(p, log_n);  zpow[k] = z^(2^k) (ext), k < log_n
#ifndef ZK_DEVICE_FUNCS_ONLY   // the kernels of openings + FRI (not needed by units that only use DotAcc)
struct ZPowers { u64 a[32], b[32]; };
static __global__ void ext_pow_bitrev_table_kernel(u64 *wa, u64 *wb, int log_n, ZPowers zp) {
    u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >> log_n) return;
    u32 e = bitrev32(p, log_n);
    gl2 acc = gl2_make(1, 0);
    for (int k = 0; k < log_n; ++k)
        if ((e >> k) & 1) acc = gl2_mul(acc, gl2_make(zp.a[k], zp.b[k]));
    wa[p] = gl_canon(acc.a);
    wb[p] = gl_canon(acc.b);
}

// The same table from two small ones: z^e = z^(e mod 2^h) * (z^(2^h))^(e >> h), so a point of the big table is two gathers
// (tables of 2^h and 2^(log_n - h) entries, cache resident) and ONE extension multiply instead of log_n conditional ones
// (27 tables of 2^20 entries per segment: 1.5 ms with the bit loop).  small = [A.a | A.b | B.a | B.b].
static __global__ void ext_pow_small_tables_kernel(u64 *small, int h, int log_n, ZPowers zp) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 na = 1u << h, nb = 1u << (log_n - h);
    if (i >= na + nb) return;
    const bool second = i >= na;
    const u32 e = second ? i - na : i;
    gl2 acc = gl2_make(1, 0);
    for (int k = 0; k < (second ? log_n - h : h); ++k)
        if ((e >> k) & 1) acc = gl2_mul(acc, gl2_make(zp.a[k + (second ? h : 0)], zp.b[k + (second ? h : 0)]));
    u64 *a = second ? small + 2 * na : small;
    const u32 cnt = second ? nb : na;
    a[e] = gl_canon(acc.a);
    a[cnt + e] = gl_canon(acc.b);
}
static __global__ void ext_pow_bitrev_from_small_kernel(u64 *__restrict__ wa, u64 *__restrict__ wb, int log_n, int h,
                                                        const u64 *__restrict__ small) {
    const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >> log_n) return;
    const u32 e = bitrev32(p, log_n);
    const u32 na = 1u << h, nb = 1u << (log_n - h);
    const u32 lo = e & (na - 1), hi = e >> h;
    const u64 *A = small, *B = small + 2 * na;
    const gl2 r = gl2_mul(gl2_make(A[lo], A[na + lo]), gl2_make(B[hi], B[nb + hi]));
    wa[p] = gl_canon(r.a);
    wb[p] = gl_canon(r.b);
}

// Openings: f_col(z_t) = sum_p c[col][p] * W_t[p] for up to ZK_EVAL_MAX_POINTS points in ONE pass over the
// coefficients (zeta and g*zeta open the same columns).  Point t only covers columns [first[t], last[t]).
// partial[((t * n_cols + col) * gridDim.x + chunk) * 2 + {0,1}] = sum over the row chunk.
#define ZK_EVAL_MAX_POINTS 3
struct EvalPoints {
    const u64 *wa[ZK_EVAL_MAX_POINTS], *wb[ZK_EVAL_MAX_POINTS];
    u32 first[ZK_EVAL_MAX_POINTS], last[ZK_EVAL_MAX_POINTS];
};
// Block = ZK_EVAL_ROWS consecutive coefficient positions x a group of columns.  The weights of the block's positions are
// staged in LDS ONCE (the r02 form gave every (row chunk, column) pair its own block, which re-read 16 NP bytes of weights
// per 8-byte coefficient: 3.1x the algorithmic traffic at 2^20, PMC r03d); then every wave walks its share of the group's
// columns -- 32 coalesced coefficient loads in flight per lane, weights from LDS, delayed-reduction accumulators -- and folds
// its 64 lanes with DPP-free shuffles.  grid = (row chunks, column groups).
// Row-chunk size (fri_host.inc kEvalRows, ZK_EVAL_ROWS): every (column, chunk) pair ends in four wave reductions (fold the 96-bit
// accumulators, six shuffle steps each: ~500 instructions), which at 512 rows = 8 coefficients per lane was 62 of the kernel's 108
// instructions per coefficient (PMC r03s: 4.4 cycles per instruction, i.e. issue-bound at 2.6 TB/s); at 2048 rows it is 16.
#define ZK_EVAL_ROWS_MAX 2048     // 64 KiB of weights per block at two points, 96 KiB at three
__device__ __forceinline__ u64 wave_sum_gl(u64 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u64 o = ((u64)(u32)__shfl_down((int)(u32)(v >> 32), off) << 32) | (u32)__shfl_down((int)(u32)v, off);
        v = gl_add(v, o);
    }
    return v;                                        // lane 0 holds the sum
}
template <int NP>
__global__ void __launch_bounds__(1024)
eval_columns_partial_kernel(const u64 *__restrict__ coeffs, size_t col_stride, u32 n, u32 n_cols, u32 cols_per_group,
                            u32 rows_max, EvalPoints P, u64 *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) u64 wlds[];      // [NP][2][rows]
    const u32 rows = n < rows_max ? n : rows_max;
    const u32 lo = blockIdx.x * rows;
    const u32 c_lo = blockIdx.y * cols_per_group, c_hi = c_lo + cols_per_group < n_cols ? c_lo + cols_per_group : n_cols;
    for (u32 e = threadIdx.x; e < rows; e += blockDim.x) {
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            wlds[(2 * t) * rows + e] = P.wa[t][lo + e];
            wlds[(2 * t + 1) * rows + e] = P.wb[t][lo + e];
        }
    }
    __syncthreads();
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    for (u32 col = c_lo + wave; col < c_hi; col += n_waves) {
        const u64 *c = coeffs + (size_t)col * col_stride + lo;
        bool on[NP];
        DotAcc acc[NP][2];
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            on[t] = col >= P.first[t] && col < P.last[t];           // wave-uniform
            dot_acc_init(acc[t][0]); dot_acc_init(acc[t][1]);
        }
        constexpr u32 UN = 8;                                       // coefficient loads in flight per lane
        u32 p = lane;
        for (; p + (UN - 1) * 64 < rows; p += UN * 64) {
            u64 v[UN];
#pragma unroll
            for (u32 i = 0; i < UN; ++i) v[i] = c[p + i * 64];
#pragma unroll
            for (u32 i = 0; i < UN; ++i)
#pragma unroll
                for (int t = 0; t < NP; ++t)
                    if (on[t]) {
                        dot_acc_mac_v(acc[t][0], wlds[(2 * t) * rows + p + i * 64], v[i]);
                        dot_acc_mac_v(acc[t][1], wlds[(2 * t + 1) * rows + p + i * 64], v[i]);
                    }
        }
        for (; p < rows; p += 64) {
            const u64 v = c[p];
#pragma unroll
            for (int t = 0; t < NP; ++t)
                if (on[t]) {
                    dot_acc_mac_v(acc[t][0], wlds[(2 * t) * rows + p], v);
                    dot_acc_mac_v(acc[t][1], wlds[(2 * t + 1) * rows + p], v);
                }
        }
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            if (!on[t]) continue;
            const u64 a = wave_sum_gl(dot_acc_reduce(acc[t][0])), b = wave_sum_gl(dot_acc_reduce(acc[t][1]));
            if (lane == 0) {
                const size_t o = (((size_t)t * n_cols + col) * gridDim.x + blockIdx.x) * 2;
                partial[o] = a;
                partial[o + 1] = b;
            }
        }
    }
}
// out[(t * n_cols + col) * 2 ..] = sum of the chunks (entries of points that skip the column stay 0).  One WAVE per entry:
// the chunks of an entry are consecutive in `partial`, lanes read them coalesced and fold with shuffles (with 2048 row chunks
// at 2^20 a lane per entry walking them one by one took 0.45 ms per launch).
static __global__ void __launch_bounds__(256) eval_columns_reduce_kernel(const u64 *partial, u32 chunks, u32 n_entries, u64 *out) {
    const u32 c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (c >= n_entries) return;                       // (whole waves: blockDim is a multiple of 64)
    u64 a = 0, b = 0;
    for (u32 k = lane; k < chunks; k += 64) {
        a = gl_add(a, partial[((size_t)c * chunks + k) * 2]);
        b = gl_add(b, partial[((size_t)c * chunks + k) * 2 + 1]);
    }
    a = wave_sum_gl(a);
    b = wave_sum_gl(b);
    if (lane == 0) {
        out[2 * c] = gl_canon(a);
        out[2 * c + 1] = gl_canon(b);
    }
}

// ---- batch combination ------------------------------------------------------------------------------
// V[j] = sum_b (prod of later shifts) * (G_b(x_j) - y_b) / (x_j - z_b),   G_b = sum_k alpha^k f_{b,k}.
// G_b is linear in the committed polynomials, so it is formed where they are SHORTEST: on the n bit-reversed coefficients
// (MODE 1: one lane per coefficient index, every column read once for all batches -- half the bytes and half the multiply-adds
// of the 2n LDE values the r01-r03 form read), extended to the 2n coset points by ONE low-degree extension of the
// 2 * n_batches component columns (ntt_host.inc), and divided pointwise (MODE 2).  MODE 0 is the value-domain form (the
// columns' LDE values, one lane per point, sums and division in one kernel; ZK_FRI_COEFF_COMBINE=0).
// The loop runs over the DISTINCT columns (the zeta and g*zeta batches open the same trace / auxiliary columns, so each
// value is loaded once and feeds every batch that opens it), and the alpha-power dot products use delayed reduction: the
// four 32x32 partial products of coef * value are summed in 96-bit accumulators (v_mad_u64_u32 + carry) and folded mod p
// once per index -- 8 VALU instructions per (column, batch, component) instead of a field multiply plus a field add.
#define ZK_FRI_MAX_BATCHES 4
struct FriCombineArgs {
    int n_batches;
    int log_N;                                  // MODE 0 / 2: points;  MODE 1: log2 of the coefficient count
    const u64 *tw;                              // w_N^k, k < N/2
    u64 coset_shift;                            // g
    u32 n_cols;                                 // distinct columns
    const u64 *const *cols;                     // device array [n_cols] of column base pointers (MODE 0: LDE, natural; MODE 1: coefficients)
    const u64 *coef;                            // device array [n_cols][n_batches][2]: alpha^pos (a, b); (0,0) = not opened
    u64 y[ZK_FRI_MAX_BATCHES][2];               // reduced opening sum_k alpha^k f_k(z_b)
    u64 z[ZK_FRI_MAX_BATCHES][2];               // opening point
    u64 shift[ZK_FRI_MAX_BATCHES][2];           // alpha^(n_polys[b])
    u64 *out_a, *out_b;                         // [N] each
    u64 *g;                                     // MODE 1 out / MODE 2 in: component (b, c) at g + (2 b + c) * g_stride
    size_t g_stride;
    // MODE 0 over ONE ROW SHARD (SURVEY 8(e) level 3): the columns hold the shard's rows in leaf order, thread j = leaf
    // shard_rank * n_rows + j, whose point is the natural index bitrev(that); shard_lw = 0: the whole domain, natural order
    u32 sharded, shard_lw, shard_rank;
};

// The column pointers come out of a device array, so the compiler knows nothing about their address space and would emit
// FLAT loads (address-space check per access, counted in lgkmcnt as well, so that every wait for a scalar coefficient load
// also waited for the column loads in flight).  They are global memory: say so, and the load takes the wave-uniform base from
// an SGPR pair plus the lane's 32-bit offset.
__device__ __forceinline__ u64 fri_col_load(const u64 *col, u32 j) {
    typedef const u64 __attribute__((address_space(1))) *gcol_t;
    return ((gcol_t)col)[j];
}

template <int NB, int MODE>
__global__ void __launch_bounds__(256) fri_combine_kernel(FriCombineArgs A) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >> (A.log_N - (MODE == 0 ? A.shard_lw : 0))) return;
    gl2 sums[NB];
    if (MODE == 2) {
#pragma unroll
        for (int b = 0; b < NB; ++b) sums[b] = gl2_make(A.g[(size_t)(2 * b) * A.g_stride + j], A.g[(size_t)(2 * b + 1) * A.g_stride + j]);
    } else {
        DotAcc acc[NB][2];
#pragma unroll
        for (int b = 0; b < NB; ++b) { dot_acc_init(acc[b][0]); dot_acc_init(acc[b][1]); }
        const u64 *__restrict__ coef = A.coef;
        auto consume = [&](u32 k, u64 v) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const u64 ca = coef[((size_t)k * NB + b) * 2], cb = coef[((size_t)k * NB + b) * 2 + 1];
                // uniform across the wave: scalar registers, and a scalar branch for batches that skip this column
                const u32 a0 = __builtin_amdgcn_readfirstlane((u32)ca), a1 = __builtin_amdgcn_readfirstlane((u32)(ca >> 32));
                const u32 b0 = __builtin_amdgcn_readfirstlane((u32)cb), b1 = __builtin_amdgcn_readfirstlane((u32)(cb >> 32));
                if ((a0 | a1 | b0 | b1) != 0) {
                    dot_acc_mac(acc[b][0], a0, a1, v);
                    dot_acc_mac(acc[b][1], b0, b1, v);
                }
            }
        };
        // The columns are 8-16 MB apart and a wave that waits for one load at a time moves 512 B per HBM round trip (2.9 TB/s
        // with every wave slot taken): keep FRI_COLS_IN_FLIGHT loads in flight per lane.
        constexpr u32 FRI_COLS_IN_FLIGHT = 8;
        u32 k = 0;
        for (; k + FRI_COLS_IN_FLIGHT <= A.n_cols; k += FRI_COLS_IN_FLIGHT) {
            u64 v[FRI_COLS_IN_FLIGHT];
#pragma unroll
            for (u32 i = 0; i < FRI_COLS_IN_FLIGHT; ++i) v[i] = fri_col_load(A.cols[k + i], j);
#pragma unroll
            for (u32 i = 0; i < FRI_COLS_IN_FLIGHT; ++i) consume(k + i, v[i]);
        }
        for (; k < A.n_cols; ++k) consume(k, fri_col_load(A.cols[k], j));
#pragma unroll
        for (int b = 0; b < NB; ++b) sums[b] = gl2_make(dot_acc_reduce(acc[b][0]), dot_acc_reduce(acc[b][1]));
        if (MODE == 1) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                A.g[(size_t)(2 * b) * A.g_stride + j] = gl_canon(sums[b].a);
                A.g[(size_t)(2 * b + 1) * A.g_stride + j] = gl_canon(sums[b].b);
            }
            return;
        }
    }
    const u32 half = 1u << (A.log_N - 1);
    const u32 jx = (MODE == 0 && A.sharded) ? bitrev32((A.shard_rank << (A.log_N - A.shard_lw)) + j, A.log_N) : j;
    u64 w = A.tw[jx & (half - 1)];
    if (jx & half) w = gl_neg(w);
    const u64 x = gl_mul(w, A.coset_shift);
    // 1 / (x - z_b) for all batches from ONE extension-field inversion (prefix products, Montgomery's trick): an inversion is
    // ~100 multiplies, the three of a STARK table's opening were a third of this kernel's pointwise part
    gl2 denom[NB], pre[NB], dinv[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        denom[b] = gl2_make(gl_sub(x, A.z[b][0]), gl_neg(A.z[b][1]));
        pre[b] = b ? gl2_mul(pre[b - 1], denom[b]) : denom[0];
    }
    gl2 inv = gl2_inv_dev(pre[NB - 1]);
#pragma unroll
    for (int b = NB - 1; b > 0; --b) {
        dinv[b] = gl2_mul(inv, pre[b - 1]);
        inv = gl2_mul(inv, denom[b]);
    }
    dinv[0] = inv;
    gl2 sum = gl2_make(0, 0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        gl2 numer = gl2_sub(sums[b], gl2_make(A.y[b][0], A.y[b][1]));
        sum = gl2_mul(sum, gl2_make(A.shift[b][0], A.shift[b][1]));
        sum = gl2_add(sum, gl2_mul(numer, dinv[b]));
    }
    A.out_a[j] = gl_canon(sum.a);
    A.out_b[j] = gl_canon(sum.b);
}

// ---- commit-phase fold on bit-reversed coefficients -------------------------------------------
// out[k'] = sum_{i'} bp[i'] * c[i' * M + k'],  bp[i'] = beta^bitrev(i', arity_bits)
struct FoldPowers { u64 a[16], b[16]; };
static __global__ void fri_fold_kernel(const u64 *ca, const u64 *cb, u64 *oa, u64 *ob, u32 M, int arity,
                                FoldPowers bp) {
    u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    gl2 acc = gl2_make(0, 0);
    for (int i = 0; i < arity; ++i) {
        gl2 c = gl2_make(ca[(size_t)i * M + k], cb[(size_t)i * M + k]);
        acc = gl2_add(acc, gl2_mul(c, gl2_make(bp.a[i], bp.b[i])));
    }
    oa[k] = gl_canon(acc.a);
    ob[k] = gl_canon(acc.b);
}

// ---- the same fold on VALUES, for a layer kept in LEAF (bit-reversed) order -- the form that is local to a row shard ----------
// A commit-phase leaf is 2^ab consecutive leaf-ordered values v_i = f(x zeta^bitrev(i)), zeta = the primitive 2^ab-th root, x = the
// coset point shift * w^l' of the leaf (l' = bitrev of its index).  With f(X) = sum_j X^j f_j(X^(2^ab)) the folded polynomial is
// f' = sum_j beta^j f_j and  f'(x^(2^ab)) = sum_j (beta / x)^j u_j,  u_j = 2^-ab sum_p zeta^(-p j) f(x zeta^p)  -- a size-2^ab
// inverse DFT and a Horner step per leaf.  Exact field arithmetic: the same values as the coefficient fold above followed by its
// NTT, and they land in leaf order of the next layer (leaf s of this layer = point bitrev(s) of the next).
struct FriFoldValArgs {
    const u64 *va, *vb;     // [len_local] each: this shard's values, leaf order
    u64 *oa, *ob;           // [len_local >> ab]
    u32 n_out;              // len_local >> ab
    int ab;                 // arity bits
    u32 log_leaves;         // log2 of the GLOBAL number of leaves of this layer (layer bits - ab)
    u32 leaf_base;          // global index of this shard's first leaf
    u64 shift_inv;          // (coset shift of this layer)^-1
    u64 w_inv;              // (primitive root of order 2^(log_leaves + ab))^-1
    u64 zeta_inv_pow[16];   // zeta^-k, k < 2^ab
    u64 inv_arity;          // 2^-ab
    u64 beta[2];
};
static __global__ void fri_fold_values_kernel(FriFoldValArgs A) {
    const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= A.n_out) return;
    const int arity = 1 << A.ab;
    gl2 v[16];
    for (int i = 0; i < arity; ++i) {                       // v[p] = f(x zeta^p): stored position i holds p = bitrev(i)
        const u32 p = bitrev32((u32)i, A.ab);
        v[p] = gl2_make(A.va[(size_t)s * arity + i], A.vb[(size_t)s * arity + i]);
    }
    const u32 lp = bitrev32(A.leaf_base + s, A.log_leaves); // the leaf's point index l'
    const u64 x_inv = gl_mul(A.shift_inv, gl_pow(A.w_inv, lp));
    const gl2 t = gl2_mul_base(gl2_make(A.beta[0], A.beta[1]), x_inv);
    gl2 acc = gl2_make(0, 0);
    for (int j = arity - 1; j >= 0; --j) {                  // Horner in t over u_j (the 2^-ab factor once at the end)
        gl2 u = gl2_make(0, 0);
        for (int p = 0; p < arity; ++p) u = gl2_add(u, gl2_mul_base(v[p], A.zeta_inv_pow[(p * j) & (arity - 1)]));
        acc = gl2_add(gl2_mul(acc, t), u);
    }
    acc = gl2_mul_base(acc, A.inv_arity);
    A.oa[s] = gl_canon(acc.a);
    A.ob[s] = gl_canon(acc.b);
}
// the rows hash_rows wants for a leaf-ordered layer: column 2 i + c of leaf s = component c of value s * arity + i
static __global__ void fri_leaf_rows_kernel(const u64 *va, const u64 *vb, u64 *out, u32 n_leaves, int ab) {
    const u32 e = blockIdx.x * blockDim.x + threadIdx.x;     // e = s * arity + i
    if ((e >> ab) >= n_leaves) return;
    const u32 s = e >> ab, i = e & ((1u << ab) - 1);
    out[(size_t)(2 * i) * n_leaves + s] = va[e];
    out[(size_t)(2 * i + 1) * n_leaves + s] = vb[e];
}

// ---- proof of work -----------------------------------------------------------------------------
// candidate w = base + tid: state = inter with w at `pos`; Poseidon; accept if state[7] has
// >= bits leading zeros.  atomicMin keeps the smallest accepted candidate of the launch.
struct PowState { u64 s[12]; };
static __global__ void fri_pow_kernel(PowState inter, int pos, u64 base, u32 bits, unsigned long long *best) {
    u64 w = base + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= GL_P) return;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = (i == pos) ? w : inter.s[i];
    poseidon_permute(s);
    u64 r = gl_canon(s[7]);
    if (bits == 0 || (r >> (64 - bits)) == 0) atomicMin(best, (unsigned long long)w);
}

// The same grind for the Keccak challenger (`KeccakGoldilocksConfig`): its permutation is plonky2's hash onion ([EXT]
// hash/keccak.rs `KeccakPermutation`) -- keccak256 of the 96 state bytes, then keccak256 of each 32-byte output, the u64
// words < p taken in order -- and the response is the eighth accepted word (the element `get()` pops).  Two Keccak-f
// per candidate unless a word is rejected (2^-32 each); the host used to walk the candidates one at a time (0.2-0.4 s
// per table at 16 bits).
static __global__ void fri_pow_keccak_kernel(PowState inter, int pos, u64 base, u32 bits, unsigned long long *best) {
    u64 w = base + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= GL_P) return;
    u64 a[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) a[i] = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) a[i] = (i == pos) ? w : gl_canon(inter.s[i]);
    a[12] ^= 0x01ULL;                                  // 96-byte message: pad byte at offset 96, 0x80 at offset 135
    a[16] ^= 0x8000000000000000ULL;
    keccak_f1600(a);
    u64 r = 0;
    int got = 0;
    for (int h = 0; h < 16 && got < 8; ++h) {          // (16 hashes without eight words < p cannot happen)
        const u64 o0 = a[0], o1 = a[1], o2 = a[2], o3 = a[3];
        if (got < 8 && o0 < GL_P) { r = o0; ++got; }
        if (got < 8 && o1 < GL_P) { r = o1; ++got; }
        if (got < 8 && o2 < GL_P) { r = o2; ++got; }
        if (got < 8 && o3 < GL_P) { r = o3; ++got; }
        if (got < 8) {
#pragma unroll
            for (int i = 0; i < 25; ++i) a[i] = 0;
            a[0] = o0; a[1] = o1; a[2] = o2; a[3] = o3;
            a[4] = 0x01ULL;
            a[16] = 0x8000000000000000ULL;
            keccak_f1600(a);
        }
    }
    if (bits == 0 || (r >> (64 - bits)) == 0) atomicMin(best, (unsigned long long)w);
}

// ---- generic gather into the flat proof buffer -------------------------------------------------
struct GatherDesc { const u64 *src; u64 dst_off; u32 count; u32 pad; u64 stride; };
static __global__ void gather_words_kernel(const GatherDesc *descs, u32 n_desc, u64 *dst) {
    u32 d = blockIdx.x;
    if (d >= n_desc) return;
    GatherDesc g = descs[d];
    for (u32 i = threadIdx.x; i < g.count; i += blockDim.x) dst[g.dst_off + i] = g.src[(size_t)i * g.stride];  // raw copy (digest slots may hold bytes)
}
#endif  // ZK_DEVICE_FUNCS_ONLY
