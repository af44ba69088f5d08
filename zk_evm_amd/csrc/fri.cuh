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
#include "poseidon.cuh"

__device__ __forceinline__ gl2 gl2_mul_base(gl2 x, u64 s) { return gl2_make(gl_mul(x.a, s), gl_mul(x.b, s)); }
__device__ __forceinline__ gl2 gl2_inv_dev(gl2 x) {
    // 1/(a + bX) = (a - bX)/(a^2 - 7 b^2)
    u64 nrm = gl_sub(gl_sqr(x.a), gl_mul7(gl_sqr(x.b)));
    u64 ni = gl_inv(nrm);
    return gl2_make(gl_mul(x.a, ni), gl_mul(gl_neg(x.b), ni));
}

// W[p] = z^bitrev(p, log_n);  zpow[k] = z^(2^k) (ext), k < log_n
struct ZPowers { u64 a[32], b[32]; };
__global__ void ext_pow_bitrev_table_kernel(u64 *wa, u64 *wb, int log_n, ZPowers zp) {
    u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >> log_n) return;
    u32 e = bitrev32(p, log_n);
    gl2 acc = gl2_make(1, 0);
    for (int k = 0; k < log_n; ++k)
        if ((e >> k) & 1) acc = gl2_mul(acc, gl2_make(zp.a[k], zp.b[k]));
    wa[p] = gl_canon(acc.a);
    wb[p] = gl_canon(acc.b);
}

// partial[(col * gridDim.x + chunk) * 2 + {0,1}] = sum over the chunk of c[col][p] * W[p]
__global__ void __launch_bounds__(256)
eval_columns_partial_kernel(const u64 *__restrict__ coeffs, size_t col_stride, u32 n,
                            const u64 *__restrict__ wa, const u64 *__restrict__ wb,
                            u64 *__restrict__ partial) {
    __shared__ u64 sa[256], sb[256];
    const u64 *c = coeffs + (size_t)blockIdx.y * col_stride;
    u32 per = (n + gridDim.x - 1) / gridDim.x;
    u32 lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    gl2 acc = gl2_make(0, 0);
    for (u32 p = lo + threadIdx.x; p < hi; p += blockDim.x) {
        u64 v = c[p];
        acc = gl2_add(acc, gl2_make(gl_mul(wa[p], v), gl_mul(wb[p], v)));
    }
    sa[threadIdx.x] = acc.a; sb[threadIdx.x] = acc.b;
    __syncthreads();
    for (u32 s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            sa[threadIdx.x] = gl_add(sa[threadIdx.x], sa[threadIdx.x + s]);
            sb[threadIdx.x] = gl_add(sb[threadIdx.x], sb[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        size_t o = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
        partial[o] = sa[0];
        partial[o + 1] = sb[0];
    }
}
__global__ void eval_columns_reduce_kernel(const u64 *partial, u32 chunks, u32 n_cols, u64 *out) {
    u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    gl2 acc = gl2_make(0, 0);
    for (u32 k = 0; k < chunks; ++k)
        acc = gl2_add(acc, gl2_make(partial[((size_t)c * chunks + k) * 2], partial[((size_t)c * chunks + k) * 2 + 1]));
    out[2 * c] = gl_canon(acc.a);
    out[2 * c + 1] = gl_canon(acc.b);
}

// ---- value-domain batch combination ---------------------------------------------------------
#define ZK_FRI_MAX_BATCHES 4
struct FriCombineArgs {
    int n_batches;
    int log_N;
    const u64 *tw;                              // w_N^k, k < N/2
    u64 coset_shift;                            // g
    const u64 *const *cols[ZK_FRI_MAX_BATCHES]; // device array of column base pointers (LDE, natural)
    const u64 *apow[ZK_FRI_MAX_BATCHES];        // device array: alpha^k as (a,b) pairs
    u32 n_polys[ZK_FRI_MAX_BATCHES];
    u64 y[ZK_FRI_MAX_BATCHES][2];               // reduced opening sum_k alpha^k f_k(z_b)
    u64 z[ZK_FRI_MAX_BATCHES][2];               // opening point
    u64 shift[ZK_FRI_MAX_BATCHES][2];           // alpha^(n_polys[b])
    u64 *out_a, *out_b;                         // [N] each
};

__global__ void __launch_bounds__(256) fri_combine_kernel(FriCombineArgs A) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >> A.log_N) return;
    const u32 half = 1u << (A.log_N - 1);
    u64 w = A.tw[j & (half - 1)];
    if (j & half) w = gl_neg(w);
    const u64 x = gl_mul(w, A.coset_shift);
    gl2 sum = gl2_make(0, 0);
    for (int b = 0; b < A.n_batches; ++b) {
        gl2 acc = gl2_make(0, 0);
        const u64 *const *cols = A.cols[b];
        const u64 *ap = A.apow[b];
        for (u32 k = 0; k < A.n_polys[b]; ++k) {
            u64 v = cols[k][j];
            acc = gl2_add(acc, gl2_make(gl_mul(ap[2 * k], v), gl_mul(ap[2 * k + 1], v)));
        }
        gl2 numer = gl2_sub(acc, gl2_make(A.y[b][0], A.y[b][1]));
        gl2 denom = gl2_make(gl_sub(x, A.z[b][0]), gl_neg(A.z[b][1]));
        sum = gl2_mul(sum, gl2_make(A.shift[b][0], A.shift[b][1]));
        sum = gl2_add(sum, gl2_mul(numer, gl2_inv_dev(denom)));
    }
    A.out_a[j] = gl_canon(sum.a);
    A.out_b[j] = gl_canon(sum.b);
}

// ---- commit-phase fold on bit-reversed coefficients -------------------------------------------
// out[k'] = sum_{i'} bp[i'] * c[i' * M + k'],  bp[i'] = beta^bitrev(i', arity_bits)
struct FoldPowers { u64 a[16], b[16]; };
__global__ void fri_fold_kernel(const u64 *ca, const u64 *cb, u64 *oa, u64 *ob, u32 M, int arity,
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

// ---- proof of work -----------------------------------------------------------------------------
// candidate w = base + tid: state = inter with w at `pos`; Poseidon; accept if state[7] has
// >= bits leading zeros.  atomicMin keeps the smallest accepted candidate of the launch.
struct PowState { u64 s[12]; };
__global__ void fri_pow_kernel(PowState inter, int pos, u64 base, u32 bits, unsigned long long *best) {
    u64 w = base + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= GL_P) return;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = (i == pos) ? w : inter.s[i];
    poseidon_permute(s);
    u64 r = gl_canon(s[7]);
    if (bits == 0 || (r >> (64 - bits)) == 0) atomicMin(best, (unsigned long long)w);
}

// ---- generic gather into the flat proof buffer -------------------------------------------------
struct GatherDesc { const u64 *src; u64 dst_off; u32 count; u32 pad; u64 stride; };
__global__ void gather_words_kernel(const GatherDesc *descs, u32 n_desc, u64 *dst) {
    u32 d = blockIdx.x;
    if (d >= n_desc) return;
    GatherDesc g = descs[d];
    for (u32 i = threadIdx.x; i < g.count; i += blockDim.x) dst[g.dst_off + i] = g.src[(size_t)i * g.stride];  // raw copy (digest slots may hold bytes)
}
