// Batched Goldilocks NTT for gfx950: LDS-tiled multi-pass radix-2^k kernels.
//
// Replaces plonky2_field 1.0.0 `ifft` / `fft` / `coset_fft` / `lde` ([EXT] field/src/fft.rs,
// polynomial/mod.rs) as driven by PolynomialBatch::from_values (reference
// evm_arithmetization/src/prover.rs:100).
//
// Design (MI355X-first, not the CPU's per-column recursive FFT):
//   * value-form data is always NATURAL order, coefficient-form data is always kept in
//     BIT-REVERSED order on the device.  values -> coeffs runs its stages from the largest pair
//     distance down (natural in, bit-reversed out), coeffs -> values from the smallest up
//     (bit-reversed in, natural out), so no pass ever performs a bit-reversal permutation through
//     HBM.  Both use the Cooley-Tukey butterfly (a + w b, a - w b): the natural -> bit-reversed
//     direction is the remainder tree f mod (y^D -+ c) with one twiddle per block of 2D indices
//     instead of the textbook Gentleman-Sande (a + b, (a - b) w) -- same outputs, 4 fewer
//     instructions per butterfly (see ntt_bfly).
//   * a transform of size 2^L is split into passes of r <= 10/11 consecutive stages.  One
//     workgroup owns a tile of R = 2^r "rows" spaced d apart (d = smallest butterfly distance of
//     the pass) times T contiguous elements, stages it in LDS once, runs all r stages there in
//     radix-8 register steps, and writes it back in place: HBM traffic is one read + one write
//     per pass per element, every access a T*8-byte contiguous segment (T=1 pass: fully linear).
//   * the zero-padding of `lde` is never materialised: in bit-reversed order the padded
//     coefficient vector is (c, 0, .., 0) groups, and the first rate_bits DIT stages just
//     replicate c, so the first forward pass loads n coefficients, applies the coset factor
//     g^i, and starts at stage rate_bits.
//   * columns are independent: grid = (tiles per column) x (columns); twiddles come from one
//     per-size table shared by all columns (L2 / Infinity-Cache resident).
#pragma once
#include "gl.cuh"

#include "ntt_common.cuh"    // NttPass, the butterfly, twiddle loads (shared with ntt_swap.cuh and its CPU emulation, tests/emu/)

// One register step of K (<= 3) consecutive stages over the LDS tile.
// rows of one sub-problem: t0 + m*q, m in [0, 2^K)
//   !DIT: stage half sizes (rows) q*2^(K-1) .. q     DIT: q .. q*2^(K-1)
// All index math is 32-bit (transforms are <= 2^31 points).  DIT: a stage whose pairs are hm apart (in
// m units) has only hm distinct twiddles per sub-problem (the twiddle of pair (m, m+hm) depends on
// m mod hm); !DIT: one twiddle per block, 2^(K-1)/hm blocks per sub-problem.  Either way a radix-8 step
// issues 1+2+4 = 7 twiddle loads, not 12.
#ifndef ZK_NTT_LOADS_IN_FLIGHT
#define ZK_NTT_LOADS_IN_FLIGHT 8      // = tile elements per lane with the plan of ntt_host.inc (kThreadsShift = 3)
#endif

template <bool DIT, int K>
__device__ __forceinline__ void ntt_step(u64 *tile, const NttPass &p, int log_q, u32 base,
                                         u32 elems, u32 tid, u32 nthr) {
    const int log_t = p.log_t;
    const u32 T = 1u << log_t, q = 1u << log_q;
    const u32 nsub = elems >> K;
    const int log_D0 = p.log_d + log_q;                 // global distance of adjacent m
    const u32 stride8 = (q << log_t) * 8;               // bytes between the tile elements of adjacent m (wave-uniform)
    char *const tile_b = reinterpret_cast<char *>(tile);
    const __amdgpu_buffer_rsrc_t twr = ntt_tw_rsrc(p.tw);
    for (u32 sp = tid; sp < nsub; sp += nthr) {
        u32 u = sp & (T - 1);
        u32 w = sp >> log_t;
        u32 j = w & (q - 1);
        u32 blk = w >> log_q;
        u32 t0 = (blk << (log_q + K)) + j;
        u64 v[1 << K];
        // element m lives at ((t0 + m q) << log_t) + u = a0 + m (q << log_t): one add per address
        const u32 a0 = (t0 << log_t) + u;
#pragma unroll
        for (int m = 0; m < (1 << K); ++m) v[m] = *reinterpret_cast<u64 *>(tile_b + (a0 * 8 + m * stride8));
        const u32 x0 = base + (t0 << p.log_d) + u;      // global index of v[0]; bits [log_D0, log_D0 + K) are zero
        if (DIT) {
            // pair (x, x + D), D = D0 << lm: twiddle T_D[x mod D]; x_m mod D = g + (m mod hm) * D0
            const u32 g8 = (x0 & ((1u << log_D0) - 1)) * 8;
#pragma unroll
            for (int lm = 0; lm < K; ++lm) {
                const int hm = 1 << lm;
                const u32 lvl = (1u << (log_D0 + lm)) - 1;
#pragma unroll
                for (int mm = 0; mm < hm; ++mm) {
                    const u64 tw = ntt_tw_load(twr, g8, lvl + ((u32)mm << log_D0));
#pragma unroll
                    for (int m = mm; m < (1 << K); m += 2 * hm) ntt_bfly(v[m], v[m + hm], tw);
                }
            }
        } else {
            // natural -> bit-reversed: the block of 2D consecutive indices holding x is reduced modulo y^D -+ c, one
            // twiddle c per BLOCK: level s = log_n - 1 - log D has 2^s blocks, block j's twiddle at tw[2^s - 1 + j]
            // (block order, ntt_host.inc) -- uniform over a strided pass, consecutive lanes -> consecutive u64 in the
            // contiguous one; a radix-8 step reads 1 + 2 + 4 adjacent entries.
#pragma unroll
            for (int lm = K - 1; lm >= 0; --lm) {
                const int hm = 1 << lm;
                const int log_D = log_D0 + lm;
                const u32 lvl = (1u << (p.log_n - 1 - log_D)) - 1;
                const u32 blk8 = (x0 >> (log_D + 1)) * 8;
#pragma unroll
                for (int hg = 0; hg < (1 << (K - 1 - lm)); ++hg) {
                    const u64 tw = ntt_tw_load(twr, blk8, lvl + hg);
#pragma unroll
                    for (int mm = 0; mm < hm; ++mm) ntt_bfly(v[hg * 2 * hm + mm], v[hg * 2 * hm + mm + hm], tw);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < (1 << K); ++m) *reinterpret_cast<u64 *>(tile_b + (a0 * 8 + m * stride8)) = v[m];
    }
}

// DIT = false: stages from the largest distance down (values, natural -> coefficients, bit-reversed).
// DIT = true : stages from the smallest distance up (coefficients, bit-reversed -> values, natural).
template <bool DIT>
__global__ void __launch_bounds__(1024) ntt_pass_kernel(NttPass p) {
    extern __shared__ __attribute__((aligned(16))) u64 tile[];
    const int r = p.r, log_t = p.log_t;
    const u32 T = 1u << log_t;
    const u32 tid = threadIdx.x, nthr = blockDim.x;
    // tile -> (hi, lo_tile): base = hi * (d * R) + lo_tile * T     (d = 2^log_d)
    const int log_lo_tiles = p.log_d - log_t;
    const u32 tile_id = p.cols_fastest ? blockIdx.y : blockIdx.x;
    const u32 col_id = p.cols_fastest ? blockIdx.x : blockIdx.y;
    const u32 hi_idx = tile_id >> log_lo_tiles, lo_tile = tile_id & ((1u << log_lo_tiles) - 1);
    const u32 base = (hi_idx << (p.log_d + r)) + (lo_tile << log_t);
    const u64 *src = p.src + (size_t)col_id * p.src_stride;
    u64 *dst = p.dst + (size_t)col_id * p.dst_stride;
    const u32 elems = 1u << (r + log_t);

    // ---- load (global index x = base + t*d + u  ->  lds[t*T + u]) ----
    // ZK_NTT_LOADS_IN_FLIGHT loads per lane are issued before the first is consumed.  (Until r04 this was a rolled loop --
    // address, global_load, s_waitcnt vmcnt(0), ds_write, next -- i.e. elems / threads = 8 SERIAL HBM round trips per tile and
    // lane: the "load phase" cost more wall time than the butterflies, and only the other resident workgroup hid part of it:
    // PMC r03x had the kernel at 0.75 of the VALU issue rate and 5.3 cycles per instruction.)
    if (p.log_rep) {
        // lde's first pass (always the contiguous one, log_d = 0): every coefficient is read and scaled ONCE and written
        // to its 2^log_rep replicas in the tile (the stages that would have produced them are skipped)
        const u32 sbase = base >> p.log_rep, rep = 1u << p.log_rep, n_src = elems >> p.log_rep;
        for (u32 s0 = tid; s0 < n_src; s0 += nthr * ZK_NTT_LOADS_IN_FLIGHT) {
            u64 v[ZK_NTT_LOADS_IN_FLIGHT], sc[ZK_NTT_LOADS_IN_FLIGHT];
#pragma unroll
            for (int k = 0; k < ZK_NTT_LOADS_IN_FLIGHT; ++k) {
                const u32 se = s0 + (u32)k * nthr;
                if (se < n_src) {
                    v[k] = src[sbase + se];
                    if (p.in_scale) sc[k] = p.in_scale[sbase + se];
                }
            }
#pragma unroll
            for (int k = 0; k < ZK_NTT_LOADS_IN_FLIGHT; ++k) {
                const u32 se = s0 + (u32)k * nthr;
                if (se < n_src) {
                    const u64 w = p.in_scale ? gl_mul(v[k], sc[k]) : v[k];
                    for (u32 j = 0; j < rep; ++j) tile[(se << p.log_rep) + j] = w;
                }
            }
        }
    } else {
        for (u32 e0 = tid; e0 < elems; e0 += nthr * ZK_NTT_LOADS_IN_FLIGHT) {
            u64 v[ZK_NTT_LOADS_IN_FLIGHT], sc[ZK_NTT_LOADS_IN_FLIGHT];
#pragma unroll
            for (int k = 0; k < ZK_NTT_LOADS_IN_FLIGHT; ++k) {
                const u32 e = e0 + (u32)k * nthr;
                if (e < elems) {
                    const u32 t = e >> log_t, u = e & (T - 1);
                    const u32 x = base + (t << p.log_d) + u;
                    v[k] = src[x];
                    if (p.in_scale) sc[k] = p.in_scale[x];
                }
            }
#pragma unroll
            for (int k = 0; k < ZK_NTT_LOADS_IN_FLIGHT; ++k) {
                const u32 e = e0 + (u32)k * nthr;
                if (e < elems) tile[e] = p.in_scale ? gl_mul(v[k], sc[k]) : v[k];
            }
        }
    }
    __syncthreads();

    // ---- stages first_stage .. r-1 in radix-2^k register steps ----
    int done = p.first_stage;
    while (done < r) {
        const int k = r - done < 3 ? r - done : 3;
        const int log_q = DIT ? done : (r - done - k);
        if (k == 3) ntt_step<DIT, 3>(tile, p, log_q, base, elems, tid, nthr);
        else if (k == 2) ntt_step<DIT, 2>(tile, p, log_q, base, elems, tid, nthr);
        else ntt_step<DIT, 1>(tile, p, log_q, base, elems, tid, nthr);
        __syncthreads();
        done += k;
    }

    // ---- store ---- (the LDS reads of ZK_NTT_LOADS_IN_FLIGHT elements issued together; stores do not wait)
    for (u32 e0 = tid; e0 < elems; e0 += nthr * ZK_NTT_LOADS_IN_FLIGHT) {
        u64 v[ZK_NTT_LOADS_IN_FLIGHT], sc[ZK_NTT_LOADS_IN_FLIGHT];
#pragma unroll
        for (int k = 0; k < ZK_NTT_LOADS_IN_FLIGHT; ++k) {
            const u32 e = e0 + (u32)k * nthr;
            if (e < elems) {
                v[k] = tile[e];
                if (p.out_scale) { const u32 t = e >> log_t, u = e & (T - 1); sc[k] = p.out_scale[base + (t << p.log_d) + u]; }
            }
        }
#pragma unroll
        for (int k = 0; k < ZK_NTT_LOADS_IN_FLIGHT; ++k) {
            const u32 e = e0 + (u32)k * nthr;
            if (e < elems) {
                const u32 t = e >> log_t, u = e & (T - 1);
                const u32 x = base + (t << p.log_d) + u;
                u64 w = v[k];
                if (p.out_scale) w = gl_mul_canon(w, sc[k]);
                else if (p.apply_out_const) w = gl_mul_canon(w, p.out_const);
                else if (p.last_pass) w = gl_canon(w);          // between passes any u64 representative will do
                dst[x] = w;
            }
        }
    }
}

#include "ntt_swap.cuh"      // r05: the passes on gfx950's lane-swap instructions (v_permlane16/32_swap)

// In-place bit-reversal permutation of each column (only used by the natural<->natural API
// entry points; the commit path never calls it).
static __global__ void bitrev_permute_kernel(u64 *data, size_t stride, int log_n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> log_n) return;
    u64 *col = data + (size_t)blockIdx.y * stride;
    size_t j = bitrev32((u32)i, log_n);
    if (i < j) { u64 a = col[i], b = col[j]; col[i] = b; col[j] = a; }
}

// out[i] = c * s^(bitrev(i, log_n))    (coset power tables in coefficient (bit-reversed) order)
static __global__ void coset_table_kernel(u64 *out, int log_n, u64 s, u64 c) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> log_n) return;
    u32 e = bitrev32((u32)i, log_n);
    out[i] = gl_canon(gl_mul(c, gl_pow(s, e)));
}

// the plan autotuner's input and verdict (ntt_host.inc ntt_swap_decide): pseudo-random words, word-for-word comparison
static __global__ void ntt_tune_fill_kernel(u64 *out, size_t n, u64 seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    out[i] = z ^ (z >> 31);
}
static __global__ void ntt_tune_compare_kernel(const u64 *a, const u64 *b, size_t n, unsigned long long *mismatches) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicAdd(mismatches, 1ULL);
}

// an order-independent 64-bit digest of n words (the column-batch trial, ntt_host.inc: did a batched form produce the same output?)
static __global__ void ntt_tune_digest_kernel(const u64 *a, size_t n, u64 salt, unsigned long long *acc) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 z = a[i] + (i + 1) * 0x9E3779B97F4A7C15ULL + salt * 0xD1B54A32D192ED03ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    atomicAdd(acc, (unsigned long long)(z ^ (z >> 31)));
}

// load factors of the SECOND coset for ntt_contig_wave_kernel_dit<2> (ntt_swap.cuh): a wave's 2^11 values are the 2^11-point transform
// of its 2^10 coefficients in local bit-reversed order, whatever the tile, so value 2 i + 1 is the plain 2^10-point transform of
// c_i * w^bitrev_10(i mod 2^10), w = the root of order 2^11 -- times the first coset's own factor when there is one
static __global__ void wave_coset2_table_kernel(u64 *out, const u64 *first, int log_n, u64 w) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> log_n) return;
    const u64 f = gl_pow(w, bitrev32((u32)i & 1023u, 10));
    out[i] = gl_canon(first ? gl_mul(first[i], f) : f);
}

// block-order levels (ntt_host.inc): out[2^s - 1 + j] = (root of order 2^(s+1))^bitrev_s(j) = w^(bitrev_s(j) * N / 2^(s+1))
static __global__ void twiddle_block_levels_kernel(u64 *out, int log_size, u64 w) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // i = 2^s - 1 + j
    const size_t total = ((size_t)1 << log_size) - 1;
    if (log_size == 0) { if (i == 0) out[0] = 1; return; }
    if (i >= total) { if (i == total) out[i] = 0; return; }
    const int s = 63 - __clzll((unsigned long long)(i + 1));
    const u32 j = (u32)(i + 1 - ((size_t)1 << s));
    out[i] = gl_canon(gl_pow(w, (u64)bitrev32(j, s) << (log_size - 1 - s)));
}

// level layout (ntt_host.inc): out[D - 1 + k] = w^(k * N / (2 D)) for D = 1, 2, .., N/2 and k < D; N = 2^log_size
static __global__ void twiddle_levels_kernel(u64 *out, int log_size, u64 w) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // i = D - 1 + k
    const size_t total = ((size_t)1 << log_size) - 1;
    if (log_size == 0) { if (i == 0) out[0] = 1; return; }
    if (i >= total) { if (i == total) out[i] = 0; return; }
    const int lvl = 63 - __clzll((unsigned long long)(i + 1));     // D = 2^lvl
    const size_t k = i + 1 - ((size_t)1 << lvl);
    out[i] = gl_canon(gl_pow(w, k << (log_size - 1 - lvl)));
}

// element-wise field op (ABI-level access to the device field primitives)
static __global__ void gl_vec_op_kernel(u32 op, const u64 *a, const u64 *b, u64 *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 x = a[i], y = (op == 3 || op == 4) ? 0 : b[i], r;
    switch (op) {
        case 0: r = gl_add(x, y); break;
        case 1: r = gl_sub(x, y); break;
        case 2: r = gl_mul(x, y); break;
        case 3: r = gl_sqr(x); break;
        case 5: out[i] = gl_mul_canon(x, y); return;                     // RAW: must already be < p
        case 6: r = gl_add_canon(x, gl_mul_canon(y, y)); break;          // the NTT butterfly's two halves:
        case 7: r = gl_sub_canon(x, gl_mul_canon(y, y)); break;          //   x +- y^2 with one correction each
        case 8: r = gl_mul_fast(x, y); break;                            // the Poseidon S-box's multiply (rare correction out of line)
        default: r = gl_canon(x) == 0 ? 0 : gl_inv(x); break;
    }
    out[i] = gl_canon(r);
}
