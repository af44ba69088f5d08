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
// PAD (the contiguous pass, T = 1): tile element i lives at i + (i >> 3).  A step with q = 1 reads rows 8 apart from
// consecutive lanes (64-byte stride: eight lanes per bank pair), q = 8 two lanes per bank pair; with one pad word per
// eight the same steps touch every bank once per quarter wave.  (PMC r03n: 48 % of the LDS-active cycles were bank conflicts,
// and since the shorter field multiply the kernel no longer hides them under VALU issue.)
#ifdef ZK_NTT_DEBUG          // tools/kbench only (WRONG results): time the phases of a pass in isolation
#define ZK_NTT_DBG(bit) (p.nt & (bit))
#else
#define ZK_NTT_DBG(bit) 0
#endif
#ifndef ZK_NTT_LOADS_IN_FLIGHT
#define ZK_NTT_LOADS_IN_FLIGHT 8      // = tile elements per lane with the default plan (ntt_host.inc kThreadsShift = 3)
#endif
template <bool PAD>
__device__ __forceinline__ u32 ntt_ph(u32 i) { return PAD ? i + (i >> 3) : i; }

template <bool DIT, int K, bool PAD>
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
        // PAD (log_t = 0): element t0 + m q lives at ph(t0 + m q) = ph(t0) + ph(m q) -- no carry between the two parts: q is a
        // power of two, t0 mod q < q and t0's bits above are a multiple of 8 q / 8 ... (m q mod 8 is a multiple of q, t0 mod 8 <
        // q when q < 8, and m q mod 8 = 0 when q >= 8) -- so the pad costs ONE shift-add per sub-problem; the per-m parts are
        // wave-uniform.  (r03's form re-derived ph() for every access and lost more in index arithmetic than the conflicts cost.)
        const u32 pa0 = PAD ? t0 + (t0 >> 3) : 0;
#pragma unroll
        for (int m = 0; m < (1 << K); ++m)
            v[m] = PAD ? *reinterpret_cast<u64 *>(tile_b + (pa0 * 8 + ntt_ph<true>((u32)m << log_q) * 8))
                       : *reinterpret_cast<u64 *>(tile_b + (a0 * 8 + m * stride8));
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
        for (int m = 0; m < (1 << K); ++m) {
            if (PAD) *reinterpret_cast<u64 *>(tile_b + (pa0 * 8 + ntt_ph<true>((u32)m << log_q) * 8)) = v[m];
            else *reinterpret_cast<u64 *>(tile_b + (a0 * 8 + m * stride8)) = v[m];
        }
    }
}

__device__ __forceinline__ void ntt_stagger(const NttPass &p) {
    if (p.stagger_ticks == 0) return;
    const u32 lin = blockIdx.x + gridDim.x * blockIdx.y;
    if (lin >= p.stagger_blocks) return;
    // mode 1: every other workgroup of the first wave; mode 2: its second half (which of the two shares CUs depends on how the
    // dispatcher fills them: measured, ntt_host.inc)
    const bool late = p.stagger_mode == 1 ? (lin & 1) : lin >= (p.stagger_blocks >> 1);
    if (!late) return;
    const u64 t0 = wall_clock64();
    while (wall_clock64() - t0 < p.stagger_ticks) __builtin_amdgcn_s_sleep(32);
}

// DIT = false: stages from the largest distance down (values, natural -> coefficients, bit-reversed).
// DIT = true : stages from the smallest distance up (coefficients, bit-reversed -> values, natural).
template <bool DIT, bool PAD = false>
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
    ntt_stagger(p);

    // ---- load (global index x = base + t*d + u  ->  lds[t*T + u]) ----
    // ZK_NTT_LOADS_IN_FLIGHT loads per lane are issued before the first is consumed.  (Until r04 this was a rolled loop --
    // address, global_load, s_waitcnt vmcnt(0), ds_write, next -- i.e. elems / threads = 8 SERIAL HBM round trips per tile and
    // lane: the "load phase" cost more wall time than the butterflies, and only the other resident workgroup hid part of it:
    // PMC r03x had the kernel at 0.75 of the VALU issue rate and 5.3 cycles per instruction.)
    if (ZK_NTT_DBG(8)) {
    } else if (p.log_rep) {
        // lde's first pass (always the contiguous one, log_d = 0): every coefficient is read and scaled ONCE and written
        // to its 2^log_rep replicas in the tile (the stages that would have produced them are skipped)
        const u32 sbase = base >> p.log_rep, rep = 1u << p.log_rep, n_src = elems >> p.log_rep;
        for (u32 s0 = tid; s0 < n_src; s0 += nthr * ZK_NTT_LOADS_IN_FLIGHT) {
            u64 v[ZK_NTT_LOADS_IN_FLIGHT], sc[ZK_NTT_LOADS_IN_FLIGHT];
#pragma unroll
            for (int k = 0; k < ZK_NTT_LOADS_IN_FLIGHT; ++k) {
                const u32 se = s0 + (u32)k * nthr;
                if (se < n_src) {
                    v[k] = (p.nt & 1) ? __builtin_nontemporal_load(src + sbase + se) : src[sbase + se];
                    if (p.in_scale) sc[k] = p.in_scale[sbase + se];
                }
            }
#pragma unroll
            for (int k = 0; k < ZK_NTT_LOADS_IN_FLIGHT; ++k) {
                const u32 se = s0 + (u32)k * nthr;
                if (se < n_src) {
                    const u64 w = p.in_scale ? gl_mul(v[k], sc[k]) : v[k];
                    for (u32 j = 0; j < rep; ++j) tile[ntt_ph<PAD>((se << p.log_rep) + j)] = w;
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
                    v[k] = (p.nt & 1) ? __builtin_nontemporal_load(src + x) : src[x];
                    if (p.in_scale) sc[k] = p.in_scale[x];
                }
            }
#pragma unroll
            for (int k = 0; k < ZK_NTT_LOADS_IN_FLIGHT; ++k) {
                const u32 e = e0 + (u32)k * nthr;
                if (e < elems) tile[PAD ? ntt_ph<true>(e0) + (u32)k * (nthr + (nthr >> 3)) : e] = p.in_scale ? gl_mul(v[k], sc[k]) : v[k];   // (nthr is a multiple of 8)
            }
        }
    }
    __syncthreads();

    // ---- stages first_stage .. r-1 in radix-2^k register steps ----
    int done = ZK_NTT_DBG(2) ? r : p.first_stage;       // (kbench -DZK_NTT_DEBUG, nt & 2: no butterflies -- the memory phases alone)
    while (done < r) {
        const int k = r - done < 3 ? r - done : 3;
        const int log_q = DIT ? done : (r - done - k);
        if (k == 3) ntt_step<DIT, 3, PAD>(tile, p, log_q, base, elems, tid, nthr);
        else if (k == 2) ntt_step<DIT, 2, PAD>(tile, p, log_q, base, elems, tid, nthr);
        else ntt_step<DIT, 1, PAD>(tile, p, log_q, base, elems, tid, nthr);
        __syncthreads();
        done += k;
    }

    if (ZK_NTT_DBG(4)) return;                          // (nt & 4: no stores; nt & 8: no loads -- the butterflies alone)
    // ---- store ---- (the LDS reads of ZK_NTT_LOADS_IN_FLIGHT elements issued together; stores do not wait)
    for (u32 e0 = tid; e0 < elems; e0 += nthr * ZK_NTT_LOADS_IN_FLIGHT) {
        u64 v[ZK_NTT_LOADS_IN_FLIGHT], sc[ZK_NTT_LOADS_IN_FLIGHT];
#pragma unroll
        for (int k = 0; k < ZK_NTT_LOADS_IN_FLIGHT; ++k) {
            const u32 e = e0 + (u32)k * nthr;
            if (e < elems) {
                v[k] = tile[PAD ? ntt_ph<true>(e0) + (u32)k * (nthr + (nthr >> 3)) : e];
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
                if (p.nt & 1) __builtin_nontemporal_store(w, dst + x);
                else dst[x] = w;
            }
        }
    }
}

// ---- the strided pass as a PERSISTENT, software-pipelined kernel (r04q) -------------------------------------------------
// ntt_pass_kernel is load -> barrier -> butterflies -> barrier -> store, and the two workgroups that share a CU run those phases
// in lock step: HBM moves the tiles at ~5 TB/s with the vector units idle, then the butterflies run with HBM idle -- the two
// ADD (DESIGN section 4 "r04": 2.02 + 2.30 = 3.94 ms).  A wave's loads return in order, so a prefetch issued inside the
// butterflies is waited for by the next twiddle load; here the ORDER of issue is arranged so that it never is:
//   one workgroup per CU walks its tiles (2^9 rows x 16 contiguous elements = one radix-8 sub-problem per lane and step);
//   per tile a lane issues, in this order: the twiddles of steps 2 and 3, the EIGHT ELEMENTS OF THE NEXT TILE (straight into
//   registers, in step 1's layout: four 128-byte row segments per wave and instruction, like the tile load of the
//   generic kernel), then runs step 1 on twiddles fetched during the previous tile, issues the next tile's step-1 twiddles,
//   and goes on to steps 2 and 3 -- whose `s_waitcnt vmcnt(N)` leaves the younger loads outstanding -- and stores step 3's
//   results straight from registers (again four 128-byte segments per wave and instruction).  The next tile's elements have
//   the whole of this tile's butterflies to arrive.  LDS only carries the two exchanges between the steps (two buffers, so
//   two barriers per tile instead of five, and no LDS round trip at either end).
// Same butterflies in the same order on the same operands as ntt_pass_kernel: bit-identical output.
// Geometry is fixed: r = 9, log_t = 4, 1024 threads; no load / store factors (a strided pass never has any: they belong
// to the contiguous pass at the coefficient end).  Everything else takes ntt_pass_kernel (ntt_host.inc launch_pass).
template <bool DIT>
__device__ __forceinline__ void ntt_tw7(__amdgpu_buffer_rsrc_t twr, const NttPass &p, int log_q, u32 x0, u64 (&tw)[7]) {
    const int log_D0 = p.log_d + log_q;
    if (ZK_NTT_DBG(16)) {                                // (kbench -DZK_NTT_DEBUG: no twiddle loads)
#pragma unroll
        for (int i = 0; i < 7; ++i) tw[i] = x0 + i;
        return;
    }
    if (DIT) {
        const u32 g8 = (x0 & ((1u << log_D0) - 1)) * 8;
#pragma unroll
        for (int lm = 0; lm < 3; ++lm)
#pragma unroll
            for (int mm = 0; mm < (1 << lm); ++mm)
                tw[(1 << lm) - 1 + mm] = ntt_tw_load(twr, g8, ((1u << (log_D0 + lm)) - 1) + ((u32)mm << log_D0));
    } else {
#pragma unroll
        for (int lm = 2; lm >= 0; --lm) {
            const int log_D = log_D0 + lm;
            const u32 lvl = (1u << (p.log_n - 1 - log_D)) - 1;
            const u32 blk8 = (x0 >> (log_D + 1)) * 8;
#pragma unroll
            for (int hg = 0; hg < (1 << (2 - lm)); ++hg) tw[(1 << (2 - lm)) - 1 + hg] = ntt_tw_load(twr, blk8, lvl + hg);
        }
    }
}
template <bool DIT>
__device__ __forceinline__ void ntt_radix8(u64 (&v)[8], const u64 (&tw)[7]) {
    if (DIT) {
#pragma unroll
        for (int lm = 0; lm < 3; ++lm) {
            const int hm = 1 << lm;
#pragma unroll
            for (int mm = 0; mm < hm; ++mm)
#pragma unroll
                for (int m = mm; m < 8; m += 2 * hm) ntt_bfly(v[m], v[m + hm], tw[hm - 1 + mm]);
        }
    } else {
#pragma unroll
        for (int lm = 2; lm >= 0; --lm) {
            const int hm = 1 << lm;
#pragma unroll
            for (int hg = 0; hg < (1 << (2 - lm)); ++hg)
#pragma unroll
                for (int mm = 0; mm < hm; ++mm) ntt_bfly(v[hg * 2 * hm + mm], v[hg * 2 * hm + mm + hm], tw[(1 << (2 - lm)) - 1 + hg]);
        }
    }
}

#ifdef ZK_NTT_DEBUG
__device__ u64 zk_ntt_trace[16];                       // tools/kbench_dbg: time per body section of one wave (10 ns ticks), + bodies
__device__ __forceinline__ u64 ntt_dbg_now() { u64 t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define ZK_NTT_MARK(k) do { if (p.nt & 64) { const u64 n_ = ntt_dbg_now(); tr[k] += n_ - tlast; tlast = n_; } } while (0)
#else
#define ZK_NTT_MARK(k) do { } while (0)
#endif
#define ZK_NTT_PERSIST_R 9
#define ZK_NTT_PERSIST_LOG_T 4
template <bool DIT>
__global__ void __launch_bounds__(1024) ntt_strided_persist_kernel(NttPass p, u32 n_cols, u32 n_items) {
    extern __shared__ __attribute__((aligned(16))) u64 tile[];
    u64 *const buf_a = tile, *const buf_b = tile + (1u << (ZK_NTT_PERSIST_R + ZK_NTT_PERSIST_LOG_T));
    const u32 tid = threadIdx.x, u = tid & 15, w = tid >> 4;
    const int log_d = p.log_d;
    const int log_lo_tiles = log_d - ZK_NTT_PERSIST_LOG_T;
    const __amdgpu_buffer_rsrc_t twr = ntt_tw_rsrc(p.tw);
    // the three register steps: rows t0 + m * 2^lq, m < 8 (ntt_step with K = 3 and one sub-problem per lane)
    constexpr int lq1 = DIT ? 0 : 6, lq2 = 3, lq3 = DIT ? 6 : 0;
    const u32 r1 = ((w >> lq1) << (lq1 + 3)) + (w & ((1u << lq1) - 1));
    const u32 r2 = ((w >> lq2) << (lq2 + 3)) + (w & ((1u << lq2) - 1));
    const u32 r3 = ((w >> lq3) << (lq3 + 3)) + (w & ((1u << lq3) - 1));
    const u32 l1 = (r1 << 4) + u, l2 = (r2 << 4) + u, l3 = (r3 << 4) + u;             // LDS index of element 0 of each step
    const u32 g1 = (r1 << log_d) + u, g3 = (r3 << log_d) + u, g2 = (r2 << log_d) + u;  // the same, relative to the tile base
    const u32 G = gridDim.x;

    // item -> (column, tile base); the items of a workgroup are it, it + G, it + 2 G, ... (columns fastest: the workgroups that
    // run side by side work on the same tile index of different columns and share its twiddles in the L2).  Items past the end
    // are clamped to the workgroup's first one: their loads are issued (one path through the loop) and never used.
    auto item_base = [&](u32 item, u32 &c) {
        if (item >= n_items) item = blockIdx.x;
        c = item % n_cols;
        const u32 tl = item / n_cols;
        return ((tl >> log_lo_tiles) << (log_d + ZK_NTT_PERSIST_R)) + ((tl & ((1u << log_lo_tiles) - 1)) << ZK_NTT_PERSIST_LOG_T);
    };
    auto fetch = [&](u64 (&dst)[8], u32 c, u32 b) {
        const char *s = reinterpret_cast<const char *>(p.src + (size_t)c * p.src_stride);
#pragma unroll
        for (int m = 0; m < 8; ++m) dst[m] = ZK_NTT_DBG(8) ? (u64)(b + m) : *reinterpret_cast<const u64 *>(s + ((b + g1 + ((u32)m << (lq1 + log_d))) << 3));
    };
    auto exchange = [&](u64 (&v)[8], u64 *buf, u32 lw, int lqw, u32 lr, int lqr) {
#pragma unroll
        for (int m = 0; m < 8; ++m) buf[lw + ((u32)m << (lqw + 4))] = v[m];
        if (!ZK_NTT_DBG(32)) __syncthreads();
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = buf[lr + ((u32)m << (lqr + 4))];
    };
    auto store = [&](const u64 (&v)[8], u32 c, u32 b) {
        char *const d = reinterpret_cast<char *>(p.dst + (size_t)c * p.dst_stride);
        if (!ZK_NTT_DBG(4)) {
#pragma unroll
            for (int m = 0; m < 8; ++m)
                *reinterpret_cast<u64 *>(d + ((b + g3 + ((u32)m << (lq3 + log_d))) << 3)) = p.last_pass ? gl_canon(v[m]) : v[m];
        } else if (v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7] == 0x123456789abcdefull) *reinterpret_cast<u64 *>(d) = 1;
    };

    u32 it = blockIdx.x;
    if (it >= n_items) return;
    u32 col0, base0, col1, base1, col2, base2;             // the tile being transformed, the next one, the one after
    u64 nxa[8], nxb[8];                                    // tile data in flight: consumed alternately, reloaded two tiles ahead
    u64 t1a[7], t2a[7], t3a[7], t1b[7], t2b[7], t3b[7];    // twiddles of the tile body a / body b works on

    // The pipelined bodies come as  a (b a)*  -- an odd number: with an even count the first tile is done here, unpipelined.
    if (!((((n_items - it) + G - 1) / G) & 1)) {
        base0 = item_base(it, col0);
        fetch(nxa, col0, base0);
        ntt_tw7<DIT>(twr, p, lq1, base0 + g1, t1a);
        ntt_tw7<DIT>(twr, p, lq2, base0 + g2, t2a);
        ntt_tw7<DIT>(twr, p, lq3, base0 + g3, t3a);
        ntt_radix8<DIT>(nxa, t1a);
        exchange(nxa, buf_a, l1, lq1, l2, lq2);
        ntt_radix8<DIT>(nxa, t2a);
        exchange(nxa, buf_b, l2, lq2, l3, lq3);
        ntt_radix8<DIT>(nxa, t3a);
        store(nxa, col0, base0);
        it += G;
        __syncthreads();                                   // (buf_a / buf_b are free again)
    }
    base0 = item_base(it, col0);
    base1 = item_base(it + G, col1);
    fetch(nxa, col0, base0);
    fetch(nxb, col1, base1);
    ntt_tw7<DIT>(twr, p, lq1, base0 + g1, t1a);
    ntt_tw7<DIT>(twr, p, lq2, base0 + g2, t2a);
    ntt_tw7<DIT>(twr, p, lq3, base0 + g3, t3a);

    // One tile.  A wave's vector-memory operations complete IN ORDER (one counter, loads and stores alike), so whatever is
    // needed soon must not have been issued behind something slow.  Order of issue per body, and when each is needed:
    //     step 1                                   (its twiddles: issued after step 1 of the previous body)
    //     T: step-1 twiddles of the next tile           -> step 1 of the next body ... fast (L2), and ahead of:
    //     N: the tile two ahead, into the buffer just consumed -> top of the body after the next ... slow (HBM)
    //     steps 2, 3                               (their twiddles: issued at the end of the previous body, ahead of N)
    //     A: step-2 / step-3 twiddles of the NEXT tile  -> steps 2, 3 of the next body ... fast, and ahead of:
    //     S: this tile's stores                    -> nothing waits for them before T of the next body is needed, a body later
#ifdef ZK_NTT_DEBUG
    u64 tr[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = ntt_dbg_now();
#endif
    auto body = [&](u64 (&nx)[8], u64 (&t1)[7], u64 (&t2)[7], u64 (&t3)[7], u64 (&t1n)[7], u64 (&t2n)[7], u64 (&t3n)[7], auto tag) {
        // (the copies of the body must not be merged back into one with the buffers rotated by register copies -- copying a
        // register with a load in flight waits for it: distinct asm comments keep them apart)
        if (decltype(tag)::value == 0) asm volatile("; persist body a"); else if (decltype(tag)::value == 1) asm volatile("; persist body b"); else asm volatile("; persist body a'");
        u64 v[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = nx[m];
        base2 = item_base(it + 2 * G, col2);
        ZK_NTT_MARK(0);
        if (!ZK_NTT_DBG(2)) ntt_radix8<DIT>(v, t1);
        __builtin_amdgcn_sched_barrier(0);
        ZK_NTT_MARK(1);                                    // step 1 and its waits (the tile, its twiddles)
        ntt_tw7<DIT>(twr, p, lq1, base1 + g1, t1n);
        __builtin_amdgcn_sched_barrier(0);
        fetch(nx, col2, base2);
        __builtin_amdgcn_sched_barrier(0);
        ZK_NTT_MARK(2);                                    // issue of T and N
        exchange(v, buf_a, l1, lq1, l2, lq2);
        ZK_NTT_MARK(3);                                    // first exchange (write, barrier, read)
        if (!ZK_NTT_DBG(2)) ntt_radix8<DIT>(v, t2);
        else v[0] += t2[0] + t2[1] + t2[2] + t2[3] + t2[4] + t2[5] + t2[6] + t1[0] + t1[1] + t1[2] + t1[3] + t1[4] + t1[5] + t1[6];
        __builtin_amdgcn_sched_barrier(0);
        ZK_NTT_MARK(4);                                    // step 2 and its wait (twiddles; behind them in order: the previous N)
        exchange(v, buf_b, l2, lq2, l3, lq3);
        ZK_NTT_MARK(5);
        if (!ZK_NTT_DBG(2)) ntt_radix8<DIT>(v, t3);
        else v[0] += t3[0] + t3[1] + t3[2] + t3[3] + t3[4] + t3[5] + t3[6];
        __builtin_amdgcn_sched_barrier(0);
        ZK_NTT_MARK(6);                                    // step 3
        ntt_tw7<DIT>(twr, p, lq2, base1 + g2, t2n);
        ntt_tw7<DIT>(twr, p, lq3, base1 + g3, t3n);
        __builtin_amdgcn_sched_barrier(0);
        store(v, col0, base0);
        __builtin_amdgcn_sched_barrier(0);
        ZK_NTT_MARK(7);                                    // issue of A and S
#ifdef ZK_NTT_DEBUG
        tr[9] += 1;
#endif
        col0 = col1; base0 = base1; col1 = col2; base1 = base2;
        it += G;
    };
    // a (b a)*: the loop header joins two copies of body a, whose pending memory operations are the same -- a header joining
    // the prologue and a body takes the worst case of both wait counts and waits for the prefetch
    body(nxa, t1a, t2a, t3a, t1b, t2b, t3b, std::integral_constant<int, 2>());
    while (it < n_items) {
        body(nxb, t1b, t2b, t3b, t1a, t2a, t3a, std::integral_constant<int, 1>());
        body(nxa, t1a, t2a, t3a, t1b, t2b, t3b, std::integral_constant<int, 0>());
    }
#ifdef ZK_NTT_DEBUG
    if ((p.nt & 64) && blockIdx.x == 37 && tid == 64 * 5) for (int k = 0; k < 10; ++k) zk_ntt_trace[k] = tr[k];
#endif
}

#include "ntt_swap.cuh"      // r05: the passes on gfx950's lane-swap instructions (v_permlane16/32_swap)

// The LAST values -> coefficients pass and the FIRST coefficients -> values pass of a commitment work on the same tiles: the
// contiguous pass of the inverse transform leaves coefficients [k 2^c, (k + 1) 2^c) of a column (bit-reversed order) in tile k,
// and the contiguous pass of the low-degree extension reads exactly those to produce values [k 2^(c + rate), (k + 1) 2^(c + rate))
// of the 2^rate times longer transform.  Fused, the coefficients are written once (they are kept for the openings) and never
// read back: 80 instead of 88 bytes per trace element of `from_values` cross HBM, one launch and one load phase fewer.
//   pd: the inverse transform's contiguous pass (log_d = 0, r = c; dst = the coefficient array, out_const = 1 / n)
//   pt: the extension's contiguous pass (log_d = 0, r = c + rate, first_stage = log_rep = rate; in_scale = coset powers in
//       coefficient order; dst = the LDE array; last_pass when no strided pass follows)
// Same arithmetic in the same order as the two separate passes: bit-identical coefficients and values.
#define ZK_NTT_FUSED_MAX_PER_THREAD 8
template <bool DIT, bool PAD>
__device__ __forceinline__ void ntt_tile_stages(u64 *tile, const NttPass &p, u32 base, u32 elems, u32 tid, u32 nthr) {
    int done = p.first_stage;
    const int r = p.r;
    while (done < r) {
        const int k = r - done < 3 ? r - done : 3;
        const int log_q = DIT ? done : (r - done - k);
        if (k == 3) ntt_step<DIT, 3, PAD>(tile, p, log_q, base, elems, tid, nthr);
        else if (k == 2) ntt_step<DIT, 2, PAD>(tile, p, log_q, base, elems, tid, nthr);
        else ntt_step<DIT, 1, PAD>(tile, p, log_q, base, elems, tid, nthr);
        __syncthreads();
        done += k;
    }
}

static __global__ void __launch_bounds__(1024) ntt_fused_kernel(NttPass pd, NttPass pt) {
    extern __shared__ __attribute__((aligned(16))) u64 tile[];
    const u32 tid = threadIdx.x, nthr = blockDim.x;
    const u32 tile_id = pd.cols_fastest ? blockIdx.y : blockIdx.x;
    const u32 col_id = pd.cols_fastest ? blockIdx.x : blockIdx.y;
    const int c = pd.r, rate = pt.log_rep;
    const u32 elems_c = 1u << c, elems_v = elems_c << rate;
    const u32 base_c = tile_id << c, base_v = base_c << rate;
    const u64 *src = pd.src + (size_t)col_id * pd.src_stride;
    u64 *coeffs = pd.dst + (size_t)col_id * pd.dst_stride;
    u64 *dst = pt.dst + (size_t)col_id * pt.dst_stride;

    for (u32 e0 = tid; e0 < elems_c; e0 += nthr * ZK_NTT_LOADS_IN_FLIGHT) {        // all of a lane's loads in flight at once
        u64 v[ZK_NTT_LOADS_IN_FLIGHT];
#pragma unroll
        for (int k = 0; k < ZK_NTT_LOADS_IN_FLIGHT; ++k) { const u32 e = e0 + (u32)k * nthr; if (e < elems_c) v[k] = src[base_c + e]; }
#pragma unroll
        for (int k = 0; k < ZK_NTT_LOADS_IN_FLIGHT; ++k) { const u32 e = e0 + (u32)k * nthr; if (e < elems_c) tile[e] = v[k]; }
    }
    __syncthreads();
    ntt_tile_stages<false, false>(tile, pd, base_c, elems_c, tid, nthr);      // values -> coefficients, stages c-1 .. 0

    // coefficients leave (canonical, scaled by 1 / n); their coset-scaled copies stay in registers until every lane has read
    // its own, then go back into the tile as the 2^rate replicas the skipped stages would have produced
    u64 keep[ZK_NTT_FUSED_MAX_PER_THREAD];
#pragma unroll
    for (int k = 0; k < ZK_NTT_FUSED_MAX_PER_THREAD; ++k) {
        const u32 e = tid + (u32)k * nthr;
        if (e < elems_c) keep[k] = pt.in_scale ? pt.in_scale[base_c + e] : 1;      // the coset factors: loads in flight first
    }
#pragma unroll
    for (int k = 0; k < ZK_NTT_FUSED_MAX_PER_THREAD; ++k) {
        const u32 e = tid + (u32)k * nthr;
        if (e < elems_c) {
            const u64 cf = gl_mul_canon(tile[e], pd.out_const);
            coeffs[base_c + e] = cf;
            keep[k] = pt.in_scale ? gl_mul(cf, keep[k]) : cf;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ZK_NTT_FUSED_MAX_PER_THREAD; ++k) {
        const u32 e = tid + (u32)k * nthr;
        if (e < elems_c)
            for (u32 j = 0; j < (1u << rate); ++j) tile[(e << rate) + j] = keep[k];
    }
    __syncthreads();
    ntt_tile_stages<true, false>(tile, pt, base_v, elems_v, tid, nthr);      // coefficients -> values, stages rate .. c+rate-1

    for (u32 e0 = tid; e0 < elems_v; e0 += nthr * ZK_NTT_LOADS_IN_FLIGHT) {
        u64 v[ZK_NTT_LOADS_IN_FLIGHT];
#pragma unroll
        for (int k = 0; k < ZK_NTT_LOADS_IN_FLIGHT; ++k) { const u32 e = e0 + (u32)k * nthr; if (e < elems_v) v[k] = tile[e]; }
#pragma unroll
        for (int k = 0; k < ZK_NTT_LOADS_IN_FLIGHT; ++k) {
            const u32 e = e0 + (u32)k * nthr;
            if (e < elems_v) dst[base_v + e] = pt.last_pass ? gl_canon(v[k]) : v[k];
        }
    }
}

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
