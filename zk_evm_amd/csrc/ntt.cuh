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

struct NttPass {
    const u64 *src;      // column c at src + c*src_stride
    u64 *dst;            // column c at dst + c*dst_stride
    size_t src_stride, dst_stride;
    const u64 *tw;       // DIT: level layout tw[D - 1 + k] = (root of order 2D)^k, k < D, for D = 1 .. 2^(log_tw-1);
                         // values -> coeffs: block-order levels tw[2^s - 1 + j] = (root of order 2^(s+1))^bitrev_s(j) (ntt_host.inc)
    const u64 *in_scale; // optional per-source-index factor applied on load (coset powers)
    const u64 *out_scale;// optional per-index factor applied on store
    u64 out_const;       // constant factor applied on store when apply_out_const
    int log_tw;
    int log_n;           // transform size of the DESTINATION array
    int log_d;           // smallest butterfly distance handled by this pass
    int r;               // tile rows = 2^r (stages first_stage .. r-1 are executed)
    int log_t;           // tile cols = 2^log_t contiguous elements (<= d)
    int first_stage;     // DIT: number of leading stages already satisfied by replication
    int log_rep;         // load: dst index x reads src index x >> log_rep
    int apply_out_const;
    int last_pass;       // the values leave the transform: store canonical representatives
    int nt;              // stream the tile data with non-temporal loads / stores (the twiddle levels keep the L2)
    int cols_fastest;    // grid = (columns, tiles): consecutive workgroups run the SAME tile of different columns (ntt_host.inc)
    // Phase stagger (ntt_host.inc kNttStagger*): the workgroups that share a CU start together and would stay in lock step --
    // all loading, then all in their butterflies, then all storing -- so the memory phases (HBM at 5 TB/s, VALU idle) and the
    // compute phases (HBM idle) never overlap.  The first-resident workgroups selected by stagger_mode wait stagger_ticks
    // (100 MHz wall clock) before their load; every later workgroup inherits the phase of the slot it takes over.
    u32 stagger_ticks, stagger_mode, stagger_blocks;
};

// One radix-2 butterfly (a, b) -> (a + w b, a - w b).  BOTH directions use this Cooley-Tukey form: the canonical product
// (gl_mul_canon) lets the add and the sub run with one correction each, 11 + 4 + 4 full-rate instructions (gl.cuh)
// against 6 + 8 + 10 for the Gentleman-Sande form (a + b, (a - b) w), whose add and sub both see two lazy operands.
__device__ __forceinline__ void ntt_bfly(u64 &a, u64 &b, u64 w) {
    u64 t = gl_mul_canon(b, w);
    u64 na = gl_add_canon(a, t);
    b = gl_sub_canon(a, t);
    a = na;
}

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

// Twiddle loads as BUFFER loads: the table's base sits in a resource descriptor (four SGPRs, built once per kernel), the
// wave-uniform part of the index in the instruction's scalar offset and the lane's part in one 32-bit VGPR -- no 64-bit
// address arithmetic on the vector unit (a v_lshl_add_u64 per twiddle with global loads: 8 of a radix-8 step's 343 VALU
// instructions, all at the slow rate), and the compiler issues a step's seven loads together, ahead of the LDS reads.  Raw
// buffer, no range check in practice (num_records = 2^32 - 1; launch_pass rejects tables above 2^28 entries = 2 GiB).
// r03t, tools/kbench 116 x 2^20: values -> coefficients 1.39 -> 1.26 ms, coefficients -> values 2.64 -> 2.64 (that direction
// is not bound by its instruction count).  The same treatment of the tile loads / stores (buffer accesses for the column
// data as well) measured SLOWER in combination (1.66 ms), so those stay global accesses.
#if defined(__HIP_DEVICE_COMPILE__)
typedef u32 ntt_v2u32 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ntt_tw_rsrc(const u64 *tw) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<u64 *>(tw), 0, -1, 0x00020000);
}
__device__ __forceinline__ u64 ntt_tw_load(__amdgpu_buffer_rsrc_t r, u32 lane_byte_off, u32 uniform_index) {
    const ntt_v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(r, lane_byte_off, uniform_index * 8, 0);
    return ((u64)v.y << 32) | v.x;
}
#else
typedef const u64 *__amdgpu_buffer_rsrc_t_host;
#define __amdgpu_buffer_rsrc_t __amdgpu_buffer_rsrc_t_host
__device__ inline __amdgpu_buffer_rsrc_t ntt_tw_rsrc(const u64 *tw) { return tw; }          // (host pass: parsed, never run)
__device__ inline u64 ntt_tw_load(__amdgpu_buffer_rsrc_t r, u32 lane_byte_off, u32 uniform_index) { return r[uniform_index + lane_byte_off / 8]; }
#endif

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
#pragma unroll
        for (int m = 0; m < (1 << K); ++m)
            v[m] = PAD ? tile[ntt_ph<PAD>(((t0 + m * q) << log_t) + u)] : *reinterpret_cast<u64 *>(tile_b + (a0 * 8 + m * stride8));
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
            if (PAD) tile[ntt_ph<PAD>(((t0 + m * q) << log_t) + u)] = v[m];
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
                if (e < elems) tile[ntt_ph<PAD>(e)] = p.in_scale ? gl_mul(v[k], sc[k]) : v[k];
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
                v[k] = tile[ntt_ph<PAD>(e)];
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
