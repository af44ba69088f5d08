// NTT passes on gfx950's lane-swap instructions (r05).  Included by ntt.cuh (NttPass, ntt_bfly, ntt_tw_load live there).
//
// ntt_pass_kernel moves every tile element through LDS four or five times (load -> LDS, one read + write per radix-8 register
// step, LDS -> store) behind five or six workgroup barriers, and r04q's in-kernel timeline has those exchanges at ~4 us of an
// 11.5 us tile with nobody computing meanwhile.  CDNA4 added v_permlane16_swap / v_permlane32_swap: they exchange, between two
// registers, the halves of a wave selected by lane bit 4 / lane bit 5 -- a 2 x 2 transpose between a REGISTER index bit and a LANE
// index bit at one VALU instruction per 32-bit register.  A butterfly stage whose pairs sit in two lanes becomes a stage whose
// pairs sit in two registers of one lane.  With sixteen elements per lane, six index bits -- four register bits and lane bits
// 4, 5 -- are therefore reachable without LDS, in any order, provided the lanes' low four bits carry sixteen CONTIGUOUS elements
// (a 128-byte segment per row: exactly the tile shape the strided pass loads anyway).
//
//   strided pass, 2^R rows x 16 contiguous elements, R = 7 .. 10  (ntt_strided_swap_kernel)
//       row bits = 4 register bits + 2 lane bits + (R - 6) wave bits.  Six stages run out of registers straight after the global
//       loads (four 128-byte row segments per wave and load instruction), ONE exchange through LDS brings the wave bits into
//       registers, the last R - 6 stages store straight from registers: one LDS round trip and one barrier per tile.
//   contiguous pass, 2^10 elements PER WAVE  (ntt_contig_wave_kernel_*)
//       the 2^10 elements are 64 rows of 16: six stages on the row bits as above, the four stages inside a 128-byte segment after
//       a WAVE-LOCAL transpose through LDS (lane l takes row l; rows padded to 17 words: conflict-free both ways), a second
//       transpose back for coalesced stores.  No workgroup barrier at all: the waves of a workgroup share nothing and drift
//       through load / butterflies / store out of phase, which is what lets a CU's memory and issue phases overlap.
//       coefficients -> values with rate_bits = 1 evaluates the tile's 2^10 coefficients on BOTH cosets (x = 2 i + b is the plain
//       2^10-point transform of c_j (g w_2n^b)^j): the same twiddles for b = 0, 1, only the scale table differs, and the two
//       results of an element leave as one 16-byte store.
// Of a group's six stages the four on register bits need no exchange at all and the two on lane bits one swap each (32 swap
// instructions per 16 elements and group).
// Twiddles.  values -> coefficients: a stage's twiddle depends on the index bits ABOVE it; the four register stages come first and
// work on the group's top bits, so those are register bits too -- compile-time offsets from a wave-uniform base, i.e. SCALAR
// loads through the constant cache; the remaining stages take per-lane loads.  coefficients -> values: a stage's
// twiddle depends on the bits BELOW it and on the element's position in its segment -- per-lane loads, except the four
// in-segment stages of the contiguous kernel (positions are register indices there: scalar loads of the table's first 15 entries).
// Same butterflies on the same operands with the same twiddles as ntt_pass_kernel: bit-identical output
// (tests/test_gpu_commit.py; the index algebra alone: tests/test_ntt_swap_model.py, a Python restatement of this file).
#pragma once

#define ZK_NTT_SWAP_LOG_T 4
#define ZK_NTT_WAVE_BITS 10                 // elements per wave of the contiguous kernels
#define ZK_NTT_WAVE_LDS 1088                // 64 rows x 17 words

#if !defined(__HIP__)                       // (tests/emu/: a plain C++ compiler; the attribute exists for kernels of the HIP compiler only)
#define ZK_NTT_WAVES_PER_EU(lo, hi)
#else
#define ZK_NTT_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) u64 *ntt_const_u64p;         // constant address space: uniform loads are scalar loads
template <int LANEBIT>
__device__ __forceinline__ void ntt_lane_swap(u64 &a, u64 &b) {
    // (register a, lane bit = 1)  <->  (register b, lane bit = 0)
    const u32 alo = (u32)a, ahi = (u32)(a >> 32), blo = (u32)b, bhi = (u32)(b >> 32);
    if (LANEBIT == 4) {
        const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        a = ((u64)hi[0] << 32) | lo[0]; b = ((u64)hi[1] << 32) | lo[1];
    } else {
        const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        a = ((u64)hi[0] << 32) | lo[0]; b = ((u64)hi[1] << 32) | lo[1];
    }
}
// the wave's index in its workgroup IS wave-uniform; said so, everything derived from it (tile bases, twiddle offsets) lives in
// scalar registers and a buffer load's scalar offset needs no waterfall loop
__device__ __forceinline__ u32 ntt_uniform(u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ void ntt_wave_sync() {          // LDS traffic of ONE wave is in order: only the compiler must not reorder
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#elif defined(ZK_NTT_EMULATE)
// tests/emu/: this file compiled for the CPU, one OS thread per lane; the lanes of a wave meet in zk_emu_lane_swap / zk_emu_wave_sync
typedef const u64 *ntt_const_u64p;
void zk_emu_lane_swap(int lanebit, u64 &a, u64 &b);       // (register a, lane bit = 1) <-> (register b, lane bit = 0)
void zk_emu_wave_sync();
template <int LANEBIT> inline void ntt_lane_swap(u64 &a, u64 &b) { zk_emu_lane_swap(LANEBIT, a, b); }
inline u32 ntt_uniform(u32 x) { return x; }
inline void ntt_wave_sync() { zk_emu_wave_sync(); }
#else
typedef const u64 *ntt_const_u64p;
template <int LANEBIT> __device__ inline void ntt_lane_swap(u64 &, u64 &) {}                     // (host pass: parsed, never run)
__device__ inline u32 ntt_uniform(u32 x) { return x; }
__device__ inline void ntt_wave_sync() {}
#endif
// the butterfly with twiddle 1: gl_mul_canon(b, 1) is the canonical representative of b
__device__ __forceinline__ void ntt_bfly_one(u64 &a, u64 &b) {
    const u64 t = gl_canon(b);
    const u64 na = gl_add_canon(a, t);
    b = gl_sub_canon(a, t);
    a = na;
}
// (wave-uniform pointer)[per-lane 32-bit BYTE offset]: the form that becomes `global_load v, v_off, s[base]` -- no vector address
// arithmetic per access (column data is < 2^31 bytes per column: 2^28 points)
// (left to itself the compiler folds the uniform part into a per-lane 64-bit pointer and pays a v_lshl_add_u64 per access: the empty
// asm pins the uniform pointer in a scalar register pair -- as a GLOBAL-address-space pointer: a generic one behind an asm would
// become flat_load)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) char *ntt_gptr;
__device__ __forceinline__ u64 ntt_ld(const u64 *uniform, u32 lane_bytes) {
    ntt_gptr q = (ntt_gptr)uniform;
    asm("" : "+s"(q));
    return *reinterpret_cast<const __attribute__((address_space(1))) u64 *>(q + lane_bytes);
}
__device__ __forceinline__ void ntt_st(u64 *uniform, u32 lane_bytes, u64 x) {
    ntt_gptr q = (ntt_gptr)uniform;
    asm("" : "+s"(q));
    *reinterpret_cast<__attribute__((address_space(1))) u64 *>(q + lane_bytes) = x;
}
__device__ __forceinline__ void ntt_st2(u64 *uniform, u32 lane_bytes, u64 x, u64 y) {      // 16-byte aligned
    typedef unsigned long long ntt_u64x2 __attribute__((ext_vector_type(2)));
    ntt_gptr q = (ntt_gptr)uniform;
    asm("" : "+s"(q));
    ntt_u64x2 v = {x, y};
    *reinterpret_cast<__attribute__((address_space(1))) ntt_u64x2 *>(q + lane_bytes) = v;
}
#else
__device__ __forceinline__ u64 ntt_ld(const u64 *uniform, u32 lane_bytes) { return *reinterpret_cast<const u64 *>(reinterpret_cast<const char *>(uniform) + lane_bytes); }
__device__ __forceinline__ void ntt_st2(u64 *uniform, u32 lane_bytes, u64 x, u64 y) { u64 *q = reinterpret_cast<u64 *>(reinterpret_cast<char *>(uniform) + lane_bytes); q[0] = x; q[1] = y; }
__device__ __forceinline__ void ntt_st(u64 *uniform, u32 lane_bytes, u64 x) { *reinterpret_cast<u64 *>(reinterpret_cast<char *>(uniform) + lane_bytes) = x; }
#endif
// transpose register bit REGBIT with lane bit LANEBIT over all sixteen registers
template <int LANEBIT, int REGBIT>
__device__ __forceinline__ void ntt_swap16(u64 (&v)[16]) {
#pragma unroll
    for (int m = 0; m < 16; ++m)
        if (!(m & (1 << REGBIT))) ntt_lane_swap<LANEBIT>(v[m], v[m | (1 << REGBIT)]);
}
// one stage over the sixteen registers: pairs (m, m | 1 << BIT), twiddle w[sel(m)]
#define ZK_NTT_STAGE16N(BIT, W_OF_M)                                                        \
    _Pragma("unroll") for (int b = 0; b < NB; ++b)                                          \
    _Pragma("unroll") for (int m = 0; m < 16; ++m)                                          \
        if (!(m & (1 << (BIT)))) ntt_bfly(vv[b][m], vv[b][m | (1 << (BIT))], (W_OF_M));
#define ZK_NTT_STAGE16(BIT, W_OF_M)                                                         \
    _Pragma("unroll") for (int m = 0; m < 16; ++m)                                          \
        if (!(m & (1 << (BIT)))) ntt_bfly(v[m], v[m | (1 << (BIT))], (W_OF_M));

// ---- six stages, values -> coefficients (largest distance first) ------------------------------------------------------------
// Six index bits q5 .. q0 ("rows"); on entry register bits [3..0] = [q5 q4 q3 q2], lane bit 5 = q1, lane bit 4 = q0: the four
// register stages need no exchange at all, the two lane stages one swap each.  Stage j (j = 0: q5 .. j = 5: q0) reads the
// block-order table at first[j] + (the row bits above the stage's), first[j] = (2^(s_top + j) - 1) + (H << j): level s_top + j, H =
// the index of this six-bit group among its peers.
// On exit registers [3..0] = [q1 q0 q3 q2], lane bit 5 = q5, lane bit 4 = q4.
// STAGES < 6: only the first STAGES of them (ntt_strided_reg_kernel: the bits below are column bits); the layout then stays the
// entry layout (STAGES <= 4) or registers [3..0] = [q1 q4 q3 q2], lane 5 = q5, lane 4 = q0 (STAGES = 5).
template <int STAGES = 6>
__device__ __forceinline__ void ntt_swap_dif6(u64 (&v)[16], const u64 *tw, __amdgpu_buffer_rsrc_t twr, int s_top, u32 H, u32 l4, u32 l5) {
    const ntt_const_u64p ctw = (ntt_const_u64p)(unsigned long long)tw;
    auto first = [&](int j) { return ((1u << (s_top + j)) - 1) + (H << j); };
    // the per-lane twiddles of the last two stages are requested FIRST: a wave's vector loads return in order, and these have
    // the four scalar-twiddle stages to arrive in (issued where they are used, each stage would stall on its own loads)
    u64 w4[8], w5[8];
    if (STAGES > 4) {
        const u32 b4 = first(4);
#pragma unroll
        for (int i = 0; i < 8; ++i) w4[i] = ntt_tw_load(twr, l5 * 64, b4 + i);
    }
    if (STAGES > 5) {
        const u32 b5 = first(5);
#pragma unroll
        for (int i = 0; i < 8; ++i) w5[i] = ntt_tw_load(twr, (l5 * 16 + l4 * 8) * 8, b5 + i);
    }
    {   // q5 = register bit 3
        const u64 w = ctw[first(0)];
        ZK_NTT_STAGE16(3, w)
    }
    if (STAGES > 1) {   // q4 = register bit 2; twiddle by q5 = register bit 3
        const u32 b = first(1);
        const u64 w[2] = {ctw[b], ctw[b + 1]};
        ZK_NTT_STAGE16(2, w[m >> 3])
    }
    if (STAGES > 2) {   // q3 = register bit 1; twiddle by (q5 q4) = register bits (3 2)
        const u32 b = first(2);
        const u64 w[4] = {ctw[b], ctw[b + 1], ctw[b + 2], ctw[b + 3]};
        ZK_NTT_STAGE16(1, w[m >> 2])
    }
    if (STAGES > 3) {   // q2 = register bit 0; twiddle by (q5 q4 q3) = register bits (3 2 1)
        const u32 b = first(3);
        const u64 w[8] = {ctw[b], ctw[b + 1], ctw[b + 2], ctw[b + 3], ctw[b + 4], ctw[b + 5], ctw[b + 6], ctw[b + 7]};
        ZK_NTT_STAGE16(0, w[m >> 1])
    }
    // registers [q5 q4 q3 q2], lane 5 = q1, lane 4 = q0
    if (STAGES > 4) {   // q1 (lane 5) <-> register bit 3 (q5); twiddle by (q5 q4 q3 q2) = (lane 5, registers 2 1 0)
        ntt_swap16<5, 3>(v);
        ZK_NTT_STAGE16(3, w4[m & 7])
    }
    if (STAGES > 5) {   // q0 (lane 4) <-> register bit 2 (q4); twiddle by (q5 q4 q3 q2 q1) = (lane 5, lane 4, registers 1 0 3)
        ntt_swap16<4, 2>(v);
        ZK_NTT_STAGE16(2, w5[((m >> 1) & 1) * 4 + (m & 1) * 2 + (m >> 3)])
    }
}
// row (six bits) held by register m after ntt_swap_dif6, without the lane bits: [q3 q2 q1 q0] = registers [1 0 3 2]
__device__ __forceinline__ constexpr u32 ntt_swap_dif6_row(int m) { return ((m & 2) << 2) | ((m & 1) << 2) | ((m & 8) >> 2) | ((m & 4) >> 2); }

// ---- six stages, coefficients -> values (smallest distance first) -----------------------------------------------------------
// On entry register bits [3..0] = [q0 q1 q3 q2], lane bit 5 = q4, lane bit 4 = q5: four register stages, then one swap for each
// lane stage.  Stage k (pairs 2^k rows apart, k = 0 .. 5) reads the level table at (2^(log_d + k) - 1) + ((row mod 2^k) << log_d) +
// (position below the rows; lane part xl8 in bytes).
// On exit registers [3..0] = [q4 q5 q3 q2], lane bit 5 = q0, lane bit 4 = q1.
// NB arrays go through the same stages with the same twiddles (the two cosets of the contiguous kernel): each twiddle is loaded once.
// Every stage's twiddles are requested one stage AHEAD of their use (a wave's vector loads return in order: requested where
// they are used, each stage would start with a full cache round trip).
// STAGES < 6 (4 or 5): only the first STAGES of them (ntt_strided_reg_kernel: the bits above are column bits); the layout then
// stays the entry layout (STAGES = 4) or registers [3..0] = [q4 q1 q3 q2], lane 5 = q0, lane 4 = q5 (STAGES = 5).
template <int NB, int STAGES = 6>
__device__ __forceinline__ void ntt_swap_dit6(u64 (&vv)[NB][16], __amdgpu_buffer_rsrc_t twr, int log_d, u32 xl8, u32 l4, u32 l5) {
    static_assert(STAGES >= 4 && STAGES <= 6, "the four register stages, then one swap per lane stage");
    auto lvl = [&](int k) { return (1u << (log_d + k)) - 1; };
    const u64 w0 = ntt_tw_load(twr, xl8, lvl(0));
    const u64 w1[2] = {ntt_tw_load(twr, xl8, lvl(1)), ntt_tw_load(twr, xl8, lvl(1) + (1u << log_d))};
    {   // k = 0: q0 = register bit 3
        ZK_NTT_STAGE16N(3, w0)
    }
    u64 w2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w2[i] = ntt_tw_load(twr, xl8, lvl(2) + ((u32)i << log_d));
    {   // k = 1: q1 = register bit 2; twiddle by q0 = register bit 3
        ZK_NTT_STAGE16N(2, w1[m >> 3])
    }
    u64 w3[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w3[i] = ntt_tw_load(twr, xl8, lvl(3) + ((u32)i << log_d));
    {   // k = 2: register bit 0; twiddle by (q1 q0) = register bits (2 3)
        ZK_NTT_STAGE16N(0, w2[((m >> 2) & 1) * 2 + (m >> 3)])
    }
    u64 w4[8];
    if (STAGES > 4) {
        const u32 lo8 = xl8 + ((l5 << log_d) << 3);
#pragma unroll
        for (int i = 0; i < 8; ++i) w4[i] = ntt_tw_load(twr, lo8, lvl(4) + ((u32)(((i >> 1) & 1) * 8 + (i & 1) * 4 + ((i >> 2) & 1) * 2) << log_d));
    }
    {   // k = 3: register bit 1; twiddle by (q2 q1 q0) = register bits (0 2 3)
        ZK_NTT_STAGE16N(1, w3[(m & 1) * 4 + ((m >> 2) & 1) * 2 + (m >> 3)])
    }
    u64 w5[8];
    if (STAGES > 5) {
        const u32 lo8 = xl8 + (((l4 * 2 + l5) << log_d) << 3);
#pragma unroll
        for (int i = 0; i < 8; ++i)      // i = the butterfly's register bits (3 1 0)
            w5[i] = ntt_tw_load(twr, lo8, lvl(5) + ((u32)(((i >> 2) & 1) * 16 + ((i >> 1) & 1) * 8 + (i & 1) * 4) << log_d));
    }
    if (STAGES > 4) {   // k = 4: q4 (lane 5) <-> register bit 3 (q0); twiddle by (q3 q2 q1 q0) = (registers 1 0 2, lane 5)
        _Pragma("unroll") for (int b = 0; b < NB; ++b) ntt_swap16<5, 3>(vv[b]);
        ZK_NTT_STAGE16N(3, w4[m & 7])
    }
    if (STAGES > 5) {   // k = 5: q5 (lane 4) <-> register bit 2 (q1); twiddle by (q4 q3 q2 q1 q0) = (registers 3 1 0, lane 4, lane 5)
        _Pragma("unroll") for (int b = 0; b < NB; ++b) ntt_swap16<4, 2>(vv[b]);
        ZK_NTT_STAGE16N(2, w5[(m >> 3) * 4 + (m & 3)])
    }
}
// row bits [q3 q2 q1 q0] held by register m on ENTRY to ntt_swap_dit6 (registers [3..0] = [q0 q1 q3 q2]), without the lane bits
__device__ __forceinline__ constexpr u32 ntt_swap_dit6_row_in(int m) { return ((m & 2) << 2) | ((m & 1) << 2) | ((m & 4) >> 1) | ((m & 8) >> 3); }
// row bits [q5 q4 q3 q2] held by register m after ntt_swap_dit6 (registers [3..0] = [q4 q5 q3 q2]), without the lane bits
__device__ __forceinline__ constexpr u32 ntt_swap_dit6_row(int m) { return ((m & 4) << 3) | ((m & 8) << 1) | ((m & 2) << 2) | ((m & 1) << 2); }

// ---- the strided pass ------------------------------------------------------------------------------------------------------
// Geometry: 2^R rows (R = 7 .. 10: 2^(R - 6) waves, 2^(R + 7) bytes of LDS: 512 threads and 64 KiB at R = 9) x 16 contiguous elements; no load /
// store factors (a strided pass never has any: they belong to the contiguous pass at the coefficient end).
template <bool DIT, int R>
__global__ void __launch_bounds__(64 << (R - 6)) ntt_strided_swap_kernel(NttPass p) {
    static_assert(R >= 7 && R <= 10, "six register / lane stages + one to four wave stages");
    extern __shared__ __attribute__((aligned(16))) u64 tile[];
    const int log_d = p.log_d;
    const u32 tid = threadIdx.x, lane = tid & 63, wv = ntt_uniform(tid >> 6), u = lane & 15, l4 = (lane >> 4) & 1, l5 = lane >> 5;
    const int log_lo_tiles = log_d - ZK_NTT_SWAP_LOG_T;
    const u32 tile_id = p.cols_fastest ? blockIdx.y : blockIdx.x;
    const u32 col_id = p.cols_fastest ? blockIdx.x : blockIdx.y;
    const u32 hi_idx = tile_id >> log_lo_tiles, lo_tile = tile_id & ((1u << log_lo_tiles) - 1);
    const u32 base = (hi_idx << (log_d + R)) + (lo_tile << ZK_NTT_SWAP_LOG_T);
    const u64 *src = p.src + (size_t)col_id * p.src_stride;
    u64 *dst = p.dst + (size_t)col_id * p.dst_stride;
    const __amdgpu_buffer_rsrc_t twr = ntt_tw_rsrc(p.tw);
    constexpr int A = R - 6;                 // wave bits = the row bits below the six of the register / lane phase
    u64 vv[1][16];
    u64 (&v)[16] = vv[0];

    if (!DIT) {
        // rows t = [t(R-1) .. t0]: register bits [3..0] = t(R-1) .. t(R-4), lane 5 = t(R-5), lane 4 = t(R-6), wave = t(A-1) .. t0
        {
            // (uniform pointer + uniform offset)[per-lane 32-bit index]: the address arithmetic stays on the scalar unit
            const u64 *sb = src + base + ((size_t)wv << log_d);
            const u32 lo8 = (u + (((l5 << (R - 5)) | (l4 << (R - 6))) << log_d)) * 8;
#pragma unroll
            for (int m = 0; m < 16; ++m) v[m] = ntt_ld(sb + ((size_t)m << (R - 4 + log_d)), lo8);
        }
        // stage k (pairs 2^k rows apart): level s_k = log_n - 1 - log_d - k, block (hi_idx << (R - 1 - k)) + (t >> (k + 1))
        const int s_top = p.log_n - log_d - R;
        ntt_swap_dif6(v, p.tw, twr, s_top, hi_idx, l4, l5);
        // the twiddles of the stages after the exchange (by t >> (k + 1), t = (wave, lane 5, lane 4, register) there): requested
        // now, so that they arrive during the exchange
        auto lvl = [&](int k) { return ((1u << (s_top + R - 1 - k)) - 1) + (hi_idx << (R - 1 - k)); };
        const u32 thl = l5 * 2 + l4, thu = wv << 2;                    // t >> 4 = thu + thl: wave-uniform but for the two lane bits
        u64 bw3 = 0, bw2[2] = {0, 0}, bw1[4] = {0, 0, 0, 0}, bw0[8];
        if (A >= 4) bw3 = ntt_tw_load(twr, thl * 8, lvl(3) + thu);
        if (A >= 3) { bw2[0] = ntt_tw_load(twr, thl * 16, lvl(2) + thu * 2); bw2[1] = ntt_tw_load(twr, thl * 16, lvl(2) + thu * 2 + 1); }
        if (A >= 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) bw1[i] = ntt_tw_load(twr, thl * 32, lvl(1) + thu * 4 + i);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) bw0[i] = ntt_tw_load(twr, thl * 64, lvl(0) + thu * 8 + i);
        // LDS row of t: t with bit 0 flipped by t4 ^ t(R-2) (a half wave writes two rows that differ in t(R-2) and reads two
        // that differ in t4: 128-byte rows, 64 banks)
        {
            const u32 tb = (l5 << (R - 1)) | (l4 << (R - 2)) | wv;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                u32 t = tb | (ntt_swap_dif6_row(m) << A);
                t ^= ((t >> 4) ^ (t >> (R - 2))) & 1;
                tile[(t << 4) + u] = v[m];
            }
        }
        __syncthreads();
        // wave = t(R-1) .. t6, lane 5 = t5, lane 4 = t4, registers = [t3 t2 t1 t0]
        const u32 tb = (wv << 6) | (l5 << 5) | (l4 << 4);
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            u32 t = tb | m;
            t ^= ((t >> 4) ^ (t >> (R - 2))) & 1;
            v[m] = tile[(t << 4) + u];
        }
        // the A = R - 6 stages of the wave bits: k = A - 1 .. 0 (twiddles bw*: requested before the exchange, above)
        if (A >= 4) { ZK_NTT_STAGE16(3, bw3) }
        if (A >= 3) { ZK_NTT_STAGE16(2, bw2[m >> 3]) }
        if (A >= 2) { ZK_NTT_STAGE16(1, bw1[m >> 2]) }
        { ZK_NTT_STAGE16(0, bw0[m >> 1]) }
        {
            u64 *db = dst + base + ((size_t)(wv << 6) << log_d);
            const u32 so8 = (u + (((l5 << 5) | (l4 << 4)) << log_d)) * 8;
#pragma unroll
            for (int m = 0; m < 16; ++m) ntt_st(db + ((size_t)m << log_d), so8, p.last_pass ? gl_canon(v[m]) : v[m]);
        }
    } else {
        // register bits [3..0] = [t0 t1 t3 t2], lane bit 5 = t4, lane bit 4 = t5, wave = t(R-1) .. t6
        {
            const u64 *sb = src + base + ((size_t)(wv << 6) << log_d);
            const u32 ld8 = (u + (((l4 << 5) | (l5 << 4)) << log_d)) * 8;
#pragma unroll
            for (int m = 0; m < 16; ++m) v[m] = ntt_ld(sb + ((size_t)ntt_swap_dit6_row_in(m) << log_d), ld8);
        }
        const u32 xl8 = ((lo_tile << ZK_NTT_SWAP_LOG_T) + u) * 8;
        ntt_swap_dit6<1, 6>(vv, twr, log_d, xl8, l4, l5);
        // the twiddles of the stages after the exchange (t = (m << (R - 4)) | tb2 there; stage k: t mod 2^k = ((m mod 2^(k - R + 4)) <<
        // (R - 4)) | tb2): requested now, so that they arrive during the exchange
        const u32 tb2 = (wv << 2) | (l5 << 1) | l4;
        auto lvl = [&](int k) { return (1u << (log_d + k)) - 1; };
        const u32 lo8 = xl8 + ((tb2 << log_d) << 3);
        u64 bw0 = 0, bw1[2] = {0, 0}, bw2[4] = {0, 0, 0, 0}, bw3[8];
        if (R - 4 >= 6) bw0 = ntt_tw_load(twr, lo8, lvl(R - 4));
        if (R - 3 >= 6) { bw1[0] = ntt_tw_load(twr, lo8, lvl(R - 3)); bw1[1] = ntt_tw_load(twr, lo8, lvl(R - 3) + ((1u << (R - 4)) << log_d)); }
        if (R - 2 >= 6) {
#pragma unroll
            for (int i = 0; i < 4; ++i) bw2[i] = ntt_tw_load(twr, lo8, lvl(R - 2) + (((u32)i << (R - 4)) << log_d));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) bw3[i] = ntt_tw_load(twr, lo8, lvl(R - 1) + (((u32)i << (R - 4)) << log_d));
        // lane 5 = t0, lane 4 = t1.  LDS row of t: bit 0 flipped by t1
        {
            const u32 tb = ((wv << 6) | (l4 << 1) | l5) ^ l4;
#pragma unroll
            for (int m = 0; m < 16; ++m) tile[((tb | ntt_swap_dit6_row(m)) << 4) + u] = v[m];
        }
        __syncthreads();
        // registers [3..0] = the top four row bits, wave / lane 5 / lane 4 = the R - 4 below: t = (m << (R - 4)) | tb
        const u32 tb = (wv << 2) | (l5 << 1) | l4;
        {
            const u32 tp = tb ^ l5;
#pragma unroll
            for (int m = 0; m < 16; ++m) v[m] = tile[(((u32)m << (R - 4)) | tp) * 16 + u];
        }
        // register bit j is row bit R - 4 + j: a stage of this phase where that is >= 6 (twiddles bw*: requested before the exchange)
        if (R - 4 >= 6) { ZK_NTT_STAGE16(0, bw0) }
        if (R - 3 >= 6) { ZK_NTT_STAGE16(1, bw1[m & 1]) }
        if (R - 2 >= 6) { ZK_NTT_STAGE16(2, bw2[m & 3]) }
        { ZK_NTT_STAGE16(3, bw3[m & 7]) }
        {
            u64 *db = dst + base + ((size_t)(wv << 2) << log_d);
            const u32 so8 = (u + (((l5 << 1) | l4) << log_d)) * 8;
#pragma unroll
            for (int m = 0; m < 16; ++m) ntt_st(db + ((size_t)m << (R - 4 + log_d)), so8, p.last_pass ? gl_canon(v[m]) : v[m]);
        }
    }
}

// ---- the strided pass with few row bits: one WAVE per tile, no LDS ------------------------------------------------------------------
// R <= 6 row bits: a wave's 2^10 elements are 2^R rows x 2^(10 - R) CONTIGUOUS columns (R = 4: the sixteen rows of a column in the
// lane's sixteen registers, the 64 lanes 512 contiguous bytes of a row), i.e. the six-bit group of ntt_swap_dif6 / _dit6 with
// 6 - R of its bits being column bits: below the rows for values -> coefficients (the group's LAST stages fall away, and with them
// the swaps: R <= 4 needs none), above them for coefficients -> values (again the last stages).  Loads and stores straight from /
// to global memory in the group's entry / exit layout: no LDS, no barrier, four independent waves per workgroup.
// values -> coefficients: R = 1 .. 6 (every twiddle scalar up to R = 4); coefficients -> values: R = 4 .. 6 (below that column bits
// would sit in registers and a lane's twiddle position would differ per register).  Needs log_d >= 10 - R.
template <bool DIT, int R>
__global__ void __launch_bounds__(256) ntt_strided_reg_kernel(NttPass p) {
    static_assert(R >= 1 && R <= 6 && (!DIT || R >= 4), "see above");
    constexpr int C = 6 - R;                       // column bits inside the six-bit group
    const int log_d = p.log_d;
    const u32 tid = threadIdx.x, lane = tid & 63, wv = ntt_uniform(tid >> 6), u = lane & 15, l4 = (lane >> 4) & 1, l5 = lane >> 5;
    const u32 tile_id = (p.cols_fastest ? blockIdx.y : blockIdx.x) * 4 + wv;
    const u32 col_id = p.cols_fastest ? blockIdx.x : blockIdx.y;
    if (((size_t)tile_id << ZK_NTT_WAVE_BITS) >> p.log_n) return;                 // (whole waves leave: nothing is shared)
    const int log_blocks = log_d - 4 - C;          // wave tiles side by side in a row of d elements
    const u32 hi_idx = tile_id >> log_blocks, lo_block = tile_id & ((1u << log_blocks) - 1);
    const u32 base = (hi_idx << (log_d + R)) + (lo_block << (4 + C));
    const u64 *src = p.src + (size_t)col_id * p.src_stride + base;
    u64 *dst = p.dst + (size_t)col_id * p.dst_stride + base;
    const __amdgpu_buffer_rsrc_t twr = ntt_tw_rsrc(p.tw);
    u64 vv[1][16];
    u64 (&v)[16] = vv[0];
    // element of the group's index q (six bits) and lane column u: values -> coefficients q = [row | column bits], the other way
    // q = [column bits | row]
    auto offset = [&](u32 q) {
        const u32 row = DIT ? (q & ((1u << R) - 1)) : (q >> C), ch = DIT ? (q >> R) : (q & ((1u << C) - 1));
        return (row << log_d) + (ch << 4);
    };
    // A group index is (bits that come from the register number m) | (bits that come from the lane), on disjoint positions, and
    // offset() maps every bit of q to its own address bit: offset(qm | ql) = offset(qm) + offset(ql) -- a wave-uniform pointer plus one
    // 32-bit lane offset per access (ntt_ld / ntt_st), no 64-bit address arithmetic per register.
    auto ld = [&](u32 qm, u32 lane_off8) { return ntt_ld(src + offset(qm), lane_off8); };
    auto st = [&](u32 qm, u32 lane_off8, u64 x) { ntt_st(dst + offset(qm), lane_off8, p.last_pass ? gl_canon(x) : x); };
    if constexpr (!DIT) {
        // entry: registers [3..0] = [q5 q4 q3 q2], lane 5 = q1, lane 4 = q0
        const u32 ql = (l5 << 1) | l4;
        const u32 in8 = (offset(ql) + u) * 8;
#pragma unroll
        for (int m = 0; m < 16; ++m) v[m] = ld((u32)m << 2, in8);
        ntt_swap_dif6<R>(v, p.tw, twr, p.log_n - log_d - R, hi_idx, l4, l5);
        // exit: R <= 4 as on entry; R = 5: registers [q1 q4 q3 q2], lane 5 = q5, lane 4 = q0; R = 6: ntt_swap_dif6_row, lanes = q5 q4
        const u32 qlo = R <= 4 ? ql : R == 5 ? ((l5 << 5) | l4) : ((l5 << 5) | (l4 << 4));
        const u32 out8 = (offset(qlo) + u) * 8;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const u32 qm = R <= 4 ? ((u32)m << 2) : R == 5 ? (((m & 4) << 2) | ((m & 2) << 2) | ((m & 1) << 2) | ((m & 8) >> 2)) : ntt_swap_dif6_row(m);
            st(qm, out8, v[m]);
        }
    } else {
        // entry: registers [3..0] = [q0 q1 q3 q2], lane 5 = q4, lane 4 = q5
        const u32 ql = (l4 << 5) | (l5 << 4);
        const u32 in8 = (offset(ql) + u) * 8;
#pragma unroll
        for (int m = 0; m < 16; ++m) v[m] = ld(ntt_swap_dit6_row_in(m), in8);
        // the lane's position below the rows: tile's column block, the group's column bits (lane bits here: C <= 2), u
        const u32 chl = C == 2 ? ((l4 << 1) | l5) : C == 1 ? l4 : 0;
        const u32 xl8 = ((lo_block << (4 + C)) + (chl << 4) + u) * 8;
        ntt_swap_dit6<1, R>(vv, twr, log_d, xl8, l4, l5);
        // exit: R = 4 as on entry; R = 5: registers [q4 q1 q3 q2], lane 5 = q0, lane 4 = q5; R = 6: ntt_swap_dit6_row, lanes = q1 q0
        const u32 qlo = R == 4 ? ql : R == 5 ? ((l4 << 5) | l5) : ((l4 << 1) | l5);
        const u32 out8 = (offset(qlo) + u) * 8;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const u32 qm = R == 4 ? ntt_swap_dit6_row_in(m) : R == 5 ? (((m & 8) << 1) | ((m & 2) << 2) | ((m & 1) << 2) | ((m & 4) >> 1)) : ntt_swap_dit6_row(m);
            st(qm, out8, v[m]);
        }
    }
}

// ---- the contiguous pass, one wave per 2^10 elements -----------------------------------------------------------------------------
// wave-local transposes: rows of 16 elements at 17 words
// (a) from the segment layout (lane = (g, u): row = the register's row bits | the lanes', column u) to one row per lane
// (b) back
__device__ __forceinline__ u32 ntt_wave_lds(u32 row, u32 col) { return row * 17 + col; }

// values -> coefficients, the LAST pass (log_d = 0, r = 10): stages 9 .. 0 on the wave's 2^10 elements, then the store factor
// (out_scale table | out_const | canonical).  256 threads = four independent waves.
static __global__ void __launch_bounds__(256) ntt_contig_wave_kernel_dif(NttPass p) {
    extern __shared__ __attribute__((aligned(16))) u64 lds_all[];            // 4 x ZK_NTT_WAVE_LDS words (ntt_host.inc)
    const u32 tid = threadIdx.x, lane = tid & 63, wv = ntt_uniform(tid >> 6), u = lane & 15, l4 = (lane >> 4) & 1, l5 = lane >> 5;
    u64 *const lds = lds_all + wv * ZK_NTT_WAVE_LDS;
    const u32 tile_id = (p.cols_fastest ? blockIdx.y : blockIdx.x) * 4 + wv;
    const u32 col_id = p.cols_fastest ? blockIdx.x : blockIdx.y;
    if (((size_t)tile_id << ZK_NTT_WAVE_BITS) >> p.log_n) return;                 // (whole waves leave: nothing below is shared)
    const u32 base = tile_id << ZK_NTT_WAVE_BITS;
    const u64 *src = p.src + (size_t)col_id * p.src_stride + base;
    u64 *dst = p.dst + (size_t)col_id * p.dst_stride + base;
    const __amdgpu_buffer_rsrc_t twr = ntt_tw_rsrc(p.tw);
    u64 v[16];
    // element e = row * 16 + u; on entry register m = row bits 5 .. 2, lane 5 = row bit 1, lane 4 = row bit 0 (a load instruction
    // reads four adjacent rows: 512 contiguous bytes); from the transposes on lane 5 = row bit 5, lane 4 = row bit 4
    const u32 rb = (l5 << 5) | (l4 << 4), rl = (l5 << 1) | l4;
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = ntt_ld(src + ((u32)m << 6), ((rl << 4) + u) * 8);
    // stage k (pairs 2^k apart): level s_k = log_n - 1 - k, block (tile << (9 - k)) + (e >> (k + 1))
    const int s_top = p.log_n - ZK_NTT_WAVE_BITS;
    ntt_swap_dif6(v, p.tw, twr, s_top, tile_id, l4, l5);
    // the twiddles of the four in-segment stages (by x >> (k + 1), x = base + lane * 16 + j after the transpose): requested now
    auto lvl = [&](int k) { return ((1u << (s_top + 9 - k)) - 1) + (tile_id << (9 - k)); };
    const u64 bw3 = ntt_tw_load(twr, lane * 8, lvl(3));
    const u64 bw2[2] = {ntt_tw_load(twr, lane * 16, lvl(2)), ntt_tw_load(twr, lane * 16, lvl(2) + 1)};
    u64 bw1[4], bw0[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) bw1[i] = ntt_tw_load(twr, lane * 32, lvl(1) + i);
#pragma unroll
    for (int i = 0; i < 8; ++i) bw0[i] = ntt_tw_load(twr, lane * 64, lvl(0) + i);
#pragma unroll
    for (int m = 0; m < 16; ++m) lds[ntt_wave_lds(rb | ntt_swap_dif6_row(m), u)] = v[m];
    ntt_wave_sync();
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = lds[ntt_wave_lds(lane, j)];
    // the lane's row is `lane`; registers = position in the segment.  stage k = 3 .. 0: twiddle by x >> (k + 1), x = base + lane * 16 + j
    // (bw*: requested before the transpose, above)
    { ZK_NTT_STAGE16(3, bw3) }
    { ZK_NTT_STAGE16(2, bw2[m >> 3]) }
    { ZK_NTT_STAGE16(1, bw1[m >> 2]) }
    { ZK_NTT_STAGE16(0, bw0[m >> 1]) }
    ntt_wave_sync();
#pragma unroll
    for (int j = 0; j < 16; ++j) lds[ntt_wave_lds(lane, j)] = v[j];
    ntt_wave_sync();
    // back to segments: register m = row bits 3 .. 0, lanes = row bits 5, 4; the store factor on coalesced addresses
    u64 sc[16];
    if (p.out_scale) {
#pragma unroll
        for (int m = 0; m < 16; ++m) sc[m] = ntt_ld(p.out_scale + base + ((u32)m << 4), ((rb << 4) + u) * 8);
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        u64 w = lds[ntt_wave_lds(rb | m, u)];
        if (p.out_scale) w = gl_mul_canon(w, sc[m]);
        else if (p.apply_out_const) w = gl_mul_canon(w, p.out_const);
        else if (p.last_pass) w = gl_canon(w);
        ntt_st(dst + ((u32)m << 4), ((rb << 4) + u) * 8, w);
    }
}

// coefficients -> values, the FIRST pass (log_d = 0): the wave's 2^10 coefficients (bit-reversed order, source index sbase + i)
// evaluated on NB = 2^rate cosets -- NB = 1: r = 10, no replication; NB = 2 (rate_bits = 1): r = 11 with the first stage free,
// value x = 2 i + b = the plain 2^10-point transform of c_i * scale_b[i], scale_0 = in_scale (may be null: ones), scale_1 =
// in_scale2 = the same coset table for shift * w_(2n) (ntt_host.inc).  Output index (sbase + i) * NB + b.
template <int NB>
__global__ void __launch_bounds__(256) ZK_NTT_WAVES_PER_EU(NB == 2 ? 3 : 4, 8) ntt_contig_wave_kernel_dit(NttPass p, const u64 *in_scale2) {
    extern __shared__ __attribute__((aligned(16))) u64 lds_all[];            // 4 x ZK_NTT_WAVE_LDS words (ntt_host.inc)
    const u32 tid = threadIdx.x, lane = tid & 63, wv = ntt_uniform(tid >> 6), u = lane & 15, l4 = (lane >> 4) & 1, l5 = lane >> 5;
    u64 *const lds = lds_all + wv * ZK_NTT_WAVE_LDS;
    const u32 tile_id = (p.cols_fastest ? blockIdx.y : blockIdx.x) * 4 + wv;
    const u32 col_id = p.cols_fastest ? blockIdx.x : blockIdx.y;
    const int log_src = p.log_n - (NB == 2 ? 1 : 0);
    if (((size_t)tile_id << ZK_NTT_WAVE_BITS) >> log_src) return;
    const u32 sbase = tile_id << ZK_NTT_WAVE_BITS;
    const u64 *src = p.src + (size_t)col_id * p.src_stride + sbase;
    u64 *dst = p.dst + (size_t)col_id * p.dst_stride + (size_t)sbase * NB;
    const __amdgpu_buffer_rsrc_t twr = ntt_tw_rsrc(p.tw);
    const ntt_const_u64p ctw = (ntt_const_u64p)(unsigned long long)p.tw;
    const u32 rb = (l5 << 5) | (l4 << 4);
    u64 vv[NB][16];
    // source index i = row * 16 + u: lane 5 = row bit 5, lane 4 = row bit 4, register m = row bits 3 .. 0; the coefficient is read
    // once and scaled once per coset (eight registers at a time: the loads in flight are what sets the kernel's register count)
#pragma unroll
    for (int h = 0; h < 16; h += 8) {
        u64 c[8], sc[NB][8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const u32 um = (u32)(h + m) << 4, lb = ((rb << 4) + u) * 8;              // i = um (uniform) + lane part
            c[m] = ntt_ld(src + um, lb);
            if (p.in_scale) sc[0][m] = ntt_ld(p.in_scale + sbase + um, lb);
            if (NB == 2) sc[NB - 1][m] = ntt_ld(in_scale2 + sbase + um, lb);
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            vv[0][h + m] = p.in_scale ? gl_mul(c[m], sc[0][m]) : c[m];
            if (NB == 2) vv[NB - 1][h + m] = gl_mul(c[m], sc[NB - 1][m]);
        }
    }
    // to one row per lane (the buffer serves one coset at a time)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b) ntt_wave_sync();
#pragma unroll
        for (int m = 0; m < 16; ++m) lds[ntt_wave_lds(rb | m, u)] = vv[b][m];
        ntt_wave_sync();
#pragma unroll
        for (int j = 0; j < 16; ++j) vv[b][j] = lds[ntt_wave_lds(lane, j)];
    }
    // the four stages inside the lane's segment: pairs 2^q apart, twiddle T_(2^q)[j mod 2^q] -- the table's first 15 entries, of
    // which T_(2^q)[0] = 1 (twiddle_levels_kernel): those 15 of the 32 butterflies skip the multiply (b * 1 = the canonical b)
#define ZK_NTT_STAGE16N_PURE(BIT, W_OF_M)                                                     \
    _Pragma("unroll") for (int b = 0; b < NB; ++b)                                            \
    _Pragma("unroll") for (int m = 0; m < 16; ++m)                                            \
        if (!(m & (1 << (BIT)))) {                                                            \
            if ((m & ((1 << (BIT)) - 1)) == 0) ntt_bfly_one(vv[b][m], vv[b][m | (1 << (BIT))]); \
            else ntt_bfly(vv[b][m], vv[b][m | (1 << (BIT))], (W_OF_M));                       \
        }
    {
        ZK_NTT_STAGE16N_PURE(0, 0)
    }
    {
        const u64 w[2] = {1, ctw[2]};
        ZK_NTT_STAGE16N_PURE(1, w[m & 1])
    }
    {
        const u64 w[4] = {1, ctw[4], ctw[5], ctw[6]};
        ZK_NTT_STAGE16N_PURE(2, w[m & 3])
    }
    {
        const u64 w[8] = {1, ctw[8], ctw[9], ctw[10], ctw[11], ctw[12], ctw[13], ctw[14]};
        ZK_NTT_STAGE16N_PURE(3, w[m & 7])
    }
#undef ZK_NTT_STAGE16N_PURE
    // back to segments in ntt_swap_dit6's entry layout: registers [3..0] = row bits [0 1 3 2], lane 5 = row bit 4, lane 4 = row bit 5
    const u32 rl = (l4 << 5) | (l5 << 4);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        ntt_wave_sync();
#pragma unroll
        for (int j = 0; j < 16; ++j) lds[ntt_wave_lds(lane, j)] = vv[b][j];
        ntt_wave_sync();
#pragma unroll
        for (int m = 0; m < 16; ++m) vv[b][m] = lds[ntt_wave_lds(ntt_swap_dit6_row_in(m) | rl, u)];
    }
    // six stages on the row bits (pairs 2^(4 + k) apart: the level table with log_d = 4 and the segment position u below)
    ntt_swap_dit6<NB, 6>(vv, twr, 4, u * 8, l4, l5);
    // lane 5 = row bit 0, lane 4 = row bit 1; registers [3..0] = row bits [4 5 3 2]; the NB values of an element leave together
    const u32 ro = (l4 << 1) | l5;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const u32 um = ntt_swap_dit6_row(m) << 4, li = (ro << 4) + u;             // i = um (uniform) + li
        if (NB == 2) {
            const u64 a = p.last_pass ? gl_canon(vv[0][m]) : vv[0][m], c = p.last_pass ? gl_canon(vv[NB - 1][m]) : vv[NB - 1][m];
            ntt_st2(dst + 2 * (size_t)um, li * 16, a, c);
        } else {
            ntt_st(dst + um, li * 8, p.last_pass ? gl_canon(vv[0][m]) : vv[0][m]);
        }
    }
}
