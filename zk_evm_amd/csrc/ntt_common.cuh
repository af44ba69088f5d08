// What every NTT pass kernel shares (ntt.cuh's LDS tile kernels, ntt_swap.cuh's lane-swap kernels -- and the CPU emulation of the
// latter, tests/emu/, which compiles ntt_swap.cuh with g++): the pass descriptor, the butterfly, the twiddle loads.
#pragma once
#include "gl.cuh"

struct NttPass {
    const u64 *src;      // column c at src + c*src_stride
    u64 *dst;            // column c at dst + c*dst_stride
    size_t src_stride, dst_stride;
    const u64 *tw;       // DIT: level layout tw[D - 1 + k] = (root of order 2D)^k, k < D, for D = 1 .. 2^(log_tw-1);
                         // values -> coeffs: block-order levels tw[2^s - 1 + j] = (root of order 2^(s+1))^bitrev_s(j) (ntt_host.inc)
    const u64 *in_scale; // optional per-source-index factor applied on load (coset powers)
    const u64 *in_scale2;// ntt_contig_wave_kernel_dit<2> only: the load factors of the second coset (ntt_swap.cuh)
    int swap_ok;         // host only: this transform was planned for the lane-swap kernels (ntt_host.inc launch_swap_pass)
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
    int cols_fastest;    // grid = (columns, tiles): consecutive workgroups run the SAME tile of different columns (ntt_host.inc)
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

