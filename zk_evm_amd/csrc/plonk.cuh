// PLONK prover kernels for the recursion layer: plonky2 1.0.0 `prove` as the reference drives it
// after every segment STARK -- `StarkWrapperCircuit::prove` / `shrink` and `root.circuit.prove`
// (evm_arithmetization/src/fixed_recursive_verifier.rs:2146, 3167-3179), circuits of 2^12..2^14 rows under
// `CircuitConfig::standard_recursion_config()` (135 wires, 80 routed, quotient degree factor 8, FRI rate_bits 3).
// [EXT] plonky2/src/plonk/{prover.rs, vanishing_poly.rs, plonk_common.rs}, gates/{gate, selectors, arithmetic_base,
// constant, public_input, noop}.rs.  The commitments, openings and FRI are the STARK path's own kernels (ntt.cuh,
// merkle.cuh, fri.cuh) at rate_bits 3; what is new here is the permutation argument and the gate-filtered quotient.
//
// MI355X mapping: both kernels are one lane per row / coset point over column-major matrices (64 lanes read 64
// consecutive u64 of one column), the wires are loaded once and shared by all challenges, the running product across
// rows is an exclusive prefix product (three-kernel block scan) instead of plonky2's sequential loop, and all
// divisions of a row share one inversion.
#pragma once
#include "gl.cuh"
#include "poseidon.cuh"

#define ZK_PLONK_MAX_CHALLENGES 2
#define ZK_PLONK_MAX_CHUNKS 16          // ceil(num_routed_wires / quotient_degree_factor): 10 for the standard config
#define ZK_PLONK_MAX_GATES 32
#define ZK_PLONK_UNUSED_SELECTOR 0xFFFFFFFFULL

struct PlonkGateDesc {
    u32 kind, param, selector_index, group_start, group_end;
    u32 slice;                                   // which of the (at most 2) quotient-kernel slices evaluates this gate
};

struct PlonkPermArgs {
    const u64 *wires; size_t wires_stride;       // [num_wires][n] witness values (natural order)
    const u64 *sigmas; size_t sigmas_stride;     // [routed][n] sigma VALUES on the subgroup
    const u64 *k_is;                             // [routed]
    const u64 *tw;                               // w_n^k, k < n/2
    u32 log_n, routed, chunk, n_chunks, n_challenges;
    u64 betas[ZK_PLONK_MAX_CHALLENGES], gammas[ZK_PLONK_MAX_CHALLENGES];
    u64 *q;                                      // [n_challenges][n_chunks][n] chunk quotients
    u64 *totals;                                 // [n_challenges][n] product of the row's chunk quotients
};

// per row: quotient_chunk_products of `wires_permutation_partial_products_and_zs`
static __global__ void __launch_bounds__(256) plonk_chunk_quotients_kernel(PlonkPermArgs A) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 n = 1u << A.log_n;
    if (i >= n) return;
    u64 x = 1;
    if (A.log_n) {
        const u32 half = n >> 1;
        x = A.tw[i & (half - 1)];
        if (i & half) x = gl_neg(x);
    }
    u64 num[ZK_PLONK_MAX_CHALLENGES][ZK_PLONK_MAX_CHUNKS], den[ZK_PLONK_MAX_CHALLENGES][ZK_PLONK_MAX_CHUNKS];
    u64 bx[ZK_PLONK_MAX_CHALLENGES];
    for (u32 c = 0; c < A.n_challenges; ++c) bx[c] = gl_mul(A.betas[c], x);
    for (u32 k = 0; k < A.n_chunks; ++k) {
        u64 pn[ZK_PLONK_MAX_CHALLENGES], pd[ZK_PLONK_MAX_CHALLENGES];
        for (u32 c = 0; c < A.n_challenges; ++c) pn[c] = pd[c] = 1;
        const u32 j1 = (k + 1) * A.chunk < A.routed ? (k + 1) * A.chunk : A.routed;
        for (u32 j = k * A.chunk; j < j1; ++j) {
            const u64 wv = A.wires[(size_t)j * A.wires_stride + i];
            const u64 sg = A.sigmas[(size_t)j * A.sigmas_stride + i];
            const u64 kj = A.k_is[j];
            for (u32 c = 0; c < A.n_challenges; ++c) {
                pn[c] = gl_mul(pn[c], gl_add(gl_add(wv, gl_mul(kj, bx[c])), A.gammas[c]));
                pd[c] = gl_mul(pd[c], gl_add(gl_add(wv, gl_mul(A.betas[c], sg)), A.gammas[c]));
            }
        }
        for (u32 c = 0; c < A.n_challenges; ++c) { num[c][k] = pn[c]; den[c][k] = pd[c]; }
    }
    for (u32 c = 0; c < A.n_challenges; ++c) {
        // one inversion for the row's n_chunks denominators: prefix products, invert, walk back
        u64 pre[ZK_PLONK_MAX_CHUNKS];
        u64 run = 1;
        for (u32 k = 0; k < A.n_chunks; ++k) { pre[k] = run; run = gl_mul(run, den[c][k]); }
        u64 inv = gl_inv(run);
        u64 total = 1;
        u64 qk[ZK_PLONK_MAX_CHUNKS];
        for (u32 k = A.n_chunks; k-- > 0;) {
            qk[k] = gl_mul(num[c][k], gl_mul(inv, pre[k]));
            inv = gl_mul(inv, den[c][k]);
        }
        for (u32 k = 0; k < A.n_chunks; ++k) {
            A.q[((size_t)c * A.n_chunks + k) * n + i] = gl_canon(qk[k]);
            total = gl_mul(total, qk[k]);
        }
        A.totals[(size_t)c * n + i] = gl_canon(total);
    }
}

// ---- exclusive prefix PRODUCT over rows (Z(x_i) = prod_{r < i} totals[r]) ----------------------------------------------
#define ZK_PSCAN_ITEMS 8
#define ZK_PSCAN_THREADS 256
static __global__ void __launch_bounds__(ZK_PSCAN_THREADS)
plonk_scan_block_kernel(const u64 *__restrict__ x, u32 n, u64 *__restrict__ incl, u64 *__restrict__ totals) {
    __shared__ u64 sh[ZK_PSCAN_THREADS];
    const u32 base = (blockIdx.x * ZK_PSCAN_THREADS + threadIdx.x) * ZK_PSCAN_ITEMS;
    u64 loc[ZK_PSCAN_ITEMS];
    u64 run = 1;
#pragma unroll
    for (int k = 0; k < ZK_PSCAN_ITEMS; ++k) {
        const u32 i = base + k;
        run = gl_mul(run, i < n ? x[i] : 1);
        loc[k] = run;
    }
    sh[threadIdx.x] = run;
    __syncthreads();
    for (u32 off = 1; off < ZK_PSCAN_THREADS; off <<= 1) {
        const u64 m = threadIdx.x >= off ? sh[threadIdx.x - off] : 1;
        __syncthreads();
        sh[threadIdx.x] = gl_mul(sh[threadIdx.x], m);
        __syncthreads();
    }
    const u64 excl = threadIdx.x ? sh[threadIdx.x - 1] : 1;
#pragma unroll
    for (int k = 0; k < ZK_PSCAN_ITEMS; ++k) {
        const u32 i = base + k;
        if (i < n) incl[i] = gl_mul(loc[k], excl);
    }
    if (threadIdx.x == ZK_PSCAN_THREADS - 1) totals[blockIdx.x] = sh[threadIdx.x];
}
static __global__ void __launch_bounds__(256) plonk_scan_totals_kernel(u64 *totals, u32 n_blocks) {
    __shared__ u64 sh[256];
    const u32 per = (n_blocks + 255) / 256;
    const u32 lo = threadIdx.x * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
    u64 run = 1;
    for (u32 i = lo; i < hi; ++i) run = gl_mul(run, totals[i]);
    sh[threadIdx.x] = run;
    __syncthreads();
    for (u32 off = 1; off < 256; off <<= 1) {
        const u64 m = threadIdx.x >= off ? sh[threadIdx.x - off] : 1;
        __syncthreads();
        sh[threadIdx.x] = gl_mul(sh[threadIdx.x], m);
        __syncthreads();
    }
    run = threadIdx.x ? sh[threadIdx.x - 1] : 1;
    for (u32 i = lo; i < hi; ++i) { const u64 v = totals[i]; totals[i] = run; run = gl_mul(run, v); }
}
// out layout (plonky2 `zs_partial_products`): Z of challenge 0, 1, ..; then the (n_chunks - 1) partial products of
// challenge 0, of challenge 1, ..   incl / block_totals: the scan pieces of challenge c at [c * n], [c * n_blocks].
static __global__ void __launch_bounds__(256)
plonk_partial_products_finish_kernel(const u64 *__restrict__ incl, const u64 *__restrict__ block_totals, u32 n_blocks,
                                     const u64 *__restrict__ q, u32 log_n, u32 n_chunks, u32 n_challenges,
                                     u64 *__restrict__ out, size_t out_stride) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 n = 1u << log_n;
    if (i >= n) return;
    for (u32 c = 0; c < n_challenges; ++c) {
        // Z(x_i) = inclusive product up to row i-1
        u64 z = 1;
        if (i) {
            const u32 p = i - 1;
            z = gl_mul(incl[(size_t)c * n + p], block_totals[(size_t)c * n_blocks + p / (ZK_PSCAN_THREADS * ZK_PSCAN_ITEMS)]);
        }
        out[(size_t)c * out_stride + i] = gl_canon(z);
        u64 acc = z;
        for (u32 k = 0; k + 1 < n_chunks; ++k) {
            acc = gl_mul(acc, q[((size_t)c * n_chunks + k) * n + i]);
            out[((size_t)n_challenges + (size_t)c * (n_chunks - 1) + k) * out_stride + i] = gl_canon(acc);
        }
    }
}

// ---- quotient: eval_vanishing_poly_base_batch / Z_H on the coset of size n * 2^qd_bits ---------------------------------
#define ZK_PLONK_MAX_INTERP_POINTS 32
struct PlonkQuotientArgs {
    const u64 *interp_domain, *interp_weights;   // CosetInterpolationGate: H (2^subgroup_bits points) and barycentric weights
    const u64 *cs; size_t cs_stride;             // constants ++ sigmas LDE  [num_constants + routed][N]
    const u64 *wires; size_t wires_stride;       // wires LDE [num_wires][N]
    const u64 *zs; size_t zs_stride;             // Zs ++ partial products LDE
    const u64 *k_is;
    const u64 *tw;                               // w_size^k, k < size/2
    const PlonkGateDesc *gates;
    const u64 *alpha_pow[ZK_PLONK_MAX_CHALLENGES];   // alpha_c^k
    u32 log_n, qd_bits, step_log;
    u32 num_constants, num_selectors, routed, num_wires, chunk, n_chunks, n_challenges, n_gates, num_gate_constraints;
    u64 betas[ZK_PLONK_MAX_CHALLENGES], gammas[ZK_PLONK_MAX_CHALLENGES];
    u64 pi_hash[4];
    u64 g_pow_n, n_inv;
    u64 *out; size_t out_stride;
    u64 *out2;                                   // second slice's partial sums (gridDim.y == 2), added by plonk_add_slices_kernel
};

// sum_k term_k * alpha_c^k for every challenge.  The terms of ONE gate share its filter, so they are summed unfiltered and
// the gate's sum is filtered and added once (`end_gate`): one multiply per term and challenge instead of two.
// (Measured and dropped, r02h: the delayed-reduction accumulator of fri.cuh here -- 8 instructions per term instead of
// 27, but v_addc on a v_mad_u64_u32 carry-out stalls, and 14 more live VGPRs: the kernel got 40 % SLOWER.)
struct PlonkAcc {
    u64 g[ZK_PLONK_MAX_CHALLENGES];               // the current gate's unfiltered sum
    u64 total[ZK_PLONK_MAX_CHALLENGES];
    const u64 *pow[ZK_PLONK_MAX_CHALLENGES];
    u32 nc;
    // (compile-time indices only: a loop over the run-time challenge count indexes g / total / pow dynamically, which puts
    //  the accumulators in scratch memory -- every term then went through a scratch load and store, r03v)
    static_assert(ZK_PLONK_MAX_CHALLENGES == 2, "PlonkAcc::add / end_gate are written out for two challenges");
    __device__ __forceinline__ void add(u32 k, u64 term) {
        g[0] = gl_add_canon(g[0], gl_mul_canon(term, pow[0][k]));
        if (nc > 1) g[1] = gl_add_canon(g[1], gl_mul_canon(term, pow[1][k]));
    }
    __device__ __forceinline__ void end_gate(u64 filt) {
        total[0] = gl_add_canon(total[0], gl_mul_canon(filt, g[0])); g[0] = 0;
        if (nc > 1) { total[1] = gl_add_canon(total[1], gl_mul_canon(filt, g[1])); g[1] = 0; }
    }
};

// wire w of this point's row
// (global-memory pointers said to be so: inside the non-inlined wide-gate evaluator the argument block arrives by reference
//  and the compiler would otherwise emit FLAT loads for every wire)
typedef const u64 __attribute__((address_space(1))) *plonk_gptr;
#define PLONK_W(w) (((plonk_gptr)A.wires)[(size_t)(w) * A.wires_stride + row])
__device__ __forceinline__ gl2 plonk_wext(const PlonkQuotientArgs &A, size_t row, u32 w) { return gl2_make(PLONK_W(w), PLONK_W(w + 1)); }

// The gates beyond the four closed-form base-field ones.  Extension elements live in D = 2 consecutive wires; a
// constraint over F_{p^2} contributes its two components as consecutive terms (`to_basefield_array`).
__device__ __noinline__ void plonk_eval_wide_gate(const PlonkQuotientArgs &A, const PlonkGateDesc &G, const u64 *consts_generic,
                                                  size_t row, u32 term, PlonkAcc &acc_io) {
    PlonkAcc acc = acc_io;                 // the accumulators in registers for the gate, written back once at the end
    const plonk_gptr consts = (plonk_gptr)consts_generic;
    const u32 n = G.param;
    auto add2 = [&](u32 k, gl2 v) { acc.add(term + 2 * k, v.a); acc.add(term + 2 * k + 1, v.b); };
    switch (G.kind) {
        case 4: {   // ArithmeticExtensionGate { num_ops }: output - (c0 m0 m1 + c1 addend)   (gates/arithmetic_extension.rs)
            const u64 c0 = consts[row], c1 = consts[A.cs_stride + row];
            for (u32 j = 0; j < n; ++j) {
                const gl2 m0 = plonk_wext(A, row, 8 * j), m1 = plonk_wext(A, row, 8 * j + 2), ad = plonk_wext(A, row, 8 * j + 4),
                          o = plonk_wext(A, row, 8 * j + 6);
                add2(j, gl2_sub(o, gl2_add(gl2_scale(gl2_mul(m0, m1), c0), gl2_scale(ad, c1))));
            }
            break;
        }
        case 5: {   // MulExtensionGate { num_ops }: output - c0 m0 m1   (gates/multiplication_extension.rs)
            const u64 c0 = consts[row];
            for (u32 j = 0; j < n; ++j) {
                const gl2 m0 = plonk_wext(A, row, 6 * j), m1 = plonk_wext(A, row, 6 * j + 2), o = plonk_wext(A, row, 6 * j + 4);
                add2(j, gl2_sub(o, gl2_scale(gl2_mul(m0, m1), c0)));
            }
            break;
        }
        case 6: {   // BaseSumGate<2> { num_limbs }: sum of limb_i 2^i - wire 0, then limb (limb - 1)   (gates/base_sum.rs)
            u64 s = 0;
            for (u32 i = n; i-- > 0;) s = gl_add(gl_add(s, s), PLONK_W(1 + i));
            acc.add(term, gl_sub(s, PLONK_W(0)));
            for (u32 i = 0; i < n; ++i) {
                const u64 l = PLONK_W(1 + i);
                acc.add(term + 1 + i, gl_mul(l, gl_sub(l, 1)));
            }
            break;
        }
        case 7: case 8: {   // ReducingGate / ReducingExtensionGate { num_coeffs }: acc_i - (acc_{i-1} alpha + coeff_i)
            const bool ext = G.kind == 8;
            const gl2 alpha = plonk_wext(A, row, 2);
            gl2 prev = plonk_wext(A, row, 4);
            const u32 accs = ext ? 6 + 2 * n : 6 + n;
            for (u32 i = 0; i < n; ++i) {
                gl2 t = gl2_mul(prev, alpha);
                if (ext) t = gl2_add(t, plonk_wext(A, row, 6 + 2 * i));
                else t.a = gl_add(t.a, PLONK_W(6 + i));
                const gl2 cur = i + 1 == n ? plonk_wext(A, row, 0) : plonk_wext(A, row, accs + 2 * i);
                add2(i, gl2_sub(cur, t));
                prev = cur;
            }
            break;
        }
        case 9: {   // ExponentiationGate { num_power_bits }   (gates/exponentiation.rs)
            const u64 base = PLONK_W(0);
            u64 last = 1;
            for (u32 i = 0; i < n; ++i) {
                const u64 prev = i == 0 ? 1 : gl_mul(last, last);
                const u64 bit = PLONK_W(1 + (n - 1 - i));
                const u64 cur = PLONK_W(2 + n + i);
                const u64 sel = gl_add(gl_mul(bit, base), gl_sub(1, bit));
                acc.add(term + i, gl_sub(cur, gl_mul(prev, sel)));
                last = cur;
            }
            acc.add(term + n, gl_sub(PLONK_W(1 + n), last));
            break;
        }
        case 10: {  // PoseidonGate   (gates/poseidon.rs): plain rounds -- the same constraint polynomials as plonky2's
                    // fast partial rounds (between S-boxes both are the same affine maps, cf. the Poseidon table AIR in airs.cuh)
            u32 k = term;
            const u64 swap = PLONK_W(24);
            acc.add(k++, gl_mul(swap, gl_sub(swap, 1)));
            u64 s[12];
            for (u32 i = 0; i < 4; ++i) {
                const u64 lhs = PLONK_W(i), rhs = PLONK_W(i + 4), d = PLONK_W(25 + i);
                acc.add(k++, gl_sub(gl_mul(swap, gl_sub(rhs, lhs)), d));
                s[i] = gl_add(lhs, d);
                s[i + 4] = gl_sub(rhs, d);
            }
            for (u32 i = 8; i < 12; ++i) s[i] = PLONK_W(i);
            for (u32 i = 0; i < 12; ++i) s[i] = gl_add_canon(s[i], ZK_RC[i]);
            int round = 0;
            for (int r = 0; r < 4; ++r) {                 // first full rounds; rounds 1..3 check their S-box inputs
                if (r) {
                    for (u32 i = 0; i < 12; ++i) {
                        const u64 w = PLONK_W(29 + 12 * (r - 1) + i);
                        acc.add(k++, gl_sub(s[i], w));
                        s[i] = w;
                    }
                }
                for (u32 i = 0; i < 12; ++i) s[i] = pos_sbox(s[i]);
                ++round;
                pos_mds<true>(s, &ZK_RCS[round * 12]);    // MDS + the next round's constants = the next S-box input
            }
            for (int r = 0; r < 22; ++r) {
                const u64 w = PLONK_W(65 + r);
                acc.add(k++, gl_sub(s[0], w));
                s[0] = pos_sbox(w);
                ++round;
                pos_mds<true>(s, &ZK_RCS[round * 12]);
            }
            for (int r = 0; r < 4; ++r) {
                for (u32 i = 0; i < 12; ++i) {
                    const u64 w = PLONK_W(87 + 12 * r + i);
                    acc.add(k++, gl_sub(s[i], w));
                    s[i] = pos_sbox(w);
                }
                ++round;
                if (r < 3) pos_mds<true>(s, &ZK_RCS[round * 12]);
                else pos_mds<false>(s, nullptr);
            }
            for (u32 i = 0; i < 12; ++i) acc.add(k++, gl_sub(s[i], PLONK_W(12 + i)));
            break;
        }
        case 11: {  // RandomAccessGate { bits, num_copies, num_extra_constants } packed as bits | copies << 8 | extra << 16
            const u32 bits = n & 0xFF, copies = (n >> 8) & 0xFF, extra = (n >> 16) & 0xFF;
            const u32 vec = 1u << bits, routed = (2 + vec) * copies + extra;
            u32 k = term;
            for (u32 c = 0; c < copies; ++c) {
                const u32 base = (2 + vec) * c;
                u64 items[32];
                for (u32 i = 0; i < vec; ++i) items[i] = PLONK_W(base + 2 + i);
                u64 rec = 0;
                for (u32 i = 0; i < bits; ++i) {
                    const u64 b = PLONK_W(routed + c * bits + i);
                    acc.add(k++, gl_mul(b, gl_sub(b, 1)));
                }
                for (u32 i = bits; i-- > 0;) rec = gl_add(gl_add(rec, rec), PLONK_W(routed + c * bits + i));
                acc.add(k++, gl_sub(rec, PLONK_W(base)));
                u32 len = vec;
                for (u32 i = 0; i < bits; ++i) {        // fold pairs by bit i (least significant first)
                    const u64 b = PLONK_W(routed + c * bits + i);
                    len >>= 1;
                    for (u32 j = 0; j < len; ++j) items[j] = gl_add(items[2 * j], gl_mul(b, gl_sub(items[2 * j + 1], items[2 * j])));
                }
                acc.add(k++, gl_sub(items[0], PLONK_W(base + 1)));
            }
            for (u32 i = 0; i < extra; ++i)
                acc.add(k++, gl_sub(consts[(size_t)i * A.cs_stride + row], PLONK_W((2 + vec) * copies + i)));
            break;
        }
        case 12: {  // PoseidonMdsGate: outputs - MDS * inputs on twelve extension elements (gates/poseidon_mds.rs)
            constexpr u32 CIRC[12] = ZK_POSEIDON_MDS_CIRC_INIT;
            for (u32 r = 0; r < 12; ++r) {
                gl2 a = gl2_make(0, 0);
                for (u32 i = 0; i < 12; ++i) a = gl2_add(a, gl2_scale(plonk_wext(A, row, 2 * ((i + r) % 12)), CIRC[i]));
                if (r == 0) a = gl2_add(a, gl2_scale(plonk_wext(A, row, 0), 8));      // MDS_MATRIX_DIAG = (8, 0, .., 0)
                add2(r, gl2_sub(plonk_wext(A, row, 24 + 2 * r), a));
            }
            break;
        }
        case 13: {  // CosetInterpolationGate { subgroup_bits, degree } packed as bits | degree << 8 (gates/coset_interpolation.rs)
            const u32 bits = n & 0xFF, deg = (n >> 8) & 0xFF, npnt = 1u << bits, ni = (npnt - 2) / (deg - 1);
            const u32 si = 1 + 2 * npnt + 4;                       // start_intermediates
            const u64 shift = PLONK_W(0);
            const gl2 point = plonk_wext(A, row, 1 + 2 * npnt), value = plonk_wext(A, row, 3 + 2 * npnt);
            const gl2 x = plonk_wext(A, row, si + 4 * ni);         // shifted evaluation point
            u32 k = 0;
            add2(k++, gl2_sub(point, gl2_scale(x, shift)));
            gl2 ev = gl2_make(0, 0), prod = gl2_make(1, 0);
            u32 lo = 0, hi = deg;
            for (u32 c = 0; c <= ni; ++c) {                         // partial_interpolate over chunk [lo, hi)
                for (u32 i = lo; i < hi; ++i) {
                    gl2 term = x;
                    term.a = gl_sub(term.a, A.interp_domain[i]);
                    const gl2 wv = gl2_scale(plonk_wext(A, row, 1 + 2 * i), A.interp_weights[i]);
                    ev = gl2_add(gl2_mul(ev, term), gl2_mul(wv, prod));
                    prod = gl2_mul(prod, term);
                }
                if (c == ni) break;
                const gl2 iev = plonk_wext(A, row, si + 2 * c), iprod = plonk_wext(A, row, si + 2 * (ni + c));
                add2(k++, gl2_sub(iev, ev));
                add2(k++, gl2_sub(iprod, prod));
                ev = iev; prod = iprod;
                lo = 1 + (deg - 1) * (c + 1);
                hi = lo + deg - 1 < npnt ? lo + deg - 1 : npnt;
            }
            add2(k++, gl2_sub(value, ev));
            break;
        }
        default: break;
    }
    acc_io = acc;
}
#undef PLONK_W

// One lane per point of the quotient coset.  For circuits of <= 2^14 rows the coset has <= 2^17 points = at most two
// waves per SIMD, and ONE wave per SIMD issues at half rate whatever its ILP (profiles/archive/r02g_ubench_single_wave_issue.txt):
// there the launch has gridDim.y == 2 and each slice evaluates the gates the host gave it (cost-balanced; slice 0 also
// the permutation terms).  The vanishing polynomial is a SUM of terms, so the slices' results just add.
static __global__ void __launch_bounds__(256) plonk_quotient_kernel(PlonkQuotientArgs A) {
    const u32 sl = blockIdx.y, nsl = gridDim.y;
    const u32 size_log = A.log_n + A.qd_bits;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> size_log) return;
    const u32 size = 1u << size_log, half = size >> 1;
    u64 w = A.tw[i & (half - 1)];
    if (i & half) w = gl_neg(w);
    const u64 x = gl_mul(w, GL_GENERATOR);                         // shifted_x = coset_shift * w_size^i
    u64 wn = 1;                                                    // (w_size^n)^i: a 2^qd_bits-th root of unity
    {
        const u32 k = i & ((1u << A.qd_bits) - 1);
        if (k) {
            const u32 idx = k << A.log_n;
            const u64 t = A.tw[idx & (half - 1)];
            wn = (idx & half) ? gl_neg(t) : t;
        }
    }
    const u64 zh = gl_sub(gl_mul(A.g_pow_n, wn), 1);               // ZeroPolyOnCoset::eval(i)
    const u64 xm1 = gl_sub(x, 1);
    const u64 it = gl_inv(gl_mul(zh, xm1));                        // x is never in H: both non-zero
    const u64 inv_zh = gl_mul(it, xm1);
    const u64 l0 = gl_mul(gl_mul(zh, gl_mul(it, zh)), A.n_inv);    // eval_l_0 = Z_H(x) / (n (x - 1))
    const size_t row = (size_t)i << A.step_log;
    const size_t row_next = (size_t)((i + (1u << A.qd_bits)) & (size - 1)) << A.step_log;
    PlonkAcc acc;
    acc.nc = A.n_challenges;
    acc.g[0] = acc.g[1] = 0; acc.total[0] = acc.total[1] = 0;
    acc.pow[0] = A.alpha_pow[0]; acc.pow[1] = A.alpha_pow[A.n_challenges > 1 ? 1 : 0];
    const u64 *sig = A.cs + (size_t)A.num_constants * A.cs_stride;
    u32 term = 0;
    // vanishing_z_1_terms: L_0(x) (Z(x) - 1)
    if (sl == 0)
        for (u32 c = 0; c < A.n_challenges; ++c) acc.add(term + c, gl_mul(l0, gl_sub(A.zs[(size_t)c * A.zs_stride + row], 1)));
    term += A.n_challenges;
    // vanishing_partial_products_terms: check_partial_products, challenge-major
    if (sl == 0) {
        u64 bx[ZK_PLONK_MAX_CHALLENGES];
        for (u32 c = 0; c < A.n_challenges; ++c) bx[c] = gl_mul(A.betas[c], x);
        const u32 per_chal = A.n_chunks;
        for (u32 k = 0; k < A.n_chunks; ++k) {
            u64 pn[ZK_PLONK_MAX_CHALLENGES], pd[ZK_PLONK_MAX_CHALLENGES];
            for (u32 c = 0; c < A.n_challenges; ++c) pn[c] = pd[c] = 1;
            const u32 j1 = (k + 1) * A.chunk < A.routed ? (k + 1) * A.chunk : A.routed;
            for (u32 j = k * A.chunk; j < j1; ++j) {
                const u64 wv = A.wires[(size_t)j * A.wires_stride + row];
                const u64 sg = sig[(size_t)j * A.cs_stride + row];
                const u64 kj = A.k_is[j];
                for (u32 c = 0; c < A.n_challenges; ++c) {
                    pn[c] = gl_mul(pn[c], gl_add(gl_add(wv, gl_mul(kj, bx[c])), A.gammas[c]));
                    pd[c] = gl_mul(pd[c], gl_add(gl_add(wv, gl_mul(A.betas[c], sg)), A.gammas[c]));
                }
            }
            for (u32 c = 0; c < A.n_challenges; ++c) {
                // accs = [Z(x), pp_0 .. pp_{m-1}, Z(g x)]
                const size_t pp0 = (size_t)A.n_challenges + (size_t)c * (A.n_chunks - 1);
                const u64 prev = k == 0 ? A.zs[(size_t)c * A.zs_stride + row] : A.zs[(pp0 + k - 1) * A.zs_stride + row];
                const u64 next = k + 1 == A.n_chunks ? A.zs[(size_t)c * A.zs_stride + row_next] : A.zs[(pp0 + k) * A.zs_stride + row];
                const u64 t = gl_sub(gl_mul(prev, pn[c]), gl_mul(next, pd[c]));
                acc.add(term + c * per_chal + k, t);
            }
        }
    }
    if (sl == 0) acc.end_gate(1);
    term += A.n_challenges * A.n_chunks;
    // gate constraints: slot j collects filter_g * constraint_{g,j} over all gates (at most one filter is non-zero on H)
    const bool many = A.num_selectors > 1;
    const u64 *consts = A.cs + (size_t)A.num_selectors * A.cs_stride;      // gate constants: selectors removed
    for (u32 g = 0; g < A.n_gates; ++g) {
        const PlonkGateDesc G = A.gates[g];
        if (G.kind == 0 || (nsl > 1 && G.slice != sl)) continue;            // NoopGate: no constraints; other slice's gate
        const u64 s = A.cs[(size_t)G.selector_index * A.cs_stride + row];
        u64 filt = 1;                                                       // compute_filter
        for (u32 r = G.group_start; r < G.group_end; ++r)
            if (r != g) filt = gl_mul(filt, gl_sub((u64)r, s));
        if (many) filt = gl_mul(filt, gl_sub(ZK_PLONK_UNUSED_SELECTOR, s));
        if (G.kind == 1) {                                                  // ConstantGate { num_consts }
            for (u32 j = 0; j < G.param; ++j)
                acc.add(term + j, gl_sub(consts[(size_t)j * A.cs_stride + row], A.wires[(size_t)j * A.wires_stride + row]));
        } else if (G.kind == 2) {                                           // PublicInputGate
            for (u32 j = 0; j < 4; ++j)
                acc.add(term + j, gl_sub(A.wires[(size_t)j * A.wires_stride + row], A.pi_hash[j]));
        } else if (G.kind == 3) {                                           // ArithmeticGate { num_ops }
            const u64 c0 = consts[row], c1 = consts[A.cs_stride + row];
            for (u32 j = 0; j < G.param; ++j) {
                const u64 *wp = A.wires + (size_t)(4 * j) * A.wires_stride + row;
                const u64 m0 = wp[0], m1 = wp[A.wires_stride], ad = wp[2 * A.wires_stride], out = wp[3 * A.wires_stride];
                const u64 v = gl_sub(out, gl_add(gl_mul(gl_mul(m0, m1), c0), gl_mul(ad, c1)));
                acc.add(term + j, v);
            }
        } else {
            plonk_eval_wide_gate(A, G, consts, row, term, acc);
        }
        acc.end_gate(filt);
    }
    u64 *out = sl ? A.out2 : A.out;
    out[i] = gl_canon(gl_mul(acc.total[0], inv_zh));
    if (A.n_challenges > 1) out[A.out_stride + i] = gl_canon(gl_mul(acc.total[1], inv_zh));
}

static __global__ void plonk_add_slices_kernel(u64 *out, const u64 *out2, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = gl_canon(gl_add(out[i], out2[i]));
}

// out[c * cap + k] = alpha_c^k
static __global__ void plonk_alpha_pow_kernel(u64 *out, u32 cap, u32 count, u32 n_challenges, u64 a0, u64 a1) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    out[j] = gl_canon(gl_pow(a0, j));
    if (n_challenges > 1) out[(size_t)cap + j] = gl_canon(gl_pow(a1, j));
}
