// Goldilocks field arithmetic for gfx950 device code (p = 2^64 - 2^32 + 1).
//
// CDNA4 has no 64x64->128 multiplier: a field multiply is four v_mad_u64_u32 plus a ~10
// instruction fold that uses 2^64 = 2^32 - 1 and 2^96 = -1 (mod p).  Values travel between ops as
// arbitrary u64 representatives ("lazy"), add/sub tolerate that, and gl_canon() is applied once
// when a value leaves the chip (store / hash output).  Semantics follow plonky2_field 1.0.0
// GoldilocksField ([EXT] field/src/goldilocks_field.rs; described in the reference at
// book/src/framework/field.md:5-19); this is an independent formulation from oracle/goldilocks.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef long long i64;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL
#define GL_GENERATOR 14293326489335486720ULL
#define GL_POW2_GENERATOR 7277203076849721926ULL

#define GL_HD __host__ __device__ __forceinline__

GL_HD u64 gl_canon(u64 a) { return a >= GL_P ? a - GL_P : a; }

// ---- portable forms (host-side table setup; also the readable statement of each algorithm) ----
GL_HD u64 gl_add_ref(u64 a, u64 b) {
    u64 s = a + b;
    u64 c = s < a ? GL_EPS : 0;
    u64 s2 = s + c;
    u64 c2 = s2 < s ? GL_EPS : 0;  // only reachable when both inputs were >= 2^64 - 2^32
    return s2 + c2;
}
GL_HD u64 gl_sub_ref(u64 a, u64 b) {
    u64 d = a - b;
    u64 br = a < b ? GL_EPS : 0;
    u64 d2 = d - br;
    u64 br2 = d2 > d ? GL_EPS : 0;
    return d2 - br2;
}
// Fold a 128-bit value hi:lo to a u64 representative.
GL_HD u64 gl_reduce128(u64 hi, u64 lo) {
    u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= GL_EPS;
    u64 t1 = ((u64)hl << 32) - hl;  // hl * (2^32 - 1)
    u64 r = t0 + t1;
    if (r < t1) r += GL_EPS;
    return r;
}
// Fold a 96-bit value (hi32:lo64).
GL_HD u64 gl_reduce96(u32 hi, u64 lo) {
    u64 t1 = ((u64)hi << 32) - hi;
    u64 r = lo + t1;
    if (r < t1) r += GL_EPS;
    return r;
}
GL_HD u64 gl_mul_ref(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 mid = (u64)a0 * b1 + (p00 >> 32);          // <= (2^32-1)^2 + 2^32 - 1 : no overflow
    u64 mid2 = (u64)a1 * b0 + (u32)mid;            // likewise
    u64 hi = (u64)a1 * b1 + (mid >> 32) + (mid2 >> 32);
    u64 lo = (mid2 << 32) | (u32)p00;
    return gl_reduce128(hi, lo);
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---- gfx950 forms -----------------------------------------------------------------------------
// Hand-scheduled carry chains on 32-bit halves.  On this chip every carry/64-bit/select VALU op
// issues at ~4.3 cycles per wave (v_mov/and/or/xor/add_u32 at ~2.4), so the goal is the fewest
// instructions and no register-pair shuffling (v_mov) around v_mad_u64_u32 results.

// [T3:T2:T1:T0] -> lazy u64, with 2^64 = 2^32 - 1 and 2^96 = -1:   [T1:T0] - T3 + T2 * (2^32 - 1).
// Head: q = [T1:T0] - T3, a borrow (probability ~2^-33 per lane) repaid by -= EPS (== += p).  Branch-free form; outputs the
// two halves of q out of place so that they can be allocated as the register pair the tail's mad wants.
#define GL_ASM_HEAD                                                                               \
    "v_sub_co_u32 %[lo], vcc, %[p0], %[t3]\n\t"                                                   \
    "v_subbrev_co_u32 %[h], vcc, 0, %[t1], vcc\n\t"                                               \
    "v_cndmask_b32_e64 %[e], 0, -1, vcc\n\t"                                                      \
    "v_sub_co_u32 %[lo], vcc, %[lo], %[e]\n\t"                                                    \
    "v_subbrev_co_u32 %[h], vcc, 0, %[h], vcc"
// Tail: r = q + T2 * (2^32 - 1) is ONE v_mad_u64_u32 (64-bit addend, carry out in vcc); the carry (2^64 == 2^32 - 1, every
// other product) is added by a second one from the 0 / 1 lane mask: three instructions where the carry-chain form (sub,
// subbrev, add, cndmask, add, addc) needs six.  No second carry: the wrapped r is < (2^32 - 1)^2.
#define GL_ASM_TAIL                                                                               \
    "v_mad_u64_u32 %[r], vcc, %[t2], -1, %[r]\n\t"                                                \
    "v_cndmask_b32_e64 %[c], 0, 1, vcc\n\t"                                                       \
    "v_mad_u64_u32 %[r], vcc, %[c], -1, %[r]"

// Chained partial products (gl_prod128 below): each v_mad_u64_u32 takes the previous product's high part as its 64-bit
// addend, so the 128-bit product needs no carry adds at all.  History, measured with tools/ubench_mul.hip in cycles per
// wave-multiply per SIMD: four independent products + six carry adds 91.2; chained with {x, 0} addend pairs (one v_mov
// each) and two carry adds 72.7; the three-product squaring 81.4 -- so a square is just gl_mul(a, a).
// The fold's FIRST correction (the borrow of [T1:T0] - T3) happens with probability ~2^-33 per lane, so in gl_mul_fast its
// three instructions sit in an unlikely block entered only when some lane of the wave borrowed: the borrow mask leaves the
// asm in an SGPR pair, the test and branch run on the scalar unit, the common path falls through (74.7 -> 58.7 cycles with
// the carry-chain fold of r02; the fold itself is now GL_ASM_TAIL).
// Used where multiplies dominate and registers are not scarce: the Poseidon S-box (gl_mul_fast).
#define GL_FOLD_HEAD(lo, h, t1, t3, p0, bm)                                                                        \
    asm("v_sub_co_u32 %[l], vcc, %[p], %[z]\n\t"         /* [h:lo] = [T1:T0] - T3 */                               \
        "v_subbrev_co_u32 %[hh], %[b], 0, %[t], vcc"                                                               \
        : [l] "=&v"(lo), [hh] "=&v"(h), [b] "=&s"(bm)                                                              \
        : [p] "v"(p0), [z] "v"(t3), [t] "v"(t1)                                                                    \
        : "vcc");                                                                                                  \
    if (__builtin_expect(bm != 0, 0)) {                    /* borrow: -= EPS (== += p) */                          \
        u32 e_;                                                                                                    \
        asm("v_cndmask_b32_e64 %[e], 0, -1, %[b]\n\t"                                                              \
            "v_sub_co_u32 %[l], vcc, %[l], %[e]\n\t"                                                               \
            "v_subbrev_co_u32 %[hh], vcc, 0, %[hh], vcc"                                                           \
            : [l] "+&v"(lo), [hh] "+&v"(h), [e] "=&v"(e_)                                                          \
            : [b] "s"(bm)                                                                                          \
            : "vcc");                                                                                              \
    }

// The 128-bit product [T3:T2:T1:T0] of two u64 as a chain of four multiply-adds:
//     P = a0 b0;  M = a0 b1 + hi(P);  M2 = a1 b0 + M  (the WHOLE pair M as the addend: the low word is what a1 b0 + lo(M)
//     gives, the high part now also carries hi(M), and the sum's bit 64 leaves in the carry mask);  H = a1 b1 + [c : hi(M2)].
// Each {x, 0} addend pair costs a v_mov; taking M as it stands saves two of the three and the 64-bit add of hi(M) (r03r).
__device__ __forceinline__ void gl_prod128(u64 a, u64 b, u32 &t0, u32 &t1, u32 &t2, u32 &t3) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 P = (u64)a0 * b0;
    u64 M = (u64)a0 * b1 + (P >> 32);                  // <= (2^32-1)^2 + 2^32 - 1: no overflow
    u32 c;
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
        "v_cndmask_b32_e64 %1, 0, 1, vcc"
        : "+v"(M), "=&v"(c)
        : "v"(a1), "v"(b0)
        : "vcc");
    const u64 H = (u64)a1 * b1 + (((u64)c << 32) | (u32)(M >> 32));
    t0 = (u32)P; t1 = (u32)M; t2 = (u32)H; t3 = (u32)(H >> 32);
}

__device__ __forceinline__ u64 gl_mul_fast(u64 a, u64 b) {
    u32 t0, t1, t2, t3, lo, h, c;
    gl_prod128(a, b, t0, t1, t2, t3);
    u64 bm;
    GL_FOLD_HEAD(lo, h, t1, t3, t0, bm)
    u64 r = ((u64)h << 32) | lo;
    asm(GL_ASM_TAIL : [r] "+v"(r), [c] "=&v"(c) : [t2] "v"(t2) : "vcc");
    return r;
}

// The general-purpose multiply: branch-free (all eleven fold instructions inline, one asm statement).  Everything but the
// Poseidon S-box uses this one: the table AIRs hold hundreds of live values, and the unlikely blocks of gl_mul_fast cost
// the 86-column Cpu AIR its occupancy (4 -> 2 waves per SIMD, its quotient 7 -> 17 ms) and the Arithmetic AIR 15 %, while
// gaining nothing measurable elsewhere.
__device__ __forceinline__ u64 gl_mul(u64 a, u64 b) {
    u32 t0, t1, t2, t3, lo, h, e;
    gl_prod128(a, b, t0, t1, t2, t3);
    asm(GL_ASM_HEAD
        : [lo] "=&v"(lo), [h] "=&v"(h), [e] "=&v"(e)
        : [p0] "v"(t0), [t1] "v"(t1), [t3] "v"(t3)
        : "vcc");
    u64 r = ((u64)h << 32) | lo;
    asm(GL_ASM_TAIL : [r] "+v"(r), [c] "=&v"(e) : [t2] "v"(t2) : "vcc");
    return r;
}

__device__ __forceinline__ u64 gl_sqr(u64 a) { return gl_mul(a, a); }

// gl_mul with the result folded into [0, p) (two compares on top of the lazy form: the only non-canonical outputs of the
// reduce are 0xFFFFFFFF:lo with lo >= 1 from the no-carry branch, and adding EPS to those wraps them to 0:lo-1 -- the
// same add the carry branch needs, so the two conditions are merged on the scalar unit).  A canonical product lets the
// add / sub that consume it use ONE correction instead of two (gl_add_canon / gl_sub_canon): 11 + 4 + 4 full-rate
// instructions for a decimation-in-time butterfly (22 + 5 + 5 with the carry-chain folds of r02).
// (The first correction is skipped by a wave-uniform s_cbranch_vccz INSIDE the asm statement: as a compiler-visible
// branch it measured 2 % slower in the NTT's register step.)
__device__ __forceinline__ u64 gl_mul_canon(u64 a, u64 b) {
    u32 t0, t1, t2, t3, lo, h, e;
    gl_prod128(a, b, t0, t1, t2, t3);
    u64 sa;
    asm("v_sub_co_u32 %[lo], vcc, %[p0], %[t3]\n\t"
        "v_subbrev_co_u32 %[h], vcc, 0, %[t1], vcc\n\t"
        "s_cbranch_vccz 1f\n\t"                          /* no lane borrowed (all but ~2^-27 of the waves) */
        "v_cndmask_b32_e64 %[e], 0, -1, vcc\n\t"
        "v_sub_co_u32 %[lo], vcc, %[lo], %[e]\n\t"
        "v_subbrev_co_u32 %[h], vcc, 0, %[h], vcc\n"
        "1:"
        : [lo] "=&v"(lo), [h] "=&v"(h), [e] "=&v"(e)
        : [p0] "v"(t0), [t1] "v"(t1), [t3] "v"(t3)
        : "vcc");
    u64 r = ((u64)h << 32) | lo;
    // r += T2 * (2^32 - 1); then += EPS (mod 2^64) when that carried (2^64 == EPS) or when r >= p (r - p == r + EPS mod 2^64):
    // either way the result is < p
    asm("v_mad_u64_u32 %[r], vcc, %[t2], -1, %[r]\n\t"
        "v_cmp_gt_u64_e64 %[sa], %[r], %[pm1]\n\t"
        "s_or_b64 vcc, vcc, %[sa]\n\t"
        "v_cndmask_b32_e64 %[c], 0, 1, vcc\n\t"
        "v_mad_u64_u32 %[r], vcc, %[c], -1, %[r]"
        : [r] "+v"(r), [c] "=&v"(e), [sa] "=&s"(sa)
        : [t2] "v"(t2), [pm1] "s"(GL_P - 1)
        : "vcc", "scc");
    return r;
}

// a, b arbitrary u64 representatives; result arbitrary representative of a+b.  (The second wrap is as rare as gl_mul's first
// correction, but moving it into an unlikely block changed nothing measurable -- 909.5 vs 910.5 ms per segment, the AIR
// kernels if anything slower -- so add / sub keep the branch-free form.)
__device__ __forceinline__ u64 gl_add(u64 a, u64 b) {
    u32 lo, hi, c;
    u64 cm;                                             // carry mask, handed to the next statement in an SGPR pair
    asm("v_add_co_u32 %[lo], vcc, %[a0], %[b0]\n\t"
        "v_addc_co_u32 %[hi], %[cm], %[a1], %[b1], vcc"
        : [lo] "=&v"(lo), [hi] "=&v"(hi), [cm] "=&s"(cm)
        : [a0] "v"((u32)a), [a1] "v"((u32)(a >> 32)), [b0] "v"((u32)b), [b1] "v"((u32)(b >> 32))
        : "vcc");
    u64 r = ((u64)hi << 32) | lo;
    // a carry is 2^64 == EPS, added by a mad from the 0 / 1 lane mask (its own carry likewise: the second wrap happens
    // only if both inputs were >= 2^64 - 2^32)
    asm("v_cndmask_b32_e64 %[c], 0, 1, %[cm]\n\t"
        "v_mad_u64_u32 %[r], vcc, %[c], -1, %[r]\n\t"
        "v_cndmask_b32_e64 %[c], 0, 1, vcc\n\t"
        "v_mad_u64_u32 %[r], vcc, %[c], -1, %[r]"
        : [r] "+v"(r), [c] "=&v"(c)
        : [cm] "s"(cm)
        : "vcc");
    return r;
}
// b must be canonical (< p): one correction is enough.  The carry is repaid by += EPS = + 2^32 - 1:  lo -= carry (borrow k),
// hi += carry & ~k, the mask arithmetic on the scalar unit (as gl_sub_canon; the halves never have to form a register pair,
// which the mad form of gl_add costs the NTT butterfly two v_mov).
__device__ __forceinline__ u64 gl_add_canon(u64 a, u64 b) {
    u32 lo, hi;
    u64 m;
    asm("v_add_co_u32 %[lo], vcc, %[a0], %[b0]\n\t"
        "v_addc_co_u32 %[hi], %[m], %[a1], %[b1], vcc\n\t"
        "v_subbrev_co_u32 %[lo], vcc, 0, %[lo], %[m]\n\t"
        "s_andn2_b64 %[m], %[m], vcc\n\t"
        "v_addc_co_u32 %[hi], vcc, 0, %[hi], %[m]"
        : [lo] "=&v"(lo), [hi] "=&v"(hi), [m] "=&s"(m)
        : [a0] "v"((u32)a), [a1] "v"((u32)(a >> 32)), [b0] "v"((u32)b), [b1] "v"((u32)(b >> 32))
        : "vcc", "scc");
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 gl_sub(u64 a, u64 b) {
    u32 lo, hi, e;
    asm("v_sub_co_u32 %[lo], vcc, %[a0], %[b0]\n\t"
        "v_subb_co_u32 %[hi], vcc, %[a1], %[b1], vcc\n\t"
        "v_cndmask_b32_e64 %[e], 0, -1, vcc\n\t"
        "v_sub_co_u32 %[lo], vcc, %[lo], %[e]\n\t"
        "v_subbrev_co_u32 %[hi], vcc, 0, %[hi], vcc\n\t"
        "v_cndmask_b32_e64 %[e], 0, -1, vcc\n\t"   // second wrap: only if b > 2^64-2^32 and a tiny
        "v_sub_co_u32 %[lo], vcc, %[lo], %[e]\n\t"
        "v_subbrev_co_u32 %[hi], vcc, 0, %[hi], vcc"
        : [lo] "=&v"(lo), [hi] "=&v"(hi), [e] "=&v"(e)
        : [a0] "v"((u32)a), [a1] "v"((u32)(a >> 32)), [b0] "v"((u32)b), [b1] "v"((u32)(b >> 32))
        : "vcc");
    return ((u64)hi << 32) | lo;
}
// b must be <= p: one correction is enough.  The borrow is repaid by -= EPS = - 2^32 + 1:  lo += borrow (carry k),
// hi -= borrow & ~k, the mask arithmetic on the scalar unit: four VALU instructions where cndmask / sub / subbrev needs five.
// (The NTT butterfly's form.  The same trick in gl_sub and gl_mul's head cost the AIR kernels more in SGPR pairs than the
// instruction saved: 638 -> 645 ms per segment, r03n.)
__device__ __forceinline__ u64 gl_sub_canon(u64 a, u64 b) {
    u32 lo, hi;
    u64 m1;
    asm("v_sub_co_u32 %[lo], vcc, %[a0], %[b0]\n\t"
        "v_subb_co_u32 %[hi], %[m1], %[a1], %[b1], vcc\n\t"
        "v_addc_co_u32 %[lo], vcc, 0, %[lo], %[m1]\n\t"
        "s_andn2_b64 %[m1], %[m1], vcc\n\t"
        "v_subbrev_co_u32 %[hi], vcc, 0, %[hi], %[m1]"
        : [lo] "=&v"(lo), [hi] "=&v"(hi), [m1] "=&s"(m1)
        : [a0] "v"((u32)a), [a1] "v"((u32)(a >> 32)), [b0] "v"((u32)b), [b1] "v"((u32)(b >> 32))
        : "vcc", "scc");
    return ((u64)hi << 32) | lo;
}
#else
GL_HD u64 gl_add(u64 a, u64 b) { return gl_add_ref(a, b); }
GL_HD u64 gl_add_canon(u64 a, u64 b) { return gl_add_ref(a, b); }
GL_HD u64 gl_sub(u64 a, u64 b) { return gl_sub_ref(a, b); }
GL_HD u64 gl_mul(u64 a, u64 b) { return gl_mul_ref(a, b); }
GL_HD u64 gl_sqr(u64 a) { return gl_mul_ref(a, a); }
GL_HD u64 gl_mul_fast(u64 a, u64 b) { return gl_mul_ref(a, b); }
GL_HD u64 gl_mul_canon(u64 a, u64 b) { return gl_canon(gl_mul_ref(a, b)); }
GL_HD u64 gl_sub_canon(u64 a, u64 b) { return gl_sub_ref(a, b); }
#endif
GL_HD u64 gl_neg(u64 a) { return gl_sub(0, a); }

GL_HD u64 gl_pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, b);
        b = gl_sqr(b);
        e >>= 1;
    }
    return r;
}
GL_HD u64 gl_sqr_n(u64 x, int n) {
#pragma unroll 1
    for (int i = 0; i < n; ++i) x = gl_sqr(x);
    return x;
}
// a^(p - 2) by an addition chain: p - 2 = 2^64 - 2^32 - 1 = (2^31 - 1) 2^33 + (2^32 - 1), and a^(2^k - 1) doubles its k with
// k squarings and one multiply -- 64 squarings + 10 multiplies where square-and-multiply over the 63 one bits needs 64 + 62.
GL_HD u64 gl_inv(u64 a) {
    const u64 t2 = gl_mul(gl_sqr(a), a);                 // a^(2^2 - 1)
    const u64 t4 = gl_mul(gl_sqr_n(t2, 2), t2);
    const u64 t8 = gl_mul(gl_sqr_n(t4, 4), t4);
    const u64 t16 = gl_mul(gl_sqr_n(t8, 8), t8);
    const u64 t24 = gl_mul(gl_sqr_n(t16, 8), t8);
    const u64 t28 = gl_mul(gl_sqr_n(t24, 4), t4);
    const u64 t30 = gl_mul(gl_sqr_n(t28, 2), t2);
    const u64 t31 = gl_mul(gl_sqr(t30), a);              // a^(2^31 - 1)
    const u64 t32 = gl_mul(gl_sqr(t31), a);              // a^(2^32 - 1)
    return gl_mul(gl_sqr_n(t31, 33), t32);
}
GL_HD u64 gl_root_of_unity(unsigned log_n) {
    u64 r = GL_POW2_GENERATOR;
    for (unsigned i = log_n; i < 32; ++i) r = gl_sqr(r);
    return gl_canon(r);
}

// ---- quadratic extension F[X]/(X^2 - 7) ------------------------------------------------------
struct gl2 {
    u64 a, b;  // a + b X
};
GL_HD gl2 gl2_make(u64 a, u64 b) { gl2 r; r.a = a; r.b = b; return r; }
GL_HD gl2 gl2_add(gl2 x, gl2 y) { return gl2_make(gl_add(x.a, y.a), gl_add(x.b, y.b)); }
GL_HD gl2 gl2_sub(gl2 x, gl2 y) { return gl2_make(gl_sub(x.a, y.a), gl_sub(x.b, y.b)); }
GL_HD u64 gl_mul7(u64 x) {  // 7x = 8x - x, as a 67-bit fold
    u32 hi = (u32)(x >> 61);
    u64 lo = x << 3;
    return gl_sub(gl_reduce96(hi, lo), x);
}
GL_HD gl2 gl2_mul(gl2 x, gl2 y) {
    u64 aa = gl_mul(x.a, y.a), bb = gl_mul(x.b, y.b);
    u64 ab = gl_mul(x.a, y.b), ba = gl_mul(x.b, y.a);
    return gl2_make(gl_add(aa, gl_mul7(bb)), gl_add(ab, ba));
}
GL_HD gl2 gl2_scale(gl2 x, u64 s) { return gl2_make(gl_mul(x.a, s), gl_mul(x.b, s)); }
GL_HD gl2 gl2_canon(gl2 x) { return gl2_make(gl_canon(x.a), gl_canon(x.b)); }

GL_HD u32 bitrev32(u32 x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
    u32 r = 0;
    for (unsigned i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
#endif
}
