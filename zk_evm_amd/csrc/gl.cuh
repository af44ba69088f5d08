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

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL
#define GL_GENERATOR 14293326489335486720ULL
#define GL_POW2_GENERATOR 7277203076849721926ULL

#define GL_HD __host__ __device__ __forceinline__

GL_HD u64 gl_canon(u64 a) { return a >= GL_P ? a - GL_P : a; }

// a, b arbitrary u64 representatives; result arbitrary representative of a+b.
GL_HD u64 gl_add(u64 a, u64 b) {
    u64 s = a + b;
    u64 c = s < a ? GL_EPS : 0;
    u64 s2 = s + c;
    u64 c2 = s2 < s ? GL_EPS : 0;  // only reachable when both inputs were >= 2^64 - 2^32
    return s2 + c2;
}
// b must be canonical (< p): one correction is enough.
GL_HD u64 gl_add_canon(u64 a, u64 b) {
    u64 s = a + b;
    return s + (s < a ? GL_EPS : 0);
}
GL_HD u64 gl_sub(u64 a, u64 b) {
    u64 d = a - b;
    u64 br = a < b ? GL_EPS : 0;
    u64 d2 = d - br;
    u64 br2 = d2 > d ? GL_EPS : 0;
    return d2 - br2;
}
GL_HD u64 gl_neg(u64 a) { return gl_sub(0, a); }

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

GL_HD u64 gl_mul(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 mid = (u64)a0 * b1 + (p00 >> 32);          // <= (2^32-1)^2 + 2^32 - 1 : no overflow
    u64 mid2 = (u64)a1 * b0 + (u32)mid;            // likewise
    u64 hi = (u64)a1 * b1 + (mid >> 32) + (mid2 >> 32);
    u64 lo = (mid2 << 32) | (u32)p00;
    return gl_reduce128(hi, lo);
}
GL_HD u64 gl_sqr(u64 a) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32);
    u64 p00 = (u64)a0 * a0;
    u64 p01 = (u64)a0 * a1;
    u64 mid = p01 + (p00 >> 32);
    u64 mid2 = p01 + (u32)mid;
    u64 hi = (u64)a1 * a1 + (mid >> 32) + (mid2 >> 32);
    u64 lo = (mid2 << 32) | (u32)p00;
    return gl_reduce128(hi, lo);
}

GL_HD u64 gl_pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, b);
        b = gl_sqr(b);
        e >>= 1;
    }
    return r;
}
GL_HD u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }
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
