// Arithmetic table witness generation on the device: one thread per operation restates
// `Operation::to_rows` (evm_arithmetization/src/arithmetic/mod.rs:253-359) and the per-operation generators
// addcy.rs:31-65, mul.rs:72-121, modular.rs:211-382, divmod.rs:24-84, shift.rs:41-85, byte.rs:101-200 with
// fixed-width integers: 256-bit operands as 4 x u64, the 512-bit dividend of the modular operations as 8 x u64
// (binary long division), the reference's i64 limb polynomials as i64 arrays.  The range-check columns are added
// afterwards by range_counter_kernel / range_histogram_kernel (arithmetic_stark.rs:130-156).
#pragma once
#include "gl.cuh"

namespace arith {

// arithmetic/columns.rs:25-118
enum : int {
    IS_ADD = 0, IS_MUL, IS_SUB, IS_DIV, IS_MOD, IS_ADDMOD, IS_MULMOD, IS_ADDFP254, IS_MULFP254, IS_SUBFP254, IS_SUBMOD,
    IS_LT, IS_GT, IS_BYTE, IS_SHL, IS_SHR, IS_RANGE_CHECK, OPCODE_COL,
    IN0 = 18, IN1 = 34, IN2 = 50, OUT = 66, AUX0 = 82, AUX1 = 98, NUM_COLS = 116, NL = 16
};
#define ARITH_AUX_ABS_MAX (1ll << 20)

struct U256 { u64 w[4]; };

__device__ __forceinline__ bool u256_is_zero(const U256 &a) { return (a.w[0] | a.w[1] | a.w[2] | a.w[3]) == 0; }
__device__ __forceinline__ bool u256_lt(const U256 &a, const U256 &b) {
    for (int i = 3; i >= 0; --i)
        if (a.w[i] != b.w[i]) return a.w[i] < b.w[i];
    return false;
}
__device__ __forceinline__ u64 u256_add(U256 &r, const U256 &a, const U256 &b) {   // returns the carry
    u64 c = 0;
    for (int i = 0; i < 4; ++i) {
        const u64 s = a.w[i] + b.w[i], s2 = s + c;
        c = (s < a.w[i]) | (s2 < s);
        r.w[i] = s2;
    }
    return c;
}
__device__ __forceinline__ u64 u256_sub(U256 &r, const U256 &a, const U256 &b) {   // wrapping; returns the borrow
    u64 br = 0;
    for (int i = 0; i < 4; ++i) {
        const u64 d = a.w[i] - b.w[i], d2 = d - br;
        br = (a.w[i] < b.w[i]) | (d < br);
        r.w[i] = d2;
    }
    return br;
}
__device__ __forceinline__ i64 limb16(const u64 *w, int i) { return (i64)((w[i >> 2] >> (16 * (i & 3))) & 0xFFFF); }
__device__ __forceinline__ void put256(u64 *row, int start, const U256 &v) {
    for (int i = 0; i < NL; ++i) row[start + i] = (u64)limb16(v.w, i);
}
__device__ __forceinline__ u64 fwrap(i64 c) { return c < 0 ? GL_P - (u64)(-c) : (u64)c; }

// 256 x 256 -> 512
__device__ inline void mul_256(u64 *p, const U256 &a, const U256 &b) {
    for (int i = 0; i < 8; ++i) p[i] = 0;
    for (int i = 0; i < 4; ++i) {
        u64 carry = 0;
        for (int j = 0; j < 4; ++j) {
            const unsigned __int128 t = (unsigned __int128)a.w[i] * b.w[j] + p[i + j] + carry;
            p[i + j] = (u64)t;
            carry = (u64)(t >> 64);
        }
        p[i + 4] = carry;
    }
}

// q (8 words), r = num / d, num % d; d != 0
__device__ inline void divmod_512(u64 *q, U256 &r, const u64 *num, const U256 &d) {
    u64 rem[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) q[i] = 0;
    int top = 511;
    while (top >= 0 && !((num[top >> 6] >> (top & 63)) & 1)) --top;
    for (int bit = top; bit >= 0; --bit) {
        for (int i = 4; i > 0; --i) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);
        rem[0] = (rem[0] << 1) | ((num[bit >> 6] >> (bit & 63)) & 1);
        bool ge = rem[4] != 0;
        if (!ge) {
            ge = true;
            for (int i = 3; i >= 0; --i)
                if (rem[i] != d.w[i]) { ge = rem[i] > d.w[i]; break; }
        }
        if (ge) {
            u64 br = 0;
            for (int i = 0; i < 4; ++i) {
                const u64 x = rem[i] - d.w[i], x2 = x - br;
                br = (rem[i] < d.w[i]) | (x < br);
                rem[i] = x2;
            }
            rem[4] -= br;
            q[bit >> 6] |= 1ull << (bit & 63);
        }
    }
    for (int i = 0; i < 4; ++i) r.w[i] = rem[i];
}

// utils.rs:278-302 pol_remove_root_2exp::<16>: q[0..len-2]
__device__ __forceinline__ void remove_root_2exp(i64 *q, const i64 *a, int len) {
    q[0] = -(a[0] >> 16);
    for (int d = 1; d < len - 1; ++d) q[d] = (q[d - 1] - a[d]) >> 16;
}

// addcy.rs:31-65
__device__ inline void gen_addcy(u64 *row, int filt, const U256 &a, const U256 &b) {
    put256(row, IN0, a);
    put256(row, IN1, b);
    U256 r, z = {{0, 0, 0, 0}};
    if (filt == IS_ADD) {
        z.w[0] = u256_add(r, a, b);
        put256(row, AUX0, z); put256(row, OUT, r);
    } else if (filt == IS_SUB) {
        z.w[0] = u256_sub(r, a, b);
        put256(row, AUX0, z); put256(row, OUT, r);
    } else if (filt == IS_LT) {
        z.w[0] = u256_sub(r, a, b);
        put256(row, AUX0, r); put256(row, OUT, z);
    } else {
        z.w[0] = u256_sub(r, b, a);
        put256(row, AUX0, r); put256(row, OUT, z);
    }
}

// mul.rs:72-109; left / right are the limbs already in the row
__device__ inline void gen_mul_limbs(u64 *row, int left, int right) {
    i64 prod[NL], aux[NL];
    i64 cy = 0;
    for (int d = 0; d < NL; ++d) {
        i64 s = 0;
        for (int i = 0; i <= d; ++i) s += (i64)row[left + i] * (i64)row[right + d - i];
        const i64 t = s + cy;
        cy = t >> 16;
        row[OUT + d] = (u64)(t & 0xFFFF);
        prod[d] = s - (t & 0xFFFF);
    }
    remove_root_2exp(aux, prod, NL);
    aux[NL - 1] = -cy;
    for (int i = 0; i < NL; ++i) {
        const i64 c = aux[i] + ARITH_AUX_ABS_MAX;
        row[AUX0 + i] = (u64)(c & 0xFFFF);
        row[AUX1 + i] = (u64)((c >> 16) & 0xFFFF);
    }
}

// modular.rs:211-341.  pol: the 2N-1 input coefficients; (mag, neg): the integer they evaluate to at 2^16;
// modulus: the value of the limbs at lv[mod_start..].  Writes nv and returns output / quotient limbs.
__device__ inline void gen_modular_op(const u64 *lv, u64 *nv, int filt, const i64 *pol, const u64 *mag, bool neg, U256 modulus,
                                      i64 *out_limbs, i64 *quo_limbs) {
    i64 mod_limbs[NL];
    for (int i = 0; i < NL; ++i) mod_limbs[i] = limb16(modulus.w, i);
    const bool mod_is_zero = u256_is_zero(modulus);
    bool big_mod = false;                       // modulus = 2^256 (DIV / SHR by zero)
    if (mod_is_zero) {
        if (filt == IS_DIV || filt == IS_SHR) big_mod = true;
        else { modulus.w[0] = 1; mod_limbs[0] = 1; }
    }
    U256 output;
    u64 q[8];
    bool qneg = false;
    if (big_mod) {
        for (int i = 0; i < 4; ++i) output.w[i] = mag[i];
        for (int i = 0; i < 8; ++i) q[i] = 0;
    } else {
        U256 r;
        divmod_512(q, r, mag, modulus);
        if (!neg) output = r;
        else {                                  // input = -mag: the non-negative residue and the floor quotient
            qneg = true;
            if (u256_is_zero(r)) output = r;
            else {
                u256_sub(output, modulus, r);
                for (int i = 0; i < 8; ++i) if (++q[i]) break;
            }
            bool qz = true;
            for (int i = 0; i < 8; ++i) qz &= q[i] == 0;
            if (qz) qneg = false;
        }
    }
    for (int i = 0; i < NL; ++i) out_limbs[i] = limb16(output.w, i);
    for (int i = 0; i < 2 * NL; ++i) quo_limbs[i] = qneg ? -limb16(q, i) : limb16(q, i);
    U256 red;                                   // 2^256 - modulus + output
    if (big_mod) red = output; else u256_sub(red, output, modulus);
    i64 constr[2 * NL], aux[2 * NL];
    for (int k = 0; k < 2 * NL; ++k) {
        i64 c = k < 2 * NL - 1 ? pol[k] : 0;
        if (k < NL) c -= out_limbs[k];
        const int j0 = k - (2 * NL - 1) > 0 ? k - (2 * NL - 1) : 0;
        for (int j = j0; j < NL && j <= k; ++j) c -= quo_limbs[k - j] * mod_limbs[j];
        constr[k] = c;
    }
    remove_root_2exp(aux, constr, 2 * NL);
    for (int i = 0; i < 2 * NL - 1; ++i) {
        const i64 c = aux[i] + ARITH_AUX_ABS_MAX;
        nv[35 + i] = (u64)(c & 0xFFFF);          // MODULAR_AUX_INPUT_LO
        nv[66 + i] = (u64)((c >> 16) & 0xFFFF);  // MODULAR_AUX_INPUT_HI
    }
    if (filt == IS_SUBMOD || filt == IS_SUBFP254) {
        if (qneg) {
            for (int i = 0; i < NL; ++i) quo_limbs[i] += 0xFFFF;
            quo_limbs[NL] = 1;
        } else quo_limbs[NL] = 0;
    }
    nv[34] = mod_is_zero;                        // MODULAR_MOD_IS_ZERO
    for (int i = 0; i < NL; ++i) nv[18 + i] = (u64)limb16(red.w, i);   // MODULAR_OUT_AUX_RED
    nv[97] = mod_is_zero ? lv[IS_DIV] + lv[IS_SHR] : 0;               // MODULAR_DIV_DENOM_IS_ZERO
}

// modular.rs:343-382
__device__ inline void gen_modular(u64 *lv, u64 *nv, int filt, const U256 &a, const U256 &b, const U256 &m) {
    put256(lv, IN0, a); put256(lv, IN1, b); put256(lv, IN2, m);
    i64 pol[2 * NL - 1];
    u64 mag[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool neg = false;
    if (filt == IS_ADDMOD || filt == IS_ADDFP254) {
        for (int i = 0; i < 2 * NL - 1; ++i) pol[i] = i < NL ? limb16(a.w, i) + limb16(b.w, i) : 0;
        U256 s;
        mag[4] = u256_add(s, a, b);
        for (int i = 0; i < 4; ++i) mag[i] = s.w[i];
    } else if (filt == IS_SUBMOD || filt == IS_SUBFP254) {
        for (int i = 0; i < 2 * NL - 1; ++i) pol[i] = i < NL ? limb16(a.w, i) - limb16(b.w, i) : 0;
        U256 d;
        neg = u256_lt(a, b);
        if (neg) u256_sub(d, b, a); else u256_sub(d, a, b);
        for (int i = 0; i < 4; ++i) mag[i] = d.w[i];
    } else {
        for (int k = 0; k < 2 * NL - 1; ++k) {
            i64 s = 0;
            for (int i = (k >= NL ? k - NL + 1 : 0); i < NL && i <= k; ++i) s += limb16(a.w, i) * limb16(b.w, k - i);
            pol[k] = s;
        }
        mul_256(mag, a, b);
    }
    i64 out[NL], quo[2 * NL];
    gen_modular_op(lv, nv, filt, pol, mag, neg, m, out, quo);
    for (int i = 0; i < NL; ++i) lv[OUT + i] = fwrap(out[i]);
    for (int i = 0; i < 2 * NL; ++i) lv[AUX0 + i] = fwrap(quo[i]);     // MODULAR_QUO_INPUT
}

// divmod.rs:24-66: lv[OUT] already holds the result; num / modulus are the values at in_start / mod_start
__device__ inline void gen_divmod_regs(u64 *lv, u64 *nv, int filt, const U256 &num, const U256 &modulus) {
    i64 pol[2 * NL - 1];
    for (int i = 0; i < 2 * NL - 1; ++i) pol[i] = i < NL ? limb16(num.w, i) : 0;
    u64 mag[8] = {num.w[0], num.w[1], num.w[2], num.w[3], 0, 0, 0, 0};
    i64 out[NL], quo[2 * NL];
    gen_modular_op(lv, nv, filt, pol, mag, false, modulus, out, quo);
    for (int i = 0; i < 2 * NL; ++i) lv[AUX0 + i] = 0;
    const i64 *src = (filt == IS_DIV || filt == IS_SHR) ? out : quo;
    for (int i = 0; i < NL; ++i) lv[AUX0 + i] = fwrap(src[i]);
}

// byte.rs:109-200
__device__ inline void gen_byte(u64 *row, const U256 &idx, const U256 &val) {
    put256(row, IN0, idx); put256(row, IN1, val);
    const u64 i0 = idx.w[0];
    for (int i = 0; i < 5; ++i) row[AUX0 + i] = (i0 >> i) & 1;
    row[AUX0 + 5] = (i0 & 0xFFFF) >> 5;
    u64 hi_sum = row[AUX0 + 5];
    for (int i = 1; i < NL; ++i) hi_sum += row[IN0 + i];
    const u64 inv = hi_sum ? gl_canon(gl_inv(hi_sum)) : 1;
    for (int k = 0; k < 4; ++k) row[91 + k] = (inv >> (16 * k)) & 0xFFFF;
    row[90] = hi_sum ? 1 : 0;
    int i = 3, src = IN1, dest = AUX1;
    for (;;) {
        const int lvl = 1 << i;
        src += ((i0 >> (i + 1)) & 1) ? 0 : lvl;
        for (int k = 0; k < lvl; ++k) row[dest + k] = row[src + k];
        if (i == 0) break;
        src = dest; dest += lvl; --i;
    }
    const u64 t = row[dest], lo = t & 0xFF, hi = t >> 8;
    row[88] = lo << 8;
    row[89] = hi;
    const u64 out = (i0 & 1) ? lo : hi;
    row[AUX1 + 15] = out;
    const bool small = idx.w[1] == 0 && idx.w[2] == 0 && idx.w[3] == 0 && i0 < 32;
    for (int k = 0; k < NL; ++k) row[OUT + k] = 0;
    row[OUT] = small ? out : 0;
}

__device__ __forceinline__ u32 rows_of(int code) {
    return (code == IS_DIV || code == IS_MOD || code == IS_SHR || code == IS_ADDMOD || code == IS_MULMOD ||
            code == IS_SUBMOD || code == IS_ADDFP254 || code == IS_MULFP254 || code == IS_SUBFP254) ? 2u : 1u;
}

}  // namespace arith

// ops: n_ops x 18 words = code (the IS_* column of the operation, 16 = range check), opcode (range check only),
// input0, input1, input2, result (range check only) as four 64-bit little-endian limbs each; row_of[i]: first row.
static __global__ void __launch_bounds__(64)
arithmetic_trace_kernel(const u64 *__restrict__ ops, const u32 *__restrict__ row_of, const u32 *__restrict__ order, u32 n_ops,
                        u64 *__restrict__ out, size_t cs) {
    using namespace arith;
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n_ops) return;
    const u32 t = order[tid];
    const u64 *o = ops + (size_t)t * 18;
    const int code = (int)o[0];
    U256 a, b, c, res;
    for (int i = 0; i < 4; ++i) { a.w[i] = o[2 + i]; b.w[i] = o[6 + i]; c.w[i] = o[10 + i]; res.w[i] = o[14 + i]; }
    u64 r1[NUM_COLS], r2[NUM_COLS];
    for (int i = 0; i < NUM_COLS; ++i) r1[i] = r2[i] = 0;
    if (code == IS_RANGE_CHECK) {
        r1[IS_RANGE_CHECK] = 1;
        r1[OPCODE_COL] = o[1];
        put256(r1, IN0, a); put256(r1, IN1, b); put256(r1, IN2, c); put256(r1, OUT, res);
    } else {
        r1[code] = 1;
        if (code == IS_ADD || code == IS_SUB || code == IS_LT || code == IS_GT) gen_addcy(r1, code, a, b);
        else if (code == IS_MUL) {
            put256(r1, IN0, a); put256(r1, IN1, b);
            gen_mul_limbs(r1, IN0, IN1);
        } else if (code == IS_DIV || code == IS_MOD) {
            u64 q[8];
            U256 r = {{0, 0, 0, 0}}, quot = {{0, 0, 0, 0}};
            if (!u256_is_zero(b)) {
                const u64 num[8] = {a.w[0], a.w[1], a.w[2], a.w[3], 0, 0, 0, 0};
                divmod_512(q, r, num, b);
                for (int i = 0; i < 4; ++i) quot.w[i] = q[i];
            }
            put256(r1, IN0, a); put256(r1, IN1, b); put256(r1, OUT, code == IS_DIV ? quot : r);
            gen_divmod_regs(r1, r2, code, a, b);
        } else if (code == IS_SHL || code == IS_SHR) {
            const bool big = a.w[1] || a.w[2] || a.w[3] || a.w[0] > 255;     // shift = input0, value = input1
            const u32 sh = (u32)a.w[0];
            U256 pw = {{0, 0, 0, 0}}, r = {{0, 0, 0, 0}};
            if (!big) {
                pw.w[sh >> 6] = 1ull << (sh & 63);
                const u32 ws = sh >> 6, bs = sh & 63;
                for (int i = 0; i < 4; ++i) {
                    if (code == IS_SHL) {
                        const int s = i - (int)ws;
                        u64 v = s >= 0 ? b.w[s] << bs : 0;
                        if (bs && s - 1 >= 0) v |= b.w[s - 1] >> (64 - bs);
                        r.w[i] = v;
                    } else {
                        const u32 s = i + ws;
                        u64 v = s < 4 ? b.w[s] >> bs : 0;
                        if (bs && s + 1 < 4) v |= b.w[s + 1] << (64 - bs);
                        r.w[i] = v;
                    }
                }
            }
            put256(r1, IN0, a); put256(r1, IN1, b); put256(r1, OUT, r); put256(r1, IN2, pw);
            if (code == IS_SHL) gen_mul_limbs(r1, IN1, IN2);
            else gen_divmod_regs(r1, r2, IS_SHR, b, pw);
        } else if (code == IS_BYTE) gen_byte(r1, a, b);
        else {
            U256 m = c;
            if (code == IS_ADDFP254 || code == IS_MULFP254 || code == IS_SUBFP254)
                m = U256{{0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};   // BN_BASE
            gen_modular(r1, r2, code, a, b, m);
        }
    }
    const u32 row = row_of[t];
    for (int i = 0; i < NUM_COLS; ++i) out[(size_t)i * cs + row] = r1[i];
    if (rows_of(code) == 2)
        for (int i = 0; i < NUM_COLS; ++i) out[(size_t)i * cs + row + 1] = r2[i];
}
