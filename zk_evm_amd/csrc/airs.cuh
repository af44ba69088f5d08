// Table AIRs: device restatements of each table's `Stark::eval_packed_generic` from the reference
// tree (evm_arithmetization/src/*/..._stark.rs).  Constraints are yielded in the reference's order
// (the Horner accumulation in ConstraintConsumer makes the order parity-critical).
// Column indices follow the reference's `#[repr(C)]` column structs.
#pragma once
#include "quotient.cuh"
#include "poseidon.cuh"

// no table constraints (lookup / CTL checks only) -- used by the generic-machinery tests
struct AirNone {
    static constexpr u32 COLUMNS = 0;
    template <class CONS>
    __device__ static __forceinline__ void eval(const RowView &, const RowView &, CONS &, const u64 *) {}
};

// MemoryContinuationStark (MemBefore / MemAfter): memory_continuation/memory_continuation_stark.rs:110-122,
// columns memory_continuation/columns.rs:7-23 (FILTER = 0, 12 columns).
struct AirMemContinuation {
    static constexpr u32 COLUMNS = 12;
    template <class CONS>
    __device__ static __forceinline__ void eval(const RowView &lv, const RowView &, CONS &c, const u64 *) {
        Fe filter = lv[0];
        c.constraint(filter * (filter - FE_ONE));  // the filter must be binary
    }
};

// LogicStark: logic.rs:249-303; columns logic.rs:46-71: op {is_and, is_or, is_xor} = 0..2,
// input0 bits 3..258, input1 bits 259..514, result limbs 515..522 (8 x 32-bit).
struct AirLogic {
    static constexpr u32 COLUMNS = 523;
    template <class CONS>
    __device__ static __forceinline__ void eval(const RowView &lv, const RowView &, CONS &c, const u64 *) {
        constexpr u32 IN0 = 3, IN1 = 3 + 256, RES = 3 + 512;
        Fe is_and = lv[0], is_or = lv[1], is_xor = lv[2];
        c.constraint(is_and * (is_and - FE_ONE));
        c.constraint(is_or * (is_or - FE_ONE));
        c.constraint(is_xor * (is_xor - FE_ONE));
        Fe all_flags = is_and + is_or + is_xor;
        c.constraint(all_flags * (all_flags - FE_ONE));
        Fe sum_coeff = is_or + is_xor;
        Fe and_coeff = is_and - is_or - is_xor * fe(2);
        // The 512 bit checks and the 8 limb recompositions read the same 512 bit columns: one visit (constraint_at keeps the
        // reference's positions: 256 + 256 bit checks, then the limbs), the three sums of a limb by Horner from the top bit
        // instead of a multiply by 2^i per term -- the same field values.
        for (u32 limb = 0; limb < 8; ++limb) {
            Fe x, y, x_land_y;
            for (int i = 31; i >= 0; --i) {
                const u32 k = 32 * limb + (u32)i;
                const Fe xb = lv[IN0 + k], yb = lv[IN1 + k];
                c.constraint_at(k, xb * (xb - FE_ONE));
                c.constraint_at(256 + k, yb * (yb - FE_ONE));
                x = x + x + xb;
                y = y + y + yb;
                x_land_y = x_land_y + x_land_y + xb * yb;
            }
            const Fe x_op_y = sum_coeff * (x + y) + and_coeff * x_land_y;
            c.constraint_at(512 + limb, lv[RES + limb] - x_op_y);
        }
        c.advance(512 + 8);
    }
};

// MemoryStark: memory/memory_stark.rs:474-626; columns memory/columns.rs:13-94 (30 columns, in
// `#[repr(C)]` order: filter 0, timestamp 1, timestamp_inv 2, is_read 3, addr_context 4,
// addr_segment 5, addr_virtual 6, value_limbs 7..14, context/segment/virtual_first_change 15..17,
// initialize_aux 18, preinitialized_segments 19, preinitialized_segments_aux 20, stale_contexts 21,
// is_pruned 22, stale_context_frequencies 23, is_stale 24, maybe_in_mem_after 25,
// mem_after_filter 26, range_check 27, counter 28, frequencies 29).
struct AirMemory {
    static constexpr u32 COLUMNS = 30;
    template <class CONS>
    __device__ static __forceinline__ void eval(const RowView &lv, const RowView &nv, CONS &c, const u64 *) {
        // Segment ids, unscaled (memory/segments.rs:14,41,79,81)
        constexpr u64 SEG_CODE = 0, SEG_TRIE_DATA = 12, SEG_ACCOUNTS_LL = 34, SEG_STORAGE_LL = 35;
        const Fe one = FE_ONE;
        Fe timestamp = lv[1], addr_context = lv[4], addr_segment = lv[5], addr_virtual = lv[6];
        Fe timestamp_inv = lv[2], is_stale = lv[24], maybe_in_mem_after = lv[25], mem_after_filter = lv[26];
        Fe initialize_aux = lv[18], preinit = lv[19], preinit_aux = lv[20];
        Fe next_timestamp = nv[1], next_is_read = nv[3];
        Fe next_addr_context = nv[4], next_addr_segment = nv[5], next_addr_virtual = nv[6];
        Fe filter = lv[0];
        c.constraint(filter * (filter - one));
        Fe is_dummy = one - filter, is_write = one - lv[3];
        c.constraint(is_dummy * is_write);
        Fe cfc = lv[15], sfc = lv[16], vfc = lv[17];
        Fe address_unchanged = one - cfc - sfc - vfc;
        Fe range_check = lv[27];
        Fe not_cfc = one - cfc, not_sfc = one - sfc, not_vfc = one - vfc, not_au = one - address_unchanged;
        c.constraint(cfc * not_cfc);
        c.constraint(sfc * not_sfc);
        c.constraint(vfc * not_vfc);
        c.constraint(address_unchanged * not_au);
        c.constraint_transition(sfc * (next_addr_context - addr_context));
        c.constraint_transition(vfc * (next_addr_context - addr_context));
        c.constraint_transition(vfc * (next_addr_segment - addr_segment));
        c.constraint_transition(address_unchanged * (next_addr_context - addr_context));
        c.constraint_transition(address_unchanged * (next_addr_segment - addr_segment));
        c.constraint_transition(address_unchanged * (next_addr_virtual - addr_virtual));
        Fe computed_range_check = cfc * (next_addr_context - addr_context - one) +
                                  sfc * (next_addr_segment - addr_segment - one) +
                                  vfc * (next_addr_virtual - addr_virtual - one) +
                                  address_unchanged * (next_timestamp - timestamp);
        c.constraint_transition(range_check - computed_range_check);
        c.constraint_transition(preinit_aux - (next_addr_segment - fe(SEG_ACCOUNTS_LL)) * (next_addr_segment - fe(SEG_STORAGE_LL)));
        c.constraint_transition(preinit - (next_addr_segment - fe(SEG_CODE)) * (next_addr_segment - fe(SEG_TRIE_DATA)) * preinit_aux);
        c.constraint_transition(initialize_aux - preinit * not_au * next_is_read);
        for (u32 i = 0; i < 8; ++i) {
            c.constraint_transition(next_is_read * address_unchanged * (nv[7 + i] - lv[7 + i]));
            c.constraint_transition(initialize_aux * nv[7 + i]);
        }
        c.constraint_transition(maybe_in_mem_after + filter * not_au * (is_stale - one));
        c.constraint(mem_after_filter * (mem_after_filter - one));
        for (u32 i = 0; i < 8; ++i) c.constraint((mem_after_filter - maybe_in_mem_after) * preinit * lv[7 + i]);
        c.constraint(timestamp * (timestamp * timestamp_inv - one));
        Fe rc1 = lv[28], rc2 = nv[28];
        c.constraint_first_row(rc1);
        c.constraint_transition(rc2 - rc1 - one);
    }
};

// BytePackingStark: byte_packing/byte_packing_stark.rs:296-352; columns byte_packing/columns.rs:12-40
// (is_read 0, index_len 1..32, addr_context 33, addr_segment 34, addr_virtual 35, timestamp 36,
// value_bytes 37..68, range_counter 69, rc_frequencies 70).
struct AirBytePacking {
    static constexpr u32 COLUMNS = 71;
    template <class CONS>
    __device__ static __forceinline__ void eval(const RowView &lv, const RowView &nv, CONS &c, const u64 *) {
        // Every column is loaded ONCE: the 32 value bytes stay in registers for the 496 pairwise products
        // index_len[i] * value_bytes[j] (j > i), which are yielded as 31 scaled runs (one dot product over the bytes per
        // i and challenge, one multiply by index_len[i]) at their reference positions; the order-free constraint_at puts
        // everything else at its position too.  Positions (byte_packing_stark.rs:296-352): 0 first_row(rc), 1 transition
        // (incr), 2 last_row(rc), 3 filter binary, 4 first_row(filter - 1), 5 is_read binary, 6..37 index_len binary,
        // 38 transition(next_filter), 39..534 the pairs -- 535 in all.
        constexpr u32 NUM_BYTES = 32, IDX = 1, VAL = 37;
        const Fe one = FE_ONE;
        Fe val[NUM_BYTES];
#pragma unroll
        for (u32 j = 0; j < NUM_BYTES; ++j) val[j] = lv[VAL + j];
        Fe current_filter;
        u32 pair_pos = 39;
#pragma unroll
        for (u32 i = 0; i < NUM_BYTES; ++i) {
            const Fe idx = lv[IDX + i];
            current_filter += idx;
            c.constraint_at(6 + i, idx * (idx - one));
            if (i + 1 < NUM_BYTES) c.template constraint_scaled_run_at<NUM_BYTES>(pair_pos, idx, val, (int)i + 1);
            pair_pos += NUM_BYTES - 1 - i;
        }
        Fe next_filter;
#pragma unroll
        for (u32 i = 0; i < NUM_BYTES; ++i) next_filter += nv[IDX + i];
        const Fe rc1 = lv[69], rc2 = nv[69];
        const Fe incr = rc2 - rc1;
        const Fe is_read = lv[0];
        c.constraint_at_first_row(0, rc1);
        c.constraint_at_transition(1, incr * incr - incr);
        c.constraint_at_last_row(2, rc1 - fe(255));  // BYTE_RANGE_MAX - 1
        c.constraint_at(3, current_filter * (current_filter - one));
        c.constraint_at_first_row(4, current_filter - one);
        c.constraint_at(5, is_read * (is_read - one));
        c.constraint_at_transition(38, next_filter * (next_filter - current_filter));
        c.advance(535);
    }
};

// ArithmeticStark: arithmetic/arithmetic_stark.rs:203-252 and its operation modules
// (mul.rs:123-185, addcy.rs:98-172, divmod.rs:86-145, modular.rs:382-612, byte.rs:201-296,
// shift.rs:85-128, polynomial helpers utils.rs), columns arithmetic/columns.rs:24-120.
// 116 columns: op flags 0..16, OPCODE 17, six 16-limb registers 18..113, RANGE_COUNTER 114,
// RC_FREQUENCIES 115.  Constraints are yielded in exactly the reference's order.
struct AirArithmetic {
    // Register / memory discipline: this AIR multiplies 16/32-limb polynomials (five 32x16 and three 16x16
    // schoolbook products per point).  Holding all operands in arrays costs 300+ VGPRs (scratch); re-reading them per
    // term made the kernel L2-bandwidth bound (14k loads per wave, the two LDE rows of a wave do not fit L1).  So
    // each product is a sliding-window convolution: the 16 limbs of one operand sit in registers, the other operand
    // streams through a 16-entry register window (one load per output coefficient), the loop over the output
    // coefficient is a real wave-uniform loop, and the 16-term inner product is an unrolled delayed-reduction dot
    // product (DotAcc, 8 VALU instructions per term).  The order of the yielded constraints is exactly the reference's.
    static constexpr u32 COLUMNS = 116;
    static constexpr u32 NL = 16;
    enum { IS_ADD = 0, IS_MUL, IS_SUB, IS_DIV, IS_MOD, IS_ADDMOD, IS_MULMOD, IS_ADDFP254, IS_MULFP254,
           IS_SUBFP254, IS_SUBMOD, IS_LT, IS_GT, IS_BYTE, IS_SHL, IS_SHR, IS_RANGE_CHECK, OPCODE_COL };
    enum { IN0 = 18, IN1 = 34, IN2 = 50, OUT = 66, AUX0 = 82, AUX1 = 98, RANGE_COUNTER = 114 };
    static constexpr u64 BASE = 1ULL << 16, OFFSET = 1ULL << 20;
    static constexpr u64 OVERFLOW_INV = 18446462594437939201ULL;  // 2^-16 (addcy.rs:67)

    // f(d, sum_j a(d - j) * m[j]) for d in [d_lo, d_hi); a(i) is read for 0 <= i < na and is 0 outside
    template <class FA, class F>
    __device__ static __forceinline__ void conv16(const Fe (&m)[NL], FA a, u32 na, u32 d_lo, u32 d_hi, F f) {
        Fe win[NL];                                     // win[j] = a(d - j)
#pragma unroll
        for (u32 j = 0; j < NL; ++j) win[j] = (d_lo >= j && d_lo - j < na) ? a(d_lo - j) : Fe();
#pragma unroll 1
        for (u32 d = d_lo; d < d_hi; ++d) {
            DotAcc p;
            dot_acc_init(p);
#pragma unroll
            for (u32 j = 0; j < NL; ++j) dot_acc_mac_v(p, win[j].v, m[j].v);
            f(d, Fe(dot_acc_reduce(p)));
#pragma unroll
            for (u32 j = NL - 1; j > 0; --j) win[j] = win[j - 1];
            win[0] = d + 1 < na ? a(d + 1) : Fe();
        }
    }

    // addcy.rs:98-151; x, y, z, given_cy: accessors i -> Fe
    template <class CONS, class FX, class FY, class FZ, class FC>
    __device__ static __forceinline__ void addcy(CONS &c, Fe filt, FX x, FY y, FZ z, FC given_cy, bool two_row) {
        Fe cy;
#pragma unroll 1
        for (u32 i = 0; i < NL; ++i) {
            Fe t = cy + x(i) + y(i) - z(i);
            Fe v = filt * t * (fe(BASE) - t);
            if (two_row) c.constraint_transition(v); else c.constraint(v);
            cy = t * fe(OVERFLOW_INV);
        }
        if (two_row) {
            c.constraint_transition(filt * (cy - given_cy(0)));
#pragma unroll 1
            for (u32 i = 1; i < NL; ++i) c.constraint_transition(filt * given_cy(i));
        } else {
            Fe g0 = given_cy(0);
            c.constraint(filt * g0 * (g0 - FE_ONE));
            c.constraint(filt * (cy - g0));
#pragma unroll 1
            for (u32 i = 1; i < NL; ++i) c.constraint(filt * given_cy(i));
        }
    }

    // (x - beta) * s(x) coefficient d of the aux polynomial a(i) (pol_adjoin_root): a(d-1) - beta * a(d)
    template <class FA>
    __device__ static __forceinline__ Fe adjoin(FA a, u32 d) {
        Fe aux_d = a(d);
        if (d == 0) return -(fe(BASE) * aux_d);
        return a(d - 1) - fe(BASE) * aux_d;
    }

    // mul.rs:123-173 (eval_packed_generic_mul); left: accessor, right_s: first column of the right operand
    template <class RV, class CONS, class FL>
    __device__ static __forceinline__ void mul(const RV &lv, CONS &c, Fe filt, FL left, u32 right_s) {
        auto aux = [&](u32 d) { return lv[AUX0 + d] + lv[AUX1 + d] * fe(BASE) - fe(OFFSET); };   // 2^20 offset undone
        Fe r[NL];
#pragma unroll
        for (u32 i = 0; i < NL; ++i) r[i] = lv[right_s + i];
        conv16(r, left, NL, 0, NL, [&](u32 d, Fe p) {                    // pol_mul_lo
            Fe cp = p - lv[OUT + d];
            cp -= adjoin(aux, d);
            c.constraint(filt * cp);
        });
    }

    // modular.rs:419-501 (modular_constr_poly incl. check_reduced), split in two: the constraints it yields
    // itself (modular_checks), and the polynomial it returns, produced coefficient by coefficient (modular_cp).
    //   output(i): 16 limbs; mod_s: first column of the modulus; quot(i): nq limbs (16 or 32)
    template <class RV>
    __device__ static __forceinline__ void load_modulus(const RV &lv, const RV &nv, u32 mod_s, Fe (&m)[NL]) {
#pragma unroll
        for (u32 i = 0; i < NL; ++i) m[i] = lv[mod_s + i];
        m[0] += nv[34];                                                 // + MODULAR_MOD_IS_ZERO
    }
    template <class RV, class CONS, class FO, class FQ>
    __device__ static __forceinline__ void modular_checks(const RV &lv, const RV &nv, CONS &c, Fe filt, FO output,
                                                          u32 mod_s, FQ quot, u32 nq) {
        Fe mod_is_zero = nv[34];                                        // MODULAR_MOD_IS_ZERO
        c.constraint_transition(filt * (mod_is_zero * mod_is_zero - mod_is_zero));
        Fe limb_sum;
#pragma unroll 1
        for (u32 i = 0; i < NL; ++i) limb_sum += lv[mod_s + i];
        c.constraint_transition(filt * limb_sum * mod_is_zero);
        Fe div_denom_is_zero = nv[97];                                  // MODULAR_DIV_DENOM_IS_ZERO
        Fe div_or_shr = lv[IS_DIV] + lv[IS_SHR];
        c.constraint_transition(filt * (mod_is_zero * div_or_shr - div_denom_is_zero));
        {   // check_reduced (modular.rs:382-414): output[0] += div_denom_is_zero inside
            Fe ilt0 = FE_ONE - mod_is_zero * div_or_shr;
            addcy(c, filt, [&](u32 i) { return i == 0 ? lv[mod_s] + mod_is_zero : lv[mod_s + i]; },
                  [&](u32 i) { return nv[18 + i]; },                    // MODULAR_OUT_AUX_RED in nv
                  [&](u32 i) { return i == 0 ? output(0) + div_denom_is_zero : output(i); },
                  [&](u32 i) { return i == 0 ? ilt0 : Fe(); }, true);
        }
        // prod = q(x) * m(x)  (pol_mul_wide2): degrees 0 .. 3*NL-2; the top NL-1 must vanish
        Fe m[NL];
        load_modulus(lv, nv, mod_s, m);
        conv16(m, quot, nq, 2 * NL, 3 * NL - 1, [&](u32, Fe p) { c.constraint_transition(filt * p); });
    }
    // f(d, coefficient d of  q*m + output + (x - beta) * s(x)) for d < 2*NL
    template <class RV, class FO, class FQ, class F>
    __device__ static __forceinline__ void modular_cp(const RV &lv, const RV &nv, FO output, u32 mod_s, FQ quot,
                                                      u32 nq, F f) {
        Fe m[NL];
        load_modulus(lv, nv, mod_s, m);
        // aux[i] = nv[35+i] - 2^20 + 2^16 * nv[66+i] (i < 31), aux[31] = 0
        auto aux = [&](u32 i) { return i < 2 * NL - 1 ? nv[35 + i] - fe(OFFSET) + fe(BASE) * nv[66 + i] : Fe(); };
        conv16(m, quot, nq, 0, 2 * NL, [&](u32 d, Fe p) {
            if (d < NL) p += output(d);
            f(d, p + adjoin(aux, d));
        });
    }

    // divmod.rs:86-116 (eval_packed_divmod_helper): the quotient input has 16 limbs (upper half zero)
    template <class RV, class CONS>
    __device__ static __forceinline__ void divmod_helper(const RV &lv, const RV &nv, CONS &c, Fe filt, u32 num_s,
                                                         u32 den_s, u32 quo_s, u32 rem_s) {
        c.constraint_last_row(filt);
        auto quo = [&](u32 i) { return lv[quo_s + i]; };
        auto rem = [&](u32 i) { return lv[rem_s + i]; };
        modular_checks(lv, nv, c, filt, rem, den_s, quo, NL);
        modular_cp(lv, nv, rem, den_s, quo, NL, [&](u32 i, Fe cp) {
            Fe v = i < NL ? cp - lv[num_s + i] : cp;
            c.constraint_transition(filt * v);
        });
    }

    // ---- the constraint families, in the reference's order (arithmetic_stark.rs:203-252); positions in brackets --------
    // part0: flags, opcode, range counter [0, 22), MUL [22, 38), ADD / SUB / LT / GT [38, 170)
    template <class RV, class CONS>
    __device__ static __forceinline__ void part_head(const RV &lv, const RV &nv, CONS &c) {
        const Fe one = FE_ONE;
        Fe all_flags;
        for (u32 f = 0; f <= IS_RANGE_CHECK; ++f) { Fe fl = lv[f]; c.constraint(fl * (fl - one)); all_flags += fl; }
        c.constraint(all_flags * (all_flags - one));
        c.constraint((one - lv[IS_RANGE_CHECK]) * lv[OPCODE_COL]);
        Fe rc1 = lv[RANGE_COUNTER], rc2 = nv[RANGE_COUNTER];
        c.constraint_first_row(rc1);
        Fe incr = rc2 - rc1;
        c.constraint_transition(incr * incr - incr);
        c.constraint_last_row(rc1 - fe(65535));
        auto in0 = [&](u32 i) { return lv[IN0 + i]; };
        auto in1 = [&](u32 i) { return lv[IN1 + i]; };
        auto out = [&](u32 i) { return lv[OUT + i]; };
        auto aux = [&](u32 i) { return lv[AUX0 + i]; };
        mul(lv, c, lv[IS_MUL], in0, IN1);
        addcy(c, lv[IS_ADD], in0, in1, out, aux, false);
        addcy(c, lv[IS_SUB], in1, out, in0, aux, false);
        addcy(c, lv[IS_LT], in1, aux, in0, out, false);
        addcy(c, lv[IS_GT], in0, aux, in1, out, false);
    }
    // modular.rs:542-612, in four pieces: [336, 420) filter / BN254 modulus / sign / sub checks, [420, 502) add + mul checks and
    // the add polynomial, [502, 534) the sub polynomial, [534, 566) the mul polynomial
    struct ModularFilters { Fe bn, add_f, sub_f, mul_f, sign, sign_ffff; };
    template <class RV>
    __device__ static __forceinline__ ModularFilters modular_filters(const RV &lv) {
        ModularFilters F;
        F.bn = lv[IS_ADDFP254] + lv[IS_MULFP254] + lv[IS_SUBFP254];
        F.add_f = lv[IS_ADDMOD] + lv[IS_ADDFP254];
        F.sub_f = lv[IS_SUBMOD] + lv[IS_SUBFP254];
        F.mul_f = lv[IS_MULMOD] + lv[IS_MULFP254];
        F.sign = lv[AUX0 + NL];                                       // quo_input(NL)
        F.sign_ffff = fe(0xFFFF) * F.sign;
        return F;
    }
    template <class RV, class CONS>
    __device__ static __forceinline__ void part_modular_a(const RV &lv, const RV &nv, CONS &c) {
        const Fe one = FE_ONE;
        // BN254 base-field modulus, 16-bit limbs (extension_tower.rs:25-30)
        constexpr u64 BN[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
        const ModularFilters F = modular_filters(lv);
        Fe filt = lv[IS_ADDMOD] + lv[IS_SUBMOD] + lv[IS_MULMOD] + F.bn;
        c.constraint_last_row(filt);
#pragma unroll
        for (u32 i = 0; i < NL; ++i) c.constraint_transition(F.bn * (lv[IN2 + i] - fe((BN[i / 4] >> (16 * (i % 4))) & 0xFFFF)));
        auto quo_input = [&](u32 i) { return lv[AUX0 + i]; };      // 2 * NL limbs
        auto out = [&](u32 i) { return lv[OUT + i]; };
        // submod_constr_poly (modular.rs:515-539): quotient with the sign limb folded in
        auto q_sub = [&](u32 i) { return i < NL ? quo_input(i) - F.sign_ffff : (i == NL ? Fe() : quo_input(i)); };
        c.constraint(F.sub_f * F.sign * (F.sign - one));
#pragma unroll 1
        for (u32 i = NL; i < 2 * NL; ++i) c.constraint(F.sub_f * q_sub(i));
        modular_checks(lv, nv, c, F.sub_f, out, IN2, q_sub, 2 * NL);
    }
    template <class RV, class CONS>
    __device__ static __forceinline__ void part_modular_b(const RV &lv, const RV &nv, CONS &c) {
        const ModularFilters F = modular_filters(lv);
        auto quo_input = [&](u32 i) { return lv[AUX0 + i]; };
        auto out = [&](u32 i) { return lv[OUT + i]; };
        modular_checks(lv, nv, c, F.add_f + F.mul_f, out, IN2, quo_input, 2 * NL);
        modular_cp(lv, nv, out, IN2, quo_input, 2 * NL, [&](u32 d, Fe cp) {       // add: input0 + input1
            c.constraint_transition(F.add_f * (d < NL ? cp - (lv[IN0 + d] + lv[IN1 + d]) : cp));
        });
    }
    template <class RV, class CONS>
    __device__ static __forceinline__ void part_modular_c(const RV &lv, const RV &nv, CONS &c) {
        const ModularFilters F = modular_filters(lv);
        auto quo_input = [&](u32 i) { return lv[AUX0 + i]; };
        auto out = [&](u32 i) { return lv[OUT + i]; };
        auto q_sub = [&](u32 i) { return i < NL ? quo_input(i) - F.sign_ffff : (i == NL ? Fe() : quo_input(i)); };
        modular_cp(lv, nv, out, IN2, q_sub, 2 * NL, [&](u32 d, Fe cp) {           // sub: input0 - input1
            c.constraint_transition(F.sub_f * (d < NL ? cp - (lv[IN0 + d] - lv[IN1 + d]) : cp));
        });
    }
    template <class RV, class CONS>
    __device__ static __forceinline__ void part_modular_d(const RV &lv, const RV &nv, CONS &c) {
        const ModularFilters F = modular_filters(lv);
        auto quo_input = [&](u32 i) { return lv[AUX0 + i]; };
        auto in0 = [&](u32 i) { return lv[IN0 + i]; };
        auto out = [&](u32 i) { return lv[OUT + i]; };
        // mul: pol_mul_wide(input0, input1), coefficient d < 2*NL - 1, subtracted from the same polynomial
        Fe m[NL], r[NL];
        load_modulus(lv, nv, IN2, m);
#pragma unroll
        for (u32 i = 0; i < NL; ++i) r[i] = lv[IN1 + i];
        auto auxn = [&](u32 i) { return i < 2 * NL - 1 ? nv[35 + i] - fe(OFFSET) + fe(BASE) * nv[66 + i] : Fe(); };
        Fe wq[NL], wi[NL];                        // two sliding windows: quotient limbs and input0 limbs
#pragma unroll
        for (u32 j = 0; j < NL; ++j) { wq[j] = j == 0 ? quo_input(0) : Fe(); wi[j] = j == 0 ? in0(0) : Fe(); }
#pragma unroll 1
        for (u32 d = 0; d < 2 * NL; ++d) {
            DotAcc p, q;
            dot_acc_init(p); dot_acc_init(q);
#pragma unroll
            for (u32 j = 0; j < NL; ++j) { dot_acc_mac_v(p, wq[j].v, m[j].v); dot_acc_mac_v(q, wi[j].v, r[j].v); }
            Fe v(dot_acc_reduce(p));
            if (d < NL) v += out(d);
            v += adjoin(auxn, d);
            if (d < 2 * NL - 1) v -= Fe(dot_acc_reduce(q));
            c.constraint_transition(F.mul_f * v);
#pragma unroll
            for (u32 j = NL - 1; j > 0; --j) { wq[j] = wq[j - 1]; wi[j] = wi[j - 1]; }
            wq[0] = d + 1 < 2 * NL ? quo_input(d + 1) : Fe();
            wi[0] = d + 1 < NL ? in0(d + 1) : Fe();
        }
    }
    // part_modular_d as two additive halves over the SAME positions [534, 566), for the tiled kernel (arith_quotient.cuh): the
    // constraint mul_f * (q*m + out + (x - beta) s - in0*in1)_d is linear in the two products, so one wave yields
    // mul_f * (q*m + out + (x - beta) s)_d and another mul_f * (-(in0*in1)_d); the consumer adds them up.
    template <class RV, class CONS>
    __device__ static __forceinline__ void part_modular_d1(const RV &lv, const RV &nv, CONS &c) {
        const ModularFilters F = modular_filters(lv);
        auto quo_input = [&](u32 i) { return lv[AUX0 + i]; };
        auto out = [&](u32 i) { return lv[OUT + i]; };
        modular_cp(lv, nv, out, IN2, quo_input, 2 * NL, [&](u32, Fe cp) { c.constraint_transition(F.mul_f * cp); });
    }
    template <class RV, class CONS>
    __device__ static __forceinline__ void part_modular_d2(const RV &lv, const RV &nv, CONS &c) {
        (void)nv;
        const ModularFilters F = modular_filters(lv);
        Fe r[NL];
#pragma unroll
        for (u32 i = 0; i < NL; ++i) r[i] = lv[IN1 + i];
        conv16(r, [&](u32 i) { return lv[IN0 + i]; }, NL, 0, 2 * NL - 1, [&](u32, Fe q) { c.constraint_transition(F.mul_f * (-q)); });
    }
    // byte.rs:201-296 [566, 608) and SHL = MUL on (IN1, IN2) [608, 624)
    template <class RV, class CONS>
    __device__ static __forceinline__ void part_byte_shl(const RV &lv, const RV &nv, CONS &c) {
        (void)nv;
        const Fe one = FE_ONE;
        auto in0 = [&](u32 i) { return lv[IN0 + i]; };
        auto in1 = [&](u32 i) { return lv[IN1 + i]; };
        auto out = [&](u32 i) { return lv[OUT + i]; };
        {
            Fe is_byte = lv[IS_BYTE];
            Fe tree[NL], auxb[6];
            for (u32 i = 0; i < NL; ++i) tree[i] = lv[AUX1 + i];
            for (u32 i = 0; i < 6; ++i) auxb[i] = lv[AUX0 + i];
            Fe idx0_lo5;
            for (u32 i = 0; i < 5; ++i) {
                Fe bit = auxb[i];
                c.constraint(is_byte * (bit * bit - bit));
                idx0_lo5 += bit * fe(1ULL << i);
            }
            Fe idx0_hi = auxb[5] * fe(32);
            c.constraint(is_byte * (in0(0) - (idx0_lo5 + idx0_hi)));
            Fe bit = auxb[4];
            for (u32 i = 0; i < 8; ++i) c.constraint(is_byte * (tree[i] - (bit * in1(i) + (one - bit) * in1(i + 8))));
            bit = auxb[3];
            for (u32 i = 0; i < 4; ++i) c.constraint(is_byte * (tree[i + 8] - (bit * tree[i] + (one - bit) * tree[i + 4])));
            bit = auxb[2];
            for (u32 i = 0; i < 2; ++i) c.constraint(is_byte * (tree[i + 12] - (bit * tree[i + 8] + (one - bit) * tree[i + 10])));
            bit = auxb[1];
            Fe limb = bit * tree[12] + (one - bit) * tree[13];
            c.constraint(is_byte * (tree[14] - limb));
            const Fe base8 = fe(256);
            Fe lo_byte = lv[88], hi_byte = lv[89];
            c.constraint(is_byte * (lo_byte + base8 * (base8 * hi_byte - limb)));
            bit = auxb[0];
            Fe t = bit * lo_byte + (one - bit) * base8 * hi_byte;
            c.constraint(is_byte * (base8 * tree[15] - t));
            Fe hi_limb_sum = lv[87];
            for (u32 i = 1; i < NL; ++i) hi_limb_sum += in0(i);
            Fe idx_is_large = lv[90];
            c.constraint(is_byte * (idx_is_large * idx_is_large - idx_is_large));
            c.constraint(is_byte * hi_limb_sum * (idx_is_large - one));
            Fe hi_inv = lv[91] + lv[92] * fe(1ULL << 16) + lv[93] * fe(1ULL << 32) + lv[94] * fe(1ULL << 48);
            c.constraint(is_byte * (hi_limb_sum * hi_inv - idx_is_large));
            c.constraint(is_byte * (out(0) - (one - idx_is_large) * tree[15]));
            for (u32 i = 1; i < NL; ++i) c.constraint(is_byte * out(i));
        }
        mul(lv, c, lv[IS_SHL], in1, IN2);
    }
    // first positions of the families (707 constraints in all; see part_* above)
    enum { POS_HEAD = 0, POS_DIV = 170, POS_MOD = 253, POS_MODULAR_A = 336, POS_MODULAR_B = 420, POS_MODULAR_C = 502,
           POS_MODULAR_D = 534, POS_BYTE_SHL = 566, POS_SHR = 624, N_CONSTRAINTS = 707 };

    template <class CONS>
    __device__ static __forceinline__ void eval(const RowView &lv, const RowView &nv, CONS &c, const u64 *) {
        part_head(lv, nv, c);
        divmod_helper(lv, nv, c, lv[IS_DIV], IN0, IN1, OUT, AUX0);
        divmod_helper(lv, nv, c, lv[IS_MOD], IN0, IN1, AUX0, OUT);
        part_modular_a(lv, nv, c);
        part_modular_b(lv, nv, c);
        part_modular_c(lv, nv, c);
        part_modular_d(lv, nv, c);
        part_byte_shl(lv, nv, c);
        // shift.rs:85-128: SHL = MUL on (IN1, IN2) (above); SHR = DIV helper on (IN1, IN2, OUT, AUX0)
        divmod_helper(lv, nv, c, lv[IS_SHR], IN1, IN2, OUT, AUX0);
    }
};

// 96-bit integer accumulator for sums of up to 2^32 weighted u64 representatives: `step` is one Horner step S <- 2 S + v
// (32 steps of values < 2^64 stay below 2^96), `add` a plain S <- S + v; `fold` reduces to a lazy field element.
struct Acc96 {
    u64 lo;
    u32 hi;
    __device__ __forceinline__ Acc96() : lo(0), hi(0) {}
    __device__ __forceinline__ void step(u64 v) {
        hi = (hi << 1) | (u32)(lo >> 63);
        lo <<= 1;
        const u64 s = lo + v;
        hi += s < v ? 1u : 0u;
        lo = s;
    }
    __device__ __forceinline__ void add(u64 v) {
        const u64 s = lo + v;
        hi += s < v ? 1u : 0u;
        lo = s;
    }
    __device__ __forceinline__ u64 fold() const { return fold96(lo, hi); }
};

// KeccakStark: keccak/keccak_stark.rs:266-426 + keccak/round_flags.rs:14-60 (xor/andn as polynomials:
// keccak/logic.rs:15-53); columns keccak/columns.rs:7-134 (2431 columns: 24 round flags, TIMESTAMP,
// A 25x2 limbs, C and C' 5x64 bits, A' 5x5x64 bits, A'' 25x2 limbs, A''[0,0] bits, A'''[0,0] limbs).
struct AirKeccak {
    static constexpr u32 COLUMNS = 2431;
    static constexpr u32 ROUNDS = 24, TIMESTAMP = 24;
    __device__ static __forceinline__ u32 reg_a(u32 x, u32 y) { return 25 + (x * 5 + y) * 2; }
    __device__ static __forceinline__ u32 reg_c(u32 x, u32 z) { return 75 + x * 64 + z; }
    __device__ static __forceinline__ u32 reg_c_prime(u32 x, u32 z) { return 395 + x * 64 + z; }
    __device__ static __forceinline__ u32 reg_a_prime(u32 x, u32 y, u32 z) { return 715 + x * 320 + y * 64 + z; }
    __device__ static __forceinline__ u32 reg_b(u32 x, u32 y, u32 z) {
        // rotation offsets R[a][b] (keccak/columns.rs:32-38)
        constexpr unsigned char R[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};
        u32 a = (x + 3 * y) % 5, b = x;
        return reg_a_prime(a, b, (z + 64 - R[a][b]) % 64);
    }
    __device__ static __forceinline__ u32 reg_a_pp(u32 x, u32 y) { return 2315 + x * 10 + y * 2; }
    __device__ static __forceinline__ u32 reg_a_ppp(u32 x, u32 y) { return (x == 0 && y == 0) ? 2429 : reg_a_pp(x, y); }
    __device__ static __forceinline__ Fe xor_gen(Fe x, Fe y) { return x + y - x * (y + y); }
    __device__ static __forceinline__ Fe xor3_gen(Fe x, Fe y, Fe z) { return xor_gen(x, xor_gen(y, z)); }
    __device__ static __forceinline__ Fe andn_gen(Fe x, Fe y) { return (FE_ONE - x) * y; }

    template <class CONS>
    __device__ static void eval(const RowView &lv, const RowView &nv, CONS &c, const u64 *) {
        const Fe one = FE_ONE;
        // round_flags.rs
        Fe local_any, next_any;
        for (u32 i = 0; i < ROUNDS; ++i) {
            Fe f = lv[i];
            c.constraint(f * (f - one));
            local_any += f;
            next_any += nv[i];
        }
        c.constraint_first_row(local_any * (lv[0] - one));
        for (u32 i = 1; i < ROUNDS; ++i) c.constraint_first_row(local_any * lv[i]);
        Fe last_round_flag = lv[ROUNDS - 1];
        Fe padding = (next_any - one) * local_any * (last_round_flag - one);
        for (u32 i = 0; i < ROUNDS; ++i)
            c.constraint_transition(next_any * (nv[(i + 1) % ROUNDS] - lv[i]) + padding);
        c.constraint_transition(next_any * (local_any - one));
        // keccak_stark.rs
        Fe not_final_step = one - last_round_flag;
        c.constraint(local_any * not_final_step * (nv[TIMESTAMP] - lv[TIMESTAMP]));
        for (u32 x = 0; x < 5; ++x)
            for (u32 z = 0; z < 64; ++z) {
                Fe xr = xor3_gen(lv[reg_c(x, z)], lv[reg_c((x + 4) % 5, z)], lv[reg_c((x + 1) % 5, (z + 63) % 64)]);
                c.constraint(lv[reg_c_prime(x, z)] - xr);
            }
        // The next three constraint families of the reference -- 50 x "A = bits of A' ^ C ^ C'", 320 x "sum_y A'[x,y,z] - C'[x,z]
        // in {0, 2, 4}", 50 x "A'' = bits of B ^ (~B & B)" -- are yielded through constraint_at: this kernel is bound by its
        // loads (2431 columns, most of them read once per family), and the dot-product consumer does not care about order.
        //   * A and the sums visit A'[x, 0..4, z] together; xor3_gen(A', C, C') = xor_gen(A', xor_gen(C, C')) and the inner
        //     xor does not depend on y: 7 loads and 6 multiplies per (x, z) instead of 5 x 3 + 6 loads and 10 + 2;
        //   * the chi step reads the five lanes B[0..4, y, z] of a row once for its five output bits (5 loads per (y, z)
        //     instead of 15).
        // Same field values at the same positions of the alpha-combination as the reference's order.
        // r03: the 32-bit recompositions sum_z 2^z v_z are INTEGER Horner sums in 96-bit accumulators (Acc96: five instructions a
        // step, one fold to the field at the end) instead of two lazy field additions per bit, and they are split by linearity:
        // xor_gen(a, t) = a + t - 2 a t  ->  sum 2^z a  +  sum 2^z t (independent of y: once per x)  -  sum 2^z (a * 2t); chi's
        // xor_gen(b0, n) with n = andn(b1, b2) likewise.  Same field values (the integers are sums of representatives), ~30 % fewer
        // instructions in a kernel that issues at 3.8 cycles per instruction.
        for (u32 x = 0; x < 5; ++x)
            for (u32 half = 0; half < 2; ++half) {
                Acc96 sa[5], sp[5], st;
                for (int z = 31; z >= 0; --z) {
                    const u32 zz = 32 * half + (u32)z;
                    const Fe cp = lv[reg_c_prime(x, zz)];
                    const Fe t = xor_gen(lv[reg_c(x, zz)], cp), t2 = t + t;
                    st.step(t.v);
                    Acc96 sum;
                    for (u32 y = 0; y < 5; ++y) {
                        const Fe a = lv[reg_a_prime(x, y, zz)];
                        sum.add(a.v);
                        sa[y].step(a.v);
                        sp[y].step((a * t2).v);
                    }
                    const Fe diff = Fe(sum.fold()) - cp;
                    c.constraint_at(50 + x * 64 + zz, diff * (diff - fe(2)) * (diff - fe(4)));
                }
                const Fe tt(st.fold());
                for (u32 y = 0; y < 5; ++y)
                    c.constraint_at(x * 10 + y * 2 + half, Fe(sa[y].fold()) + tt - Fe(sp[y].fold()) - lv[reg_a(x, y) + half]);
            }
        c.advance(50 + 320);
        for (u32 y = 0; y < 5; ++y)
            for (u32 half = 0; half < 2; ++half) {
                Acc96 sb[5], sn[5], sm[5];
                for (int z = 31; z >= 0; --z) {
                    const u32 zz = 32 * half + (u32)z;
                    Fe b[5];
                    for (u32 x = 0; x < 5; ++x) { b[x] = lv[reg_b(x, y, zz)]; sb[x].step(b[x].v); }
                    for (u32 x = 0; x < 5; ++x) {
                        const Fe b2 = b[(x + 2) % 5];
                        const Fe n = b2 - b[(x + 1) % 5] * b2;               // andn_gen(b1, b2) = (1 - b1) b2
                        sn[x].step(n.v);
                        sm[x].step((b[x] * (n + n)).v);                      // xor_gen(b0, n) = b0 + n - b0 * 2n
                    }
                }
                for (u32 x = 0; x < 5; ++x)
                    c.constraint_at(x * 10 + y * 2 + half, Fe(sb[x].fold()) + Fe(sn[x].fold()) - Fe(sm[x].fold()) - lv[reg_a_pp(x, y) + half]);
            }
        c.advance(50);
        for (u32 half = 0; half < 2; ++half) {
            Fe acc;
            for (int z = 32 * half + 31; z >= (int)(32 * half); --z) acc = acc + acc + lv[2365 + z];
            c.constraint(acc - lv[reg_a_pp(0, 0) + half]);
        }
        // A'''[0,0] = A''[0,0] xor RC (keccak/constants.rs)
        constexpr u64 RC[24] = {
            0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
            0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
            0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
            0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
            0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
            0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
        for (u32 half = 0; half < 2; ++half) {
            Fe acc;
            for (int z = 32 * half + 31; z >= (int)(32 * half); --z) {
                Fe rc_bit;
                for (u32 r = 0; r < ROUNDS; ++r)
                    if ((RC[r] >> z) & 1) rc_bit += lv[r];
                acc = acc + acc + xor_gen(lv[2365 + z], rc_bit);
            }
            c.constraint(acc - lv[2429 + half]);
        }
        Fe not_last_round = one - last_round_flag;
        for (u32 x = 0; x < 5; ++x)
            for (u32 y = 0; y < 5; ++y) {
                c.constraint_transition(not_last_round * (lv[reg_a_ppp(x, y)] - nv[reg_a(x, y)]));
                c.constraint_transition(not_last_round * (lv[reg_a_ppp(x, y) + 1] - nv[reg_a(x, y) + 1]));
            }
    }
};

// KeccakSpongeStark: keccak_sponge/keccak_sponge_stark.rs:546-715; columns keccak_sponge/columns.rs:31-95
// (438 columns: is_full_input_block 0, context 1, segment 2, virt 3, timestamp 4,
// already_absorbed_bytes 5, is_padding_byte 6..141, original_rate_u32s 142..175,
// original_capacity_u32s 176..191, block_bytes 192..327, xored_rate_u32s 328..361,
// partial_updated_state_u32s 362..403, updated_digest_state_bytes 404..435, range_counter 436,
// rc_frequencies 437).
struct AirKeccakSponge {
    static constexpr u32 COLUMNS = 438;
    template <class CONS>
    __device__ static void eval(const RowView &lv, const RowView &nv, CONS &c, const u64 *) {
        constexpr u32 RATE = 136, RATE_U32 = 34, CAP_U32 = 16, DIG_U32 = 8;
        constexpr u32 PAD = 6, ORATE = 142, OCAP = 176, BLOCK = 192, PARTIAL = 362, DIGEST = 404, RC = 436;
        const Fe one = FE_ONE;
        Fe rc1 = lv[RC], rc2 = nv[RC];
        c.constraint_first_row(rc1);
        Fe incr = rc2 - rc1;
        c.constraint_transition(incr * incr - incr);
        c.constraint_last_row(rc1 - fe(255));
        Fe full = lv[0];
        c.constraint(full * (full - one));
        for (u32 i = 0; i < RATE; ++i) { Fe p = lv[PAD + i]; c.constraint(p * (p - one)); }
        Fe is_final = lv[PAD + RATE - 1];
        for (u32 i = 1; i < RATE; ++i) c.constraint(lv[PAD + i - 1] * (lv[PAD + i] - one));
        c.constraint(is_final * full);
        Fe absorbed = lv[5];
        c.constraint_first_row(absorbed);
        for (u32 i = 0; i < RATE_U32; ++i) c.constraint_first_row(lv[ORATE + i]);
        for (u32 i = 0; i < CAP_U32; ++i) c.constraint_first_row(lv[OCAP + i]);
        c.constraint_transition(is_final * nv[5]);
        for (u32 i = 0; i < RATE_U32; ++i) c.constraint_transition(is_final * nv[ORATE + i]);
        for (u32 i = 0; i < CAP_U32; ++i) c.constraint_transition(is_final * nv[OCAP + i]);
        for (u32 k = 1; k <= 4; ++k) c.constraint_transition(full * (lv[k] - nv[k]));
        for (u32 k = 0; k < DIG_U32; ++k) {
            Fe cur = lv[DIGEST + 4 * k];
            for (u32 i = 1; i < 4; ++i) cur += lv[DIGEST + 4 * k + i] * fe(1ULL << (8 * i));
            c.constraint_transition(full * (nv[ORATE + k] - cur));
        }
        for (u32 k = 0; k < RATE_U32 - DIG_U32; ++k) c.constraint_transition(full * (nv[ORATE + DIG_U32 + k] - lv[PARTIAL + k]));
        for (u32 k = 0; k < CAP_U32; ++k) c.constraint_transition(full * (nv[OCAP + k] - lv[PARTIAL + (RATE_U32 - DIG_U32) + k]));
        c.constraint_transition(full * (absorbed + fe(RATE) - nv[5]));
        Fe single = lv[PAD + RATE - 1] - lv[PAD + RATE - 2];
        c.constraint_transition(single * (lv[BLOCK + RATE - 1] - fe(0x81)));
        for (u32 i = 0; i + 1 < RATE; ++i) {
            Fe pi = lv[PAD + i];
            Fe first = i > 0 ? pi - lv[PAD + i - 1] : pi;
            Fe bb = lv[BLOCK + i];
            c.constraint_transition(first * (bb - one));
            c.constraint_transition(pi * (first - one) * bb);
        }
        c.constraint_transition(is_final * (single - one) * (lv[BLOCK + RATE - 1] - fe(0x80)));
        Fe is_dummy = one - full - is_final;
        c.constraint_transition(is_dummy * (nv[0] + nv[PAD + RATE - 1]));
    }
};

// CpuStark: cpu/cpu_stark.rs:594-626 calling, in this order, byte_unpacking, clock, contextops,
// control_flow, decode, dup_swap, gas, halt, jumps, membus, memio, modfp254, pc, push0, shift,
// simple_logic (not, eq_iszero), stack, syscalls_exceptions (each cpu/<module>.rs `eval_packed`).
// Columns cpu/columns/{mod.rs:56-97, ops.rs:6-47, general.rs}: context 0, code_context 1,
// program_counter 2, stack_len 3, is_kernel_mode 4, gas 5, op flags 6..23 (eth_mainnet: no poseidon),
// opcode_bits 24..31, general union 32..39, clock 40, mem_channels[3] 41..79, partial_channel 80..84.
// air_consts = { halt_final pc, init pc, syscall_jumptable, exception_jumptable } -- kernel labels that
// only the reference's assembler can produce (cpu/control_flow.rs:37-47, syscalls_exceptions.rs:68,73).
// ERIGON = true is the `cdk_erigon` build of the table: one more operation flag, `poseidon`, after
// jumpdest_keccak_general (cpu/columns/ops.rs:22-25) -- every later column moves by one (86 columns) -- and the
// differences of contextops.rs:26-27, control_flow.rs:11-23, decode.rs:10,41-42, gas.rs:30-31, jumps.rs:124-150 (no
// JUMPDEST-bit read), stack.rs:106-119,353-369.
template <bool ERIGON>
struct AirCpuT {
    static constexpr u32 X = ERIGON ? 1 : 0;
    static constexpr u32 COLUMNS = 85 + X;
    static constexpr u32 N_OPS = 18 + X;
    enum { CTX = 0, CODE_CTX, PC, STACK_LEN, KERNEL, GAS };
    enum { BINARY_OP = 6, TERNARY_OP, FP254_OP, EQ_ISZERO, LOGIC_OP, NOT_POP, SHIFT, JUMPDEST_KECCAK_GENERAL,
           POSEIDON = 14 /* only when ERIGON */, JUMPS = 14 + X, PUSH_PROVER_INPUT, DUP_SWAP, CONTEXT_OP, M_OP_32BYTES,
           EXIT_KERNEL, M_OP_GENERAL, PC_PUSH0, SYSCALL, EXCEPTION };
    enum { BITS = 24 + X, GEN = 32 + X, CLOCK = 40 + X, CH0 = 41 + X, CH1 = 54 + X, CH2 = 67 + X, PARTIAL = 80 + X };
    enum { USED = 0, IS_READ = 1, ACTX = 2, ASEG = 3, AVIRT = 4, VAL = 5 };
    enum { STACK_INV = 36 + X, STACK_INV_AUX = 37 + X, STACK_INV_AUX_2 = 38 + X, STACK_LEN_BOUNDS_AUX = 39 + X };
    static constexpr u64 SEG_STACK = 1, SEG_SHIFT_TABLE = 13, SEG_JUMPDEST_BITS = 14, SEG_CODE = 0;
    __device__ static __forceinline__ u32 ch(u32 k) { return CH0 + 13 * k; }

    // stack.rs:173-282 (eval_packed_one)
    template <class CONS>
    __device__ static void stack_one(const RowView &lv, const RowView &nv, CONS &c, Fe filt, u32 num_pops,
                                     bool pushes, bool disable) {
        const Fe one = FE_ONE;
        if (num_pops > 0) {
            for (u32 i = 1; i < num_pops; ++i) {
                u32 b = ch(i);
                c.constraint(filt * (lv[b + USED] - one));
                c.constraint(filt * (lv[b + IS_READ] - one));
                c.constraint(filt * (lv[b + ACTX] - lv[CTX]));
                c.constraint(filt * (lv[b + ASEG] - fe(SEG_STACK)));
                c.constraint(filt * (lv[b + AVIRT] - (lv[STACK_LEN] - fe(i + 1))));
            }
            c.constraint(filt * lv[PARTIAL + USED]);
            if (!pushes) {
                Fe len_diff = lv[STACK_LEN] - fe(num_pops);
                Fe nf = len_diff * filt;
                c.constraint_transition(nf * (nv[CH0 + USED] - one));
                c.constraint_transition(nf * (nv[CH0 + IS_READ] - one));
                c.constraint_transition(nf * (nv[CH0 + ACTX] - nv[CTX]));
                c.constraint_transition(nf * (nv[CH0 + ASEG] - fe(SEG_STACK)));
                c.constraint_transition(nf * (nv[CH0 + AVIRT] - (nv[STACK_LEN] - one)));
                c.constraint(filt * (len_diff * lv[STACK_INV] - lv[STACK_INV_AUX]));
                c.constraint_transition(filt * (lv[STACK_INV_AUX] - one) * nv[CH0 + USED]);
            }
        } else if (pushes) {
            Fe nf = lv[STACK_LEN] * filt;
            c.constraint(nf * (lv[PARTIAL + USED] - one));
            c.constraint(nf * lv[PARTIAL + IS_READ]);
            c.constraint(nf * (lv[PARTIAL + ACTX] - lv[CTX]));
            c.constraint(nf * (lv[PARTIAL + ASEG] - fe(SEG_STACK)));
            c.constraint(nf * (lv[PARTIAL + AVIRT] - (lv[STACK_LEN] - one)));
            c.constraint(filt * (lv[STACK_LEN] * lv[STACK_INV] - lv[STACK_INV_AUX]));
            c.constraint(filt * (lv[STACK_INV_AUX] - one) * lv[PARTIAL + USED]);
        } else {
            c.constraint(filt * nv[CH0 + USED]);
            for (u32 i = 0; i < 8; ++i) c.constraint(filt * (lv[CH0 + VAL + i] - nv[CH0 + VAL + i]));
            c.constraint(filt * lv[PARTIAL + USED]);
        }
        if (disable) {
            u32 lo = num_pops > 1 ? num_pops : 1, hi = 3 - (pushes ? 1 : 0);
            for (u32 i = lo; i < hi; ++i) c.constraint(filt * lv[ch(i) + USED]);
        }
        c.constraint_transition(filt * (nv[STACK_LEN] - (lv[STACK_LEN] - fe(num_pops) + fe(pushes ? 1 : 0))));
    }

    template <class CONS>
    __device__ static void eval(const RowView &lv, const RowView &nv, CONS &c, const u64 *K) {
        const Fe one = FE_ONE;
        const Fe halt_pc(K[0]), start_pc(K[1]), syscall_jumptable(K[2]), exception_jumptable(K[3]);
        Fe b[8];
        for (u32 i = 0; i < 8; ++i) b[i] = lv[BITS + i];
        // ---- byte_unpacking.rs ----
        {
            Fe filt = lv[M_OP_32BYTES] * (b[5] - one);
            Fe len = one;
            for (u32 i = 0; i < 5; ++i) len += b[i] * fe(1ULL << i);
            c.constraint(filt * (nv[CH0 + VAL] - lv[CH0 + VAL] - len));
            c.constraint(filt * (nv[CH0 + VAL + 1] - lv[CH0 + VAL + 1]));
            c.constraint(filt * (nv[CH0 + VAL + 2] - lv[CH0 + VAL + 2]));
            for (u32 i = 3; i < 8; ++i) c.constraint(filt * nv[CH0 + VAL + i]);
        }
        // ---- clock.rs ----
        c.constraint_first_row(lv[CLOCK] - one);
        c.constraint_transition(nv[CLOCK] - lv[CLOCK] - one);
        // ---- contextops.rs ----
        {
            Fe dctx = nv[CTX] - lv[CTX];
            for (u32 op = BINARY_OP; op <= EXCEPTION; ++op)
                if (op != CONTEXT_OP) c.constraint_transition(lv[op] * dctx);
            Fe cop = lv[CONTEXT_OP];
            c.constraint_transition(cop * (b[0] - one) * dctx);
            // get
            Fe filt = cop * (one - b[0]);
            c.constraint(filt * (nv[CH0 + VAL + 2] - lv[CTX]));
            for (u32 i = 0; i < 8; ++i) if (i != 2) c.constraint(filt * nv[CH0 + VAL + i]);
            Fe pruning = lv[GEN];
            c.constraint(filt * pruning);
            c.constraint(filt * (nv[STACK_LEN] - (lv[STACK_LEN] + one)));
            c.constraint(filt * lv[CH1 + USED]);
            c.constraint(filt * nv[CH0 + USED]);
            // set
            filt = cop * b[0];
            c.constraint(filt * (lv[CH0 + VAL + 2] - nv[CTX]));
            for (u32 i = 1; i < 8; ++i) if (i != 2) c.constraint(filt * lv[CH0 + VAL + i]);
            c.constraint(cop * pruning * (pruning - one));
            c.constraint(filt * (pruning - lv[CH0 + VAL]));
            c.constraint(cop * (lv[STACK_INV_AUX] * b[0] - lv[STACK_INV_AUX_2]));
            for (u32 i = 0; i < 8; ++i) c.constraint(cop * lv[STACK_INV_AUX_2] * (nv[CH0 + VAL + i] - lv[CH2 + VAL + i]));
            c.constraint(filt * lv[CH1 + USED]);
            c.constraint(filt * nv[CH0 + USED]);
            // tail
            Fe stack_len = nv[STACK_LEN] - (one - b[0]);
            c.constraint(cop * (stack_len * lv[STACK_INV] - lv[STACK_INV_AUX]));
            c.constraint(cop * (lv[STACK_INV_AUX] - lv[CH2 + USED]));
            Fe nf = cop * lv[STACK_INV_AUX];
            c.constraint(nf * (lv[CH2 + IS_READ] - b[0]));
            c.constraint(nf * (lv[CH2 + ACTX] - nv[CTX]));
            c.constraint(nf * (lv[CH2 + ASEG] - fe(SEG_STACK)));
            c.constraint(nf * (lv[CH2 + AVIRT] - (stack_len - one)));
        }
        // ---- control_flow.rs ----
        Fe is_cpu, is_cpu_next;
        for (u32 op = BINARY_OP; op <= EXCEPTION; ++op) { is_cpu += lv[op]; is_cpu_next += nv[op]; }
        const Fe next_halt = one - is_cpu_next;
        {
            c.constraint_transition(is_cpu * (is_cpu_next + next_halt - one));
            Fe native = lv[BINARY_OP] + lv[TERNARY_OP] + lv[FP254_OP] + lv[EQ_ISZERO] + lv[LOGIC_OP] + lv[NOT_POP] +
                        lv[SHIFT] + lv[JUMPDEST_KECCAK_GENERAL];
            if (ERIGON) native += lv[POSEIDON];
            native = native + lv[PC_PUSH0] + lv[DUP_SWAP] + lv[CONTEXT_OP] + lv[M_OP_GENERAL];
            Fe dpc = lv[PC] - nv[PC] + one, dk = lv[KERNEL] - nv[KERNEL];
            c.constraint_transition(native * dpc);
            c.constraint_transition(native * dk);
            Fe is_pi = lv[PUSH_PROVER_INPUT] * b[7];
            c.constraint_transition(is_pi * dpc);
            c.constraint_transition(is_pi * dk);
            c.constraint(lv[PUSH_PROVER_INPUT] * ((lv[KERNEL] + lv[GEN]) - one));
            Fe last_noncpu = (is_cpu - one) * is_cpu_next;
            c.constraint_transition(last_noncpu * (nv[PC] - start_pc));
            c.constraint_transition(last_noncpu * (nv[KERNEL] - one));
            c.constraint_transition(last_noncpu * nv[STACK_LEN]);
        }
        // ---- decode.rs ----
        {
            Fe km = lv[KERNEL];
            c.constraint(km * (km - one));
            for (u32 i = 0; i < 8; ++i) c.constraint(b[i] * (b[i] - one));
            // OPCODES: (opcode, block_length, kernel_only, flag column)
            // (the cdk_erigon POSEIDON block 0x22-0x23 sits second in the list; without the feature its slot is skipped)
            constexpr u32 NOC = 6;
            constexpr u32 OC[NOC] = {0x14, 0x22, 0x56, 0x80, 0xf6, 0xf9};
            constexpr u32 BL[NOC] = {1, 1, 1, 5, 1, 0};
            constexpr bool KO[NOC] = {false, true, false, false, true, true};
            constexpr u32 COL[NOC] = {EQ_ISZERO, POSEIDON, JUMPS, DUP_SWAP, CONTEXT_OP, EXIT_KERNEL};
            constexpr u32 COMBINED[11] = {LOGIC_OP, FP254_OP, BINARY_OP, TERNARY_OP, SHIFT, M_OP_GENERAL,
                                          JUMPDEST_KECCAK_GENERAL, NOT_POP, PC_PUSH0, M_OP_32BYTES, PUSH_PROVER_INPUT};
            Fe flag_sum;
            for (u32 k = 0; k < NOC; ++k) {
                if (k == 1 && !ERIGON) continue;
                Fe f = lv[COL[k]]; c.constraint(f * (f - one)); flag_sum += f;
            }
            for (u32 k = 0; k < 11; ++k) { Fe f = lv[COMBINED[k]]; c.constraint(f * (f - one)); flag_sum += f; }
            c.constraint(flag_sum * (flag_sum - one));
            for (u32 k = 0; k < NOC; ++k) {
                if (k == 1 && !ERIGON) continue;
                Fe unavailable = KO[k] ? one - km : Fe();
                Fe mismatch;
                for (int i = 7; i >= (int)BL[k]; --i) mismatch += ((OC[k] >> i) & 1) ? one - b[i] : b[i];
                c.constraint(lv[COL[k]] * (unavailable + mismatch));
            }
            Fe opcode, high3;
            for (int i = 7; i >= 0; --i) { opcode += b[i] * fe(1ULL << i); if (i >= 5) high3 += b[i] * fe(1ULL << i); }
            c.constraint((km - one) * lv[FP254_OP]);
            c.constraint(lv[TERNARY_OP] * b[1] * (km - one));
            c.constraint((km - one) * lv[M_OP_GENERAL]);
            c.constraint((opcode - fe(0xfb)) * (opcode - fe(0xfc)) * lv[M_OP_GENERAL]);
            c.constraint((km - one) * lv[JUMPDEST_KECCAK_GENERAL] * (one - b[1]));
            c.constraint((opcode - fe(0x21)) * (opcode - fe(0x5b)) * lv[JUMPDEST_KECCAK_GENERAL]);
            c.constraint((opcode - fe(0x58)) * (opcode - fe(0x5f)) * lv[PC_PUSH0]);
            c.constraint((opcode - fe(0x19)) * (opcode - fe(0x50)) * lv[NOT_POP]);
            c.constraint((km - one) * lv[M_OP_32BYTES]);
            c.constraint((high3 - fe(0xc0)) * (opcode - fe(0xf8)) * lv[M_OP_32BYTES]);
            c.constraint((opcode - fe(0xee)) * (high3 - fe(0x60)) * lv[PUSH_PROVER_INPUT]);
            c.constraint(lv[PUSH_PROVER_INPUT] * b[7] * (km - one));
        }
        // ---- dup_swap.rs ----
        {
            Fe n = b[0] + b[1] * fe(2) + b[2] * fe(4) + b[3] * fe(8);
            auto constrain_chan = [&](bool is_read, Fe f, Fe offset, u32 base) {
                c.constraint(f * (lv[base + USED] - one));
                c.constraint(f * (lv[base + IS_READ] - fe(is_read ? 1 : 0)));
                c.constraint(f * (lv[base + ACTX] - lv[CTX]));
                c.constraint(f * (lv[base + ASEG] - fe(SEG_STACK)));
                c.constraint(f * (lv[base + AVIRT] - (lv[STACK_LEN] - one - offset)));
            };
            Fe f = lv[DUP_SWAP] * (one - b[4]);
            for (u32 i = 0; i < 8; ++i) c.constraint(f * (lv[CH1 + VAL + i] - lv[CH0 + VAL + i]));
            constrain_chan(false, f, Fe(), CH1);
            for (u32 i = 0; i < 8; ++i) c.constraint(f * (lv[CH2 + VAL + i] - nv[CH0 + VAL + i]));
            constrain_chan(true, f, n, CH2);
            c.constraint_transition(f * (nv[STACK_LEN] - lv[STACK_LEN] - one));
            c.constraint(f * nv[CH0 + USED]);
            f = lv[DUP_SWAP] * b[4];
            Fe n1 = n + one;
            for (u32 i = 0; i < 8; ++i) c.constraint(f * (lv[CH0 + VAL + i] - lv[CH2 + VAL + i]));
            constrain_chan(false, f, n1, CH2);
            for (u32 i = 0; i < 8; ++i) c.constraint(f * (lv[CH1 + VAL + i] - nv[CH0 + VAL + i]));
            constrain_chan(true, f, n1, CH1);
            c.constraint(f * (nv[STACK_LEN] - lv[STACK_LEN]));
            c.constraint(f * nv[CH0 + USED]);
            c.constraint(lv[DUP_SWAP] * lv[PARTIAL + USED]);
        }
        // ---- gas.rs ----
        {
            // SIMPLE_OPCODES in struct field order: (column, cost)
            constexpr u32 NG = 10;                      // slot 4 = poseidon (KERNEL_ONLY_INSTR = 0), cdk_erigon only
            constexpr u32 GC[NG] = {FP254_OP, EQ_ISZERO, LOGIC_OP, SHIFT, POSEIDON, DUP_SWAP, CONTEXT_OP, M_OP_32BYTES, M_OP_GENERAL, PC_PUSH0};
            constexpr u64 GV[NG] = {0, 3, 3, 3, 0, 3, 0, 0, 0, 2};
            Fe gfilt, gas_used;
            for (u32 k = 0; k < NG; ++k) {
                if (k == 4 && !ERIGON) continue;
                Fe f = lv[GC[k]]; gfilt += f; gas_used += fe(GV[k]) * f;
            }
            c.constraint_transition(gfilt * (nv[GAS] - (lv[GAS] + gas_used)));
            Fe gas_diff = nv[GAS] - lv[GAS];
            for (u32 k = 0; k < NG; ++k) {
                if (k == 4 && !ERIGON) continue;
                c.constraint_transition(lv[GC[k]] * (gas_diff - fe(GV[k])));
            }
            c.constraint_transition(lv[JUMPS] * (gas_diff - (fe(8) + b[0] * fe(2))));
            Fe cost_filter = b[0] + b[4] - b[0] * b[4];
            c.constraint_transition(lv[BINARY_OP] * (gas_diff - (fe(5) + cost_filter * (fe(3) - fe(5)))));
            c.constraint_transition(lv[TERNARY_OP] * (gas_diff - (fe(8) - b[1] * fe(8))));
            c.constraint_transition(lv[NOT_POP] * (gas_diff - ((one - b[0]) * fe(2) + b[0] * fe(3))));
            c.constraint_transition(lv[JUMPDEST_KECCAK_GENERAL] * (gas_diff - (b[1] * fe(1) + (one - b[1]) * fe(0))));
            c.constraint_transition(lv[PUSH_PROVER_INPUT] * (gas_diff - ((one - b[7]) * fe(3) + b[7] * fe(0))));
            c.constraint_transition((is_cpu - one) * is_cpu_next * nv[GAS]);
        }
        // ---- halt.rs ----
        {
            Fe halt_state = one - is_cpu;
            c.constraint(halt_state * (halt_state - one));
            c.constraint_transition(halt_state * (next_halt - one));
            c.constraint(halt_state * (lv[KERNEL] - one));
            for (u32 i = 0; i < 3; ++i) c.constraint(halt_state * lv[ch(i) + USED]);
            c.constraint_last_row(halt_state - one);
            c.constraint(halt_state * (lv[PC] - halt_pc));
        }
        // ---- jumps.rs ----
        {
            Fe f = lv[EXIT_KERNEL];
            c.constraint_transition(f * (lv[CH0 + VAL] - nv[PC]));
            c.constraint_transition(f * (lv[CH0 + VAL + 1] - nv[KERNEL]));
            c.constraint_transition(f * (lv[CH0 + VAL + 6] - nv[GAS]));
            c.constraint(f * lv[CH0 + VAL + 7]);
            f = lv[JUMPS];
            Fe is_jump = f * (one - b[0]), is_jumpi = f * b[0];
            Fe len_diff = lv[STACK_LEN] - one - b[0];
            Fe nf = len_diff * f;
            c.constraint_transition(nf * (nv[CH0 + USED] - one));
            c.constraint_transition(nf * (nv[CH0 + IS_READ] - one));
            c.constraint_transition(nf * (nv[CH0 + ACTX] - nv[CTX]));
            c.constraint_transition(nf * (nv[CH0 + ASEG] - fe(SEG_STACK)));
            c.constraint_transition(nf * (nv[CH0 + AVIRT] - (nv[STACK_LEN] - one)));
            c.constraint(f * (len_diff * lv[STACK_INV] - lv[STACK_INV_AUX]));
            c.constraint_transition(f * (lv[STACK_INV_AUX] - one) * nv[CH0 + USED]);
            c.constraint(is_jump * (lv[CH1 + VAL] - one));
            for (u32 i = 1; i < 8; ++i) c.constraint(is_jump * lv[CH1 + VAL + i]);
            Fe sj = lv[GEN], cond_sum_pinv = lv[GEN + 1];
            c.constraint(f * sj * (sj - one));
            Fe cond_sum, dst_hi_sum;
            for (u32 i = 0; i < 8; ++i) cond_sum += lv[CH1 + VAL + i];
            for (u32 i = 1; i < 8; ++i) dst_hi_sum += lv[CH0 + VAL + i];
            c.constraint(f * (sj - one) * cond_sum);
            c.constraint(f * (cond_sum_pinv * cond_sum - sj));
            c.constraint(f * sj * dst_hi_sum);
            if (!ERIGON) {                                  // "We skip jump destinations verification with cdk_erigon"
                c.constraint(f * (lv[CH2 + VAL] - one));
                c.constraint(f * (lv[CH2 + USED] - sj * (one - lv[KERNEL])));
                c.constraint(f * (lv[CH2 + IS_READ] - one));
                c.constraint(f * (lv[CH2 + ACTX] - lv[CTX]));
                c.constraint(f * (lv[CH2 + ASEG] - fe(SEG_JUMPDEST_BITS)));
                c.constraint(f * (lv[CH2 + AVIRT] - lv[CH0 + VAL]));
            }
            c.constraint(f * lv[PARTIAL + USED]);
            c.constraint(is_jump * lv[CH1 + USED]);
            c.constraint_transition(is_jump * (nv[STACK_LEN] - lv[STACK_LEN] + one));
            c.constraint_transition(is_jumpi * (nv[STACK_LEN] - lv[STACK_LEN] + fe(2)));
            c.constraint_transition(f * (sj - one) * (nv[PC] - (lv[PC] + one)));
            c.constraint_transition(f * sj * (nv[PC] - lv[CH0 + VAL]));
        }
        // ---- membus.rs ----
        c.constraint(lv[CODE_CTX] - (one - lv[KERNEL]) * lv[CTX]);
        for (u32 i = 0; i < 3; ++i) { Fe u = lv[ch(i) + USED]; c.constraint(u * (u - one)); }
        { Fe u = lv[PARTIAL + USED]; c.constraint(u * (u - one)); }
        // ---- memio.rs ----
        {
            Fe f = lv[M_OP_GENERAL] * b[0];   // load: address in channel 0 (virt, segment, ctx)
            c.constraint(f * (lv[CH1 + USED] - one));
            c.constraint(f * (lv[CH1 + IS_READ] - one));
            c.constraint(f * (lv[CH1 + ACTX] - lv[CH0 + VAL + 2]));
            c.constraint(f * (lv[CH1 + ASEG] - lv[CH0 + VAL + 1]));
            c.constraint(f * (lv[CH1 + AVIRT] - lv[CH0 + VAL]));
            for (u32 i = 0; i < 8; ++i) c.constraint(f * (lv[CH1 + VAL + i] - nv[CH0 + VAL + i]));
            c.constraint(f * lv[CH2 + USED]);
            c.constraint(f * lv[PARTIAL + USED]);
            stack_one(lv, nv, c, f, 1, true, false);                    // MLOAD_GENERAL_OP
            f = lv[M_OP_GENERAL] * (b[0] - one);                        // store: address in channel 1
            c.constraint(f * (lv[PARTIAL + USED] - one));
            c.constraint(f * lv[PARTIAL + IS_READ]);
            c.constraint(f * (lv[PARTIAL + ACTX] - lv[CH1 + VAL + 2]));
            c.constraint(f * (lv[PARTIAL + ASEG] - lv[CH1 + VAL + 1]));
            c.constraint(f * (lv[PARTIAL + AVIRT] - lv[CH1 + VAL]));
            c.constraint(f * lv[CH2 + USED]);
            c.constraint(f * (lv[CH1 + USED] - one));
            c.constraint(f * (lv[CH1 + IS_READ] - one));
            c.constraint(f * (lv[CH1 + ACTX] - lv[CTX]));
            c.constraint(f * (lv[CH1 + ASEG] - fe(SEG_STACK)));
            c.constraint(f * (lv[CH1 + AVIRT] - (lv[STACK_LEN] - fe(2))));
            Fe mg = lv[M_OP_GENERAL];
            c.constraint(mg * ((lv[STACK_LEN] - fe(2)) * lv[STACK_INV] - lv[STACK_INV_AUX]));
            Fe is_top_read = lv[STACK_INV_AUX] * (one - b[0]);
            c.constraint(mg * (lv[STACK_INV_AUX_2] - is_top_read));
            Fe nf = mg * lv[STACK_INV_AUX_2];
            c.constraint_transition(nf * (nv[CH0 + USED] - one));
            c.constraint_transition(nf * (nv[CH0 + IS_READ] - one));
            c.constraint_transition(nf * (nv[CH0 + ACTX] - nv[CTX]));
            c.constraint_transition(nf * (nv[CH0 + ASEG] - fe(SEG_STACK)));
            c.constraint_transition(nf * (nv[CH0 + AVIRT] - (nv[STACK_LEN] - one)));
            c.constraint(mg * (lv[STACK_INV_AUX] - one) * nv[CH0 + USED]);
            c.constraint(mg * b[0] * nv[CH0 + USED]);
        }
        // ---- modfp254.rs ---- (BN254 base-field modulus, 32-bit limbs)
        {
            constexpr u64 PL[8] = {0xd87cfd47, 0x3c208c16, 0x6871ca8d, 0x97816a91, 0x8181585d, 0xb85045b6, 0xe131a029, 0x30644e72};
            for (u32 i = 0; i < 8; ++i) c.constraint(lv[FP254_OP] * (lv[CH2 + VAL + i] - fe(PL[i])));
        }
        // ---- pc.rs / push0.rs ----
        {
            Fe f = lv[PC_PUSH0] * (one - b[0]);
            c.constraint(f * (nv[CH0 + VAL] - lv[PC]));
            for (u32 i = 1; i < 8; ++i) c.constraint(f * nv[CH0 + VAL + i]);
            f = lv[PC_PUSH0] * b[0];
            for (u32 i = 0; i < 8; ++i) c.constraint(f * nv[CH0 + VAL + i]);
        }
        // ---- shift.rs ----
        {
            Fe sh = lv[SHIFT];
            Fe hz = lv[CH2 + USED];
            c.constraint(sh * hz * (lv[CH2 + IS_READ] - one));
            Fe hsum;
            for (u32 i = 1; i < 8; ++i) hsum += lv[CH0 + VAL + i];
            c.constraint(sh * (hsum * lv[GEN] - (one - hz)));
            c.constraint(sh * hsum * hz);
            c.constraint(sh * lv[CH2 + ACTX]);
            c.constraint(sh * (lv[CH2 + ASEG] - fe(SEG_SHIFT_TABLE)));
            c.constraint(sh * (lv[CH2 + AVIRT] - lv[CH0 + VAL]));
        }
        // ---- simple_logic: not.rs, eq_iszero.rs ----
        {
            Fe f = lv[NOT_POP] * b[0];
            for (u32 i = 0; i < 8; ++i) c.constraint(f * (nv[CH0 + VAL + i] + lv[CH0 + VAL + i] - fe(0xFFFFFFFFULL)));
            stack_one(lv, nv, c, f, 1, true, true);                     // BASIC_UNARY_OP
            Fe ef = lv[EQ_ISZERO];
            Fe eqf = ef * (one - b[0]), izf = ef * b[0];
            Fe equal = nv[CH0 + VAL], unequal = one - equal;
            c.constraint(ef * equal * unequal);
            for (u32 i = 1; i < 8; ++i) c.constraint(ef * nv[CH0 + VAL + i]);
            for (u32 i = 0; i < 8; ++i) c.constraint(izf * lv[CH1 + VAL + i]);
            Fe dot;
            for (u32 i = 0; i < 8; ++i) {
                Fe diff = lv[CH0 + VAL + i] - lv[CH1 + VAL + i];
                c.constraint(ef * equal * diff);
            }
            for (u32 i = 0; i < 8; ++i) dot += (lv[CH0 + VAL + i] - lv[CH1 + VAL + i]) * lv[GEN + i];
            c.constraint(ef * (dot - unequal));
            stack_one(lv, nv, c, eqf, 2, true, true);
            stack_one(lv, nv, c, izf, 1, true, true);
        }
        // ---- stack.rs eval_packed ----
        {
            // STACK_BEHAVIORS / MIGHT_OVERFLOW in struct field order; num_pops < 0 = None
            constexpr int NP[18] = {2, 3, 2, -1, 2, -1, 2, -1, -1, 0, -1, -1, 2, 1, -1, 0, 0, 0};
            constexpr bool PU[18] = {1, 1, 1, 0, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1, 1, 1};
            constexpr bool DI[18] = {1, 1, 1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 0, 0};
            constexpr bool OV[18] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
            for (u32 k = 0; k < 18; ++k) {                       // (cdk_erigon's poseidon flag: None / false, no term)
                Fe op = lv[BINARY_OP + k + (k >= 8 ? X : 0)];
                if (NP[k] >= 0) stack_one(lv, nv, c, op, (u32)NP[k], PU[k], DI[k]);
                if (OV[k]) {
                    Fe diff = nv[STACK_LEN] - fe(1025);
                    c.constraint_transition(op * (diff * lv[STACK_LEN_BOUNDS_AUX] - (one - nv[KERNEL])));
                }
            }
            stack_one(lv, nv, c, lv[JUMPDEST_KECCAK_GENERAL] * b[1], 0, false, true);          // JUMPDEST_OP
            stack_one(lv, nv, c, lv[JUMPDEST_KECCAK_GENERAL] * (one - b[1]), 2, true, true);   // KECCAK_GENERAL_OP
            if (ERIGON) {
                stack_one(lv, nv, c, lv[POSEIDON] * (one - b[0]), 3, true, true);               // POSEIDON_OP
                stack_one(lv, nv, c, lv[POSEIDON] * b[0], 2, true, true);                       // POSEIDON_GENERAL_OP
            }
            Fe npop = lv[NOT_POP];
            c.constraint(npop * ((lv[STACK_LEN] - one) * lv[STACK_INV] - lv[STACK_INV_AUX]));
            Fe is_top_read = lv[STACK_INV_AUX] * (one - b[0]);
            c.constraint(npop * (lv[STACK_INV_AUX_2] - is_top_read));
            Fe nf = npop * lv[STACK_INV_AUX_2];
            c.constraint_transition(nf * (nv[CH0 + USED] - one));
            c.constraint_transition(nf * (nv[CH0 + IS_READ] - one));
            c.constraint_transition(nf * (nv[CH0 + ACTX] - nv[CTX]));
            c.constraint_transition(nf * (nv[CH0 + ASEG] - fe(SEG_STACK)));
            c.constraint_transition(nf * (nv[CH0 + AVIRT] - (nv[STACK_LEN] - one)));
            c.constraint(npop * (lv[STACK_INV_AUX_2] - one) * nv[CH0 + USED]);
            Fe pf = npop * (b[0] - one);
            c.constraint(pf * lv[CH1 + USED]);
            c.constraint(pf * lv[CH2 + USED]);
            c.constraint(pf * lv[PARTIAL + USED]);
            c.constraint_transition(pf * (nv[STACK_LEN] - lv[STACK_LEN] + one));
        }
        // ---- syscalls_exceptions.rs ----
        {
            Fe fs = lv[SYSCALL], fex = lv[EXCEPTION];
            Fe tf = fs + fex;
            c.constraint(fs * (fs - one));
            c.constraint(fex * (fex - one));
            Fe exc_code = lv[GEN] + lv[GEN + 1] * fe(2) + lv[GEN + 2] * fe(4);
            const Fe stop = fe(6);                                       // EXC_STOP_CODE
            c.constraint(fex * (exc_code - stop) * lv[KERNEL]);
            for (u32 i = 0; i < 3; ++i) { Fe bit = lv[GEN + i]; c.constraint(fex * bit * (bit - one)); }
            Fe opcode;
            for (u32 i = 0; i < 8; ++i) opcode += b[i] * fe(1ULL << i);
            Fe op_handler = syscall_jumptable + opcode * fe(3);          // BYTES_PER_OFFSET = 3
            Fe exc_handler = exception_jumptable + exc_code * fe(3);
            c.constraint(tf * lv[CH1 + USED]);
            c.constraint(tf * (lv[CH1 + IS_READ] - one));
            c.constraint(tf * lv[CH1 + ACTX]);
            c.constraint(tf * (lv[CH1 + ASEG] - fe(SEG_CODE)));
            c.constraint(fs * (lv[CH1 + AVIRT] - op_handler));
            c.constraint(fex * (lv[CH1 + AVIRT] - exc_handler));
            for (u32 i = 1; i < 8; ++i) c.constraint(tf * lv[CH1 + VAL + i]);
            c.constraint(tf * lv[CH2 + USED]);
            c.constraint_transition(tf * (nv[PC] - lv[CH1 + VAL]));
            c.constraint_transition(tf * (nv[KERNEL] - one));
            c.constraint_transition(tf * nv[GAS]);
            c.constraint(fs * (nv[CH0 + VAL] - (lv[PC] + one)));
            c.constraint(fex * (nv[CH0 + VAL] - lv[PC]));
            c.constraint(fs * (nv[CH0 + VAL + 1] - lv[KERNEL]));
            c.constraint(tf * (nv[CH0 + VAL + 6] - lv[GAS]));
            c.constraint(tf * nv[CH0 + VAL + 7]);
            c.constraint(fex * (exc_code - stop) * nv[CH0 + VAL + 1]);
            for (u32 i = 2; i < 6; ++i) c.constraint(tf * nv[CH0 + VAL + i]);
        }
    }
};

using AirCpu = AirCpuT<false>;
using AirCpuErigon = AirCpuT<true>;

// PoseidonStark (`cdk_erigon` feature): poseidon/poseidon_stark.rs:445-690, columns poseidon/columns.rs:14-94.
// The reference evaluates the partial rounds through plonky2's sparse "fast" factorisation; the plain round function
// used here (constant vector, S-box on word 0, full MDS) yields the identical constraint polynomials -- between two
// S-boxes both forms are the same affine map of (state after the first full rounds, S-box outputs so far): the fast
// form is a rewriting of the plain one valid for ANY function on word 0, hence also with the S-box outputs as free
// symbols -- and on this chip the plain MDS layer (24 v_mad_u64_u32 per row with inline constants,
// next round's constants as the accumulators' start values: pos_mds) is also the cheaper one.
struct AirPoseidon {
    static constexpr u32 COLUMNS = 322;
    enum : u32 { CONTEXT = 0, SEGMENT, VIRT, TIMESTAMP, LEN, ALREADY_ABSORBED, IS_FINAL_INPUT_LEN = 6, IS_FULL_INPUT_BLOCK = 14,
                 INPUT = 15, CUBED_FULL = 27, CUBED_PARTIAL = 123, FULL_SBOX_0 = 145, PARTIAL_SBOX = 181, FULL_SBOX_1 = 203,
                 DIGEST = 251, OUTPUT_PARTIAL = 259, PINV = 267, INPUT_BYTES = 271, IS_SIMPLE_OP = 319,
                 IS_FIRST_ROW_GENERAL_OP = 320, NOT_PADDING = 321 };
    template <class CONS>
    __device__ static __forceinline__ void eval(const RowView &lv, const RowView &nv, CONS &c, const u64 *) {
        constexpr u32 W = 12, RATE = 8, DG = 4;
        const Fe one = FE_ONE;
        Fe is_full = lv[IS_FULL_INPUT_BLOCK];
        c.constraint(is_full * (is_full - one));
        Fe is_final;
        for (u32 i = 0; i < RATE; ++i) is_final += lv[IS_FINAL_INPUT_LEN + i];
        c.constraint(is_final * (is_final - one));
        for (u32 i = 0; i < RATE; ++i) { Fe f = lv[IS_FINAL_INPUT_LEN + i]; c.constraint(f * (f - one)); }
        Fe first_general = lv[IS_FIRST_ROW_GENERAL_OP];
        c.constraint(first_general * (first_general - one));
        c.constraint(is_final * is_full);
        Fe absorbed = lv[ALREADY_ABSORBED], len = lv[LEN];
        c.constraint_first_row(absorbed);
        for (u32 i = RATE; i < W; ++i) c.constraint_first_row(len * lv[INPUT + i]);
        c.constraint_transition(is_final * nv[ALREADY_ABSORBED]);
        Fe nlen = nv[LEN];
        for (u32 i = RATE; i < W; ++i) c.constraint_transition(nlen * is_final * nv[INPUT + i]);
        for (u32 col = CONTEXT; col <= TIMESTAMP; ++col) c.constraint_transition(is_full * (lv[col] - nv[col]));
        c.constraint_transition(is_full * (absorbed + fe(56) - nv[ALREADY_ABSORBED]));   // FELT_MAX_BYTES * RATE
        for (u32 i = 0; i < W - RATE; ++i)
            c.constraint_transition(is_full * (lv[DIGEST + 2 * i] + lv[DIGEST + 2 * i + 1] * fe(1ull << 32) - nv[INPUT + RATE + i]));
        Fe is_dummy = one - is_full - is_final;
        Fe next_is_final;
        for (u32 i = 0; i < RATE; ++i) next_is_final += nv[IS_FINAL_INPUT_LEN + i];
        c.constraint_transition(is_dummy * (nv[IS_FULL_INPUT_BLOCK] + next_is_final));
        Fe offset = len - absorbed;
        for (u32 i = 0; i < RATE; ++i) c.constraint(len * lv[IS_FINAL_INPUT_LEN + i] * (offset - fe(56 - i)));

        // ---- the permutation, plain rounds ----
        u64 s[12];
#pragma unroll
        for (u32 i = 0; i < W; ++i) s[i] = gl_add(lv[INPUT + i].v, ZK_RC[i]);
        int round = 0;
#pragma unroll 1
        for (u32 r = 0; r < 4; ++r) {
#pragma unroll
            for (u32 i = 0; i < W; ++i) {
                if (r != 0) {
                    Fe sbox_in = lv[FULL_SBOX_0 + W * (r - 1) + i];
                    c.constraint(Fe(s[i]) - sbox_in);
                    s[i] = sbox_in.v;
                }
                Fe cube = lv[CUBED_FULL + W * r + i];
                c.constraint(Fe(gl_mul(gl_sqr(s[i]), s[i])) - cube);
                s[i] = gl_mul(s[i], gl_sqr(cube.v));
            }
            ++round;
            pos_mds<true>(s, &ZK_RCS[round * 12]);
        }
#pragma unroll 1
        for (u32 r = 0; r < 22; ++r) {
            Fe sbox_in = lv[PARTIAL_SBOX + r];
            c.constraint(Fe(s[0]) - sbox_in);
            Fe cube = lv[CUBED_PARTIAL + r];
            c.constraint(Fe(gl_mul(gl_sqr(sbox_in.v), sbox_in.v)) - cube);
            s[0] = gl_mul(gl_sqr(cube.v), sbox_in.v);
            ++round;
            pos_mds<true>(s, &ZK_RCS[round * 12]);
        }
#pragma unroll 1
        for (u32 r = 0; r < 4; ++r) {
#pragma unroll
            for (u32 i = 0; i < W; ++i) {
                Fe sbox_in = lv[FULL_SBOX_1 + W * r + i];
                c.constraint(Fe(s[i]) - sbox_in);
                Fe cube = lv[CUBED_FULL + W * (4 + r) + i];
                c.constraint(Fe(gl_mul(gl_sqr(sbox_in.v), sbox_in.v)) - cube);
                s[i] = gl_mul(sbox_in.v, gl_sqr(cube.v));
            }
            ++round;
            if (r < 3) pos_mds<true>(s, &ZK_RCS[round * 12]);
            else pos_mds<false>(s, nullptr);
        }
        for (u32 i = 0; i < DG; ++i)
            c.constraint(Fe(s[i]) - (lv[DIGEST + 2 * i] + lv[DIGEST + 2 * i + 1] * fe(1ull << 32)));
        for (u32 i = DG; i < W; ++i) c.constraint(Fe(s[i]) - lv[OUTPUT_PARTIAL + i - DG]);
        for (u32 i = 0; i < DG; ++i)
            c.constraint(((lv[DIGEST + 2 * i + 1] - fe(0xFFFFFFFFull)) * lv[PINV + i] - one) * lv[DIGEST + 2 * i]);
    }
};
