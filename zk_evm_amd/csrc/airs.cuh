// Table AIRs: device restatements of each table's `Stark::eval_packed_generic` from the reference
// tree (evm_arithmetization/src/*/..._stark.rs).  Constraints are yielded in the reference's order
// (the Horner accumulation in ConstraintConsumer makes the order parity-critical).
// Column indices follow the reference's `#[repr(C)]` column structs.
#pragma once
#include "quotient.cuh"

// no table constraints (lookup / CTL checks only) -- used by the generic-machinery tests
struct AirNone {
    static constexpr u32 COLUMNS = 0;
    __device__ static __forceinline__ void eval(const RowView &, const RowView &, Consumer &, const u64 *) {}
};

// MemoryContinuationStark (MemBefore / MemAfter): memory_continuation/memory_continuation_stark.rs:110-122,
// columns memory_continuation/columns.rs:7-23 (FILTER = 0, 12 columns).
struct AirMemContinuation {
    static constexpr u32 COLUMNS = 12;
    __device__ static __forceinline__ void eval(const RowView &lv, const RowView &, Consumer &c, const u64 *) {
        Fe filter = lv[0];
        c.constraint(filter * (filter - FE_ONE));  // the filter must be binary
    }
};

// LogicStark: logic.rs:249-303; columns logic.rs:46-71: op {is_and, is_or, is_xor} = 0..2,
// input0 bits 3..258, input1 bits 259..514, result limbs 515..522 (8 x 32-bit).
struct AirLogic {
    static constexpr u32 COLUMNS = 523;
    __device__ static __forceinline__ void eval(const RowView &lv, const RowView &, Consumer &c, const u64 *) {
        constexpr u32 IN0 = 3, IN1 = 3 + 256, RES = 3 + 512;
        Fe is_and = lv[0], is_or = lv[1], is_xor = lv[2];
        c.constraint(is_and * (is_and - FE_ONE));
        c.constraint(is_or * (is_or - FE_ONE));
        c.constraint(is_xor * (is_xor - FE_ONE));
        Fe all_flags = is_and + is_or + is_xor;
        c.constraint(all_flags * (all_flags - FE_ONE));
        Fe sum_coeff = is_or + is_xor;
        Fe and_coeff = is_and - is_or - is_xor * fe(2);
        for (u32 i = 0; i < 256; ++i) { Fe b = lv[IN0 + i]; c.constraint(b * (b - FE_ONE)); }
        for (u32 i = 0; i < 256; ++i) { Fe b = lv[IN1 + i]; c.constraint(b * (b - FE_ONE)); }
        for (u32 limb = 0; limb < 8; ++limb) {
            Fe x, y, x_land_y;
            for (u32 i = 0; i < 32; ++i) {
                Fe xb = lv[IN0 + 32 * limb + i], yb = lv[IN1 + 32 * limb + i];
                Fe w = fe(1ULL << i);
                x += xb * w;
                y += yb * w;
                x_land_y += xb * yb * w;
            }
            Fe x_op_y = sum_coeff * (x + y) + and_coeff * x_land_y;
            c.constraint(lv[RES + limb] - x_op_y);
        }
    }
};
