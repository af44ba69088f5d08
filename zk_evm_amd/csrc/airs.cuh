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

// MemoryStark: memory/memory_stark.rs:474-626; columns memory/columns.rs:13-94 (30 columns, in
// `#[repr(C)]` order: filter 0, timestamp 1, timestamp_inv 2, is_read 3, addr_context 4,
// addr_segment 5, addr_virtual 6, value_limbs 7..14, context/segment/virtual_first_change 15..17,
// initialize_aux 18, preinitialized_segments 19, preinitialized_segments_aux 20, stale_contexts 21,
// is_pruned 22, stale_context_frequencies 23, is_stale 24, maybe_in_mem_after 25,
// mem_after_filter 26, range_check 27, counter 28, frequencies 29).
struct AirMemory {
    static constexpr u32 COLUMNS = 30;
    __device__ static __forceinline__ void eval(const RowView &lv, const RowView &nv, Consumer &c, const u64 *) {
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
    __device__ static __forceinline__ void eval(const RowView &lv, const RowView &nv, Consumer &c, const u64 *) {
        constexpr u32 NUM_BYTES = 32, IDX = 1, VAL = 37;
        const Fe one = FE_ONE;
        Fe rc1 = lv[69], rc2 = nv[69];
        c.constraint_first_row(rc1);
        Fe incr = rc2 - rc1;
        c.constraint_transition(incr * incr - incr);
        c.constraint_last_row(rc1 - fe(255));  // BYTE_RANGE_MAX - 1
        Fe current_filter;
        for (u32 i = 0; i < NUM_BYTES; ++i) current_filter += lv[IDX + i];
        c.constraint(current_filter * (current_filter - one));
        c.constraint_first_row(current_filter - one);
        Fe is_read = lv[0];
        c.constraint(is_read * (is_read - one));
        for (u32 i = 0; i < NUM_BYTES; ++i) { Fe idx = lv[IDX + i]; c.constraint(idx * (idx - one)); }
        Fe next_filter;
        for (u32 i = 0; i < NUM_BYTES; ++i) next_filter += nv[IDX + i];
        c.constraint_transition(next_filter * (next_filter - current_filter));
        for (u32 i = 0; i + 1 < NUM_BYTES; ++i) {
            Fe idx = lv[IDX + i];
            for (u32 j = i + 1; j < NUM_BYTES; ++j) c.constraint(idx * lv[VAL + j]);
        }
    }
};
