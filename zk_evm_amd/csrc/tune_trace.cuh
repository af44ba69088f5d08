// tune_trace.cuh -- the filter columns of the trial segment's synthetic traces (tune_host.inc): table t's lookup / cross-table-lookup
// filter columns made binary / one-hot row by row, the shapes tools/benchlib.py synthetic_segment_traces writes (the helper-column
// kernels reject a non-binary filter exactly like starky's debug assert).  Its own header so that tests/emu/tune_trace_emu.cpp can
// run it on the CPU (tests/test_tune_trace_emulated.py compares the columns it touches with the Python generator's).
#pragma once
#include "gl.cuh"

// table t's filter columns, row by row (one thread per row): the shapes tools/benchlib.py synthetic_segment_traces writes
static __global__ void tune_trace_filters_kernel(u64 *tr, size_t stride, u32 n, int table, u64 seed) {
    const u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    u64 h = seed + (u64)row * 0x9E3779B97F4A7C15ULL;
    auto next = [&] { h += 0x9E3779B97F4A7C15ULL; u64 z = h; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); };
    auto put = [&](u32 col, u64 v) { tr[(size_t)col * stride + row] = v; };
    auto one_hot = [&](u32 first, u32 count, u32 extra) { const u32 pick = (u32)(next() % (count + extra)); for (u32 k = 0; k < count; ++k) put(first + k, pick == k); };
    switch (table) {
        case 0: one_hot(0, 17, 1); break;                                   // Arithmetic: op flags + IS_RANGE_CHECK
        case 1: one_hot(1, 32, 1); break;                                   // BytePacking: index_len
        case 2: {                                                           // Cpu: op flags, then the binary columns
            one_hot(6, 18, 1);
            const u64 b = next();
            for (u32 k = 0; k < 9; ++k) put(24 + k, (b >> k) & 1);
            put(41, (b >> 9) & 1); put(54, (b >> 10) & 1); put(67, (b >> 11) & 1); put(80, (b >> 12) & 1);
            break;
        }
        case 3: { const u64 b = next(); put(0, b & 1); put(23, (b >> 1) & 1); break; }      // Keccak: first / last round flags
        case 4: {                                                           // KeccakSponge: none / full block / final block of length ln
            const u32 kind = (u32)(next() % 3), ln = (u32)(next() % 136);
            put(0, kind == 1);
            for (u32 i = 0; i < 136; ++i) put(6 + i, kind == 2 && ln <= i);
            break;
        }
        case 5: one_hot(0, 3, 1); break;                                    // Logic ops
        case 6: {                                                           // Memory
            const u64 b = next();
            put(0, b & 1); put(22, (b >> 1) & 1); put(24, (b >> 2) & 1); put(26, (b >> 3) & 1);
            one_hot(15, 2, 2);
            put(1, (b >> 4) & 1); put(2, (b >> 4) & 1);                  // timestamp = timestamp_inv in {0, 1}: mem_before's filter binary
            break;
        }
        default: put(0, next() & 1); break;                                 // MemBefore / MemAfter filter
    }
}
