// TEST INFRASTRUCTURE.  zk_evm_amd/csrc/tune_trace.cuh (the filter columns of the trial segment's synthetic traces, written while
// this repository had no GPU access) compiled by g++ against the shim in tests/emu/hip and run one "thread" after the other: for
// each of the nine tables, a [columns][rows] matrix pre-filled with a sentinel goes through the kernel and is printed as
//   table <t> col <c> touched <rows written> min <m> max <M>            for every column the kernel wrote
//   table <t> group <first> <count> max_row_sum <s>                      (helper for the one-hot checks: see the Python test)
// tests/test_tune_trace_emulated.py compares the touched columns with the binary columns of tools/benchlib.py's generator.
//   g++ -std=c++17 -O1 -I tests/emu -I zk_evm_amd/csrc tests/emu/tune_trace_emu.cpp -o /tmp/tune_trace_emu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

thread_local EmuIdx threadIdx, blockIdx, blockDim, gridDim;
void __syncthreads() {}

#include "tune_trace.cuh"

int main() {
    static const unsigned COLS[9] = {116, 71, 85, 2431, 438, 523, 30, 12, 12};
    const unsigned n = 1000;                                   // not a multiple of the block size: the row guard is exercised
    const unsigned long long SENT = 0xDEADBEEFCAFEF00DULL;
    for (int t = 0; t < 9; ++t) {
        std::vector<unsigned long long> tr((size_t)COLS[t] * n, SENT);
        for (unsigned b = 0; b < (n + 255) / 256; ++b)
            for (unsigned th = 0; th < 256; ++th) {
                threadIdx = {th, 0, 0}; blockIdx = {b, 0, 0}; blockDim = {256, 1, 1}; gridDim = {(n + 255) / 256, 1, 1};
                tune_trace_filters_kernel(tr.data(), n, n, t, 0xF117E500ULL + t);
            }
        for (unsigned c = 0; c < COLS[t]; ++c) {
            unsigned touched = 0;
            unsigned long long mn = ~0ULL, mx = 0;
            for (unsigned r = 0; r < n; ++r) {
                const unsigned long long v = tr[(size_t)c * n + r];
                if (v == SENT) continue;
                ++touched; mn = v < mn ? v : mn; mx = v > mx ? v : mx;
            }
            if (touched) printf("table %d col %u touched %u min %llu max %llu\n", t, c, touched, mn, mx);
        }
        // one row per line for the Python side's row-wise checks (only touched columns; '.' = untouched)
        for (unsigned r = 0; r < 64; ++r) {
            printf("table %d row %u :", t, r);
            for (unsigned c = 0; c < COLS[t]; ++c) { const unsigned long long v = tr[(size_t)c * n + r]; if (v != SENT) printf(" %u=%llu", c, v); }
            printf("\n");
        }
    }
    return 0;
}
