// TEST INFRASTRUCTURE -- hipemu: the two rocPRIM device algorithms csrc/memtrace_host.inc calls, as stream-ordered host operations
// (a stable LSD sort on the key bits [begin_bit, end_bit) with payload, an exclusive scan).  Same calling convention: a first call with
// a null temporary buffer returns the size wanted.  rocPRIM itself is a third-party library validated on hardware since r03; what the
// emulation checks is the library's use of it (buffers, sizes, stream order).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace hipemu { void enqueue_host(hipStream_t stream, std::function<void()> fn); }
namespace rocprim {
template <class T> struct plus { T operator()(const T &a, const T &b) const { return a + b; } };

template <class Key, class Value>
hipError_t radix_sort_pairs(void *tmp, size_t &bytes, const Key *keys_in, Key *keys_out, const Value *values_in, Value *values_out, size_t n,
                            unsigned begin_bit, unsigned end_bit, hipStream_t stream) {
    if (!tmp) { bytes = 256; return hipSuccess; }
    hipemu::enqueue_host(stream, [=]() {
        std::vector<size_t> order(n);
        std::iota(order.begin(), order.end(), (size_t)0);
        const unsigned width = end_bit - begin_bit;
        const Key mask = width >= sizeof(Key) * 8 ? ~(Key)0 : (Key)((((Key)1) << width) - 1);
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return ((keys_in[a] >> begin_bit) & mask) < ((keys_in[b] >> begin_bit) & mask); });
        std::vector<Key> k(n);
        std::vector<Value> v(n);
        for (size_t i = 0; i < n; ++i) { k[i] = keys_in[order[i]]; v[i] = values_in[order[i]]; }
        std::copy(k.begin(), k.end(), keys_out);
        std::copy(v.begin(), v.end(), values_out);
    });
    return hipSuccess;
}
template <class In, class Out, class Init, class Op>
hipError_t exclusive_scan(void *tmp, size_t &bytes, In in, Out out, Init init, size_t n, Op op, hipStream_t stream) {
    if (!tmp) { bytes = 256; return hipSuccess; }
    hipemu::enqueue_host(stream, [=]() {
        auto acc = static_cast<typename std::remove_reference<decltype(out[0])>::type>(init);
        for (size_t i = 0; i < n; ++i) { const auto x = in[i]; out[i] = acc; acc = op(acc, x); }
    });
    return hipSuccess;
}
}  // namespace rocprim
