// host Poseidon permutation microbenchmark (transcript path): us per permutation on this CPU
#include <cstdio>
#include <chrono>
#include "../../zk_evm_amd/csrc/host_hash.hpp"
int main() {
    u64 s[12];
    for (int i = 0; i < 12; ++i) s[i] = i * 0x9E3779B97F4A7C15ULL;
    const int N = 200000;
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) zkhost::poseidon_permute(s);
        auto t1 = std::chrono::steady_clock::now();
        printf("%.3f us/perm  %llx\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N, (unsigned long long)s[0]);
    }
}
