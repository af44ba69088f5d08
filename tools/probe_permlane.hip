// tools/probe_permlane.hip -- what v_permlane16_swap / v_permlane32_swap do on this chip, stated as the one sentence
// csrc/ntt_swap.cuh relies on: "(first operand, lane bit = 1) <-> (second operand, lane bit = 0)", lane bit 4 for permlane16,
// lane bit 5 for permlane32.  Prints PASS / FAIL per instruction and, on FAIL, what each lane received.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe_permlane tools/probe_permlane.hip && tools/probe_permlane
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void probe(unsigned *out) {
    const unsigned lane = threadIdx.x;
    const unsigned a = 1000 + lane, b = 2000 + lane;                 // first operand, second operand
    const auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    const auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r16[0]; out[64 + lane] = r16[1]; out[128 + lane] = r32[0]; out[192 + lane] = r32[1];
}

int main() {
    unsigned *d = nullptr, h[256];
    if (hipMalloc(&d, sizeof h) != hipSuccess) { printf("no device\n"); return 2; }
    probe<<<1, 64>>>(d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
    int bad = 0;
    for (int t = 0; t < 2; ++t) {
        const int bit = t == 0 ? 4 : 5;
        bool ok = true;
        for (unsigned l = 0; l < 64; ++l) {
            const unsigned p = l ^ (1u << bit), hi = (l >> bit) & 1;
            const unsigned want_a = hi ? 2000 + p : 1000 + l;        // a's lanes with the bit set receive the partner's b
            const unsigned want_b = hi ? 2000 + l : 1000 + p;        // b's lanes with the bit clear receive the partner's a
            ok = ok && h[t * 128 + l] == want_a && h[t * 128 + 64 + l] == want_b;
        }
        printf("v_permlane%d_swap: %s\n", t == 0 ? 16 : 32, ok ? "PASS (first operand's lanes with the bit set <-> second operand's lanes with it clear)" : "FAIL");
        if (!ok) {
            ++bad;
            for (unsigned l = 0; l < 64; ++l) printf("  lane %2u: first %u second %u\n", l, h[t * 128 + l], h[t * 128 + 64 + l]);
        }
    }
    return bad;
}
