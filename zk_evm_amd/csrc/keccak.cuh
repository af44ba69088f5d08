// Keccak-f[1600] / Keccak-256 for gfx950, one sponge per lane (25 lanes-of-64-bit in VGPRs).
// Used for plonky2 `KeccakHash<25>` ([EXT] plonky2/src/hash/keccak.rs), the hasher of
// `KeccakGoldilocksConfig` (reference evm_arithmetization/tests/simple_transfer.rs:30).
#pragma once
#include "gl.cuh"

static __constant__ u64 ZK_KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

__device__ __forceinline__ u64 rotl64(u64 x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

__device__ __forceinline__ void keccak_f1600(u64 (&a)[25]) {
    constexpr int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43,
                             25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
#pragma unroll 1
    for (int rnd = 0; rnd < 24; ++rnd) {
        u64 c[5], b[25];
#pragma unroll
        for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
        for (int x = 0; x < 5; ++x) {
            u64 dx = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
#pragma unroll
            for (int y = 0; y < 5; ++y) a[x + 5 * y] ^= dx;
        }
#pragma unroll
        for (int x = 0; x < 5; ++x)
#pragma unroll
            for (int y = 0; y < 5; ++y)
                b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(a[x + 5 * y], ROT[x + 5 * y]);
#pragma unroll
        for (int y = 0; y < 5; ++y)
#pragma unroll
            for (int x = 0; x < 5; ++x)
                a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= ZK_KECCAK_RC[rnd];
    }
}

// Streaming Keccak-256 absorber over a sequence of u64 words (the LE encoding of canonical field
// elements), 17 words per 136-byte block, original-Keccak padding (0x01 .. 0x80).
struct Keccak256Words {
    u64 st[25];
    int pos;  // words absorbed into the current block
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < 25; ++i) st[i] = 0;
        pos = 0;
    }
    // `pos` must be compile-time trackable by the caller for register allocation; the generic
    // version below uses a switch-free XOR through a rotating select.
    __device__ __forceinline__ void absorb_at(int idx, u64 w) {
#pragma unroll
        for (int i = 0; i < 17; ++i)
            if (i == idx) st[i] ^= w;
    }
};
