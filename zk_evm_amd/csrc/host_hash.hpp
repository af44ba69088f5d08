// Host-side Poseidon / Keccak permutations and the Fiat-Shamir Challenger of the product library.
//
// The transcript is a few thousand field elements per segment: it stays on the
// host, exactly where the reference keeps it (`Challenger::new()` at
// evm_arithmetization/src/prover.rs:118, `challenger.compact()` at prover.rs:320).  Semantics:
// plonky2 1.0.0 `iop/challenger.rs` ([EXT]): rate 8 / width 12 duplex sponge in overwrite mode,
// outputs popped from the end of the buffer.  This is product code; it shares nothing with oracle/.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "gl.cuh"
#include "../../include/poseidon_constants.h"

namespace zkhost {

// Host permutation for the transcript.  A wide table's opening set is ~10^4 elements (Keccak: 1200 permutations
// per proof, 2150 per segment), so this is on the critical path between kernels: 128-bit products (one mulq each), the MDS
// rows over a doubled state array (no modulo in the inner loop), constants added once per round, and the blocked schedule
// of the device permutation for the partial rounds (poseidon_permute below).
inline u64 mul_host(u64 a, u64 b) {
    const unsigned __int128 p = (unsigned __int128)a * b;
    return gl_reduce128((u64)(p >> 64), (u64)p);
}
// MDS layer on the split state (lo / hi 32-bit halves, doubled to 24 entries): s[r] = sum_i C[i] * x[r + i] + 8 x[0] [r = 0]
inline void mds_scalar(u64 (&s)[12], const u64 (&lo)[24], const u64 (&hi)[24]) {
    static const u32 CIRC[12] = ZK_POSEIDON_MDS_CIRC_INIT;
    for (int r = 0; r < 12; ++r) {
        // 12 terms of (< 2^32) * (<= 41) per half: no overflow
        u64 al = 0, ah = 0;
#pragma GCC unroll 12
        for (int i = 0; i < 12; ++i) {
            al += lo[r + i] * CIRC[i];
            ah += hi[r + i] * CIRC[i];
        }
        if (r == 0) { al += lo[0] * 8; ah += hi[0] * 8; }
        const u64 t = al + (ah << 32);
        const u32 top = (u32)(ah >> 32) + (t < al ? 1u : 0u);
        s[r] = gl_canon(gl_reduce96(top, t));
    }
}
#if defined(__x86_64__)
// the same with vpmuludq: four rows per vector, 72 multiplies instead of 288
__attribute__((target("avx2"))) inline void mds_avx2(u64 (&s)[12], const u64 (&lo)[24], const u64 (&hi)[24]) {
    static const u32 CIRC[12] = ZK_POSEIDON_MDS_CIRC_INIT;
    alignas(32) u64 al[12], ah[12];
    for (int g = 0; g < 3; ++g) {
        __m256i vl = _mm256_setzero_si256(), vh = _mm256_setzero_si256();
#pragma GCC unroll 12
        for (int i = 0; i < 12; ++i) {
            const __m256i c = _mm256_set1_epi64x((long long)CIRC[i]);
            vl = _mm256_add_epi64(vl, _mm256_mul_epu32(_mm256_loadu_si256((const __m256i *)(lo + 4 * g + i)), c));
            vh = _mm256_add_epi64(vh, _mm256_mul_epu32(_mm256_loadu_si256((const __m256i *)(hi + 4 * g + i)), c));
        }
        _mm256_store_si256((__m256i *)(al + 4 * g), vl);
        _mm256_store_si256((__m256i *)(ah + 4 * g), vh);
    }
    al[0] += lo[0] * 8;
    ah[0] += hi[0] * 8;
    for (int r = 0; r < 12; ++r) {
        const u64 t = al[r] + (ah[r] << 32);
        const u32 top = (u32)(ah[r] >> 32) + (t < al[r] ? 1u : 0u);
        s[r] = gl_canon(gl_reduce96(top, t));
    }
}
#endif
#if defined(__x86_64__)
// the 12 x 12 product of a block (poseidon_permute below) on the 32-bit halves: al[r] = sum_j T[j][r] * lo[j] etc., four rows
// per vector; T = the matrix transposed, one u64 lane per entry, rows 12 / 13 = the columns multiplying d1 / d2
struct Block3Tables { alignas(32) u64 t[14][12]; };
__attribute__((target("avx2"))) inline void block3_avx2(u64 (&al)[12], u64 (&ah)[12], const Block3Tables &T,
                                                        const u64 (&lo)[14], const u64 (&hi)[14]) {
    for (int g = 0; g < 3; ++g) {
        __m256i vl = _mm256_setzero_si256(), vh = _mm256_setzero_si256();
#pragma GCC unroll 14
        for (int j = 0; j < 14; ++j) {
            const __m256i c = _mm256_load_si256((const __m256i *)(T.t[j] + 4 * g));
            vl = _mm256_add_epi64(vl, _mm256_mul_epu32(_mm256_set1_epi64x((long long)lo[j]), c));
            vh = _mm256_add_epi64(vh, _mm256_mul_epu32(_mm256_set1_epi64x((long long)hi[j]), c));
        }
        _mm256_storeu_si256((__m256i *)(al + 4 * g), vl);
        _mm256_storeu_si256((__m256i *)(ah + 4 * g), vh);
    }
}
#endif
inline u64 sbox_host(u64 x) {
    const u64 x2 = mul_host(x, x), x4 = mul_host(x2, x2);
    return mul_host(mul_host(x, x2), x4);
}
inline u64 red128(unsigned __int128 v) { return gl_canon(gl_reduce128((u64)(v >> 64), (u64)v)); }
struct RcSplitHost { u64 lo, hi; };
// The schedule of the device permutation (csrc/poseidon.cuh, pos_block3; constants and the proof that it equals the plain
// 30 rounds: tools/gen_poseidon_constants.py): full rounds 0 .. 2 as they are, then the 23 linear layers up to round 25 as
// seven blocks of three (one 12 x 12 product with the entries of MDS^3 < 2^21, here as 128-bit sums) and two plain rounds --
// 8 + 2 matrix products instead of 23 for the middle of the permutation: 1.32 -> 0.8 us on the GPU box's EPYC 9575F.
inline void poseidon_permute(u64 (&s)[12]) {
    static const u64 RC[ZK_POSEIDON_ROUNDS * 12] = ZK_POSEIDON_RC_INIT;
    static const u32 CIRC[12] = ZK_POSEIDON_MDS_CIRC_INIT;
    static const u32 M2R0[12] = ZK_POSEIDON_M2_ROW0_INIT, M2C0[12] = ZK_POSEIDON_M2_COL0_INIT;
    static const u32 M3[144] = ZK_POSEIDON_M3_INIT;
    static const RcSplitHost K3[ZK_POSEIDON_BLOCK3_COUNT * 12] = ZK_POSEIDON_RCS3_INIT;
    static const RcSplitHost KZ3[ZK_POSEIDON_BLOCK3_COUNT] = ZK_POSEIDON_RCS3Z_INIT;
#if defined(__x86_64__)
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    static const Block3Tables T3 = [] {
        Block3Tables t;
        for (int i = 0; i < 12; ++i) {
            for (int j = 0; j < 12; ++j) t.t[j][i] = M3[i * 12 + j];
            t.t[12][i] = M2C0[i];
            t.t[13][i] = CIRC[(12 - i) % 12] + (i == 0 ? 8u : 0u);
        }
        return t;
    }();
#endif
    typedef unsigned __int128 u128;
    auto mds = [&](u64 (&x)[12]) {                     // x <- MDS x  (x: S-box layer already applied)
        u64 lo[24], hi[24];
        for (int i = 0; i < 12; ++i) {
            lo[i] = lo[i + 12] = x[i] & 0xFFFFFFFFULL;
            hi[i] = hi[i + 12] = x[i] >> 32;
        }
#if defined(__x86_64__)
        if (have_avx2) { mds_avx2(x, lo, hi); return; }
#endif
        mds_scalar(x, lo, hi);
    };
    for (int i = 0; i < 12; ++i) s[i] = gl_canon(s[i]);
    for (int round = 0; round < ZK_POSEIDON_HALF_FULL_ROUNDS - 1; ++round) {
        for (int i = 0; i < 12; ++i) s[i] = sbox_host(gl_add_ref(s[i], RC[round * 12 + i]));
        mds(s);
    }
    for (int i = 0; i < 12; ++i) s[i] = sbox_host(gl_add_ref(s[i], RC[(ZK_POSEIDON_HALF_FULL_ROUNDS - 1) * 12 + i]));
    for (int b = 0; b < ZK_POSEIDON_BLOCK3_COUNT; ++b) {
        const int r = ZK_POSEIDON_HALF_FULL_ROUNDS - 1 + 3 * b;
        if (b) s[0] = sbox_host(s[0]);
        u128 a = RC[(r + 1) * 12], z = KZ3[b].lo | (KZ3[b].hi << 32);
        for (int j = 0; j < 12; ++j) {
            a += (u128)s[j] * (CIRC[j] + (j == 0 ? 8u : 0u));          // row 0 of MDS
            z += (u128)s[j] * M2R0[j];
        }
        const u64 w0 = red128(a);
        const u64 d1 = gl_canon(gl_sub_ref(sbox_host(w0), w0));
        const u64 z0 = red128(z + (u128)d1 * (CIRC[0] + 8u));
        const u64 d2 = gl_canon(gl_sub_ref(sbox_host(z0), z0));
        u64 out[12];
#if defined(__x86_64__)
        if (have_avx2) {
            u64 lo[14], hi[14], al[12], ah[12];
            for (int j = 0; j < 12; ++j) { lo[j] = s[j] & 0xFFFFFFFFULL; hi[j] = s[j] >> 32; }
            lo[12] = d1 & 0xFFFFFFFFULL; hi[12] = d1 >> 32;
            lo[13] = d2 & 0xFFFFFFFFULL; hi[13] = d2 >> 32;
            block3_avx2(al, ah, T3, lo, hi);               // sums < 2^58
            for (int i = 0; i < 12; ++i)
                out[i] = red128((u128)al[i] + ((u128)ah[i] << 32) + (K3[b * 12 + i].lo | (K3[b * 12 + i].hi << 32)));
        } else
#endif
        for (int i = 0; i < 12; ++i) {
            u128 acc = (K3[b * 12 + i].lo | (K3[b * 12 + i].hi << 32)) + (u128)d1 * M2C0[i] +
                       (u128)d2 * (CIRC[(12 - i) % 12] + (i == 0 ? 8u : 0u));            // MDS[i][0]
            for (int j = 0; j < 12; ++j) acc += (u128)s[j] * M3[i * 12 + j];
            out[i] = red128(acc);
        }
        for (int i = 0; i < 12; ++i) s[i] = out[i];
    }
    // rounds 24, 25 (partial; the constants of round 24 are in), then the four full rounds
    int round = ZK_POSEIDON_HALF_FULL_ROUNDS - 1 + 3 * ZK_POSEIDON_BLOCK3_COUNT;
    s[0] = sbox_host(s[0]);
    mds(s);
    ++round;
    for (int i = 0; i < 12; ++i) s[i] = gl_add_ref(s[i], RC[round * 12 + i]);
    s[0] = sbox_host(s[0]);
    mds(s);
    for (++round; round < ZK_POSEIDON_ROUNDS; ++round) {
        for (int i = 0; i < 12; ++i) s[i] = sbox_host(gl_add_ref(s[i], RC[round * 12 + i]));
        mds(s);
    }
}
inline u64 rotl(u64 x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

inline void keccak_f1600(u64 (&a)[25]) {
    static const u64 RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    // rho offsets by (x, y) and the pi destination, derived rather than tabulated
    for (int rnd = 0; rnd < 24; ++rnd) {
        u64 c[5];
        for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; ++x) {
            u64 d = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
            for (int y = 0; y < 5; ++y) a[x + 5 * y] ^= d;
        }
        u64 b[25];
        b[0] = a[0];
        int x = 1, y = 0;
        for (int t = 0; t < 24; ++t) {  // rho offset of the t-th lane on the (x,y) walk = (t+1)(t+2)/2
            int r = ((t + 1) * (t + 2) / 2) % 64;
            int nx = y, ny = (2 * x + 3 * y) % 5;
            b[nx + 5 * ny] = rotl(a[x + 5 * y], r);
            x = nx; y = ny;
        }
        for (int yy = 0; yy < 5; ++yy)
            for (int xx = 0; xx < 5; ++xx)
                a[xx + 5 * yy] = b[xx + 5 * yy] ^ (~b[(xx + 1) % 5 + 5 * yy] & b[(xx + 2) % 5 + 5 * yy]);
        a[0] ^= RC[rnd];
    }
}

inline void keccak256(const uint8_t *in, size_t len, uint8_t out[32]) {
    u64 st[25] = {0};
    uint8_t blk[136];
    while (true) {
        size_t take = len < 136 ? len : 136;
        memset(blk, 0, sizeof blk);
        memcpy(blk, in, take);
        bool last = len < 136;
        if (last) { blk[take] ^= 0x01; blk[135] ^= 0x80; }
        for (int i = 0; i < 17; ++i) { u64 w; memcpy(&w, blk + 8 * i, 8); st[i] ^= w; }
        keccak_f1600(st);
        if (last) break;
        in += 136; len -= 136;
    }
    memcpy(out, st, 32);
}

// [EXT] plonky2 hash/keccak.rs KeccakPermutation: hash onion, words >= p rejected
inline void keccak_permutation(u64 (&s)[12]) {
    uint8_t cur[96], h[32];
    for (int i = 0; i < 12; ++i) { u64 w = gl_canon(s[i]); memcpy(cur + 8 * i, &w, 8); }
    size_t len = 96;
    int got = 0;
    while (got < 12) {
        keccak256(cur, len, h);
        memcpy(cur, h, 32);
        len = 32;
        for (int k = 0; k < 4 && got < 12; ++k) {
            u64 w; memcpy(&w, h + 8 * k, 8);
            if (w < GL_P) s[got++] = w;
        }
    }
}

struct Challenger {
    uint32_t hasher = 0;
    u64 state[12] = {0};
    u64 in[8];
    int n_in = 0;
    u64 out[8];
    int n_out = 0;

    void permute() { if (hasher == 0) poseidon_permute(state); else keccak_permutation(state); }
    void duplexing() {
        for (int i = 0; i < n_in; ++i) state[i] = in[i];
        n_in = 0;
        permute();
        memcpy(out, state, sizeof out);
        n_out = 8;
    }
    void observe(u64 e) {
        n_out = 0;
        in[n_in++] = gl_canon(e);
        if (n_in == 8) duplexing();
    }
    void observe_slice(const u64 *e, size_t n) { for (size_t i = 0; i < n; ++i) observe(e[i]); }
    // GenericHashOut::to_vec: Poseidon = 4 elements; BytesHash<25> = 7,7,7,4-byte LE chunks
    void observe_hash(const u64 *slot) {
        if (hasher == 0) { observe_slice(slot, 4); return; }
        const uint8_t *b = reinterpret_cast<const uint8_t *>(slot);
        for (int k = 0; k < 4; ++k) {
            u64 w = 0;
            memcpy(&w, b + 7 * k, k < 3 ? 7 : 4);
            observe(w);
        }
    }
    void observe_cap(const u64 *slots, size_t n) { for (size_t i = 0; i < n; ++i) observe_hash(slots + 4 * i); }
    u64 get() {
        if (n_in != 0 || n_out == 0) duplexing();
        return out[--n_out];
    }
    void get_ext(u64 (&e)[2]) { e[0] = get(); e[1] = get(); }
    void compact(u64 *state_out) {
        if (n_in != 0) duplexing();
        n_out = 0;
        if (state_out) memcpy(state_out, state, sizeof state);
    }
};

// host extension-field helpers (tiny, transcript-side only)
struct Ext { u64 a, b; };
inline Ext ext_mul(Ext x, Ext y) {
    u64 aa = gl_mul_ref(x.a, y.a), bb = gl_mul_ref(x.b, y.b);
    u64 ab = gl_mul_ref(x.a, y.b), ba = gl_mul_ref(x.b, y.a);
    u64 bb7 = gl_mul_ref(bb, 7);
    return Ext{gl_canon(gl_add_ref(aa, bb7)), gl_canon(gl_add_ref(ab, ba))};
}
inline Ext ext_add(Ext x, Ext y) { return Ext{gl_canon(gl_add_ref(x.a, y.a)), gl_canon(gl_add_ref(x.b, y.b))}; }
inline Ext ext_pow(Ext b, u64 e) {
    Ext r{1, 0};
    while (e) { if (e & 1) r = ext_mul(r, b); b = ext_mul(b, b); e >>= 1; }
    return r;
}

}  // namespace zkhost
