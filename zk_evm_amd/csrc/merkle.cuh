// Merkle-cap commitment kernels for gfx950 (plonky2 `MerkleTree::new`, [EXT]
// plonky2/src/hash/merkle_tree.rs; reference call path evm_arithmetization/src/prover.rs:100).
//
// Layout decisions (MI355X-first):
//   * the LDE matrix stays COLUMN-major ([C][N], natural row order) - the layout the NTT writes.
//     A lane hashes one row: for a fixed column, 64 lanes read 64 consecutive u64 -> every load is
//     a fully coalesced 512-byte wavefront request.  plonky2's `transpose` + row-major leaves are
//     never materialised.
//   * `reverse_index_bits_in_place` becomes an index computation: the lane that hashed natural
//     row j stores its 32-byte digest at leaf slot bitrev(j).
//   * digests: level-concatenated 32-byte slots (level 0 = leaves), inner levels are one thread
//     per parent; the cap is the last 2^cap_height slots.
#pragma once
#include "gl.cuh"
#include "poseidon.cuh"
#include "keccak.cuh"

// Column c of the matrix starts at cols + (GATHER ? col_off[c] : c * col_stride): the GATHER form
// hashes rows of a matrix whose columns are arbitrary strided views (FRI commit-phase leaves).
template <bool GATHER>
__device__ __forceinline__ size_t col_offset(u32 c, size_t col_stride, const u64 *col_off) {
    return GATHER ? (size_t)col_off[c] : (size_t)c * col_stride;
}

// Poseidon `hash_or_noop` of each row; digest of row j -> slot (bitrev ? bitrev(j) : j).
template <bool GATHER>
__global__ void __launch_bounds__(256)
poseidon_hash_rows_kernel(const u64 *__restrict__ cols, size_t col_stride, const u64 *__restrict__ col_off,
                          u32 n_cols, size_t n_rows, int log_rows, int do_bitrev,
                          u64 *__restrict__ digests) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_rows) return;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = 0;
    const u64 *p = cols + j;
    if (n_cols <= 4) {
        for (u32 c = 0; c < n_cols; ++c) s[c] = gl_canon(p[col_offset<GATHER>(c, col_stride, col_off)]);
    } else {
        // ONE inlined permutation (46 KB of code: two copies would not fit the 64 KB instruction cache a CU pair shares)
#pragma unroll 1
        for (u32 c = 0; c < n_cols; c += 8) {
            if (c + 8 <= n_cols) {
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] = p[col_offset<GATHER>(c + i, col_stride, col_off)];
            } else {                                   // ragged last chunk: overwrite mode keeps the other elements
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (c + i < n_cols) s[i] = p[col_offset<GATHER>(c + i, col_stride, col_off)];
            }
            poseidon_permute(s);
        }
    }
    size_t slot = do_bitrev ? (size_t)bitrev32((u32)j, log_rows) : j;
    ulonglong2 *o = reinterpret_cast<ulonglong2 *>(digests + 4 * slot);
    o[0] = make_ulonglong2(gl_canon(s[0]), gl_canon(s[1]));
    o[1] = make_ulonglong2(gl_canon(s[2]), gl_canon(s[3]));
}

// The same for matrices with few rows (FRI commit-phase leaves, the LDE of a 2^12-row table): one row per 16-lane group
// and the lane-cooperative permutation (poseidon.cuh), so that 2^12 rows are 1024 waves instead of 64 and a sponge step
// costs the ~16 us latency of the cooperative permutation instead of the ~58 us of a lone wave's.  Lane e < 8 of a group
// loads column c + e of the chunk being absorbed.
template <bool GATHER>
__global__ void __launch_bounds__(256)
poseidon_hash_rows_coop_kernel(const u64 *__restrict__ cols, size_t col_stride, const u64 *__restrict__ col_off,
                               u32 n_cols, u32 n_rows, int log_rows, int do_bitrev, u64 *__restrict__ digests) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 j = t >> 4, e = t & 15;
    if (j >= n_rows) return;                           // whole groups leave together
    const u64 *p = cols + j;
    const size_t slot = do_bitrev ? (size_t)bitrev32(j, log_rows) : j;
    if (n_cols <= 4) {                                 // hash_or_noop: the elements themselves, zero padded
        if (e < 4) digests[4 * slot + e] = e < n_cols ? gl_canon(p[col_offset<GATHER>(e, col_stride, col_off)]) : 0;
        return;
    }
    u64 s = 0;
    for (u32 c = 0; c < n_cols; c += 8) {
        if (e < 8 && c + e < n_cols) s = p[col_offset<GATHER>(c + e, col_stride, col_off)];
        s = poseidon_permute_coop(s, e, threadIdx.x & 63);
    }
    if (e < 4) digests[4 * slot + e] = gl_canon(s);
}

// Poseidon `two_to_one`: parent[i] = P(child[2i] || child[2i+1] || 0^4)[0..4]
static __global__ void __launch_bounds__(256)
poseidon_merkle_level_kernel(const u64 *__restrict__ child, u64 *__restrict__ parent, size_t n_parent) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parent) return;
    const ulonglong2 *c = reinterpret_cast<const ulonglong2 *>(child + 8 * i);
    ulonglong2 a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
    u64 s[12] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a3.y, 0, 0, 0, 0};
    poseidon_permute(s);
    ulonglong2 *o = reinterpret_cast<ulonglong2 *>(parent + 4 * i);
    o[0] = make_ulonglong2(gl_canon(s[0]), gl_canon(s[1]));
    o[1] = make_ulonglong2(gl_canon(s[2]), gl_canon(s[3]));
}

// two_to_one for the small levels near the cap: one parent per 16-lane group (poseidon.cuh, cooperative permutation)
static __global__ void __launch_bounds__(256)
poseidon_merkle_level_coop_kernel(const u64 *__restrict__ child, u64 *__restrict__ parent, u32 n_parent) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 node = t >> 4, e = t & 15;
    if (node >= n_parent) return;                      // whole groups leave together
    u64 s = e < 8 ? child[(size_t)8 * node + e] : 0;
    s = poseidon_permute_coop(s, e, threadIdx.x & 63);
    if (e < 4) parent[(size_t)4 * node + e] = gl_canon(s);
}

// The same level of SEVERAL trees in one launch (r05: the trace commitments of a segment are independent of the transcript, so
// the latency-bound small levels -- 16-45 us each whatever their size -- are paid once per level instead of once per tree and
// level; merkle_host.inc merkle_levels_batched).  blockIdx.y = tree; every tree has n_parent nodes at this level.
#define ZK_MERKLE_MULTI_MAX 16
struct MerkleTrees { const u64 *child[ZK_MERKLE_MULTI_MAX]; u64 *parent[ZK_MERKLE_MULTI_MAX]; };
static __global__ void __launch_bounds__(256)
poseidon_merkle_level_multi_kernel(MerkleTrees t, size_t n_parent) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parent) return;
    const ulonglong2 *c = reinterpret_cast<const ulonglong2 *>(t.child[blockIdx.y] + 8 * i);
    ulonglong2 a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
    u64 s[12] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a3.y, 0, 0, 0, 0};
    poseidon_permute(s);
    ulonglong2 *o = reinterpret_cast<ulonglong2 *>(t.parent[blockIdx.y] + 4 * i);
    o[0] = make_ulonglong2(gl_canon(s[0]), gl_canon(s[1]));
    o[1] = make_ulonglong2(gl_canon(s[2]), gl_canon(s[3]));
}
static __global__ void __launch_bounds__(256)
poseidon_merkle_level_multi_coop_kernel(MerkleTrees t, u32 n_parent) {
    const u32 th = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 node = th >> 4, e = th & 15;
    if (node >= n_parent) return;                      // whole groups leave together
    u64 s = e < 8 ? t.child[blockIdx.y][(size_t)8 * node + e] : 0;
    s = poseidon_permute_coop(s, e, threadIdx.x & 63);
    if (e < 4) t.parent[blockIdx.y][(size_t)4 * node + e] = gl_canon(s);
}

static __global__ void poseidon_permute_states_kernel(u64 *states, size_t n_states) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_states) return;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) s[k] = states[12 * i + k];
    poseidon_permute(s);
#pragma unroll
    for (int k = 0; k < 12; ++k) states[12 * i + k] = gl_canon(s[k]);
}

static __global__ void keccak_f1600_states_kernel(u64 *states, size_t n_states) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_states) return;
    u64 a[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) a[k] = states[25 * i + k];
    keccak_f1600(a);
#pragma unroll
    for (int k = 0; k < 25; ++k) states[25 * i + k] = a[k];
}

// store the first 25 bytes of the Keccak state into a zero-padded 32-byte slot
__device__ __forceinline__ void keccak25_store(const u64 (&a)[25], u64 *slot) {
    ulonglong2 *o = reinterpret_cast<ulonglong2 *>(slot);
    o[0] = make_ulonglong2(a[0], a[1]);
    o[1] = make_ulonglong2(a[2], a[3] & 0xFFULL);
}

// KeccakHash<25>::hash_or_noop of each row (elements encoded as canonical LE u64).
template <bool GATHER>
__global__ void __launch_bounds__(256)
keccak_hash_rows_kernel(const u64 *__restrict__ cols, size_t col_stride, const u64 *__restrict__ col_off,
                        u32 n_cols, size_t n_rows, int log_rows, int do_bitrev,
                        u64 *__restrict__ digests) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_rows) return;
    const u64 *p = cols + j;
    u64 a[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) a[i] = 0;
    size_t slot = do_bitrev ? (size_t)bitrev32((u32)j, log_rows) : j;
    if (n_cols * 8 <= 25) {  // noop: raw bytes.  (Compile-time indices only: one run-time index into a[] puts the whole
                             // sponge state in scratch memory for every Keccak-f round of the kernel.)
#pragma unroll
        for (u32 c = 0; c < 3; ++c)
            if (c < n_cols) a[c] = gl_canon(p[col_offset<GATHER>(c, col_stride, col_off)]);
        ulonglong2 *o = reinterpret_cast<ulonglong2 *>(digests + 4 * slot);
        o[0] = make_ulonglong2(a[0], a[1]);
        o[1] = make_ulonglong2(a[2], 0);
        return;
    }
    u32 c = 0;
    for (; c + 17 <= n_cols; c += 17) {  // full 136-byte blocks
#pragma unroll
        for (int i = 0; i < 17; ++i) a[i] ^= gl_canon(p[col_offset<GATHER>(c + i, col_stride, col_off)]);
        keccak_f1600(a);
    }
    // final (possibly empty) partial block + padding; message is word aligned
    u32 rem = n_cols - c;
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        if ((u32)i < rem) a[i] ^= gl_canon(p[col_offset<GATHER>(c + i, col_stride, col_off)]);
        if ((u32)i == rem) a[i] ^= 0x01ULL;
    }
    a[16] ^= 0x8000000000000000ULL;
    keccak_f1600(a);
    keccak25_store(a, digests + 4 * slot);
}

// KeccakHash<25>::two_to_one: keccak256(left[0..25] || right[0..25])[0..25]
static __global__ void __launch_bounds__(256)
keccak_merkle_level_kernel(const u64 *__restrict__ child, u64 *__restrict__ parent, size_t n_parent) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parent) return;
    const u64 *l = child + 8 * i, *r = l + 4;
    u64 a[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) a[k] = 0;
    // 50-byte message: bytes 0..24 = left, 25..49 = right (right shifted by one byte)
    u64 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3] & 0xFF;
    a[0] = l[0]; a[1] = l[1]; a[2] = l[2];
    a[3] = (l[3] & 0xFF) | (r0 << 8);
    a[4] = (r0 >> 56) | (r1 << 8);
    a[5] = (r1 >> 56) | (r2 << 8);
    a[6] = (r2 >> 56) | (r3 << 8) | (0x01ULL << 16);  // byte 50 = 0x01 pad
    a[16] ^= 0x8000000000000000ULL;
    keccak_f1600(a);
    keccak25_store(a, parent + 4 * i);
}
