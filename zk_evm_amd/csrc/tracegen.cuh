// Trace finalisation on the device, first table: the Keccak table.
// Reference: evm_arithmetization/src/keccak/keccak_stark.rs:65-234 (`generate_trace_rows`): 24 rows of 2431
// columns per permutation (467 kB of trace from 200 bytes of input), zero rows up to the padded height.
//
// MI355X mapping: one lane per ROW, so that for every column a wave writes 64 consecutive u64 (the trace is
// column-major).  The lane recomputes the state entering its round from the permutation input (<= 23 rounds of
// keccak-f on 25 registers: noise next to the 2431 stores) instead of chaining rows through memory.
#pragma once
#include "gl.cuh"
#include "poseidon.cuh"

#define ZK_KECCAK_COLUMNS 2431
__device__ static const u64 ZK_KTRACE_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000AULL, 0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
// rotation offsets R[x][y] (keccak/columns.rs:43-49)
__device__ static const unsigned char ZK_KTRACE_R[5][5] = {
    {0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};

__device__ __forceinline__ u64 rotl64(u64 v, unsigned r) { return r ? (v << r) | (v >> (64 - r)) : v; }

// the intermediate values of one round, in the table's own terms (a[x][y] indexed as the reference's reg_a(x, y))
struct KeccakRound {
    u64 c[5], cp[5], ap[5][5], app[5][5];
};
__device__ __forceinline__ void keccak_round_parts(const u64 (&a)[5][5], KeccakRound &k) {
#pragma unroll
    for (int x = 0; x < 5; ++x) k.c[x] = a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4];
#pragma unroll
    for (int x = 0; x < 5; ++x) k.cp[x] = k.c[x] ^ k.c[(x + 4) % 5] ^ rotl64(k.c[(x + 1) % 5], 1);   // C[x+1, z-1]
#pragma unroll
    for (int x = 0; x < 5; ++x)
#pragma unroll
        for (int y = 0; y < 5; ++y) k.ap[x][y] = a[x][y] ^ k.c[x] ^ k.cp[x];
    // B[x, y, z] = A'[(x + 3y) % 5, x, z - R]  (reg_b): B[x][y] = rotl(A'[a][b], R[a][b])
#pragma unroll
    for (int y = 0; y < 5; ++y) {
        u64 b[5];
#pragma unroll
        for (int x = 0; x < 5; ++x) { const int aa = (x + 3 * y) % 5; b[x] = rotl64(k.ap[aa][x], ZK_KTRACE_R[aa][x]); }
#pragma unroll
        for (int x = 0; x < 5; ++x) k.app[x][y] = b[x] ^ (~b[(x + 1) % 5] & b[(x + 2) % 5]);
    }
}

// inputs: [n_perms][25] (reference order: input[y * 5 + x]); out: column-major, column c at out + c * stride
static __global__ void __launch_bounds__(256)
keccak_trace_kernel(const u64 *__restrict__ inputs, const u64 *__restrict__ timestamps, u32 n_perms, u32 n_rows,
                    u64 *__restrict__ out, size_t stride) {
    const u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    auto put = [&](u32 col, u64 v) { out[(size_t)col * stride + row] = v; };
    const u32 perm = row / 24, rnd = row % 24;
    if (perm >= n_perms) {                              // padding rows are all zero
        for (u32 c = 0; c < ZK_KECCAK_COLUMNS; ++c) put(c, 0);
        return;
    }
    u64 a[5][5];
#pragma unroll
    for (int x = 0; x < 5; ++x)
#pragma unroll
        for (int y = 0; y < 5; ++y) a[x][y] = inputs[(size_t)perm * 25 + y * 5 + x];
    KeccakRound k;
    for (u32 r = 0; r < rnd; ++r) {                     // state entering this row's round
        keccak_round_parts(a, k);
#pragma unroll
        for (int x = 0; x < 5; ++x)
#pragma unroll
            for (int y = 0; y < 5; ++y) a[x][y] = k.app[x][y];
        a[0][0] ^= ZK_KTRACE_RC[r];
    }
    keccak_round_parts(a, k);
    for (u32 s = 0; s < 24; ++s) put(s, s == rnd ? 1 : 0);                      // reg_step
    put(24, timestamps[perm]);                                                   // TIMESTAMP
    for (int x = 0; x < 5; ++x)
        for (int y = 0; y < 5; ++y) {
            const u32 ra = 25 + (x * 5 + y) * 2;                                 // reg_a
            put(ra, a[x][y] & 0xFFFFFFFFULL);
            put(ra + 1, a[x][y] >> 32);
            const u32 rpp = 2315 + x * 10 + y * 2;                               // reg_a_prime_prime
            put(rpp, k.app[x][y] & 0xFFFFFFFFULL);
            put(rpp + 1, k.app[x][y] >> 32);
            for (u32 z = 0; z < 64; ++z) put(715 + x * 320 + y * 64 + z, (k.ap[x][y] >> z) & 1);   // reg_a_prime
        }
    for (int x = 0; x < 5; ++x)
        for (u32 z = 0; z < 64; ++z) {
            put(75 + x * 64 + z, (k.c[x] >> z) & 1);                             // reg_c
            put(395 + x * 64 + z, (k.cp[x] >> z) & 1);                           // reg_c_prime
        }
    for (u32 z = 0; z < 64; ++z) put(2365 + z, (k.app[0][0] >> z) & 1);          // reg_a_prime_prime_0_0_bit
    const u64 appp = k.app[0][0] ^ ZK_KTRACE_RC[rnd];
    put(2429, appp & 0xFFFFFFFFULL);                                             // reg_a_prime_prime_prime(0, 0)
    put(2430, appp >> 32);
}

// ---- range-check finalisation ------------------------------------------------------------------------
// `generate_range_checks` of the Arithmetic / BytePacking / KeccakSponge tables (arithmetic_stark.rs:130-156,
// byte_packing_stark.rs:254-283, keccak_sponge_stark.rs:503-533 -- the same code three times):
//   counter[i] = min(i, range_max - 1);  frequencies[x] = number of cells of the checked columns equal to x.
static __global__ void range_counter_kernel(u64 *__restrict__ counter, u64 *__restrict__ freq, u32 n, u32 range_max) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    counter[i] = i < range_max ? i : range_max - 1;
    freq[i] = 0;
}
// grid: (row blocks, checked columns).  Values below ZK_RC_LDS_BINS go through a per-block LDS histogram -- that is all
// of a byte range, and for the 16-bit range of the Arithmetic table it takes the hot cells (0, 1, the 15 / 16 of the
// offset auxiliary limbs), which as same-address global atomics would serialise in L2; the rest are spread over the
// range and go to global atomics directly.
#define ZK_RC_LDS_BINS 4096
static __global__ void __launch_bounds__(256)
range_histogram_kernel(const u64 *__restrict__ cols, size_t stride, u32 n, u32 rows_per_block, u32 range_max,
                       unsigned long long *__restrict__ freq, int *__restrict__ err_flag) {
    __shared__ u32 bins[ZK_RC_LDS_BINS];
    const u32 lds_bins = range_max < ZK_RC_LDS_BINS ? range_max : ZK_RC_LDS_BINS;
    for (u32 b = threadIdx.x; b < lds_bins; b += blockDim.x) bins[b] = 0;
    __syncthreads();
    const u64 *c = cols + (size_t)blockIdx.y * stride;
    const u32 lo = blockIdx.x * rows_per_block, hi = lo + rows_per_block < n ? lo + rows_per_block : n;
    u32 zeros = 0;
    for (u32 i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const u64 x = gl_canon(c[i]);
        if (x >= range_max) { atomicExch(err_flag, 1); continue; }   // the reference asserts
        if (x == 0) ++zeros;
        else if (x < lds_bins) atomicAdd(&bins[(u32)x], 1u);
        else atomicAdd(&freq[x], 1ULL);
    }
    if (zeros) atomicAdd(&bins[0], zeros);
    __syncthreads();
    for (u32 b = threadIdx.x; b < lds_bins; b += blockDim.x)
        if (bins[b]) atomicAdd(&freq[b], (unsigned long long)bins[b]);
}

// ---- Logic table ----------------------------------------------------------------------------------------
// `LogicStark::generate_trace_rows` / `Operation::into_row` (evm_arithmetization/src/logic.rs:165-240): per
// operation a one-hot flag (0 AND, 1 OR, 2 XOR), the 2 x 256 input bits and the 8 x 32-bit result limbs; zero rows up
// to the padded height.  ops: [n_ops][9] = {operator, input0 limbs[4], input1 limbs[4]} (64-bit little-endian limbs,
// `U256.0`).  One lane per row, 523 coalesced column stores.
static __global__ void __launch_bounds__(256)
logic_trace_kernel(const u64 *__restrict__ ops, u32 n_ops, u32 n_rows, u64 *__restrict__ out, size_t stride) {
    const u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    auto put = [&](u32 col, u64 v) { out[(size_t)col * stride + row] = v; };
    if (row >= n_ops) {
        for (u32 c = 0; c < 523; ++c) put(c, 0);
        return;
    }
    const u64 *o = ops + (size_t)row * 9;
    const u32 op = (u32)o[0];
    for (u32 k = 0; k < 3; ++k) put(k, k == op ? 1 : 0);
    for (u32 l = 0; l < 4; ++l) {
        const u64 a = o[1 + l], b = o[5 + l];
        const u64 r = op == 0 ? (a & b) : (op == 1 ? (a | b) : (a ^ b));
        for (u32 i = 0; i < 64; ++i) {
            put(3 + 64 * l + i, (a >> i) & 1);
            put(259 + 64 * l + i, (b >> i) & 1);
        }
        put(515 + 2 * l, r & 0xFFFFFFFFULL);
        put(516 + 2 * l, r >> 32);
    }
}

// ---- MemBefore / MemAfter table -----------------------------------------------------------------------------
// `mem_before_values_to_rows` + `MemoryContinuationStark::generate_trace`
// (memory_continuation/memory_continuation_stark.rs:53-98): FILTER = 1, (context, segment, virt), eight 32-bit value
// limbs; zero rows up to the padded height.  entries: [n][7] = {context, segment, virt, value as 4 x 64-bit LE limbs}.
static __global__ void mem_continuation_trace_kernel(const u64 *__restrict__ entries, u32 n, u32 n_rows, u64 *__restrict__ out,
                                              size_t stride) {
    const u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const bool live = row < n;
    const u64 *e = live ? entries + (size_t)row * 7 : nullptr;       // (`entries` is null for an empty table: no arithmetic on it -- UBSan, r06c)
    out[row] = live ? 1 : 0;
    for (u32 k = 0; k < 3; ++k) out[(size_t)(1 + k) * stride + row] = live ? e[k] : 0;
    for (u32 l = 0; l < 4; ++l) {
        const u64 v = live ? e[3 + l] : 0;
        out[(size_t)(4 + 2 * l) * stride + row] = v & 0xFFFFFFFFULL;
        out[(size_t)(5 + 2 * l) * stride + row] = v >> 32;
    }
}

// ---- BytePacking table --------------------------------------------------------------------------------------
// `BytePackingStark::generate_trace_rows` / `generate_row_for_op` (byte_packing/byte_packing_stark.rs:194-251); the
// range-check columns are added afterwards by range_counter_kernel / range_histogram_kernel.
// ops: [n][10] = {is_read, context, segment, virt, timestamp, len (1..32), bytes as 4 x u64 (byte k of the sequence
// at bits 8*(k%8) of word k/8)}.  value_bytes[i] = bytes[len - 1 - i].
static __global__ void byte_packing_trace_kernel(const u64 *__restrict__ ops, u32 n_ops, u32 n_rows, u64 *__restrict__ out,
                                          size_t stride) {
    const u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    auto put = [&](u32 col, u64 v) { out[(size_t)col * stride + row] = v; };
    if (row >= n_ops) {
        for (u32 c = 0; c < 71; ++c) put(c, 0);
        return;
    }
    const u64 *o = ops + (size_t)row * 10;
    const u32 len = (u32)o[5];
    put(0, o[0]);
    for (u32 i = 0; i < 32; ++i) put(1 + i, i + 1 == len ? 1 : 0);        // index_len
    put(33, o[1]); put(34, o[2]); put(35, o[3]); put(36, o[4]);
    for (u32 i = 0; i < 32; ++i) {
        u64 b = 0;
        if (i < len) { const u32 k = len - 1 - i; b = (o[6 + k / 8] >> (8 * (k % 8))) & 0xFF; }
        put(37 + i, b);
    }
    put(69, 0); put(70, 0);
}

// ---- KeccakSponge table ---------------------------------------------------------------------------------------
// `KeccakSpongeStark::generate_rows_for_op` / `generate_common_fields` (keccak_sponge/keccak_sponge_stark.rs:298-494).
// The blocks of one input are chained through the sponge state, so one lane walks one operation and writes its
// len/136 + 1 rows (the table is small: <= 2^13 rows in production, scripts/prove_stdio.rs:94); padding rows and the
// range-check columns come from a memset and the range kernels.
// ops: [n][7] = {context, segment, virt, timestamp, input length, byte offset into `data`, first row}.
__device__ __forceinline__ void keccakf_trace(u64 (&st)[25]) {      // plain keccak-f[1600] on the standard lane order
    u64 a[5][5];
#pragma unroll
    for (int x = 0; x < 5; ++x)
#pragma unroll
        for (int y = 0; y < 5; ++y) a[x][y] = st[y * 5 + x];
    KeccakRound k;
    for (u32 r = 0; r < 24; ++r) {
        keccak_round_parts(a, k);
#pragma unroll
        for (int x = 0; x < 5; ++x)
#pragma unroll
            for (int y = 0; y < 5; ++y) a[x][y] = k.app[x][y];
        a[0][0] ^= ZK_KTRACE_RC[r];
    }
#pragma unroll
    for (int x = 0; x < 5; ++x)
#pragma unroll
        for (int y = 0; y < 5; ++y) st[y * 5 + x] = a[x][y];
}
static __global__ void keccak_sponge_trace_kernel(const u64 *__restrict__ ops, const unsigned char *__restrict__ data, u32 n_ops,
                                           u64 *__restrict__ out, size_t stride) {
    const u32 op = blockIdx.x * blockDim.x + threadIdx.x;
    if (op >= n_ops) return;
    const u64 *o = ops + (size_t)op * 7;
    const u64 len = o[4];
    const unsigned char *in = data + o[5];
    u32 row = (u32)o[6];
    auto put = [&](u32 col, u64 v) { out[(size_t)col * stride + row] = v; };
    u64 st[25];
    for (int i = 0; i < 25; ++i) st[i] = 0;
    for (u64 absorbed = 0;; absorbed += 136, ++row) {
        const u64 left = len - absorbed;
        const bool full = left >= 136;
        put(0, full ? 1 : 0);
        put(1, o[0]); put(2, o[1]); put(3, o[2]); put(4, o[3]); put(5, absorbed);
        unsigned char blk[136];
        for (u32 i = 0; i < 136; ++i) {
            unsigned char b = i < left ? in[absorbed + i] : 0;
            if (!full) {                                              // pad10*1
                if (i == left) b = left == 135 ? 0x81 : 0x01;
                else if (i == 135) b = 0x80;
            }
            blk[i] = b;
            put(192 + i, b);                                          // block_bytes
            put(6 + i, (!full && i >= left) ? 1 : 0);                 // is_padding_byte
        }
        for (u32 i = 0; i < 50; ++i) {                                // original rate / capacity u32s
            const u64 w = (st[i / 2] >> (32 * (i % 2))) & 0xFFFFFFFFULL;
            put(i < 34 ? 142 + i : 176 + (i - 34), w);
        }
        for (u32 i = 0; i < 17; ++i) {                                // xor the block into the rate
            u64 w = 0;
            for (u32 j = 0; j < 8; ++j) w |= (u64)blk[8 * i + j] << (8 * j);
            st[i] ^= w;
            put(328 + 2 * i, st[i] & 0xFFFFFFFFULL);                  // xored_rate_u32s
            put(329 + 2 * i, st[i] >> 32);
        }
        keccakf_trace(st);
        for (u32 i = 8; i < 50; ++i) put(362 + (i - 8), (st[i / 2] >> (32 * (i % 2))) & 0xFFFFFFFFULL);   // partial_updated_state
        for (u32 i = 0; i < 32; ++i) put(404 + i, (st[i / 8] >> (8 * (i % 8))) & 0xFF);                  // updated_digest_state_bytes
        put(436, 0); put(437, 0);
        if (!full) break;
    }
}

// ---- Poseidon table (`cdk_erigon`) -----------------------------------------------------------------------------
// `PoseidonStark::generate_trace_rows` / `generate_perm` (poseidon/poseidon_stark.rs:183-405).  One lane per unit: an
// operation (a general operation walks its 56-byte blocks in sequence: the digest of one row is the capacity of the
// next) or one padding row (the permutation of the zero state).  The table is zero-filled beforehand.
// tab: [n_ops][16] = {kind, a1..a12 (simple: the 12 inputs; general: context, segment, virt, timestamp, len, bytes),
//                     byte offset into `data`, first row, unused}
__device__ inline void poseidon_table_perm_row(u64 *__restrict__ out, size_t cs, u32 row, u64 (&s)[12]) {
    enum : u32 { INPUT = 15, CUBED_FULL = 27, CUBED_PARTIAL = 123, FULL_SBOX_0 = 145, PARTIAL_SBOX = 181, FULL_SBOX_1 = 203,
                 DIGEST = 251, OUTPUT_PARTIAL = 259, PINV = 267 };
    auto put = [&](u32 col, u64 v) { out[(size_t)col * cs + row] = v; };
#pragma unroll
    for (u32 i = 0; i < 12; ++i) { s[i] = gl_canon(s[i]); put(INPUT + i, s[i]); s[i] = gl_add(s[i], ZK_RC[i]); }
    int round = 0;
#pragma unroll 1
    for (u32 r = 0; r < 4; ++r) {
#pragma unroll
        for (u32 i = 0; i < 12; ++i) {
            const u64 x = gl_canon(s[i]);
            if (r != 0) put(FULL_SBOX_0 + 12 * (r - 1) + i, x);
            const u64 cube = gl_canon(gl_mul(gl_sqr(x), x));
            put(CUBED_FULL + 12 * r + i, cube);
            s[i] = gl_mul(x, gl_sqr(cube));
        }
        ++round;
        pos_mds<true>(s, &ZK_RCS[round * 12]);
    }
#pragma unroll 1
    for (u32 r = 0; r < 22; ++r) {
        const u64 x = gl_canon(s[0]);
        put(PARTIAL_SBOX + r, x);
        const u64 cube = gl_canon(gl_mul(gl_sqr(x), x));
        put(CUBED_PARTIAL + r, cube);
        s[0] = gl_mul(x, gl_sqr(cube));
        ++round;
        pos_mds<true>(s, &ZK_RCS[round * 12]);
    }
#pragma unroll 1
    for (u32 r = 0; r < 4; ++r) {
#pragma unroll
        for (u32 i = 0; i < 12; ++i) {
            const u64 x = gl_canon(s[i]);
            put(FULL_SBOX_1 + 12 * r + i, x);
            const u64 cube = gl_canon(gl_mul(gl_sqr(x), x));
            put(CUBED_FULL + 12 * (4 + r) + i, cube);
            s[i] = gl_mul(x, gl_sqr(cube));
        }
        ++round;
        if (r < 3) pos_mds<true>(s, &ZK_RCS[round * 12]);
        else pos_mds<false>(s, nullptr);
    }
#pragma unroll
    for (u32 i = 0; i < 12; ++i) s[i] = gl_canon(s[i]);
    for (u32 i = 0; i < 4; ++i) {
        const u64 lo = s[i] & 0xFFFFFFFFull, hi = s[i] >> 32;
        const u64 d = gl_canon(gl_sub(hi, 0xFFFFFFFFull));
        put(PINV + i, d ? gl_canon(gl_inv(d)) : 0);
        put(DIGEST + 2 * i, lo);
        put(DIGEST + 2 * i + 1, hi);
    }
    for (u32 i = 4; i < 12; ++i) put(OUTPUT_PARTIAL + i - 4, s[i]);
}

static __global__ void __launch_bounds__(64)
poseidon_table_trace_kernel(const u64 *__restrict__ tab, const unsigned char *__restrict__ data, u32 n_ops, u32 rows_used, u32 n_rows,
                            u64 *__restrict__ out, size_t cs) {
    enum : u32 { CONTEXT = 0, SEGMENT, VIRT, TIMESTAMP, LEN, ALREADY_ABSORBED, IS_FINAL_INPUT_LEN = 6, IS_FULL_INPUT_BLOCK = 14,
                 INPUT_BYTES = 271, IS_SIMPLE_OP = 319, IS_FIRST_ROW_GENERAL_OP = 320, NOT_PADDING = 321 };
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 n_pad = n_rows - rows_used;
    if (t >= n_ops + n_pad) return;
    auto put = [&](u32 col, u32 row, u64 v) { out[(size_t)col * cs + row] = v; };
    u64 s[12];
    if (t >= n_ops) {                                            // padding row
#pragma unroll
        for (u32 i = 0; i < 12; ++i) s[i] = 0;
        poseidon_table_perm_row(out, cs, rows_used + (t - n_ops), s);
        return;
    }
    const u64 *o = tab + (size_t)t * 16;
    const u32 row0 = (u32)o[14];
    if (o[0] == 0) {                                             // generate_row_for_simple_op
#pragma unroll
        for (u32 i = 0; i < 12; ++i) s[i] = o[1 + i];
        poseidon_table_perm_row(out, cs, row0, s);
        put(IS_FINAL_INPUT_LEN + 7, row0, 1);
        put(NOT_PADDING, row0, 1);
        put(IS_SIMPLE_OP, row0, 1);
        return;
    }
    const u64 len = o[5];
    const u32 n_blocks = (u32)(o[6] / 56), last_non_padding = (u32)(len % 56);
    const unsigned char *in = data + o[13];
    u64 cap[4] = {0, 0, 0, 0}, absorbed = 0;
    for (u32 k = 0; k < n_blocks; ++k) {                         // generate_rows_for_general_op
        const u32 row = row0 + k;
        const unsigned char *blk = in + (size_t)k * 56;
        for (u32 i = 0; i < 8; ++i) {
            u64 v = 0;
            for (u32 j = 0; j < 7; ++j) v |= (u64)blk[7 * i + j] << (8 * j);
            s[i] = v;
            for (u32 j = 0; j < 6; ++j) put(INPUT_BYTES + 6 * i + j, row, blk[7 * i + 1 + j]);
        }
#pragma unroll
        for (u32 i = 0; i < 4; ++i) s[8 + i] = cap[i];
        const bool is_last = k + 1 == n_blocks;
        if (is_last) put(IS_FINAL_INPUT_LEN + last_non_padding, row, 1);
        else put(IS_FULL_INPUT_BLOCK, row, 1);
        put(CONTEXT, row, o[1]); put(SEGMENT, row, o[2]); put(VIRT, row, o[3]); put(TIMESTAMP, row, o[4]);
        put(LEN, row, len); put(ALREADY_ABSORBED, row, absorbed);
        put(NOT_PADDING, row, 1);
        poseidon_table_perm_row(out, cs, row, s);
        absorbed += is_last ? last_non_padding : 56;
#pragma unroll
        for (u32 i = 0; i < 4; ++i) cap[i] = s[i];              // digest limbs recombined = the canonical word
    }
    put(IS_FIRST_ROW_GENERAL_OP, row0, 1);
}
