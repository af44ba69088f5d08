// Micro-benchmark: issue rate of the integer VALU instructions a Goldilocks/Poseidon kernel can be
// built from, on gfx950.  Each kernel runs ITER iterations of 8 independent dependency chains of
// one instruction; reports wave-instructions per cycle per SIMD assuming 2.4 GHz is NOT needed:
// we report ns per (wave-instruction) per SIMD at full occupancy and the ratio to v_add_u32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITER 4096

#define BODY8(ASM)                                                    \
    for (int it = 0; it < ITER; ++it) {                                \
        ASM(r0) ASM(r1) ASM(r2) ASM(r3) ASM(r4) ASM(r5) ASM(r6) ASM(r7) \
    }

#define DEF_KERNEL32(NAME, INSN)                                                          \
    __global__ void NAME(uint32_t *out, uint32_t a, uint32_t b) {                         \
        uint32_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4,    \
                 r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;                                   \
        uint32_t x = a + threadIdx.x, y = b;                                              \
        for (int it = 0; it < ITER; ++it) {                                               \
            asm volatile(INSN : "+v"(r0) : "v"(x), "v"(y));                               \
            asm volatile(INSN : "+v"(r1) : "v"(x), "v"(y));                               \
            asm volatile(INSN : "+v"(r2) : "v"(x), "v"(y));                               \
            asm volatile(INSN : "+v"(r3) : "v"(x), "v"(y));                               \
            asm volatile(INSN : "+v"(r4) : "v"(x), "v"(y));                               \
            asm volatile(INSN : "+v"(r5) : "v"(x), "v"(y));                               \
            asm volatile(INSN : "+v"(r6) : "v"(x), "v"(y));                               \
            asm volatile(INSN : "+v"(r7) : "v"(x), "v"(y));                               \
        }                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7; \
    }

#define DEF_KERNEL64(NAME, INSN)                                                          \
    __global__ void NAME(uint32_t *out, uint32_t a, uint32_t b) {                         \
        uint64_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4,    \
                 r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;                                   \
        uint32_t x = a + threadIdx.x, y = b;                                              \
        for (int it = 0; it < ITER; ++it) {                                               \
            asm volatile(INSN : "+v"(r0) : "v"(x), "v"(y));                               \
            asm volatile(INSN : "+v"(r1) : "v"(x), "v"(y));                               \
            asm volatile(INSN : "+v"(r2) : "v"(x), "v"(y));                               \
            asm volatile(INSN : "+v"(r3) : "v"(x), "v"(y));                               \
            asm volatile(INSN : "+v"(r4) : "v"(x), "v"(y));                               \
            asm volatile(INSN : "+v"(r5) : "v"(x), "v"(y));                               \
            asm volatile(INSN : "+v"(r6) : "v"(x), "v"(y));                               \
            asm volatile(INSN : "+v"(r7) : "v"(x), "v"(y));                               \
        }                                                                                 \
        uint64_t z = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)z ^ (uint32_t)(z >> 32);   \
    }

DEF_KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
DEF_KERNEL32(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
DEF_KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
DEF_KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
DEF_KERNEL32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
DEF_KERNEL32(k_dot4_u32_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
DEF_KERNEL32(k_dot2_u32_u16, "v_dot2_u32_u16 %0, %1, %2, %0")
DEF_KERNEL32(k_perm_b32, "v_perm_b32 %0, %0, %1, %2")
DEF_KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
DEF_KERNEL32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
DEF_KERNEL32(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 22")
DEF_KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
DEF_KERNEL32(k_pk_mad_u16, "v_pk_mad_u16 %0, %1, %2, %0")
DEF_KERNEL32(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
DEF_KERNEL32(k_mad_u16, "v_mad_u16 %0, %1, %2, %0")
DEF_KERNEL32(k_fma_f32, "v_fma_f32 %0, %1, %2, %0")
DEF_KERNEL32(k_pk_fma_f32_as32, "v_fmac_f32 %0, %1, %2")
DEF_KERNEL32(k_mov_b32, "v_mov_b32 %0, %1")
DEF_KERNEL32(k_and_b32, "v_and_b32 %0, %0, %1")
DEF_KERNEL32(k_or_b32, "v_or_b32 %0, %0, %1")
DEF_KERNEL32(k_xor_b32, "v_xor_b32 %0, %0, %1")
DEF_KERNEL32(k_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
DEF_KERNEL32(k_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
DEF_KERNEL32(k_sub_u32, "v_sub_u32 %0, %0, %1")
DEF_KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL32(k_cndmask_e64, "v_cndmask_b32_e64 %0, 0, -1, vcc")
DEF_KERNEL32(k_add_co_only, "v_add_co_u32 %0, vcc, %0, %1")
DEF_KERNEL32(k_addc_co_only, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
DEF_KERNEL32(k_cmp_lt_u32, "v_cmp_lt_u32 vcc, %0, %1\n v_mov_b32 %0, %2")
DEF_KERNEL32(k_bfi, "v_bfi_b32 %0, %0, %1, %2")
DEF_KERNEL32(k_add_lshl, "v_add_lshl_u32 %0, %0, %1, 3")
DEF_KERNEL32(k_min_u32, "v_min_u32 %0, %0, %1")
DEF_KERNEL64(k_cmp_lt_u64, "v_cmp_lt_u64 vcc, %0, %0")
DEF_KERNEL64(k_mov_b64, "v_mov_b64 %0, %0")
DEF_KERNEL64(k_lshrrev_b64, "v_lshrrev_b64 %0, 1, %0")
DEF_KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
DEF_KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 3, %0")
DEF_KERNEL64(k_lshlrev_b64, "v_lshlrev_b64 %0, 1, %0")
DEF_KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %0, %0")
DEF_KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %0, %0")
DEF_KERNEL64(k_pk_add_u32x, "v_pk_add_f32 %0, %0, %0")
__global__ void k_add_co_pair(uint32_t *out, uint32_t a, uint32_t b) {
    uint32_t lo[8], hi[8];
    for (int i = 0; i < 8; ++i) { lo[i] = threadIdx.x + i; hi[i] = i; }
    uint32_t x = a + threadIdx.x, y = b;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc"
                         : "+v"(lo[i]), "+v"(hi[i]) : "v"(x), "v"(y) : "vcc");
    }
    uint32_t z = 0;
    for (int i = 0; i < 8; ++i) z ^= lo[i] ^ hi[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = z;
}

typedef void (*kern_t)(uint32_t *, uint32_t, uint32_t);
struct Item { const char *name; kern_t k; int insns; };

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    double clk_ghz = prop.clockRate / 1e6;
    printf("device %s, %d CUs, clock %.2f GHz\n", prop.name, cus, clk_ghz);
    // 8 waves per SIMD: 256 threads/block x 8 blocks per CU
    int blocks = cus * 8, threads = 256;
    uint32_t *out;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    std::vector<Item> items = {
        {"v_add_u32", k_add_u32, 1}, {"v_add3_u32", k_add3_u32, 1},
        {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
        {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_mul_u32_u24", k_mul_u32_u24, 1},
        {"v_dot4_u32_u8", k_dot4_u32_u8, 1}, {"v_dot2_u32_u16", k_dot2_u32_u16, 1},
        {"v_perm_b32", k_perm_b32, 1}, {"v_alignbit_b32", k_alignbit, 1},
        {"v_lshl_add_u32", k_lshl_add_u32, 1}, {"v_bfe_u32", k_bfe_u32, 1}, {"v_and_or_b32", k_and_or, 1},
        {"v_pk_mad_u16", k_pk_mad_u16, 1}, {"v_pk_mul_lo_u16", k_pk_mul_lo_u16, 1}, {"v_mad_u16", k_mad_u16, 1},
        {"v_fma_f32", k_fma_f32, 1}, {"v_fmac_f32", k_pk_fma_f32_as32, 1},
        {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_lshl_add_u64", k_lshl_add_u64, 1},
        {"v_lshlrev_b64", k_lshlrev_b64, 1}, {"v_fma_f64", k_fma_f64, 1}, {"v_pk_fma_f32", k_pk_fma_f32, 1},
        {"v_pk_add_f32", k_pk_add_u32x, 1}, {"add_co+addc (64b add)", k_add_co_pair, 2},
        {"v_mov_b32", k_mov_b32, 1}, {"v_and_b32", k_and_b32, 1}, {"v_or_b32", k_or_b32, 1}, {"v_xor_b32", k_xor_b32, 1},
        {"v_lshlrev_b32", k_lshlrev_b32, 1}, {"v_lshrrev_b32", k_lshrrev_b32, 1}, {"v_sub_u32", k_sub_u32, 1},
        {"v_cndmask_b32 (e32)", k_cndmask, 1}, {"v_cndmask_b32_e64 const", k_cndmask_e64, 1},
        {"v_add_co_u32", k_add_co_only, 1}, {"v_addc_co_u32", k_addc_co_only, 1},
        {"v_cmp_lt_u32+v_mov", k_cmp_lt_u32, 2}, {"v_bfi_b32", k_bfi, 1},
        {"v_add_lshl_u32", k_add_lshl, 1}, {"v_min_u32", k_min_u32, 1}, {"v_cmp_lt_u64", k_cmp_lt_u64, 1},
        {"v_mov_b64", k_mov_b64, 1}, {"v_lshrrev_b64", k_lshrrev_b64, 1},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto &it : items) {
        it.k<<<blocks, threads>>>(out, 1, 2);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 5; ++rep) it.k<<<blocks, threads>>>(out, 1, 2);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        // wave-instructions issued per SIMD: waves per SIMD (8) * ITER * 8 chains * insns
        double wi_per_simd = 8.0 * ITER * 8 * it.insns;
        double cycles = ms * 1e-3 * clk_ghz * 1e9;
        printf("%-24s %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", it.name, ms, cycles / wi_per_simd);
    }
    return 0;
}
