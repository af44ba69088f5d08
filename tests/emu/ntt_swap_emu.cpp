// TEST INFRASTRUCTURE.  The lane-swap NTT kernels (zk_evm_amd/csrc/ntt_swap.cuh) compiled for the CPU from their own source and run
// with one OS thread per lane: threadIdx / blockIdx are thread-local, __syncthreads is a pthread barrier over the workgroup, the
// v_permlane16/32_swap exchange and the wave-local LDS synchronisation are rendezvous of a wave's 64 threads, `extern __shared__`
// arrays are process globals (one workgroup runs at a time).  Each kernel is compared with the plain definition of its pass -- the
// stages of ntt.cuh's ntt_pass_kernel / ntt_step, one butterfly at a time, written here without any of the kernels' bookkeeping.
// What this cannot check: the hardware's semantics of the two swap instructions (assumed: odd rows / upper half of the first
// operand against even rows / lower half of the second, as LLVM documents) and anything about speed.
//   g++ -std=c++17 -O1 -pthread -DZK_NTT_EMULATE -I tests/emu -I zk_evm_amd/csrc tests/emu/ntt_swap_emu.cpp -o /tmp/ntt_swap_emu
#include <hip/hip_runtime.h>

#include <pthread.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>

thread_local EmuIdx threadIdx, blockIdx, blockDim, gridDim;

// ---- one workgroup at a time ------------------------------------------------------------------------------------------------
static pthread_barrier_t g_block_barrier;
struct WaveRendezvous {
    pthread_barrier_t bar;
    unsigned long long a[64], b[64];
};
static std::vector<WaveRendezvous *> g_waves;
static thread_local WaveRendezvous *t_wave;
static thread_local unsigned t_lane;

void __syncthreads() { pthread_barrier_wait(&g_block_barrier); }
void zk_emu_wave_sync() { pthread_barrier_wait(&t_wave->bar); }
void zk_emu_lane_swap(int lanebit, unsigned long long &a, unsigned long long &b) {
    WaveRendezvous *w = t_wave;
    w->a[t_lane] = a; w->b[t_lane] = b;
    pthread_barrier_wait(&w->bar);
    const unsigned partner = t_lane ^ (1u << lanebit);
    if ((t_lane >> lanebit) & 1) a = w->b[partner];       // (register a, lane bit = 1) <-> (register b, lane bit = 0)
    else b = w->a[partner];
    pthread_barrier_wait(&w->bar);
}

__attribute__((aligned(16))) unsigned long long tile[1 << 14];        // `extern __shared__` of the strided kernel (R <= 10)
__attribute__((aligned(16))) unsigned long long lds_all[4 * 1088];    // ... of the wave kernels

#include "gl.cuh"
#include "ntt_common.cuh"
#include "ntt_swap.cuh"

static void run_grid(unsigned gx, unsigned gy, unsigned nthr, const std::function<void()> &kernel) {
    for (unsigned by = 0; by < gy; ++by)
        for (unsigned bx = 0; bx < gx; ++bx) {
            pthread_barrier_init(&g_block_barrier, nullptr, nthr);
            const unsigned nw = nthr / 64;
            g_waves.clear();
            for (unsigned w = 0; w < nw; ++w) { auto *r = new WaveRendezvous; pthread_barrier_init(&r->bar, nullptr, 64); g_waves.push_back(r); }
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nthr; ++t)
                th.emplace_back([=, &kernel] {
                    threadIdx = {t, 0, 0}; blockIdx = {bx, by, 0}; blockDim = {nthr, 1, 1}; gridDim = {gx, gy, 1};
                    t_wave = g_waves[t / 64]; t_lane = t % 64;
                    kernel();
                });
            for (auto &t : th) t.join();
            for (auto *r : g_waves) { pthread_barrier_destroy(&r->bar); delete r; }
            pthread_barrier_destroy(&g_block_barrier);
        }
}

// ---- the definition of a pass ---------------------------------------------------------------------------------------------
static u64 rnd_state = 0x9E3779B97F4A7C15ULL;
static u64 rnd() { rnd_state ^= rnd_state << 13; rnd_state ^= rnd_state >> 7; rnd_state ^= rnd_state << 17; return rnd_state; }
static void bfly_ref(u64 &a, u64 &b, u64 w) {
    const u64 t = gl_canon(gl_mul_ref(gl_canon(b), gl_canon(w)));
    const u64 na = gl_canon(gl_add_ref(gl_canon(a), t));
    b = gl_canon(gl_sub_ref(gl_canon(a), t));
    a = na;
}
static void reference_pass(std::vector<u64> &d, int log_n, int log_d, int r, const std::vector<u64> &tw, bool dit, int first_stage) {
    for (int s = 0; s < r - first_stage; ++s) {
        const int k = dit ? first_stage + s : r - 1 - s;
        const int log_D = log_d + k;
        const size_t D = (size_t)1 << log_D;
        for (size_t x = 0; x < ((size_t)1 << log_n); ++x) {
            if (x & D) continue;
            const u64 w = dit ? tw[D - 1 + (x & (D - 1))] : tw[(((size_t)1 << (log_n - 1 - log_D)) - 1) + (x >> (log_D + 1))];
            bfly_ref(d[x], d[x + D], w);
        }
    }
}
static std::vector<u64> level_table(int log_size) {          // ntt_host.inc get_twiddle_levels
    std::vector<u64> tw((size_t)1 << log_size, 0);
    for (int lv = 0; lv < log_size; ++lv) {
        const size_t D = (size_t)1 << lv;
        const u64 w = gl_root_of_unity(lv + 1);
        u64 x = 1;
        for (size_t k = 0; k < D; ++k) { tw[D - 1 + k] = x; x = gl_canon(gl_mul_ref(x, w)); }
    }
    return tw;
}
static bool same(const std::vector<u64> &got, const std::vector<u64> &want, const char *what) {
    for (size_t i = 0; i < want.size(); ++i)
        if (gl_canon(got[i]) != gl_canon(want[i])) { printf("MISMATCH %s at %zu: %llx != %llx\n", what, i, (unsigned long long)got[i], (unsigned long long)want[i]); return false; }
    return true;
}

template <bool DIT, int R>
static bool check_strided(int log_d, int extra_hi, unsigned n_cols) {
    const int log_n = log_d + R + extra_hi;
    const size_t n = (size_t)1 << log_n;
    std::vector<u64> src(n * n_cols), tw(n), dst(n * n_cols, 0);
    for (auto &x : src) x = rnd();
    for (auto &x : tw) x = gl_canon(rnd());
    NttPass p = {};
    p.src = src.data(); p.dst = dst.data(); p.src_stride = p.dst_stride = n; p.tw = tw.data(); p.log_tw = log_n; p.log_n = log_n;
    p.log_d = log_d; p.r = R; p.log_t = 4; p.cols_fastest = 1; p.last_pass = DIT ? 1 : 0;
    const unsigned tiles = (unsigned)(n >> (R + 4));
    run_grid(n_cols, tiles, 64u << (R - 6), [&] { ntt_strided_swap_kernel<DIT, R>(p); });
    bool ok = true;
    for (unsigned c = 0; c < n_cols && ok; ++c) {
        std::vector<u64> want(src.begin() + c * n, src.begin() + (c + 1) * n), got(dst.begin() + c * n, dst.begin() + (c + 1) * n);
        reference_pass(want, log_n, log_d, R, tw, DIT, 0);
        char what[96];
        snprintf(what, sizeof what, "strided %s R=%d log_d=%d log_n=%d col %u", DIT ? "DIT" : "DIF", R, log_d, log_n, c);
        ok = same(got, want, what);
        if (DIT) for (u64 x : got) ok = ok && x < GL_P;          // last pass: canonical representatives
    }
    printf("%s strided %s R=%d log_d=%d log_n=%d cols=%u\n", ok ? "ok  " : "FAIL", DIT ? "DIT" : "DIF", R, log_d, log_n, n_cols);
    return ok;
}

template <bool DIT, int R>
static bool check_reg(int log_d, int extra_hi, unsigned n_cols) {              // ntt_strided_reg_kernel: one wave per tile, no LDS
    const int log_n = log_d + R + extra_hi;
    const size_t n = (size_t)1 << log_n;
    std::vector<u64> src(n * n_cols), tw(n), dst(n * n_cols, 0);
    for (auto &x : src) x = rnd();
    for (auto &x : tw) x = gl_canon(rnd());
    NttPass p = {};
    p.src = src.data(); p.dst = dst.data(); p.src_stride = p.dst_stride = n; p.tw = tw.data(); p.log_tw = log_n; p.log_n = log_n;
    p.log_d = log_d; p.r = R; p.cols_fastest = 1; p.last_pass = DIT ? 1 : 0;
    const unsigned blocks = (unsigned)(((n >> 10) + 3) / 4);
    run_grid(n_cols, blocks, 256, [&] { ntt_strided_reg_kernel<DIT, R>(p); });
    bool ok = true;
    for (unsigned c = 0; c < n_cols && ok; ++c) {
        std::vector<u64> want(src.begin() + c * n, src.begin() + (c + 1) * n), got(dst.begin() + c * n, dst.begin() + (c + 1) * n);
        reference_pass(want, log_n, log_d, R, tw, DIT, 0);
        char what[96];
        snprintf(what, sizeof what, "register-only strided %s R=%d log_d=%d log_n=%d col %u", DIT ? "DIT" : "DIF", R, log_d, log_n, c);
        ok = same(got, want, what);
        if (DIT) for (u64 x : got) ok = ok && x < GL_P;
    }
    printf("%s register-only strided %s R=%d log_d=%d log_n=%d cols=%u\n", ok ? "ok  " : "FAIL", DIT ? "DIT" : "DIF", R, log_d, log_n, n_cols);
    return ok;
}

static bool check_contig_dif(int log_n, int factor) {        // factor 0: canonical only, 1: out_const, 2: out_scale
    const size_t n = (size_t)1 << log_n;
    std::vector<u64> src(n), tw(n), dst(n, 0), scale(n);
    for (auto &x : src) x = rnd();
    for (auto &x : tw) x = gl_canon(rnd());
    for (auto &x : scale) x = gl_canon(rnd());
    NttPass p = {};
    p.src = src.data(); p.dst = dst.data(); p.src_stride = p.dst_stride = n; p.tw = tw.data(); p.log_tw = log_n; p.log_n = log_n;
    p.log_d = 0; p.r = 10; p.cols_fastest = 1; p.last_pass = 1;
    if (factor == 1) { p.apply_out_const = 1; p.out_const = gl_canon(rnd()); }
    if (factor == 2) p.out_scale = scale.data();
    const unsigned blocks = (unsigned)(((n >> 10) + 3) / 4);
    run_grid(1, blocks, 256, [&] { ntt_contig_wave_kernel_dif(p); });
    std::vector<u64> want = src;
    reference_pass(want, log_n, 0, 10, tw, false, 0);
    for (size_t i = 0; i < n; ++i) {
        if (factor == 1) want[i] = gl_canon(gl_mul_ref(gl_canon(want[i]), p.out_const));
        if (factor == 2) want[i] = gl_canon(gl_mul_ref(gl_canon(want[i]), scale[i]));
    }
    bool ok = same(dst, want, "contiguous DIF");
    for (u64 x : dst) ok = ok && x < GL_P;
    printf("%s contiguous values->coefficients log_n=%d store factor %d\n", ok ? "ok  " : "FAIL", log_n, factor);
    return ok;
}

template <int NB>
static bool check_contig_dit(int log_src, bool with_scale, bool last) {
    const int log_n = log_src + (NB == 2 ? 1 : 0);
    const size_t ns = (size_t)1 << log_src, n = (size_t)1 << log_n;
    std::vector<u64> src(ns), s0(ns), s1(ns), dst(n, 0);
    const std::vector<u64> tw = level_table(log_n);
    for (auto &x : src) x = rnd();
    for (auto &x : s0) x = gl_canon(rnd());
    const u64 w2048 = gl_root_of_unity(11);
    for (size_t i = 0; i < ns; ++i)                          // ntt.cuh wave_coset2_table_kernel
        s1[i] = gl_canon(gl_mul_ref(with_scale ? s0[i] : 1, gl_pow(w2048, bitrev32((u32)i & 1023u, 10))));
    NttPass p = {};
    p.src = src.data(); p.dst = dst.data(); p.src_stride = ns; p.dst_stride = n; p.tw = tw.data(); p.log_tw = log_n; p.log_n = log_n;
    p.log_d = 0; p.r = 10 + (NB == 2 ? 1 : 0); p.cols_fastest = 1; p.last_pass = last;
    p.in_scale = with_scale ? s0.data() : nullptr; p.in_scale2 = NB == 2 ? s1.data() : nullptr;
    p.log_rep = p.first_stage = NB == 2 ? 1 : 0;
    const unsigned blocks = (unsigned)(((ns >> 10) + 3) / 4);
    run_grid(1, blocks, 256, [&] { ntt_contig_wave_kernel_dit<NB>(p, p.in_scale2); });
    std::vector<u64> want(n);
    for (size_t se = 0; se < ns; ++se) {                     // the skipped stages replicate the scaled coefficient
        const u64 v = with_scale ? gl_canon(gl_mul_ref(gl_canon(src[se]), s0[se])) : gl_canon(src[se]);
        for (int j = 0; j < NB; ++j) want[se * NB + j] = v;
    }
    reference_pass(want, log_n, 0, p.r, tw, true, p.first_stage);
    bool ok = same(dst, want, "contiguous DIT");
    if (last) for (u64 x : dst) ok = ok && x < GL_P;
    printf("%s contiguous coefficients->values cosets=%d log_src=%d in_scale=%d last_pass=%d\n", ok ? "ok  " : "FAIL", NB, log_src, with_scale, last);
    return ok;
}

int main(int argc, char **argv) {
    const bool quick = argc > 1 && argv[1][0] == 'q';
    bool ok = true;
    ok &= check_strided<false, 7>(4, 1, 2);
    ok &= check_strided<true, 7>(5, 0, 1);
    ok &= check_strided<false, 9>(4, 0, 1);
    ok &= check_strided<true, 9>(4, 1, 1);
    if (!quick) {
        ok &= check_strided<false, 8>(5, 1, 1);
        ok &= check_strided<true, 8>(4, 0, 2);
        ok &= check_strided<false, 9>(6, 1, 1);
        ok &= check_strided<false, 10>(4, 1, 1);
        ok &= check_strided<true, 10>(5, 0, 1);
    }
    ok &= check_reg<false, 1>(10, 0, 1);
    ok &= check_reg<false, 2>(10, 1, 2);
    ok &= check_reg<false, 3>(10, 0, 1);
    ok &= check_reg<false, 4>(10, 0, 1);
    ok &= check_reg<false, 4>(7, 1, 1);
    ok &= check_reg<false, 5>(10, 0, 1);
    ok &= check_reg<false, 5>(5, 1, 2);
    ok &= check_reg<false, 6>(10, 0, 1);
    ok &= check_reg<false, 6>(4, 2, 1);
    ok &= check_reg<true, 4>(11, 0, 1);
    ok &= check_reg<true, 4>(6, 1, 2);
    ok &= check_reg<true, 5>(11, 0, 1);
    ok &= check_reg<true, 5>(5, 2, 1);
    ok &= check_reg<true, 6>(11, 0, 1);
    ok &= check_reg<true, 6>(4, 1, 1);
    ok &= check_contig_dif(10, 0);
    ok &= check_contig_dif(12, 1);
    ok &= check_contig_dif(11, 2);
    ok &= check_contig_dit<1>(10, false, true);
    ok &= check_contig_dit<1>(12, true, false);
    ok &= check_contig_dit<2>(10, true, true);
    ok &= check_contig_dit<2>(12, true, false);
    ok &= check_contig_dit<2>(11, false, false);
    printf(ok ? "ALL OK\n" : "FAILED\n");
    return ok ? 0 : 1;
}
