// TEST INFRASTRUCTURE.  Whole transforms on the CPU from the library's own kernel sources: the twiddle / coset tables are built by
// the library's table kernels, `PolynomialBatch::from_values`' two transforms (values -> coefficients, coefficients -> values on
// the coset of twice the size) are run pass by pass ONCE through the LDS tile kernels (ntt.cuh ntt_pass_kernel: the r01-r04 code
// that every GPU parity test has pinned) and ONCE through the lane-swap kernels (ntt_swap.cuh) in the plan ntt_host.inc gives them,
// and the two results are compared word for word -- what tools/kbench's checksums compare on the GPU.  Also: the result is the
// transform (a few outputs against a direct evaluation of the polynomial).
// Threads: tests/emu/ntt_swap_emu.cpp's scheme (one OS thread per lane, barriers); kernels without any cross-thread traffic (the
// table builders) run their "threads" one after the other.
//   g++ -std=c++17 -O1 -pthread -DZK_NTT_EMULATE -I tests/emu -I zk_evm_amd/csrc tests/emu/ntt_plan_emu.cpp -o /tmp/ntt_plan_emu
#include <hip/hip_runtime.h>

#include <pthread.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <thread>
#include <type_traits>
#include <vector>

thread_local EmuIdx threadIdx, blockIdx, blockDim, gridDim;
static pthread_barrier_t g_block_barrier;
struct WaveRendezvous { pthread_barrier_t bar; unsigned long long a[64], b[64]; };
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
    if ((t_lane >> lanebit) & 1) a = w->b[partner];
    else b = w->a[partner];
    pthread_barrier_wait(&w->bar);
}
__attribute__((aligned(16))) unsigned long long tile[1 << 14];
__attribute__((aligned(16))) unsigned long long lds_all[4 * 1088];

#include "ntt.cuh"

static void run_grid(unsigned gx, unsigned gy, unsigned nthr, const std::function<void()> &kernel) {
    for (unsigned by = 0; by < gy; ++by)
        for (unsigned bx = 0; bx < gx; ++bx) {
            pthread_barrier_init(&g_block_barrier, nullptr, nthr);
            g_waves.clear();
            for (unsigned w = 0; w < (nthr + 63) / 64; ++w) { auto *r = new WaveRendezvous; pthread_barrier_init(&r->bar, nullptr, 64); g_waves.push_back(r); }
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
static void run_grid_seq(size_t n_threads, const std::function<void()> &kernel) {     // kernels whose threads never meet
    blockDim = {256, 1, 1}; gridDim = {(unsigned)((n_threads + 255) / 256), 1, 1};
    for (size_t i = 0; i < n_threads; ++i) { blockIdx = {(unsigned)(i / 256), 0, 0}; threadIdx = {(unsigned)(i % 256), 0, 0}; kernel(); }
}

// ---- what ntt_host.inc does, for one column batch (restated: that file drives the HIP runtime) ---------------------------------------
struct Plan { int log_d, r; };
static std::vector<u64> levels(int log_size, bool block_order_inverse) {
    std::vector<u64> t((size_t)1 << log_size);
    if (block_order_inverse) { const u64 w = gl_canon(gl_inv(gl_root_of_unity(log_size))); run_grid_seq(t.size(), [&] { twiddle_block_levels_kernel(t.data(), log_size, w); }); }
    else { const u64 w = gl_root_of_unity(log_size); run_grid_seq(t.size(), [&] { twiddle_levels_kernel(t.data(), log_size, w); }); }
    return t;
}
template <bool DIT>
static void launch(NttPass p, unsigned n_cols, bool swap_kernels) {
    const size_t n = (size_t)1 << p.log_n;
    p.cols_fastest = 1;
    if (swap_kernels && p.log_d >= 4 && p.r >= 7) {
        p.log_t = 4;
        const unsigned tiles = (unsigned)(n >> (p.r + 4)), nthr = 64u << (p.r - 6);
        switch (p.r) {
            case 7: run_grid(n_cols, tiles, nthr, [&] { ntt_strided_swap_kernel<DIT, 7>(p); }); break;
            case 8: run_grid(n_cols, tiles, nthr, [&] { ntt_strided_swap_kernel<DIT, 8>(p); }); break;
            case 9: run_grid(n_cols, tiles, nthr, [&] { ntt_strided_swap_kernel<DIT, 9>(p); }); break;
            default: run_grid(n_cols, tiles, nthr, [&] { ntt_strided_swap_kernel<DIT, 10>(p); }); break;
        }
        return;
    }
    if (swap_kernels && p.log_d >= 4 && p.r >= 1 && p.r <= 6 && p.log_d >= 10 - p.r && (!DIT || p.r >= 4)) {      // one wave per tile, no LDS
        const unsigned blocks = (unsigned)(((n >> 10) + 3) / 4);
        if constexpr (DIT) {
            switch (p.r) {
                case 4: run_grid(n_cols, blocks, 256, [&] { ntt_strided_reg_kernel<true, 4>(p); }); break;
                case 5: run_grid(n_cols, blocks, 256, [&] { ntt_strided_reg_kernel<true, 5>(p); }); break;
                default: run_grid(n_cols, blocks, 256, [&] { ntt_strided_reg_kernel<true, 6>(p); }); break;
            }
        } else {
            switch (p.r) {
                case 1: run_grid(n_cols, blocks, 256, [&] { ntt_strided_reg_kernel<false, 1>(p); }); break;
                case 2: run_grid(n_cols, blocks, 256, [&] { ntt_strided_reg_kernel<false, 2>(p); }); break;
                case 3: run_grid(n_cols, blocks, 256, [&] { ntt_strided_reg_kernel<false, 3>(p); }); break;
                case 4: run_grid(n_cols, blocks, 256, [&] { ntt_strided_reg_kernel<false, 4>(p); }); break;
                case 5: run_grid(n_cols, blocks, 256, [&] { ntt_strided_reg_kernel<false, 5>(p); }); break;
                default: run_grid(n_cols, blocks, 256, [&] { ntt_strided_reg_kernel<false, 6>(p); }); break;
            }
        }
        return;
    }
    if (swap_kernels && p.log_d == 0 && (DIT ? p.r - p.log_rep == 10 && p.first_stage == p.log_rep : p.r == 10)) {
        if (!DIT) { run_grid(n_cols, (unsigned)(((n >> 10) + 3) / 4), 256, [&] { ntt_contig_wave_kernel_dif(p); }); return; }
        const unsigned blocks = (unsigned)(((n >> p.r) + 3) / 4);
        if (p.log_rep == 1) run_grid(n_cols, blocks, 256, [&] { ntt_contig_wave_kernel_dit<2>(p, p.in_scale2); });
        else run_grid(n_cols, blocks, 256, [&] { ntt_contig_wave_kernel_dit<1>(p, nullptr); });
        return;
    }
    // the tile kernel as launch_pass sets it up (kTileElemBits 13, kThreadsShift 3)
    if (p.log_d == 0) p.log_t = 0;
    else { int lt = 13 - p.r; if (lt < 0) lt = 0; if (lt > p.log_d) lt = p.log_d; p.log_t = lt; }
    const size_t elems = (size_t)1 << (p.r + p.log_t);
    unsigned nthr = (unsigned)(elems >> 3);
    if (nthr < 64) nthr = 64;
    if (nthr > 1024) nthr = 1024;
    size_t tiles = n / elems;
    if (tiles == 0) tiles = 1;
    run_grid(n_cols, (unsigned)tiles, nthr, [&] { ntt_pass_kernel<DIT>(p); });
}
static void values_to_coeffs(const u64 *src, u64 *dst, unsigned n_cols, int log_n, const std::vector<Plan> &plan, const u64 *tw, bool swap_kernels) {
    const size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < plan.size(); ++i) {
        NttPass p = {};
        p.src = i == 0 ? src : dst; p.src_stride = n; p.dst = dst; p.dst_stride = n;
        p.tw = tw; p.log_tw = log_n; p.log_n = log_n; p.log_d = plan[i].log_d; p.r = plan[i].r;
        if (i + 1 == plan.size()) { p.last_pass = 1; p.apply_out_const = 1; p.out_const = gl_canon(gl_inv((u64)1 << log_n)); }
        launch<false>(p, n_cols, swap_kernels);
    }
}
static void coeffs_to_values(const u64 *src, u64 *dst, unsigned n_cols, int log_n, int rate, const std::vector<Plan> &plan, const u64 *tw,
                             const u64 *in_scale, const u64 *in_scale2, bool swap_kernels) {
    const int L = log_n + rate;
    for (size_t k = 0; k < plan.size(); ++k) {
        const size_t i = plan.size() - 1 - k;
        NttPass p = {};
        p.dst = dst; p.dst_stride = (size_t)1 << L; p.tw = tw; p.log_tw = L; p.log_n = L; p.log_d = plan[i].log_d; p.r = plan[i].r;
        if (k == 0) { p.src = src; p.src_stride = (size_t)1 << log_n; p.in_scale = in_scale; p.in_scale2 = in_scale2; p.log_rep = rate; p.first_stage = rate; }
        else { p.src = dst; p.src_stride = (size_t)1 << L; }
        p.last_pass = k + 1 == plan.size();
        launch<true>(p, n_cols, swap_kernels);
    }
}

static u64 rnd_state = 0x243F6A8885A308D3ULL;
static u64 rnd() { rnd_state ^= rnd_state << 13; rnd_state ^= rnd_state >> 7; rnd_state ^= rnd_state << 17; return rnd_state; }

static bool check(int log_n, unsigned n_cols, std::vector<Plan> old_dif, std::vector<Plan> new_dif, std::vector<Plan> old_dit, std::vector<Plan> new_dit) {
    const int rate = 1, L = log_n + rate;
    const size_t n = (size_t)1 << log_n, N = (size_t)1 << L;
    const std::vector<u64> tw_dif = levels(log_n, true), tw_dit = levels(L, false);
    std::vector<u64> coset(n), coset2(n);
    run_grid_seq(n, [&] { coset_table_kernel(coset.data(), log_n, gl_canon(GL_GENERATOR), 1); });
    run_grid_seq(n, [&] { wave_coset2_table_kernel(coset2.data(), coset.data(), log_n, gl_root_of_unity(11)); });
    std::vector<u64> vals(n * n_cols);
    for (auto &x : vals) x = rnd();
    std::vector<u64> c_old(n * n_cols), c_new(n * n_cols), v_old(N * n_cols), v_new(N * n_cols);
    values_to_coeffs(vals.data(), c_old.data(), n_cols, log_n, old_dif, tw_dif.data(), false);
    values_to_coeffs(vals.data(), c_new.data(), n_cols, log_n, new_dif, tw_dif.data(), true);
    bool ok = c_old == c_new;
    if (!ok) for (size_t i = 0; i < c_old.size(); ++i) if (c_old[i] != c_new[i]) { printf("coefficients differ at %zu: %llx != %llx\n", i, (unsigned long long)c_old[i], (unsigned long long)c_new[i]); break; }
    coeffs_to_values(c_old.data(), v_old.data(), n_cols, log_n, rate, old_dit, tw_dit.data(), coset.data(), nullptr, false);
    coeffs_to_values(c_old.data(), v_new.data(), n_cols, log_n, rate, new_dit, tw_dit.data(), coset.data(), coset2.data(), true);
    if (v_old != v_new) {
        ok = false;
        for (size_t i = 0; i < v_old.size(); ++i) if (v_old[i] != v_new[i]) { printf("extension values differ at %zu: %llx != %llx\n", i, (unsigned long long)v_old[i], (unsigned long long)v_new[i]); break; }
    }
    // and it IS the transform: coefficient j (stored at bitrev(j)) and a few values f(g w^x) against direct evaluation, column 0
    auto coeff = [&](size_t j) { return c_old[bitrev32((u32)j, log_n)]; };
    for (int t = 0; t < 3 && ok; ++t) {
        const size_t x = rnd() % n;                              // f(w_n^x) = vals[x]
        const u64 pt = gl_pow(gl_root_of_unity(log_n), x);
        u64 acc = 0;
        for (size_t j = n; j-- > 0;) acc = gl_canon(gl_add_ref(gl_canon(gl_mul_ref(acc, pt)), coeff(j)));
        ok = ok && acc == gl_canon(vals[x]);
        const size_t y = rnd() % N;                              // f(g w_2n^y) = v[y]
        const u64 pt2 = gl_canon(gl_mul_ref(gl_canon(GL_GENERATOR), gl_pow(gl_root_of_unity(L), y)));
        acc = 0;
        for (size_t j = n; j-- > 0;) acc = gl_canon(gl_add_ref(gl_canon(gl_mul_ref(acc, pt2)), coeff(j)));
        ok = ok && acc == v_old[y];
    }
    printf("%s 2^%d rows x %u columns: lane-swap plan == tile plan (coefficients, extension), == the polynomial\n", ok ? "ok  " : "FAIL", log_n, n_cols);
    return ok;
}

int main(int argc, char **argv) {
    const bool quick = argc > 1 && argv[1][0] == 'q';
    bool ok = true;
    // plans as ntt_host.inc makes them (tests/test_ntt_plan_cpu.py pins those): {log_d, r}, largest distance first
    // small tables: the wave kernel + the register-only strided pass (values -> coefficients), the tile plan's own strided pass through
    // the register-only kernel where the extension has fewer than four row bits left
    ok &= check(13, 3, {{7, 6}, {0, 7}}, {{10, 3}, {0, 10}}, {{8, 6}, {0, 8}}, {{8, 6}, {0, 8}});
    ok &= check(14, 2, {{8, 6}, {0, 8}}, {{10, 4}, {0, 10}}, {{9, 6}, {0, 9}}, {{11, 4}, {0, 11}});
    ok &= check(16, 1, {{8, 8}, {0, 8}}, {{10, 6}, {0, 10}}, {{9, 8}, {0, 9}}, {{11, 6}, {0, 11}});
    ok &= check(17, quick ? 1 : 2, {{9, 8}, {0, 9}}, {{10, 7}, {0, 10}}, {{9, 9}, {0, 9}}, {{11, 7}, {0, 11}});
    if (!quick) ok &= check(18, 1, {{9, 9}, {0, 9}}, {{10, 8}, {0, 10}}, {{10, 9}, {0, 10}}, {{11, 8}, {0, 11}});
    if (argc > 1 && argv[1][0] == 'b') {          // one-off (minutes of thread rendezvous): the 2^19 and 2^20 plans, R = 9 and 10
        ok &= check(19, 1, {{10, 9}, {0, 10}}, {{10, 9}, {0, 10}}, {{11, 9}, {0, 11}}, {{11, 9}, {0, 11}});
        ok &= check(20, 1, {{11, 9}, {0, 11}}, {{10, 10}, {0, 10}}, {{12, 9}, {0, 12}}, {{11, 10}, {0, 11}});
    }
    printf(ok ? "ALL OK\n" : "FAILED\n");
    return ok ? 0 : 1;
}
