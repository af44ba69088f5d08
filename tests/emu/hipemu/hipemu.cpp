// TEST INFRASTRUCTURE -- hipemu: the CPU implementation behind tests/emu/hipemu/hip/hip_runtime.h (read that header first).
//
// Execution of a kernel:  hipemu::launch() wraps the grid in an Op.  Running the Op hands the blocks of the grid to a pool of OS
// threads (HIPEMU_THREADS, default: the machine's cores); a worker runs ONE block at a time, its threads as fibers on the worker's
// own stacks, round-robin, each until it finishes or blocks in __syncthreads() / a wave operation.  A barrier opens when every thread
// of the block (wave) that has not yet returned has arrived -- which is what the hardware does at wave granularity.  A round in which
// nothing can run and no barrier opens is a deadlock (divergent barriers): reported and aborted, never a hang.
//
// Streams (HIPEMU_ASYNC=1): an Op is appended to its stream's queue and runs when the host WAITS for something that depends on it.
// Dependencies are exactly HIP's: order within a stream, hipStreamWaitEvent, the legacy rule that nothing is ordered against a
// hipStreamNonBlocking stream except through events.  Copies from PAGEABLE host memory snapshot their source at enqueue time (the
// real runtime stages them before returning); copies from pinned memory and every copy TO host memory happen when the Op runs.
#include "hip/hip_runtime.h"
#undef threadIdx
#undef blockIdx
#undef blockDim
#undef gridDim

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPEMU_ASAN 1
#endif
#if __has_feature(thread_sanitizer)
#define HIPEMU_TSAN 1
#endif
#endif
#if defined(__SANITIZE_ADDRESS__) && !defined(HIPEMU_ASAN)
#define HIPEMU_ASAN 1
#endif
#if defined(__SANITIZE_THREAD__) && !defined(HIPEMU_TSAN)
#define HIPEMU_TSAN 1
#endif
// (tests/emu/build_emu.py --san tsan compiles THIS file without -fsanitize=thread and passes -DHIPEMU_TSAN=1)
#ifdef HIPEMU_ASAN
extern "C" void __sanitizer_start_switch_fiber(void **fake_stack_save, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fake_stack_save, const void **bottom_old, size_t *size_old);
#endif
#ifdef HIPEMU_TSAN
extern "C" void *__tsan_get_current_fiber(void);
extern "C" void *__tsan_create_fiber(unsigned flags);
extern "C" void __tsan_destroy_fiber(void *fiber);
extern "C" void __tsan_switch_to_fiber(void *fiber, unsigned flags);
extern "C" void __tsan_acquire(void *addr);
extern "C" void __tsan_release(void *addr);
// Under ThreadSanitizer every fiber is a thread of its own and a switch is NOT a synchronisation (flag 1 = no_sync): what orders the
// threads of a block is what orders them on the hardware -- __syncthreads(), the wave operations, the start and the end of the block --
// each annotated as release + acquire on an object of the worker.  A kernel that reads LDS (or global memory) another thread of its
// block wrote, with no barrier in between, is then reported as a data race: a missing __syncthreads() shows up without a GPU.
// (Lanes of ONE wave that rely on lock-step execution without ntt_wave_sync would be reported too; the library has no such place.)
#define HIPEMU_NO_TSAN __attribute__((no_sanitize("thread")))
#define HIPEMU_TSAN_RELEASE(p) __tsan_release((void *)(p))
#define HIPEMU_TSAN_ACQUIRE(p) __tsan_acquire((void *)(p))
static constexpr unsigned kTsanSwitchFlags = 1;
#else
#define HIPEMU_NO_TSAN
#define HIPEMU_TSAN_RELEASE(p) ((void)0)
#define HIPEMU_TSAN_ACQUIRE(p) ((void)0)
#endif

namespace hipemu {

static long env_long(const char *name, long dflt) {
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    char *end = nullptr;
    const long x = strtol(v, &end, 10);
    return end == v ? dflt : x;
}
static const bool kAsync = env_long("HIPEMU_ASYNC", 0) != 0;
static const long kThreads = env_long("HIPEMU_THREADS", 0);
static const long kVerbose = env_long("HIPEMU_VERBOSE", 0);
static std::atomic<uint64_t> g_counters[8];          // launches, blocks, fibers, switches, copies, mallocs, late ops, streams
static std::atomic<long> g_fail_launch_at{0}, g_launch_seq{0};
static std::atomic<long> g_fail_malloc_at{env_long("HIPEMU_FAIL_MALLOC_AT", 0)}, g_fail_malloc_from{env_long("HIPEMU_FAIL_MALLOC_FROM", 0)}, g_malloc_seq{0};

[[noreturn]] static void die(const char *fmt, const char *a = "", const char *b = "") {
    fprintf(stderr, "hipemu: ");
    fprintf(stderr, fmt, a, b);
    fprintf(stderr, "\n");
    fflush(stderr);
    abort();
}

// ---- fibers ---------------------------------------------------------------------------------------------------------------------
extern "C" void hipemu_ctx_switch(void **save_sp, void *to_sp);
#if defined(__x86_64__)
asm(R"(
    .text
    .globl hipemu_ctx_switch
    .type hipemu_ctx_switch, @function
hipemu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_ctx_switch, .-hipemu_ctx_switch
)");
#else
#error "hipemu's fiber switch is written for x86-64"
#endif

static constexpr size_t kStackBytes = 256 * 1024;     // (a kernel thread's frame: registers arrays, a Poseidon state, inlined AIRs)
enum FiberState : uint8_t { F_RUNNABLE, F_WAIT_BLOCK, F_WAIT_WAVE, F_DONE };

struct Wave {
    char sync = 0;
    uint64_t xchg[2][64];
    uint32_t stamp[2][64];
    unsigned live = 0, waiting = 0;
};
struct Fiber {
    ThreadState ts;
    void *sp = nullptr;
    char *stack = nullptr;
    FiberState state = F_DONE;
    uint32_t wave_ops = 0;
#ifdef HIPEMU_ASAN
    void *asan_fake = nullptr;
#endif
#ifdef HIPEMU_TSAN
    void *tsan = nullptr;
#endif
};
struct Worker {                                        // one per OS thread that runs blocks
    std::vector<Fiber> fibers;                         // stacks are kept between blocks
    std::vector<Wave> waves;
    std::vector<char> dyn;
    void *sched_sp = nullptr;                          // the worker's own context (run_block), resumed when the block is over
    const std::function<void()> *body = nullptr;
    Fiber *cur = nullptr;
    unsigned n = 0, n_waves = 0, live = 0, block_waiting = 0;
    bool deadlock = false, from_main = false;
    uint64_t switches = 0;
    char sync_start = 0, sync_end = 0, sync_block = 0;      // (TSan: addresses of the block's synchronisation objects)
    const char *kernel = "";
#ifdef HIPEMU_ASAN
    const void *sched_bottom = nullptr;
    size_t sched_size = 0;
    void *sched_fake = nullptr;
#endif
#ifdef HIPEMU_TSAN
    void *sched_tsan = nullptr;
#endif
    ~Worker() {
        for (auto &f : fibers) {
            if (f.stack) munmap(f.stack, kStackBytes);
#ifdef HIPEMU_TSAN
            if (f.tsan) __tsan_destroy_fiber(f.tsan);
#endif
        }
    }
};
thread_local ThreadState *t_cur = nullptr;
static thread_local Worker *t_worker = nullptr;

// Fiber -> fiber, directly (the worker's own context only starts a block and takes over again when it is finished or stuck).
HIPEMU_NO_TSAN static inline void switch_fiber(Worker *w, Fiber *from, Fiber *to) {
    ++w->switches;
    w->cur = to;
    t_cur = &to->ts;
#ifdef HIPEMU_ASAN
    __sanitizer_start_switch_fiber(from->state == F_DONE ? nullptr : &from->asan_fake, to->stack, kStackBytes);
#endif
#ifdef HIPEMU_TSAN
    __tsan_switch_to_fiber(to->tsan, kTsanSwitchFlags);
#endif
    hipemu_ctx_switch(&from->sp, to->sp);
#ifdef HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(from->asan_fake, nullptr, nullptr);
#endif
}
HIPEMU_NO_TSAN static void switch_to_main(Worker *w, Fiber *from) {
    t_cur = nullptr;
#ifdef HIPEMU_ASAN
    __sanitizer_start_switch_fiber(from->state == F_DONE ? nullptr : &from->asan_fake, w->sched_bottom, w->sched_size);
#endif
#ifdef HIPEMU_TSAN
    __tsan_switch_to_fiber(w->sched_tsan, kTsanSwitchFlags);
#endif
    hipemu_ctx_switch(&from->sp, w->sched_sp);
#ifdef HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(from->asan_fake, nullptr, nullptr);
#endif
}
HIPEMU_NO_TSAN static bool open_barriers(Worker *w) {
    bool opened = false;
    if (w->block_waiting && w->block_waiting == w->live) {
        for (unsigned t = 0; t < w->n; ++t) if (w->fibers[t].state == F_WAIT_BLOCK) w->fibers[t].state = F_RUNNABLE;
        w->block_waiting = 0;
        opened = true;
    }
    for (unsigned v = 0; v < w->n_waves; ++v) {
        Wave &wv = w->waves[v];
        if (wv.waiting && wv.waiting == wv.live) {
            for (unsigned t = v * 64; t < w->n && t < (v + 1) * 64; ++t) if (w->fibers[t].state == F_WAIT_WAVE) w->fibers[t].state = F_RUNNABLE;
            wv.waiting = 0;
            opened = true;
        }
    }
    return opened;
}
// The running fiber has just blocked (or finished): run the next one that can run; returns when this fiber is runnable again.
HIPEMU_NO_TSAN static void yield_blocked(Worker *w) {
    Fiber *f = w->cur;
    const unsigned me = (unsigned)(f - w->fibers.data()), n = w->n;
    for (;;) {
        for (unsigned k = 1; k <= n; ++k) {
            unsigned t = me + k;
            if (t >= n) t -= n;
            if (w->fibers[t].state != F_RUNNABLE) continue;
            if (t == me) return;                       // a barrier this fiber was the last to reach
            switch_fiber(w, f, &w->fibers[t]);
            return;                                    // (resumed: somebody made this fiber runnable and switched here)
        }
        if (open_barriers(w)) continue;
        w->deadlock = w->live > 0;                     // nothing can run: the block is finished, or stuck
        switch_to_main(w, f);
        return;
    }
}
HIPEMU_NO_TSAN static void fiber_entry() {
    Worker *w = t_worker;
    HIPEMU_TSAN_ACQUIRE(&w->sync_start);
#ifdef HIPEMU_ASAN
    {
        const void *b = nullptr; size_t sz = 0;
        __sanitizer_finish_switch_fiber(nullptr, &b, &sz);
        if (w->from_main) { w->sched_bottom = b; w->sched_size = sz; }
    }
#endif
    w->from_main = false;
    (*w->body)();
    HIPEMU_TSAN_RELEASE(&w->sync_end);
    Fiber *f = w->cur;
    f->state = F_DONE;
    --w->live;
    --w->waves[f->ts.wave].live;
    yield_blocked(w);
    die("a finished fiber was resumed");
}

HIPEMU_NO_TSAN static void run_block(Worker *w, dim3 grid, dim3 block, dim3 bidx, size_t lds, const std::function<void()> &body, const char *name) {
    const unsigned n = block.x * block.y * block.z;
    if (n == 0 || n > 1024) die("kernel %s launched with a bad block size", name);
    w->n = n; w->live = n; w->block_waiting = 0; w->body = &body; w->kernel = name; w->deadlock = false;
    if (w->fibers.size() < n) w->fibers.resize(n);     // (never while a fiber of this worker is alive)
    const unsigned n_waves = (n + 63) / 64;
    w->n_waves = n_waves;
    w->waves.assign(n_waves, Wave());
    if (w->dyn.size() < lds + 64) w->dyn.resize(lds + 64);
#ifdef HIPEMU_TSAN
    w->sched_tsan = __tsan_get_current_fiber();
#endif
    for (unsigned t = 0; t < n; ++t) {
        Fiber &f = w->fibers[t];
        if (!f.stack) {
            void *m = mmap(nullptr, kStackBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
            if (m == MAP_FAILED) die("no memory for a fiber stack");
            f.stack = (char *)m;
            mprotect(f.stack, 4096, PROT_NONE);        // guard page: a kernel thread that overruns its stack faults instead of corrupting a neighbour
#ifdef HIPEMU_TSAN
            f.tsan = __tsan_create_fiber(0);
#endif
        }
        f.ts.threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        f.ts.blockIdx = bidx; f.ts.blockDim = block; f.ts.gridDim = grid;
        f.ts.lane = t & 63; f.ts.wave = t >> 6;
        f.state = F_RUNNABLE;
        f.wave_ops = 0;
        ++w->waves[t >> 6].live;
        uintptr_t top = ((uintptr_t)f.stack + kStackBytes) & ~(uintptr_t)15;
        void **sp = (void **)top;
        *--sp = nullptr;                               // fiber_entry's (never used) return address
        *--sp = (void *)&fiber_entry;                  // popped by hipemu_ctx_switch's `ret`
        for (int i = 0; i < 6; ++i) *--sp = nullptr;   // rbp rbx r12 r13 r14 r15
        f.sp = sp;
    }
    // into fiber 0; control comes back here when no fiber can run any more
    Fiber *first = &w->fibers[0];
    w->cur = first;
    t_cur = &first->ts;
    w->from_main = true;
    HIPEMU_TSAN_RELEASE(&w->sync_start);
#ifdef HIPEMU_ASAN
    __sanitizer_start_switch_fiber(&w->sched_fake, first->stack, kStackBytes);
#endif
#ifdef HIPEMU_TSAN
    __tsan_switch_to_fiber(first->tsan, kTsanSwitchFlags);
#endif
    hipemu_ctx_switch(&w->sched_sp, first->sp);
#ifdef HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(w->sched_fake, nullptr, nullptr);
#endif
    w->cur = nullptr;
    t_cur = nullptr;
    HIPEMU_TSAN_ACQUIRE(&w->sync_end);
    if (w->deadlock) {
        fprintf(stderr, "hipemu: DEADLOCK in kernel %s, block (%u,%u,%u): %u threads alive, %u at __syncthreads", name, bidx.x, bidx.y, bidx.z, w->live, w->block_waiting);
        for (unsigned v = 0; v < n_waves; ++v) if (w->waves[v].waiting) fprintf(stderr, ", wave %u: %u of %u at a wave operation", v, w->waves[v].waiting, w->waves[v].live);
        fprintf(stderr, " (a barrier some threads never reach)\n");
        abort();
    }
    g_counters[1].fetch_add(1, std::memory_order_relaxed);
    g_counters[2].fetch_add(n, std::memory_order_relaxed);
    g_counters[3].fetch_add(w->switches, std::memory_order_relaxed);
    w->switches = 0;
}

void *dyn_lds() {
    Worker *w = t_worker;
    return (void *)(((uintptr_t)w->dyn.data() + 63) & ~(uintptr_t)63);
}
HIPEMU_NO_TSAN void sync_threads() {
    Worker *w = t_worker;
    if (!w || !w->cur) die("__syncthreads outside a kernel");
    HIPEMU_TSAN_RELEASE(&w->sync_block);
    w->cur->state = F_WAIT_BLOCK;
    ++w->block_waiting;
    yield_blocked(w);
    HIPEMU_TSAN_ACQUIRE(&w->sync_block);
}
HIPEMU_NO_TSAN void wave_sync() {
    Worker *w = t_worker;
    if (!w || !w->cur) die("a wave operation outside a kernel");
    Fiber *f = w->cur;
    HIPEMU_TSAN_RELEASE(&w->waves[f->ts.wave].sync);
    f->state = F_WAIT_WAVE;
    ++w->waves[f->ts.wave].waiting;
    yield_blocked(w);
    HIPEMU_TSAN_ACQUIRE(&w->waves[f->ts.wave].sync);
}
HIPEMU_NO_TSAN uint64_t wave_exchange(uint64_t mine, unsigned src) {
    Worker *w = t_worker;
    if (!w || !w->cur) die("a wave operation outside a kernel");
    Fiber *f = w->cur;
    Wave &wv = w->waves[f->ts.wave];
    const uint32_t k = ++f->wave_ops;                  // this lane's k-th wave operation: buffers alternate, one barrier per operation
    wv.xchg[k & 1][f->ts.lane] = mine;
    wv.stamp[k & 1][f->ts.lane] = k;
    wave_sync();
    if (src >= 64 || wv.stamp[k & 1][src] != k) return mine;      // that lane has returned, or does not exist: the hardware leaves the destination alone
    return wv.xchg[k & 1][src];
}

// ---- the pool that runs grids -------------------------------------------------------------------------------------------------------
struct GridJob {
    dim3 grid, block;
    size_t lds;
    const std::function<void()> *body;
    const char *name;
    std::atomic<uint64_t> next{0};
    uint64_t total = 0;
    std::atomic<unsigned> active{0};
};
struct Pool {
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    GridJob *job = nullptr;
    uint64_t job_seq = 0;
    bool stop = false;
    unsigned n_threads = 1;
    Pool() {
        long n = kThreads > 0 ? kThreads : (long)std::thread::hardware_concurrency();
        if (n < 1) n = 1;
        if (n > 64) n = 64;
        n_threads = (unsigned)n;
        for (unsigned i = 1; i < n_threads; ++i) threads.emplace_back([this] { loop(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(m); stop = true; }
        cv_work.notify_all();
        for (auto &t : threads) t.join();
    }
    static void work(GridJob *j) {
        static thread_local Worker worker;
        t_worker = &worker;
        for (;;) {
            const uint64_t b = j->next.fetch_add(1, std::memory_order_relaxed);
            if (b >= j->total) break;
            const dim3 bidx((unsigned)(b % j->grid.x), (unsigned)((b / j->grid.x) % j->grid.y), (unsigned)(b / ((uint64_t)j->grid.x * j->grid.y)));
            run_block(&worker, j->grid, j->block, bidx, j->lds, *j->body, j->name);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            GridJob *j = nullptr;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return stop || (job && job_seq != seen); });
                if (stop) return;
                j = job; seen = job_seq;
                j->active.fetch_add(1);
            }
            work(j);
            {
                std::lock_guard<std::mutex> g(m);
                j->active.fetch_sub(1);
            }
            cv_done.notify_all();
        }
    }
    std::mutex run_m;                                  // one grid at a time (host threads of two contexts take turns)
    void run(GridJob *j) {
        std::lock_guard<std::mutex> one(run_m);
        const bool fan_out = j->total > 1 && n_threads > 1;
        if (fan_out) {
            { std::lock_guard<std::mutex> g(m); job = j; ++job_seq; }
            cv_work.notify_all();
        }
        work(j);
        if (fan_out) {
            std::unique_lock<std::mutex> lk(m);
            job = nullptr;                              // late wakers find nothing; those inside are counted
            cv_done.wait(lk, [&] { return j->active.load() == 0; });
        }
    }
};
static Pool &pool() { static Pool *p = new Pool(); return *p; }     // (leaked on purpose: worker threads may outlive static destruction order)

static void run_grid(dim3 grid, dim3 block, size_t lds, const std::function<void()> &body, const char *name) {
    GridJob j;
    j.grid = grid; j.block = block; j.lds = lds; j.body = &body; j.name = name;
    j.total = (uint64_t)grid.x * grid.y * grid.z;
    if (j.total == 0) return;
    if (kVerbose > 1) fprintf(stderr, "hipemu: %s <<<(%u,%u,%u),(%u,%u,%u),%zu>>>\n", name, grid.x, grid.y, grid.z, block.x, block.y, block.z, lds);
    if (t_worker && t_worker->cur) die("kernel %s launched from inside a kernel", name);
    pool().run(&j);
    t_worker = nullptr;
}

// ---- streams, events, operations ------------------------------------------------------------------------------------------------------
struct Stream;
struct Op {
    std::function<void()> fn;                          // what to do (empty for markers)
    hipemu_event *wait = nullptr;                      // != null: a hipStreamWaitEvent marker
    uint64_t wait_seq = 0;                             // ... for this recording of the event
    hipemu_stream *wait_stream = nullptr;              // ... which went to this stream (the event may be recorded elsewhere again before the wait runs)
    hipemu_event *record = nullptr;                    // != null: a hipEventRecord marker
    uint64_t record_seq = 0;
};
}  // namespace hipemu
struct hipemu_stream {
    std::deque<hipemu::Op> q;
    bool non_blocking = false;
    bool draining = false;
    uint64_t id = 0;
};
struct hipemu_event {
    hipemu_stream *stream = nullptr;                   // where the latest record went
    uint64_t recorded = 0, completed = 0;              // sequence numbers of hipEventRecord calls / of those whose marker has run
    std::chrono::steady_clock::time_point when;
};
namespace hipemu {

static std::recursive_mutex g_rt;                      // the runtime's own state (streams, events, registries)
static hipemu_stream g_null_stream;
static std::set<hipemu_stream *> g_streams;
static std::set<hipemu_event *> g_events;
static std::map<uintptr_t, size_t> g_pinned, g_device;
static thread_local hipError_t t_last_error = hipSuccess;

static hipemu_stream *S(hipStream_t s) { return s ? s : &g_null_stream; }
static bool is_pinned(const void *p) {
    auto it = g_pinned.upper_bound((uintptr_t)p);
    if (it == g_pinned.begin()) return false;
    --it;
    return (uintptr_t)p < it->first + it->second;
}

static void drain(hipemu_stream *s, size_t upto);      // run the first `upto` queued operations of s (and what they wait for)
static void drain_all(hipemu_stream *s) { drain(s, s->q.size()); }
static void wait_event_record(hipemu_event *e, uint64_t seq, hipemu_stream *t = nullptr) {
    if (e->completed >= seq) return;
    if (!t) t = e->stream;
    if (!t) return;
    size_t upto = 0;
    for (size_t i = 0; i < t->q.size(); ++i)
        if (t->q[i].record == e && t->q[i].record_seq <= seq) upto = i + 1;
    if (upto) drain(t, upto);
}
static void drain(hipemu_stream *s, size_t upto) {
    if (s->draining) return;                           // (a cycle of waits cannot be built with HIP's API; re-entry is a no-op)
    s->draining = true;
    while (upto > 0 && !s->q.empty()) {
        Op op = std::move(s->q.front());
        s->q.pop_front();
        --upto;
        if (op.wait) wait_event_record(op.wait, op.wait_seq, op.wait_stream);
        if (op.fn) op.fn();
        if (op.record) { op.record->completed = std::max(op.record->completed, op.record_seq); op.record->when = std::chrono::steady_clock::now(); }
        g_counters[6].fetch_add(1, std::memory_order_relaxed);
    }
    s->draining = false;
}
static void device_sync() {
    std::vector<hipemu_stream *> all(g_streams.begin(), g_streams.end());
    std::reverse(all.begin(), all.end());              // (the youngest streams first: the side lanes before the main one)
    for (auto *s : all) drain_all(s);
    drain_all(&g_null_stream);
}
// the legacy default stream is ordered against every BLOCKING stream
static void null_stream_barrier(hipemu_stream *s) {
    if (s != &g_null_stream) { if (!s->non_blocking) drain_all(&g_null_stream); return; }
    for (auto *t : g_streams) if (!t->non_blocking) drain_all(t);
}
static void enqueue(hipStream_t st, Op op) {
    hipemu_stream *s = S(st);
    if (!kAsync) {
        if (op.fn) op.fn();
        if (op.record) { op.record->completed = op.record_seq; op.record->when = std::chrono::steady_clock::now(); }
        return;
    }
    std::lock_guard<std::recursive_mutex> g(g_rt);
    null_stream_barrier(s);
    s->q.push_back(std::move(op));
}

void enqueue_host(hipStream_t st, std::function<void()> fn) {      // a stream-ordered host-side stand-in for a device library call (rocprim/)
    Op op;
    op.fn = std::move(fn);
    enqueue(st, std::move(op));
}

void launch(dim3 grid, dim3 block, size_t lds, hipStream_t st, std::function<void()> body, const char *name) {
    g_counters[0].fetch_add(1, std::memory_order_relaxed);
    const uint64_t total = (uint64_t)grid.x * grid.y * grid.z;
    const unsigned n = block.x * block.y * block.z;
    if (total == 0 || n == 0 || n > 1024 || grid.y > 65535 || grid.z > 65535 || lds > 160 * 1024) { t_last_error = hipErrorInvalidValue; return; }   // what the real launch refuses
    {
        const long at = g_fail_launch_at.load();
        if (at > 0 && g_launch_seq.fetch_add(1) + 1 == at) { t_last_error = hipErrorLaunchFailure; return; }      // (test control: this launch is refused)
    }
    Op op;
    auto shared = std::make_shared<std::function<void()>>(std::move(body));
    op.fn = [=]() { run_grid(grid, block, lds, *shared, name); };
    enqueue(st, std::move(op));
}

}  // namespace hipemu

using namespace hipemu;

uint32_t hipemu_readfirstlane_checked(uint32_t v) {
    // The library uses readfirstlane only as a "this is wave-uniform" hint.  Checking it costs a wave rendezvous per call and would
    // change which lanes must be converged; HIPEMU_CHECK_UNIFORM=1 turns the check on for the runs that want it.
    static const bool check = env_long("HIPEMU_CHECK_UNIFORM", 0) != 0;
    if (!check) return v;
    const uint64_t first = wave_exchange(v, t_cur->lane & ~63u);      // lane 0 of the wave (if it is alive)
    if ((uint32_t)first != v) die("readfirstlane of a value that is NOT wave-uniform in kernel %s", t_worker ? t_worker->kernel : "?");
    return v;
}
hipemu_u32x2 hipemu_permlane_swap(unsigned lanebit, uint32_t a, uint32_t b) {
    const unsigned lane = t_cur->lane;
    const uint64_t got = wave_exchange(((uint64_t)b << 32) | a, lane ^ (1u << lanebit));
    if ((lane >> lanebit) & 1) a = (uint32_t)(got >> 32);      // odd rows of vdst take the even rows' vsrc
    else b = (uint32_t)got;
    return hipemu_u32x2{a, b};
}
// csrc/ntt_swap.cuh, ZK_NTT_EMULATE branch
void zk_emu_lane_swap(int lanebit, unsigned long long &a, unsigned long long &b) {
    const unsigned lane = t_cur->lane, partner = lane ^ (1u << lanebit);
    const bool upper = (lane >> lanebit) & 1;
    const uint64_t got = wave_exchange(upper ? a : b, partner);      // (register a, lane bit = 1) <-> (register b, lane bit = 0)
    if (upper) a = got; else b = got;
}
void zk_emu_wave_sync() { wave_sync(); }

extern "C" {

void hipemu_fail_malloc_at(long nth) { g_fail_malloc_at = nth; g_fail_malloc_from = 0; g_malloc_seq = 0; }
void hipemu_fail_launch_at(long nth) { g_fail_launch_at = nth; g_launch_seq = 0; }
void hipemu_fail_malloc_from(long nth) { g_fail_malloc_from = nth; g_fail_malloc_at = 0; g_malloc_seq = 0; }
void hipemu_counters(uint64_t out[8]) { for (int i = 0; i < 8; ++i) out[i] = g_counters[i].load(); }

hipError_t hipGetDeviceCount(int *count) { if (!count) return hipErrorInvalidValue; *count = (int)env_long("HIPEMU_DEVICES", 1); return hipSuccess; }
hipError_t hipSetDevice(int device) { return device >= 0 && device < env_long("HIPEMU_DEVICES", 1) ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDevice(int *device) { if (device) *device = 0; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *prop, int device) {
    if (!prop || device < 0) return hipErrorInvalidValue;
    memset(prop, 0, sizeof *prop);
    snprintf(prop->name, sizeof prop->name, "hipemu (CPU emulation)");
    snprintf(prop->gcnArchName, sizeof prop->gcnArchName, "gfx950-emulated");
    prop->totalGlobalMem = (size_t)16 << 30;
    prop->multiProcessorCount = 256; prop->warpSize = 64; prop->maxThreadsPerBlock = 1024; prop->sharedMemPerBlock = 160 * 1024;
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { std::lock_guard<std::recursive_mutex> g(g_rt); device_sync(); return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { if (least) *least = 1; if (greatest) *greatest = -1; return hipSuccess; }
hipError_t hipMemGetInfo(size_t *f, size_t *t) { if (f) *f = (size_t)8 << 30; if (t) *t = (size_t)16 << 30; return hipSuccess; }

hipError_t hipGetLastError(void) { const hipError_t e = t_last_error; t_last_error = hipSuccess; return e; }
hipError_t hipPeekAtLastError(void) { return t_last_error; }
const char *hipGetErrorString(hipError_t e) {
    switch (e) {
        case hipSuccess: return "no error";
        case hipErrorInvalidValue: return "invalid argument";
        case hipErrorOutOfMemory: return "out of memory";
        case hipErrorInvalidDevice: return "invalid device ordinal";
        case hipErrorInvalidResourceHandle: return "invalid resource handle";
        case hipErrorNotReady: return "device not ready";
        case hipErrorLaunchFailure: return "unspecified launch failure";
        default: return "unknown error";
    }
}
const char *hipGetErrorName(hipError_t e) { return hipGetErrorString(e); }

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags) {
    if (!s) return hipErrorInvalidValue;
    std::lock_guard<std::recursive_mutex> g(g_rt);
    auto *st = new hipemu_stream();
    st->non_blocking = (flags & hipStreamNonBlocking) != 0;
    st->id = g_counters[7].fetch_add(1) + 1;
    g_streams.insert(st);
    *s = st;
    return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t *s) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int) { return hipStreamCreateWithFlags(s, flags); }
hipError_t hipStreamDestroy(hipStream_t s) {
    if (!s) return hipErrorInvalidResourceHandle;
    std::lock_guard<std::recursive_mutex> g(g_rt);
    if (!g_streams.count(s)) return hipErrorInvalidResourceHandle;
    drain_all(s);                                       // (the real call lets queued work finish, then releases the stream)
    for (auto *e : g_events) if (e->stream == s) e->stream = nullptr;
    g_streams.erase(s);
    delete s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
    std::lock_guard<std::recursive_mutex> g(g_rt);
    if (s && !g_streams.count(s)) return hipErrorInvalidResourceHandle;
    null_stream_barrier(S(s));
    drain_all(S(s));
    return hipSuccess;
}
hipError_t hipStreamQuery(hipStream_t s) { std::lock_guard<std::recursive_mutex> g(g_rt); return S(s)->q.empty() ? hipSuccess : hipErrorNotReady; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    if (!e) return hipErrorInvalidResourceHandle;
    std::lock_guard<std::recursive_mutex> g(g_rt);
    if (!g_events.count(e)) return hipErrorInvalidResourceHandle;
    if (e->recorded == 0 || !kAsync) return hipSuccess;     // never recorded: a no-op, as in HIP
    Op op;
    op.wait = e; op.wait_seq = e->recorded; op.wait_stream = e->stream;
    enqueue(s, std::move(op));
    return hipSuccess;
}

hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) {
    if (!e) return hipErrorInvalidValue;
    std::lock_guard<std::recursive_mutex> g(g_rt);
    *e = new hipemu_event();
    g_events.insert(*e);
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) {
    if (!e) return hipErrorInvalidResourceHandle;
    std::lock_guard<std::recursive_mutex> g(g_rt);
    if (!g_events.count(e)) return hipErrorInvalidResourceHandle;
    if (e->completed < e->recorded) wait_event_record(e, e->recorded);      // (queued markers point at it)
    g_events.erase(e);
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    if (!e) return hipErrorInvalidResourceHandle;
    std::lock_guard<std::recursive_mutex> g(g_rt);
    if (!g_events.count(e) || (s && !g_streams.count(s))) return hipErrorInvalidResourceHandle;
    e->stream = S(s);
    Op op;
    op.record = e; op.record_seq = ++e->recorded;
    enqueue(s, std::move(op));
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    if (!e) return hipErrorInvalidResourceHandle;
    std::lock_guard<std::recursive_mutex> g(g_rt);
    if (!g_events.count(e)) return hipErrorInvalidResourceHandle;
    wait_event_record(e, e->recorded);
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e) { std::lock_guard<std::recursive_mutex> g(g_rt); return e && e->completed >= e->recorded ? hipSuccess : hipErrorNotReady; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    if (!ms || !a || !b) return hipErrorInvalidValue;
    std::lock_guard<std::recursive_mutex> g(g_rt);
    if (a->recorded == 0 || b->recorded == 0) return hipErrorInvalidResourceHandle;
    if (a->completed < a->recorded || b->completed < b->recorded) return hipErrorNotReady;
    *ms = std::chrono::duration<float, std::milli>(b->when - a->when).count();
    return hipSuccess;
}

hipError_t hipMalloc(void **p, size_t bytes) {
    if (!p) return hipErrorInvalidValue;
    *p = nullptr;
    g_counters[5].fetch_add(1, std::memory_order_relaxed);
    const long at = g_fail_malloc_at.load(), from = g_fail_malloc_from.load(), seq = g_malloc_seq.fetch_add(1) + 1;
    if ((at > 0 && seq == at) || (from > 0 && seq >= from)) { t_last_error = hipErrorOutOfMemory; return hipErrorOutOfMemory; }
    if (bytes > ((size_t)12 << 30)) { t_last_error = hipErrorOutOfMemory; return hipErrorOutOfMemory; }
    void *m = nullptr;
    if (posix_memalign(&m, 256, bytes ? bytes : 1) != 0) { t_last_error = hipErrorOutOfMemory; return hipErrorOutOfMemory; }
    memset(m, 0xA5, bytes < 4096 ? bytes : 4096);       // device memory is not zeroed: make a read of a fresh block visible
    std::lock_guard<std::recursive_mutex> g(g_rt);
    g_device[(uintptr_t)m] = bytes;
    *p = m;
    return hipSuccess;
}
hipError_t hipFree(void *p) {
    if (!p) return hipSuccess;
    std::lock_guard<std::recursive_mutex> g(g_rt);
    auto it = g_device.find((uintptr_t)p);
    if (it == g_device.end()) { t_last_error = hipErrorInvalidValue; return hipErrorInvalidValue; }
    device_sync();                                      // hipFree waits for the device
    g_device.erase(it);
    free(p);
    return hipSuccess;
}
hipError_t hipMallocAsync(void **p, size_t bytes, hipStream_t) { return hipMalloc(p, bytes); }
hipError_t hipFreeAsync(void *p, hipStream_t s) { Op op; op.fn = [p]() { std::lock_guard<std::recursive_mutex> g(g_rt); g_device.erase((uintptr_t)p); free(p); }; enqueue(s, std::move(op)); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) {
    if (!p) return hipErrorInvalidValue;
    void *m = nullptr;
    if (posix_memalign(&m, 4096, bytes ? bytes : 1) != 0) return hipErrorOutOfMemory;
    memset(m, 0, bytes);
    std::lock_guard<std::recursive_mutex> g(g_rt);
    g_pinned[(uintptr_t)m] = bytes ? bytes : 1;
    *p = m;
    return hipSuccess;
}
hipError_t hipHostFree(void *p) {
    if (!p) return hipSuccess;
    std::lock_guard<std::recursive_mutex> g(g_rt);
    if (!g_pinned.erase((uintptr_t)p)) return hipErrorInvalidValue;
    device_sync();
    free(p);
    return hipSuccess;
}
hipError_t hipHostRegister(void *p, size_t bytes, unsigned) {
    if (!p || !bytes) return hipErrorInvalidValue;
    std::lock_guard<std::recursive_mutex> g(g_rt);
    g_pinned[(uintptr_t)p] = bytes;
    return hipSuccess;
}
hipError_t hipHostUnregister(void *p) {
    std::lock_guard<std::recursive_mutex> g(g_rt);
    device_sync();
    return g_pinned.erase((uintptr_t)p) ? hipSuccess : hipErrorInvalidValue;
}

static void copy2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height) {
    for (size_t r = 0; r < height; ++r) memmove((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
}
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s) {
    if ((!dst || !src) && width && height) return hipErrorInvalidValue;
    if (height > 1 && (dpitch < width || spitch < width)) return hipErrorInvalidValue;
    g_counters[4].fetch_add(1, std::memory_order_relaxed);
    Op op;
    bool pageable_src = false;
    if (kAsync && (kind == hipMemcpyHostToDevice || kind == hipMemcpyHostToHost || kind == hipMemcpyDefault)) {
        std::lock_guard<std::recursive_mutex> g(g_rt);
        pageable_src = !is_pinned(src) && !g_device.count((uintptr_t)src) && kind != hipMemcpyDefault;
    }
    if (pageable_src) {                                 // the real runtime stages pageable memory before it returns
        auto snap = std::make_shared<std::vector<char>>(width * height);
        copy2d(snap->data(), width, src, spitch, width, height);
        op.fn = [=]() { copy2d(dst, dpitch, snap->data(), width, width, height); };
    } else {
        op.fn = [=]() { copy2d(dst, dpitch, src, spitch, width, height); };
    }
    enqueue(s, std::move(op));
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s) { return hipMemcpy2DAsync(dst, bytes, src, bytes, bytes, 1, kind, s); }
hipError_t hipMemcpy2D(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind) {
    hipError_t e = hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) { return hipMemcpy2D(dst, bytes, src, bytes, bytes, 1, kind); }
hipError_t hipMemset2DAsync(void *dst, size_t pitch, int value, size_t width, size_t height, hipStream_t s) {
    if (!dst && width && height) return hipErrorInvalidValue;
    Op op;
    op.fn = [=]() { for (size_t r = 0; r < height; ++r) memset((char *)dst + r * pitch, value, width); };
    enqueue(s, std::move(op));
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t s) { return hipMemset2DAsync(dst, bytes, value, bytes, 1, s); }
hipError_t hipMemset(void *dst, int value, size_t bytes) {
    hipError_t e = hipMemsetAsync(dst, value, bytes, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
}
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }

}  // extern "C"
