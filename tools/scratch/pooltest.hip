#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <cstdint>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void info(const char *tag) { size_t f, t; hipMemGetInfo(&f, &t); printf("%-28s free %.1f GB\n", tag, f / 1e9); }
int main(int argc, char **argv) {
    uint64_t thr = argc > 1 ? strtoull(argv[1], 0, 0) : ~0ULL;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
    printf("set thr rc=%d\n", (int)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
    uint64_t got = 0; hipMemPoolGetAttribute(pool, hipMemPoolAttrReleaseThreshold, &got); printf("thr now %llx\n", (unsigned long long)got);
    info("start");
    for (size_t gb : {4, 40, 40, 20, 40}) {
        void *p = nullptr;
        double t0 = now();
        hipError_t e = hipMallocAsync(&p, gb << 30, st);
        hipStreamSynchronize(st);
        double t1 = now();
        hipMemsetAsync(p, 1, gb << 30, st); hipStreamSynchronize(st);
        double t2 = now();
        printf("alloc %zu GB rc=%d  %.3f s, memset %.3f s\n", gb, (int)e, t1 - t0, t2 - t1);
        info(" after alloc");
        hipFreeAsync(p, st); hipStreamSynchronize(st);
        info(" after free+sync");
        hipDeviceSynchronize();
        info(" after device sync");
        uint64_t r = 0, u = 0;
        hipMemPoolGetAttribute(pool, hipMemPoolAttrReservedMemCurrent, &r);
        hipMemPoolGetAttribute(pool, hipMemPoolAttrUsedMemCurrent, &u);
        printf(" pool reserved %.1f GB used %.1f GB\n", r / 1e9, u / 1e9);
    }
    // plain hipMalloc timing
    for (size_t gb : {40, 40}) { void *p; double t0 = now(); hipMalloc(&p, gb << 30); double t1 = now(); hipMemset(p, 1, gb << 30); hipDeviceSynchronize(); double t2 = now(); hipFree(p); double t3 = now();
        printf("hipMalloc %zu GB %.3f s, memset %.3f, free %.3f\n", gb, t1 - t0, t2 - t1, t3 - t2); }
    return 0;
}
