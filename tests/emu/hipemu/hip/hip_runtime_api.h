/* TEST INFRASTRUCTURE -- hipemu: the host-side runtime API of <hip/hip_runtime_api.h> as far as libzkstark and tests/cabi/ use it,
 * implemented on the CPU by tests/emu/hipemu/hipemu.cpp (see hip_runtime.h in this directory).  Plain C. */
#ifndef HIPEMU_RUNTIME_API_H
#define HIPEMU_RUNTIME_API_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum hipError_t {
    hipSuccess = 0,
    hipErrorInvalidValue = 1,
    hipErrorOutOfMemory = 2,
    hipErrorInvalidDevice = 101,
    hipErrorInvalidResourceHandle = 400,
    hipErrorNotReady = 600,
    hipErrorLaunchFailure = 719,
    hipErrorUnknown = 999
} hipError_t;

typedef enum hipMemcpyKind {
    hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4
} hipMemcpyKind;

typedef struct hipemu_stream *hipStream_t;
typedef struct hipemu_event *hipEvent_t;

typedef struct hipDeviceProp_t {
    char name[256];
    size_t totalGlobalMem;
    int multiProcessorCount;
    int warpSize;
    int maxThreadsPerBlock;
    size_t sharedMemPerBlock;
    char gcnArchName[256];
} hipDeviceProp_t;

enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum { hipEventDefault = 0, hipEventBlockingSync = 1, hipEventDisableTiming = 2 };
enum { hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };
enum { hipHostRegisterDefault = 0 };
typedef enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 } hipFuncAttribute;

hipError_t hipGetDeviceCount(int *count);
hipError_t hipSetDevice(int device);
hipError_t hipGetDevice(int *device);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *prop, int device);
hipError_t hipDeviceSynchronize(void);
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest);
hipError_t hipMemGetInfo(size_t *free_bytes, size_t *total_bytes);

hipError_t hipGetLastError(void);
hipError_t hipPeekAtLastError(void);
const char *hipGetErrorString(hipError_t e);
const char *hipGetErrorName(hipError_t e);

hipError_t hipStreamCreate(hipStream_t *s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int priority);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);

hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t start, hipEvent_t stop);

hipError_t hipMalloc(void **p, size_t bytes);
hipError_t hipFree(void *p);
hipError_t hipMallocAsync(void **p, size_t bytes, hipStream_t s);
hipError_t hipFreeAsync(void *p, hipStream_t s);
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipHostRegister(void *p, size_t bytes, unsigned flags);
hipError_t hipHostUnregister(void *p);

hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpy2D(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind);
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemset(void *dst, int value, size_t bytes);
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t s);
hipError_t hipMemset2DAsync(void *dst, size_t pitch, int value, size_t width, size_t height, hipStream_t s);

hipError_t hipFuncSetAttribute(const void *func, hipFuncAttribute attr, int value);

/* hipemu's own controls (tests): counters of what ran; failure injection.  HIPEMU_FAIL_MALLOC_AT=n makes the n-th hipMalloc of the
 * process (1-based) return hipErrorOutOfMemory; hipemu_fail_malloc_at does the same from code (0 = off). */
void hipemu_fail_malloc_at(long nth);
void hipemu_fail_malloc_from(long nth);   /* every hipMalloc from the n-th on fails (a device that is full), until reset with 0 */
void hipemu_fail_launch_at(long nth);     /* the n-th kernel launch from now on is refused (hipGetLastError: launch failure), 0 = off */
void hipemu_counters(uint64_t out[8]);     /* launches, blocks, fibers, fiber switches, copies, mallocs, deferred ops run late, streams made */

#ifdef __cplusplus
}
template <class T> static inline hipError_t hipMalloc(T **p, size_t bytes) { return hipMalloc((void **)p, bytes); }
template <class T> static inline hipError_t hipHostMalloc(T **p, size_t bytes, unsigned flags = 0) { return hipHostMalloc((void **)p, bytes, flags); }
template <class T> static inline hipError_t hipFuncSetAttribute(T *func, hipFuncAttribute attr, int value) { return hipFuncSetAttribute((const void *)func, attr, value); }
#endif
#endif
