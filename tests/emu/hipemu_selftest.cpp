// TEST INFRASTRUCTURE -- does the emulator do what it claims?  Two kernels that reverse a block's 256 values through LDS: one with the
// __syncthreads() between the write and the read, one WITHOUT.  Run in order both give... the right answer here (fibers run one after
// the other), which is exactly why a missing barrier needs the ThreadSanitizer build to be seen: there every fiber is a thread of its
// own and only barriers order them, so the second kernel is reported as a data race and the first is not.  Also: a wave shuffle, a lane
// swap, a block whose threads leave early before a barrier, deferred streams with an event wait.
//   translate.py < this file | clang++ -std=c++17 -I tests/emu/hipemu [-fsanitize=thread] ... hipemu.cpp   (tests/test_emu_gpu_suite.py)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

static __global__ void reverse_with_barrier(const unsigned long long *in, unsigned long long *out) {
    __shared__ unsigned long long buf[256];
    buf[threadIdx.x] = in[blockIdx.x * 256 + threadIdx.x];
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = buf[255 - threadIdx.x];
}
static __global__ void reverse_without_barrier(const unsigned long long *in, unsigned long long *out) {
    __shared__ unsigned long long buf[256];
    buf[threadIdx.x] = in[blockIdx.x * 256 + threadIdx.x];
    out[blockIdx.x * 256 + threadIdx.x] = buf[255 - threadIdx.x];      // a race: thread 255 - x may not have written yet
}
static __global__ void wave_ops(unsigned *out) {
    const unsigned lane = threadIdx.x & 63;
    const int a = __shfl((int)(lane * 3), (int)((lane + 1) & 63), 64);           // neighbour's value
    const int b = __shfl_down((int)lane, 4, 64);                                 // lane + 4, own beyond the end
    if (threadIdx.x >= 200) return;                                              // (some threads leave before the barrier)
    __syncthreads();
    out[threadIdx.x] = (unsigned)a * 1000u + (unsigned)b;
}

int main(int argc, char **argv) {
    const bool racy = argc > 1 && argv[1][0] == 'r';
    const size_t n = 8 * 256;
    unsigned long long *d_in = nullptr, *d_out = nullptr;
    unsigned *d_w = nullptr;
    hipMalloc(&d_in, n * 8); hipMalloc(&d_out, n * 8); hipMalloc(&d_w, 256 * 4);
    std::vector<unsigned long long> h(n), got(n);
    for (size_t i = 0; i < n; ++i) h[i] = i * 0x9E3779B97F4A7C15ULL;
    hipStream_t s1 = nullptr, s2 = nullptr;
    hipEvent_t ev = nullptr;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); hipEventCreate(&ev);
    hipMemcpyAsync(d_in, h.data(), n * 8, hipMemcpyHostToDevice, s1);
    hipEventRecord(ev, s1);
    hipStreamWaitEvent(s2, ev, 0);                       // the kernel on s2 needs the copy on s1
    if (racy) reverse_without_barrier<<<8, 256, 0, s2>>>(d_in, d_out);
    else reverse_with_barrier<<<8, 256, 0, s2>>>(d_in, d_out);
    hipMemcpyAsync(got.data(), d_out, n * 8, hipMemcpyDeviceToHost, s2);
    hipStreamSynchronize(s2);
    int bad = 0;
    for (size_t b = 0; b < 8; ++b)
        for (size_t t = 0; t < 256; ++t) bad += got[b * 256 + t] != h[b * 256 + 255 - t];
    wave_ops<<<1, 256, 0, s2>>>(d_w);
    std::vector<unsigned> w(256);
    hipMemcpyAsync(w.data(), d_w, 200 * 4, hipMemcpyDeviceToHost, s2);
    hipStreamSynchronize(s2);
    for (unsigned t = 0; t < 200; ++t) {
        const unsigned lane = t & 63, a = ((lane + 1) & 63) * 3, b = lane + 4 < 64 ? lane + 4 : lane;
        bad += w[t] != a * 1000u + b;
    }
    hipFree(d_in); hipFree(d_out); hipFree(d_w);
    printf("hipemu selftest (%s): %d wrong values\n", racy ? "kernel without its barrier" : "kernel with its barrier", bad);
    return bad ? 1 : 0;
}
