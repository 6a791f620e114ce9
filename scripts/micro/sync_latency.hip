// Host wait latency after a small kernel: hipStreamSynchronize vs polling a flag a trailing 1-thread kernel writes
// to pinned host memory.   hipcc --offload-arch=gfx950 -O2 -o /tmp/sync_latency scripts/micro/sync_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <immintrin.h>
__global__ void work(double* x, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x[i] * 1.0000001 + 1e-9;
}
__global__ void flag_kernel(volatile unsigned long long* f, unsigned long long v) {
    __atomic_store_n((unsigned long long*)f, v, __ATOMIC_RELEASE);
}
int main() {
    double* x;
    const int n = 40000;
    hipMalloc(&x, n * 8);
    hipMemset(x, 0, n * 8);
    unsigned long long* flag;
    hipHostMalloc(&flag, 64);
    *flag = 0;
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    auto now = [] { return std::chrono::steady_clock::now(); };
    const int reps = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int w = 0; w < 100; ++w) { hipLaunchKernelGGL(work, dim3(157), dim3(256), 0, s, x, n); hipStreamSynchronize(s); }
        auto t0 = now();
        unsigned long long seq = *flag;
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(work, dim3(157), dim3(256), 0, s, x, n);
            if (mode == 0) {
                hipStreamSynchronize(s);
            } else if (mode == 1) {
                ++seq;
                hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, s, flag, seq);
                while (*(volatile unsigned long long*)flag != seq) _mm_pause();
            } else {
                ++seq;
                hipStreamWriteValue64(s, flag, seq, 0);
                while (*(volatile unsigned long long*)flag != seq) _mm_pause();
            }
        }
        double us = std::chrono::duration<double, std::micro>(now() - t0).count() / reps;
        printf("mode %d (%s): %.2f us per launch+wait\n", mode, mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "flag kernel + poll" : "hipStreamWriteValue64 + poll", us);
        hipStreamSynchronize(s);
    }
    return 0;
}
