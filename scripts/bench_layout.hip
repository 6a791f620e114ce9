// micro-benchmark: per-tet state streamed by one lane per tet, SoA [component][Tpad] against 64-tet blocks
// [block][component][64]; same bytes, same loads in flight (NF independent loads per wait)
// build: hipcc --offload-arch=gfx950 -O3 scripts/bench_layout.hip -o /tmp/bench_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NF, bool BLOCKED>
__global__ void __launch_bounds__(256) k(const double* __restrict__ a, double* out, long T, long Tpad, int ncomp) {
    const long tet = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tet >= T) return;
    const double* p = BLOCKED ? a + (tet >> 6) * (long)ncomp * 64 + (tet & 63) : a + tet;
    const long s = BLOCKED ? 64 : Tpad;
    double acc[NF];
    for (int j = 0; j < NF; ++j) acc[j] = 0;
    for (int c = 0; c + NF <= ncomp; c += NF) {
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[j] += p[(long)(c + j) * s];
    }
    double r = 0;
    for (int j = 0; j < NF; ++j) r += acc[j];
    out[tet] = r;
}

// latency: a chain of dependent loads, each from a different component array (the operator chain of a pass)
template <bool BLOCKED>
__global__ void __launch_bounds__(256) chain(const double* __restrict__ a, double* out, long T, long Tpad, int ncomp,
                                             int nsteps, int hop, int start) {
    const long tet = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tet >= T) return;
    const double* p = BLOCKED ? a + (tet >> 6) * (long)ncomp * 64 + (tet & 63) : a + tet;
    const long s = BLOCKED ? 64 : Tpad;
    long c = start;
    double v = 0;
    for (int j = 0; j < nsteps; ++j) {
        v += p[c * s];
        c = (c + hop + (long)v) % ncomp;  // v is 0: the next address depends on the loaded value
    }
    out[tet] = v;
}

int main(int argc, char** argv) {
    const long T = 42288, Tpad = (T + 63) / 64 * 64;
    const int ncomp_total = 2400;  // 0.8 GB of state
    double *a, *out;
    hipMalloc(&a, (size_t)Tpad * ncomp_total * 8);
    hipMemset(a, 0, (size_t)Tpad * ncomp_total * 8);
    hipMalloc(&out, Tpad * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name, int wg, int ncomp) {
        const int reps = 20;
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3((T + wg - 1) / wg), dim3(wg), 0, 0, a, out, T, Tpad, ncomp);
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3((T + wg - 1) / wg), dim3(wg), 0, 0, a, out, T, Tpad, ncomp);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps, gb = (double)T * ncomp * 8 / 1e9;
        printf("%-34s ncomp %4d: %7.1f us  %6.0f GB/s\n", name, ncomp, us, gb / (us * 1e-6));
    };
    for (int ncomp : {36, 144, 900, 2376}) {
        run(k<18, false>, "SoA      18 in flight, wg 64", 64, ncomp);
        run(k<18, true>, "blocked  18 in flight, wg 64", 64, ncomp);
        run(k<36, false>, "SoA      36 in flight, wg 64", 64, ncomp);
        run(k<36, true>, "blocked  36 in flight, wg 64", 64, ncomp);
    }
    auto runc = [&](auto kern, const char* name, int ncomp, int hop) {
        const int reps = 20, nsteps = 40;
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3((T + 63) / 64), dim3(64), 0, 0, a, out, T, Tpad, ncomp, nsteps, hop, 0);
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3((T + 63) / 64), dim3(64), 0, 0, a, out, T, Tpad, ncomp, nsteps, hop, (i * 997 + 13) % ncomp);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-20s ncomp %4d hop %3d: %7.1f us per launch, %.2f us per dependent load\n", name, ncomp, hop,
               ms * 1e3 / reps, ms * 1e3 / reps / nsteps);
    };
    for (int hop : {1, 37, 601}) {
        runc(chain<false>, "chain SoA", 2376, hop);
        runc(chain<true>, "chain blocked", 2376, hop);
    }
    return 0;
}
