// micro-benchmark of the 32x32 tile LU + inverse (mf_kernels.h: tile_factor)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -I sanm_amd/csrc scripts/bench_tile.hip -o /tmp/bench_tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "mf_kernels.h"
using namespace sanm_hip::mfk;

template <int VARIANT>
__global__ void __launch_bounds__(256) k(double* F, double* D, int32_t* status, int nfront) {
    __shared__ double T[NB][TPAD];
    const int tid = threadIdx.x, tc = tid % NB, tr = tid / NB;
    double* Ff = F + (size_t)blockIdx.x * NB * NB;
    for (int s = 0; s < 4; ++s) T[tr + 8 * s][tc] = Ff[(tr + 8 * s) * NB + tc];
    __syncthreads();
    if (VARIANT == 0) {
        tile_factor(T, NB, tid, status);
    } else if (VARIANT == 1) {  // elimination only
        for (int j = 0; j < NB; ++j) {
            double piv = T[j][j];
            const double inv = 1.0 / piv;
            if (tc > j) {
                const double u = T[j][tc];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    int r = tr + 8 * s;
                    if (r > j) T[r][tc] -= (T[r][j] * inv) * u;
                }
            }
            __syncthreads();
        }
    } else if (VARIANT == 2) {  // load/store only
    }
    __syncthreads();
    for (int s = 0; s < 4; ++s) Ff[(tr + 8 * s) * NB + tc] = T[tr + 8 * s][tc];
}

int main() {
    const int nf = 4;
    std::vector<double> h(nf * NB * NB);
    for (int f = 0; f < nf; ++f)
        for (int i = 0; i < NB; ++i)
            for (int j = 0; j < NB; ++j) h[f * NB * NB + i * NB + j] = (i == j ? 40.0 : 0.0) + 1.0 / (1 + i + 2 * j);
    double *F, *D;
    int32_t* st;
    hipMalloc(&F, h.size() * 8);
    hipMalloc(&D, nf * 2 * NB * NB * 8);
    hipMalloc(&st, 64);
    hipMemset(st, 0, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name) {
        hipMemcpy(F, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(nf), dim3(256), 0, 0, F, D, st, nf);
        hipEventRecord(e0);
        const int reps = 200;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(nf), dim3(256), 0, 0, F, D, st, nf);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %.2f us per launch\n", name, ms * 1e3 / reps);
    };
    run(k<0>, "tile_factor (in-LDS LU)");
    run(k<1>, "elimination only");
    run(k<2>, "load/store only");
    return 0;
}
