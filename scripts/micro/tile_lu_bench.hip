// micro-benchmark of the 32x32 diagonal-tile LU on the critical path of the panel chain (mf_kernels.h: tile_factor):
//   0: the product's version (one wavefront, lane = row, pivot row by v_readlane)
//   1: same layout, pivot row by ds_bpermute (__shfl with a uniform index) instead of readlane
//   2: 64 lanes = 32 rows x 2 column halves (columns interleaved), pivot row and multiplier by ds_bpermute
// Each variant factors the same tile REPS times inside one kernel (reloading it from LDS), timed with the wall
// clock; the factors of all variants are compared bit for bit.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -I sanm_amd/csrc scripts/micro/tile_lu_bench.hip -o /tmp/tile_lu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "mf_kernels.h"
using namespace sanm_hip::mfk;

__device__ __forceinline__ double shfl_f64(double v, int lane) { return __shfl(v, lane, 64); }

// ---- variant 1 ------------------------------------------------------------------------------------------------
template <int J>
__device__ __forceinline__ void v1_step(double (&a)[NB], int r, double inv, double piv, int& nbad, double thr) {
    const double l = (r > J) ? a[J] * inv : 0.0;
    a[J] = (r > J) ? l : (r == J ? piv : a[J]);
    double inv_next = 1.0, piv_next = 1.0;
    if constexpr (J + 1 < NB) {
        a[J + 1] = __builtin_fma(-l, shfl_f64(a[J + 1], J), a[J + 1]);
        piv_next = shfl_f64(a[J + 1], J + 1);
        inv_next = pivot_reciprocal(piv_next, nbad, true, thr);
    }
#pragma unroll
    for (int c = J + 2; c < NB; ++c) a[c] = __builtin_fma(-l, shfl_f64(a[c], J), a[c]);
    if constexpr (J + 1 < NB) v1_step<J + 1>(a, r, inv_next, piv_next, nbad, thr);
}
__device__ __forceinline__ void v1_factor(double (*T)[TPAD], int tid, int32_t* status, double thr) {
    if (tid >= 64) return;
    const int r = tid & (NB - 1);
    double a[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) a[c] = T[r][c];
    int nbad = 0;
    double piv0 = shfl_f64(a[0], 0);
    const double inv0 = pivot_reciprocal(piv0, nbad, true, thr);
    v1_step<0>(a, r, inv0, piv0, nbad, thr);
    if (tid == 0 && nbad) atomicAdd(status, nbad);
    if (tid < NB) {
#pragma unroll
        for (int c = 0; c < NB; ++c) T[r][c] = a[c];
    }
}

// ---- variant 2: lane (r, h) holds columns 2 q + h, q < 16 --------------------------------------------------------
template <int J>
__device__ __forceinline__ void v2_step(double (&a)[NB / 2], int r, int h, double inv, double piv, int& nbad, double thr) {
    constexpr int HJ = J & 1, QJ = J >> 1;  // the pivot column lives in half HJ at local index QJ
    // multiplier: computed where column J is held, sent to the other half of the same row
    double l = (h == HJ && r > J) ? a[QJ] * inv : 0.0;
    if (h == HJ) a[QJ] = (r > J) ? l : (r == J ? piv : a[QJ]);
    l = shfl_f64(l, r + 32 * HJ);
    double inv_next = 1.0, piv_next = 1.0;
    if constexpr (J + 1 < NB) {
        constexpr int H1 = (J + 1) & 1, Q1 = (J + 1) >> 1;
        const double pr = shfl_f64(a[Q1], J + 32 * H1);  // row J's entry of column J + 1
        if (h == H1) a[Q1] = __builtin_fma(-l, pr, a[Q1]);
        piv_next = shfl_f64(a[Q1], J + 1 + 32 * H1);
        inv_next = pivot_reciprocal(piv_next, nbad, true, thr);
    }
    // remaining columns c >= J + 2: local q with 2 q + h >= J + 2
#pragma unroll
    for (int q = (J + 2) >> 1; q < NB / 2; ++q) {
        const double pr = shfl_f64(a[q], J + 32 * h);  // row J, same half
        const bool live = 2 * q + h >= J + 2;
        a[q] = live ? __builtin_fma(-l, pr, a[q]) : a[q];
    }
    if constexpr (J + 1 < NB) v2_step<J + 1>(a, r, h, inv_next, piv_next, nbad, thr);
}
__device__ __forceinline__ void v2_factor(double (*T)[TPAD], int tid, int32_t* status, double thr) {
    if (tid >= 64) return;
    const int r = tid & (NB - 1), h = tid >> 5;
    double a[NB / 2];
#pragma unroll
    for (int q = 0; q < NB / 2; ++q) a[q] = T[r][2 * q + h];
    int nbad = 0;
    double piv0 = shfl_f64(a[0], 0);
    const double inv0 = pivot_reciprocal(piv0, nbad, true, thr);
    v2_step<0>(a, r, h, inv0, piv0, nbad, thr);
    if (tid == 0 && nbad) atomicAdd(status, nbad);
#pragma unroll
    for (int q = 0; q < NB / 2; ++q) T[r][2 * q + h] = a[q];
}

template <int VARIANT>
__global__ void __launch_bounds__(256) k(const double* F, double* out, int32_t* status, long long* ticks, int reps) {
    __shared__ double T[NB][TPAD];
    __shared__ double S[NB][TPAD];
    const int tid = threadIdx.x, tc = tid % NB, tr = tid / NB;
    for (int s = 0; s < 4; ++s) S[tr + 8 * s][tc] = F[(tr + 8 * s) * NB + tc];
    __syncthreads();
    long long t0 = 0, acc = 0;
    for (int it = 0; it < reps; ++it) {
        for (int s = 0; s < 4; ++s) T[tr + 8 * s][tc] = S[tr + 8 * s][tc];
        __syncthreads();
        if (tid == 0) t0 = wall_clock64();
        if (VARIANT == 0) tile_factor(T, NB, tid, status, 1e-300);
        else if (VARIANT == 1) v1_factor(T, tid, status, 1e-300);
        else v2_factor(T, tid, status, 1e-300);
        if (tid == 0) acc += wall_clock64() - t0;
        __syncthreads();
    }
    for (int s = 0; s < 4; ++s) out[(tr + 8 * s) * NB + tc] = T[tr + 8 * s][tc];
    if (tid == 0) *ticks = acc;
}

int main() {
    std::vector<double> h(NB * NB);
    for (int i = 0; i < NB; ++i)
        for (int j = 0; j < NB; ++j) h[i * NB + j] = (i == j ? 4.0 : 0.0) + 1.0 / (1 + i + 2 * j) - 0.3 / (3 + 2 * i + j);
    double *F, *O;
    int32_t* st;
    long long* tk;
    hipMalloc(&F, h.size() * 8);
    hipMalloc(&O, h.size() * 8);
    hipMalloc(&st, 64);
    hipMalloc(&tk, 8);
    hipMemset(st, 0, 64);
    hipMemcpy(F, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    std::vector<double> ref, got(NB * NB);
    const int reps = 200;
    auto run = [&](auto kern, const char* name) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(256), 0, 0, F, O, st, tk, reps);
        hipLaunchKernelGGL(kern, dim3(1), dim3(256), 0, 0, F, O, st, tk, reps);
        hipDeviceSynchronize();
        long long t;
        hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost);
        hipMemcpy(got.data(), O, got.size() * 8, hipMemcpyDeviceToHost);
        bool same = true;
        if (ref.empty()) ref = got;
        else same = std::memcmp(ref.data(), got.data(), got.size() * 8) == 0;
        // wall_clock64 ticks at 100 MHz
        printf("%-44s %7.2f us per tile   bits %s\n", name, (double)t / reps / 100.0, same ? "identical" : "DIFFER");
    };
    run(k<0>, "0 product (lane = row, readlane)");
    run(k<1>, "1 lane = row, ds_bpermute");
    run(k<2>, "2 32 rows x 2 column halves, ds_bpermute");
    return 0;
}
