// HIP kernels of the multifrontal LU (see multifrontal.h for the method).
// Included by backend_hip.hip only.
//
// All fronts of one tree level are processed by the same launches; blocks that
// fall outside a front's extent exit at once.  Tiles are NB x NB (NB = 32),
// one 256-thread workgroup per tile, operands staged through LDS.
#pragma once
#include <hip/hip_runtime.h>

#include "mf_types.h"

namespace sanm_hip {
namespace mfk {

constexpr int NB = MF_NB;
constexpr int TPAD = NB + 1;  // LDS row stride (odd: no bank conflicts on column access)

__global__ void scatter_kernel(int64_t nnz, const int64_t* __restrict__ a_dst,
                               const double* __restrict__ val, double* __restrict__ store) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nnz) store[a_dst[p]] = val[p];
}

// parent[rel[i], rel[j]] += child_schur[i, j]; one child per blockIdx.y
__global__ void __launch_bounds__(256) extend_add_kernel(MfDev mf, const int32_t* __restrict__ children) {
    const MfFrontDev c = mf.fronts[children[blockIdx.y]];
    const int nb = c.m - c.k;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)nb * nb) return;
    const int i = idx / nb, j = idx % nb;
    const MfFrontDev p = mf.fronts[c.parent];
    const int32_t* rel = mf.rel + c.rel_off;
    double v = mf.front_store[c.off + (int64_t)(c.k + i) * c.m + c.k + j];
    mf.front_store[p.off + (int64_t)rel[i] * p.m + rel[j]] += v;
}

// LU of the diagonal tile of panel p (kb pivots) + inverses of the extended
// unit-lower / upper tile factors.  One workgroup per front.
__global__ void __launch_bounds__(256) diag_kernel(MfDev mf, int level_begin, int p) {
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.x]];
    const int m = f.m, r0 = p * NB;
    const int kb = min(NB, f.k - r0);
    __shared__ double T[NB][TPAD], LI[NB][TPAD], UI[NB][TPAD];
    double* F = mf.front_store + f.off;
    const int tid = threadIdx.x, tc = tid % NB, tr = tid / NB;  // tr in 0..7
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s, gr = r0 + r, gc = r0 + tc;
        T[r][tc] = (gr < m && gc < m) ? F[(int64_t)gr * m + gc] : (r == tc ? 1.0 : 0.0);
    }
    for (int j = 0; j < kb; ++j) {
        __syncthreads();
        double piv = T[j][j];
        if (!(fabs(piv) > 1e-290)) {
            if (tid == 0) atomicAdd(mf.status, 1);
            piv = 1.0;
        }
        __syncthreads();
        if (tid > j && tid < NB) T[tid][j] /= piv;
        __syncthreads();
        for (int s = 0; s < 4; ++s) {
            int r = tr + 8 * s;
            if (r > j && tc > j) T[r][tc] -= T[r][j] * T[j][tc];
        }
    }
    __syncthreads();
    // write the factored tile back
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s, gr = r0 + r, gc = r0 + tc;
        if (gr < m && gc < m) F[(int64_t)gr * m + gc] = T[r][tc];
    }
    // inverse of Lext = [[L11,0],[L21,I]] (unit lower) and Uext = [[U11,U12],[0,I]] (upper):
    // thread c < NB builds column c of each by substitution
    if (tid < NB) {
        const int c = tid;
        // Lext X = e_c  (forward)
        for (int r = 0; r < NB; ++r) {
            double v = (r == c) ? 1.0 : 0.0;
            const int lim = min(r, kb);
            for (int q = c; q < lim; ++q) v -= T[r][q] * LI[q][c];  // X[q][c] = 0 for q < c
            LI[r][c] = (r < c) ? 0.0 : v;
        }
        // Uext X = e_c  (backward)
        for (int r = NB - 1; r >= 0; --r) {
            double v = (r == c) ? 1.0 : 0.0;
            if (r < kb) {
                for (int q = r + 1; q <= c; ++q) v -= T[r][q] * UI[q][c];  // X[q][c] = 0 for q > c
                double d = T[r][r];
                v = (r > c) ? 0.0 : v / ((fabs(d) > 1e-290) ? d : 1.0);
            } else {
                v = (r == c) ? 1.0 : 0.0;
            }
            UI[r][c] = v;
        }
    }
    __syncthreads();
    double* D = mf.dinv_store + f.dinv_off + (int64_t)p * 2 * NB * NB;
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s;
        D[r * NB + tc] = LI[r][tc];
        D[NB * NB + r * NB + tc] = UI[r][tc];
    }
}

// panel tiles: blockIdx.y == 0: U panel tile (p, t) <- Linv * tile
//              blockIdx.y == 1: L panel tile (t, p) <- tile * Uinv      (t > p)
__global__ void __launch_bounds__(256) trsm_kernel(MfDev mf, int level_begin, int p) {
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.z]];
    const int m = f.m, nt = (m + NB - 1) / NB;
    const int t = p + 1 + blockIdx.x;
    if (t >= nt) return;
    const bool upanel = blockIdx.y == 0;
    __shared__ double A[NB][TPAD], B[NB][TPAD];
    double* F = mf.front_store + f.off;
    const double* D = mf.dinv_store + f.dinv_off + (int64_t)p * 2 * NB * NB + (upanel ? 0 : NB * NB);
    const int r0 = (upanel ? p : t) * NB, c0 = (upanel ? t : p) * NB;
    const int tid = threadIdx.x, tc = tid % NB, tr = tid / NB;
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s, gr = r0 + r, gc = c0 + tc;
        A[r][tc] = (gr < m && gc < m) ? F[(int64_t)gr * m + gc] : 0.0;
        B[r][tc] = D[r * NB + tc];
    }
    __syncthreads();
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s, gr = r0 + r, gc = c0 + tc;
        double acc = 0;
        if (upanel) {
#pragma unroll 8
            for (int q = 0; q < NB; ++q) acc += B[r][q] * A[q][tc];  // Linv * tile
        } else {
#pragma unroll 8
            for (int q = 0; q < NB; ++q) acc += A[r][q] * B[q][tc];  // tile * Uinv
        }
        if (gr < m && gc < m) F[(int64_t)gr * m + gc] = acc;
    }
}

// trailing update: tile(ti,tj) -= L(ti,p)[:, :kb] * U(p,tj)[:kb, :]   (ti, tj > p)
__global__ void __launch_bounds__(256) update_kernel(MfDev mf, int level_begin, int p) {
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.z]];
    const int m = f.m, nt = (m + NB - 1) / NB;
    const int ti = p + 1 + blockIdx.y, tj = p + 1 + blockIdx.x;
    if (ti >= nt || tj >= nt) return;
    const int kb = min(NB, f.k - p * NB);
    __shared__ double L[NB][TPAD], U[NB][TPAD];
    double* F = mf.front_store + f.off;
    const int tid = threadIdx.x, tc = tid % NB, tr = tid / NB;
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s;
        int gr = ti * NB + r, gc = p * NB + tc;
        L[r][tc] = (gr < m && tc < kb) ? F[(int64_t)gr * m + gc] : 0.0;
        gr = p * NB + r;
        gc = tj * NB + tc;
        U[r][tc] = (r < kb && gc < m) ? F[(int64_t)gr * m + gc] : 0.0;
    }
    __syncthreads();
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s, gr = ti * NB + r, gc = tj * NB + tc;
        if (gr < m && gc < m) {
            double acc = 0;
#pragma unroll 8
            for (int q = 0; q < NB; ++q) acc += L[r][q] * U[q][tc];
            F[(int64_t)gr * m + gc] -= acc;
        }
    }
}

// ---------------------------------------------------------------- solve --
__global__ void permute_in_kernel(int64_t n, const int32_t* __restrict__ perm,
                                  const double* __restrict__ b, double* __restrict__ w) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[perm[i]] = b[i];
}
__global__ void permute_out_kernel(int64_t n, const int32_t* __restrict__ perm,
                                   const double* __restrict__ w, double* __restrict__ x) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = w[perm[i]];
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// forward, own part: z = L11^-1 (w_own + gathered child updates); one workgroup
// per front; the running vector lives in dynamic LDS (k doubles).
__global__ void __launch_bounds__(256) fwd_own_kernel(MfDev mf, int level_begin) {
    extern __shared__ double t[];
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.x]];
    const int m = f.m, k = f.k;
    const double* F = mf.front_store + f.off;
    const int32_t* gp = mf.gat_ptr + f.gat_off;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int r = tid; r < k; r += 256) {
        double v = mf.work[f.own_start + r];
        for (int32_t s = gp[r]; s < gp[r + 1]; ++s) v += mf.upd_store[mf.gat_src[s]];
        t[r] = v;
    }
    __shared__ double tp[NB];
    const int np = (k + NB - 1) / NB;
    for (int p = 0; p < np; ++p) {
        __syncthreads();
        const int r0 = p * NB, kb = min(NB, k - r0);
        // t[r0+r] -= F[r0+r, 0:r0] . z[0:r0]   (one wave per row, 4 rows at a time)
        for (int r = wv; r < kb; r += 4) {
            const double* row = F + (int64_t)(r0 + r) * m;
            double acc = 0;
            for (int c = lane; c < r0; c += 64) acc += row[c] * t[c];
            acc = wave_sum(acc);
            if (lane == 0) tp[r] = t[r0 + r] - acc;
        }
        __syncthreads();
        // z_p = Linv_pp * tp
        if (tid < kb) {
            const double* LI = mf.dinv_store + f.dinv_off + (int64_t)p * 2 * NB * NB;
            double acc = 0;
            for (int q = 0; q <= tid; ++q) acc += LI[tid * NB + q] * tp[q];
            t[r0 + tid] = acc;
        }
    }
    __syncthreads();
    for (int r = tid; r < k; r += 256) mf.work[f.own_start + r] = t[r];
}

// forward, boundary part: upd[r-k] = gathered(r) - F[r, 0:k] . z ; one wave per row
__global__ void __launch_bounds__(256) fwd_bnd_kernel(MfDev mf, int level_begin) {
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.y]];
    const int m = f.m, k = f.k;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = k + blockIdx.x * 4 + wv;
    if (r >= m) return;
    const double* row = mf.front_store + f.off + (int64_t)r * m;
    const double* z = mf.work + f.own_start;
    double acc = 0;
    for (int c = lane; c < k; c += 64) acc += row[c] * z[c];
    acc = wave_sum(acc);
    if (lane == 0) {
        const int32_t* gp = mf.gat_ptr + f.gat_off;
        double v = 0;
        for (int32_t s = gp[r]; s < gp[r + 1]; ++s) v += mf.upd_store[mf.gat_src[s]];
        mf.upd_store[f.upd_off + r - k] = v - acc;
    }
}

// backward, coupling part: w_own[r] -= F[r, k:m] . x[bnd] ; one wave per row
__global__ void __launch_bounds__(256) bwd_bnd_kernel(MfDev mf, int level_begin) {
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.y]];
    const int m = f.m, k = f.k;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= k || m == k) return;
    const double* row = mf.front_store + f.off + (int64_t)r * m;
    const int32_t* bi = mf.bnd_idx + f.bnd_off;
    double acc = 0;
    for (int c = k + lane; c < m; c += 64) acc += row[c] * mf.work[bi[c - k]];
    acc = wave_sum(acc);
    if (lane == 0) mf.work[f.own_start + r] -= acc;
}

// backward, own part: x_own = U11^-1 t ; one workgroup per front
__global__ void __launch_bounds__(256) bwd_own_kernel(MfDev mf, int level_begin) {
    extern __shared__ double t[];
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.x]];
    const int m = f.m, k = f.k;
    const double* F = mf.front_store + f.off;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int r = tid; r < k; r += 256) t[r] = mf.work[f.own_start + r];
    __shared__ double tp[NB];
    const int np = (k + NB - 1) / NB;
    for (int p = np - 1; p >= 0; --p) {
        __syncthreads();
        const int r0 = p * NB, kb = min(NB, k - r0), c0 = r0 + kb;
        for (int r = wv; r < kb; r += 4) {
            const double* row = F + (int64_t)(r0 + r) * m;
            double acc = 0;
            for (int c = c0 + lane; c < k; c += 64) acc += row[c] * t[c];
            acc = wave_sum(acc);
            if (lane == 0) tp[r] = t[r0 + r] - acc;
        }
        __syncthreads();
        if (tid < kb) {
            const double* UI = mf.dinv_store + f.dinv_off + (int64_t)p * 2 * NB * NB + NB * NB;
            double acc = 0;
            for (int q = tid; q < kb; ++q) acc += UI[tid * NB + q] * tp[q];
            t[r0 + tid] = acc;
        }
    }
    __syncthreads();
    for (int r = tid; r < k; r += 256) mf.work[f.own_start + r] = t[r];
}

}  // namespace mfk
}  // namespace sanm_hip
