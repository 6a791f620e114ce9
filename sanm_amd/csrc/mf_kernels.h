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

// identity blocks of the augmentation: F[r, m + r] = F[m + r, r] = 1 for r < k
__global__ void aug_identity_kernel(MfDev mf) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= mf.n) return;
    const MfFrontDev f = mf.fronts[mf.own_front[i]];
    const int r = i - f.own_start;
    double* F = mf.front_store + f.off;
    F[(int64_t)r * f.ld + f.m + r] = 1.0;
    F[(int64_t)(f.m + r) * f.ld + r] = 1.0;
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
    double v = mf.front_store[c.off + (int64_t)(c.k + i) * c.ld + c.k + j];
    mf.front_store[p.off + (int64_t)rel[i] * p.ld + rel[j]] += v;
}

// In-LDS LU of a diagonal tile (kb pivots, no pivoting) by a 256-thread
// workgroup, followed by the inverses of the extended tile factors
//   Lext = [[L11,0],[L21,I]] (unit lower),  Uext = [[U11,U12],[0,I]] (upper)
// written to D = [Lext^-1 | Uext^-1].  On entry T holds the tile (synchronised);
// on exit T holds the packed factors (synchronised).
__device__ __forceinline__ void tile_factor(double (*T)[TPAD], double (*LI)[TPAD], double (*UI)[TPAD],
                                            int kb, int tid, double* D, int32_t* status) {
    const int tc = tid % NB, tr = tid / NB;  // tr in 0..7
    // right-looking elimination; column j is left unscaled during the sweep (later
    // steps never read it), so one barrier per step suffices
    for (int j = 0; j < kb; ++j) {
        double piv = T[j][j];
        if (!(fabs(piv) > 1e-290)) {
            if (tid == 0) atomicAdd(status, 1);
            piv = 1.0;
        }
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
    // scale the L columns
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s;
        if (tc < kb && r > tc) {
            double d = T[tc][tc];
            T[r][tc] /= (fabs(d) > 1e-290) ? d : 1.0;
        }
    }
    __syncthreads();
    // column c of each inverse by substitution; the inverse columns live in LDS
    // (LI / UI), lanes 0..31 of wave 0 build Lext^-1, lanes 0..31 of wave 1 Uext^-1
    const int c = tid & 63;
    if (tid < NB) {
        for (int i = 0; i < NB; ++i) {
            double v0 = (i == c) ? 1.0 : 0.0, v1 = 0, v2 = 0, v3 = 0;
            const int lim = min(i, kb);
            int q = 0;
            for (; q + 3 < lim; q += 4) {  // four independent LDS load pairs in flight
                v0 -= T[i][q] * LI[q][c];
                v1 -= T[i][q + 1] * LI[q + 1][c];
                v2 -= T[i][q + 2] * LI[q + 2][c];
                v3 -= T[i][q + 3] * LI[q + 3][c];
            }
            for (; q < lim; ++q) v0 -= T[i][q] * LI[q][c];
            LI[i][c] = (i < c) ? 0.0 : (v0 + v1) + (v2 + v3);
        }
    } else if (tid >= 64 && tid < 64 + NB) {
        for (int i = NB - 1; i >= 0; --i) {
            double v0 = (i == c) ? 1.0 : 0.0, v1 = 0, v2 = 0, v3 = 0;
            if (i < kb) {
                int q = i + 1;
                for (; q + 3 < NB; q += 4) {
                    v0 -= T[i][q] * UI[q][c];
                    v1 -= T[i][q + 1] * UI[q + 1][c];
                    v2 -= T[i][q + 2] * UI[q + 2][c];
                    v3 -= T[i][q + 3] * UI[q + 3][c];
                }
                for (; q < NB; ++q) v0 -= T[i][q] * UI[q][c];
                double d = T[i][i];
                v0 = ((v0 + v1) + (v2 + v3)) / ((fabs(d) > 1e-290) ? d : 1.0);
            }
            UI[i][c] = (i > c) ? 0.0 : v0;
        }
    }
    __syncthreads();
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s;
        D[r * NB + tc] = LI[r][tc];
        D[NB * NB + r * NB + tc] = UI[r][tc];
    }
}

// diagonal tile of panel p of every front of a level
__global__ void __launch_bounds__(256) diag_kernel(MfDev mf, int level_begin, int p) {
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.x]];
    const int m = f.ld, r0 = p * NB;  // m: extent and leading dimension of the augmented front
    const int kb = min(NB, f.k - r0);
    __shared__ double T[NB][TPAD], LI[NB][TPAD], UI[NB][TPAD];
    double* F = mf.front_store + f.off;
    const int tid = threadIdx.x, tc = tid % NB, tr = tid / NB;
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s, gr = r0 + r, gc = r0 + tc;
        T[r][tc] = (gr < m && gc < m) ? F[(int64_t)gr * m + gc] : (r == tc ? 1.0 : 0.0);
    }
    __syncthreads();
    tile_factor(T, LI, UI, kb, tid, mf.dinv_store + f.dinv_off + (int64_t)p * 2 * NB * NB, mf.status);
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s, gr = r0 + r, gc = r0 + tc;
        if (gr < m && gc < m) F[(int64_t)gr * m + gc] = T[r][tc];
    }
}

// panel tiles: blockIdx.y == 0: U panel tile (p, t) <- Linv * tile
//              blockIdx.y == 1: L panel tile (t, p) <- tile * Uinv      (t > p)
__global__ void __launch_bounds__(256) trsm_kernel(MfDev mf, int level_begin, int p) {
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.z]];
    const int m = f.ld, nt = (m + NB - 1) / NB;
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

// trailing update: tile(ti,tj) -= L(ti,p)[:, :kb] * U(p,tj)[:kb, :]   (ti, tj > p).
// (Fusing the next diagonal tile's LU into this kernel was tried: the extra
// registers of the tile LU cut the occupancy of every update workgroup and made
// the factorisation slower overall -- profiles/r01_*.)
__global__ void __launch_bounds__(256) update_kernel(MfDev mf, int level_begin, int p) {
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.z]];
    const int m = f.ld, nt = (m + NB - 1) / NB;
    const int ti = p + 1 + blockIdx.y, tj = p + 1 + blockIdx.x;
    if (ti >= nt || tj >= nt) return;
    // the (augmentation x augmentation) corner is never used
    if (ti * NB >= f.m && tj * NB >= f.m) return;
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

// forward, step 1: t = w_own + (children's update entries mapped to own rows)
__global__ void __launch_bounds__(256) fwd_gather_kernel(MfDev mf, int level_begin) {
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.y]];
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= f.k) return;
    const int32_t* gp = mf.gat_ptr + f.gat_off;
    double v = mf.work[f.own_start + r];
    for (int32_t s = gp[r]; s < gp[r + 1]; ++s) v += mf.upd_store[mf.gat_src[s]];
    mf.work[f.own_start + r] = v;
}

// forward, step 2: [z; upd] = [L11^-1; -L21 L11^-1] t (+ gathered on boundary rows);
// one wave per row, rows of all fronts of the level in one launch
__global__ void __launch_bounds__(256) fwd_mv_kernel(MfDev mf, int level_begin) {
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.y]];
    const int m = f.m, k = f.k;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= m) return;
    const double* row = mf.front_store + f.off + (int64_t)r * f.ld + m;
    const double* t = mf.work + f.own_start;
    const int cend = r < k ? r + 1 : k;  // L11^-1 is lower triangular
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int c = lane;
    for (; c + 192 < cend; c += 256) {  // 4 independent loads in flight per lane
        a0 += row[c] * t[c];
        a1 += row[c + 64] * t[c + 64];
        a2 += row[c + 128] * t[c + 128];
        a3 += row[c + 192] * t[c + 192];
    }
    for (; c < cend; c += 64) a0 += row[c] * t[c];
    double acc = wave_sum((a0 + a1) + (a2 + a3));
    if (lane == 0) {
        if (r < k) {
            mf.work2[f.own_start + r] = acc;
        } else {
            const int32_t* gp = mf.gat_ptr + f.gat_off;
            double v = acc;
            for (int32_t s = gp[r]; s < gp[r + 1]; ++s) v += mf.upd_store[mf.gat_src[s]];
            mf.upd_store[f.upd_off + r - k] = v;
        }
    }
}

// backward: x_own = [U11^-1, -U11^-1 U12] [z; x_bnd]; one wave per row
__global__ void __launch_bounds__(256) bwd_mv_kernel(MfDev mf, int level_begin) {
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.y]];
    const int m = f.m, k = f.k;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= k) return;
    const double* row = mf.front_store + f.off + (int64_t)(m + r) * f.ld;
    const double* z = mf.work2 + f.own_start;
    const int32_t* bi = mf.bnd_idx + f.bnd_off;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int c = r + lane;  // U11^-1 is upper triangular
    for (; c + 192 < k; c += 256) {
        a0 += row[c] * z[c];
        a1 += row[c + 64] * z[c + 64];
        a2 += row[c + 128] * z[c + 128];
        a3 += row[c + 192] * z[c + 192];
    }
    for (; c < k; c += 64) a0 += row[c] * z[c];
    c = k + lane;
    for (; c + 192 < m; c += 256) {
        a0 += row[c] * mf.work[bi[c - k]];
        a1 += row[c + 64] * mf.work[bi[c + 64 - k]];
        a2 += row[c + 128] * mf.work[bi[c + 128 - k]];
        a3 += row[c + 192] * mf.work[bi[c + 192 - k]];
    }
    for (; c < m; c += 64) a0 += row[c] * mf.work[bi[c - k]];
    double acc = wave_sum((a0 + a1) + (a2 + a3));
    if (lane == 0) mf.work[f.own_start + r] = acc;
}

}  // namespace mfk
}  // namespace sanm_hip
