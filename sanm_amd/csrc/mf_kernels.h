// HIP kernels of the multifrontal LU (see multifrontal.h for the method).
// Included by backend_hip.hip only.
//
// All fronts of one tree level are processed by the same launches; blocks that
// fall outside a front's extent exit at once.  Tiles are NB x NB (NB = 32),
// one 256-thread workgroup per tile, operands staged through LDS.
//
// Front layout (mf_types.h): rows / columns ordered [P pivot k | A augmentation k |
// B boundary b], leading dimension ld = 2k + b.  Per level:
//   1. panel loop on the leading 2k x 2k block (diag / trsm / update kernels):
//      LU of F[P,P] with the identity blocks turning into L11^-1 and U11^-1;
//   2. gemm1_kernel:  tmpU = L11^-1 F[P,B],  tmpL = F[B,P] U11^-1          (K = k)
//   3. gemm2_kernel:  F[B,B] -= tmpL tmpU,  F[B,A] = -tmpL L11^-1,  F[A,B] = -U11^-1 tmpU
// so the Schur complement is read and written once instead of once per panel.
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

// identity blocks of the augmentation: F[r, k + r] = F[k + r, r] = 1 for r < k
__global__ void aug_identity_kernel(MfDev mf) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= mf.n) return;
    const MfFrontDev f = mf.fronts[mf.own_front[i]];
    const int r = i - f.own_start;
    double* F = mf.front_store + f.off;
    F[(int64_t)r * f.ld + f.k + r] = 1.0;
    F[(int64_t)(f.k + r) * f.ld + r] = 1.0;
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
    double v = mf.front_store[c.off + (int64_t)(2 * c.k + i) * c.ld + 2 * c.k + j];
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
    const int ld = f.ld, m = 2 * f.k, r0 = p * NB;  // m: extent of the pivot + augmentation block
    const int kb = min(NB, f.k - r0);
    __shared__ double T[NB][TPAD], LI[NB][TPAD], UI[NB][TPAD];
    double* F = mf.front_store + f.off;
    const int tid = threadIdx.x, tc = tid % NB, tr = tid / NB;
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s, gr = r0 + r, gc = r0 + tc;
        T[r][tc] = (gr < m && gc < m) ? F[(int64_t)gr * ld + gc] : (r == tc ? 1.0 : 0.0);
    }
    __syncthreads();
    tile_factor(T, LI, UI, kb, tid, mf.dinv_store + f.dinv_off + (int64_t)p * 2 * NB * NB, mf.status);
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s, gr = r0 + r, gc = r0 + tc;
        if (gr < m && gc < m) F[(int64_t)gr * ld + gc] = T[r][tc];
    }
}

// panel tiles: blockIdx.y == 0: U panel tile (p, t) <- Linv * tile
//              blockIdx.y == 1: L panel tile (t, p) <- tile * Uinv      (t > p)
__global__ void __launch_bounds__(256) trsm_kernel(MfDev mf, int level_begin, int p) {
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.z]];
    const int ld = f.ld, m = 2 * f.k, nt = (m + NB - 1) / NB;
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
        A[r][tc] = (gr < m && gc < m) ? F[(int64_t)gr * ld + gc] : 0.0;
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
        if (gr < m && gc < m) F[(int64_t)gr * ld + gc] = acc;
    }
}

// trailing update: tile(ti,tj) -= L(ti,p)[:, :kb] * U(p,tj)[:kb, :]   (ti, tj > p).
// Look-ahead: the workgroup that owns the next diagonal tile (p+1,p+1) factors it
// right after updating it, so panels p >= 1 need no separate diagonal launch and
// that short sequential LU hides behind the other tiles of the same launch.
// (tile_factor keeps its inverse columns in LDS: with them in registers the
// fused kernel lost occupancy and the factorisation got slower; a second stream
// for the diagonal tile was tried as well -- the cross-stream events cost as
// much as they hid.)
__global__ void __launch_bounds__(256) update_kernel(MfDev mf, int level_begin, int p) {
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.z]];
    const int ld = f.ld, m = 2 * f.k, nt = (m + NB - 1) / NB;
    const int ti = p + 1 + blockIdx.y, tj = p + 1 + blockIdx.x;
    if (ti >= nt || tj >= nt) return;
    // the (augmentation x augmentation) corner is never used
    if (ti * NB >= f.k && tj * NB >= f.k) return;
    const int kb = min(NB, f.k - p * NB);
    __shared__ double L[NB][TPAD], U[NB][TPAD], T[NB][TPAD];
    double* F = mf.front_store + f.off;
    const int tid = threadIdx.x, tc = tid % NB, tr = tid / NB;
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s;
        int gr = ti * NB + r, gc = p * NB + tc;
        L[r][tc] = (gr < m && tc < kb) ? F[(int64_t)gr * ld + gc] : 0.0;
        gr = p * NB + r;
        gc = tj * NB + tc;
        U[r][tc] = (r < kb && gc < m) ? F[(int64_t)gr * ld + gc] : 0.0;
    }
    __syncthreads();
    const bool next_diag = (ti == tj) && (ti == p + 1) && ((p + 1) * NB < f.k);  // workgroup-uniform
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s, gr = ti * NB + r, gc = tj * NB + tc;
        double v = (r == tc) ? 1.0 : 0.0;  // identity padding outside the front
        if (gr < m && gc < m) {
            double acc = 0;
#pragma unroll 8
            for (int q = 0; q < NB; ++q) acc += L[r][q] * U[q][tc];
            v = F[(int64_t)gr * ld + gc] - acc;
            if (!next_diag) F[(int64_t)gr * ld + gc] = v;
        }
        if (next_diag) T[r][tc] = v;
    }
    if (!next_diag) return;
    __syncthreads();
    const int kb1 = min(NB, f.k - (p + 1) * NB);
    tile_factor(T, L, U, kb1, tid, mf.dinv_store + f.dinv_off + (int64_t)(p + 1) * 2 * NB * NB, mf.status);
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s, gr = ti * NB + r, gc = tj * NB + tc;
        if (gr < m && gc < m) F[(int64_t)gr * ld + gc] = T[r][tc];
    }
}

// C tile (64x64) of a product of two strided matrices: 256 threads, each a 4x4
// register block, K in steps of 16 through LDS.  Element (i,j) of an operand is
// p[i*ld + j] inside (rows, cols), else 0.  K range [k0, k1) in elements.
struct MatView {
    const double* p;
    int ld, rows, cols;
};
constexpr int GT = 64;   // GEMM tile edge
constexpr int GK = 16;   // GEMM K step
__device__ __forceinline__ void gemm_tile(const MatView& A, const MatView& B, int ti, int tj, int k0,
                                          int k1, double (*As)[GT + 1], double (*Bs)[GT + 4],
                                          double acc[4][4]) {
    const int tid = threadIdx.x, tx = tid % 16, ty = tid / 16;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0;
    for (int kk = k0; kk < k1; kk += GK) {
        __syncthreads();
        // A tile 64 x 16 stored transposed As[e][row]; B tile 16 x 64 as Bs[e][col]
        for (int s = 0; s < 4; ++s) {
            int idx = tid + 256 * s;       // 0..1023
            int ar = idx / GK, ae = idx % GK;  // consecutive threads along K: contiguous in a row of A
            int gr = ti * GT + ar, gc = kk + ae;
            As[ae][ar] = (gr < A.rows && gc < A.cols && gc < k1) ? A.p[(int64_t)gr * A.ld + gc] : 0.0;
            int be = idx / GT, bc = idx % GT;  // consecutive threads along the columns of B
            gr = kk + be;
            gc = tj * GT + bc;
            Bs[be][bc] = (gr < B.rows && gr < k1 && gc < B.cols) ? B.p[(int64_t)gr * B.ld + gc] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < GK; ++e) {
            double a[4], bq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[e][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) bq[j] = Bs[e][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * bq[j];
        }
    }
}

// step 2:  which = 0: tmpU (k x b) = L11^-1 F[P,B]     (L11^-1 lower: K tiles 0..ti)
//          which = 1: tmpL (b x k) = F[B,P] U11^-1     (U11^-1 upper: K tiles 0..tj)
__global__ void __launch_bounds__(256) gemm1_kernel(MfDev mf, int level_begin) {
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.z / 2]];
    const int which = blockIdx.z & 1;
    const int k = f.k, b = f.m - f.k, ld = f.ld;
    const int rows = which ? b : k, cols = which ? k : b;
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (ti * GT >= rows || tj * GT >= cols) return;
    __shared__ double As[GK][GT + 1], Bs[GK][GT + 4];
    const double* F = mf.front_store + f.off;
    double* tmp = mf.tmp_store + f.tmp_off;
    MatView A, B;
    int k1;
    if (which == 0) {
        A = {F + k, ld, k, k};                       // F[P,A] = L11^-1 (lower)
        B = {F + 2 * k, ld, k, b};                   // F[P,B]
        k1 = min(k, (ti + 1) * GT);
    } else {
        A = {F + (int64_t)2 * k * ld, ld, b, k};     // F[B,P]
        B = {F + (int64_t)k * ld, ld, k, k};         // F[A,P] = U11^-1 (upper)
        k1 = min(k, (tj + 1) * GT);
    }
    double acc[4][4];
    gemm_tile(A, B, ti, tj, 0, k1, As, Bs, acc);
    double* C = which ? tmp + (int64_t)k * b : tmp;  // tmpU: ld b ; tmpL: ld k
    const int cld = which ? k : b;
    const int tid = threadIdx.x, tx = tid % 16, ty = tid / 16;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            int r = ti * GT + ty * 4 + i, c = tj * GT + tx * 4 + j;
            if (r < rows && c < cols) C[(int64_t)r * cld + c] = acc[i][j];
        }
}

// step 3:  which = 0: F[B,B] -= tmpL tmpU                        (b x b, K = k)
//          which = 1: F[B,A]  = -tmpL L11^-1   (b x k; L11^-1 lower: K tiles tj..)
//          which = 2: F[A,B]  = -U11^-1 tmpU   (k x b; U11^-1 upper: K tiles ti..)
__global__ void __launch_bounds__(256) gemm2_kernel(MfDev mf, int level_begin) {
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.z / 3]];
    const int which = blockIdx.z % 3;
    const int k = f.k, b = f.m - f.k, ld = f.ld;
    const int rows = which == 2 ? k : b, cols = which == 1 ? k : b;
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (ti * GT >= rows || tj * GT >= cols) return;
    __shared__ double As[GK][GT + 1], Bs[GK][GT + 4];
    double* F = mf.front_store + f.off;
    const double* tmpU = mf.tmp_store + f.tmp_off;
    const double* tmpL = tmpU + (int64_t)k * b;
    MatView A, B;
    int k0 = 0;
    double* C;
    if (which == 0) {
        A = {tmpL, k, b, k};
        B = {tmpU, b, k, b};
        C = F + (int64_t)2 * k * ld + 2 * k;
    } else if (which == 1) {
        A = {tmpL, k, b, k};
        B = {F + k, ld, k, k};  // L11^-1 (lower): rows >= column
        k0 = tj * GT;
        C = F + (int64_t)2 * k * ld + k;
    } else {
        A = {F + (int64_t)k * ld, ld, k, k};  // U11^-1 (upper): columns >= row
        B = {tmpU, b, k, b};
        k0 = ti * GT;
        C = F + (int64_t)k * ld + 2 * k;
    }
    double acc[4][4];
    gemm_tile(A, B, ti, tj, k0, k, As, Bs, acc);
    const int tid = threadIdx.x, tx = tid % 16, ty = tid / 16;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            int r = ti * GT + ty * 4 + i, c = tj * GT + tx * 4 + j;
            if (r < rows && c < cols) {
                double* dst = C + (int64_t)r * ld + c;
                *dst = (which == 0) ? *dst - acc[i][j] : -acc[i][j];
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
    // row r of [L11^-1 ; -L21 L11^-1] = columns A of physical row r (own) or 2k + (r-k)
    const int pr = r < k ? r : r + k;
    const double* row = mf.front_store + f.off + (int64_t)pr * f.ld + k;
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
    // row r of [U11^-1 , -U11^-1 U12] = physical row k + r: columns P then columns B
    const double* row = mf.front_store + f.off + (int64_t)(k + r) * f.ld;
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
    const double* rowb = row + 2 * k;  // boundary columns
    const int nbnd = m - k;
    c = lane;
    for (; c + 192 < nbnd; c += 256) {
        a0 += rowb[c] * mf.work[bi[c]];
        a1 += rowb[c + 64] * mf.work[bi[c + 64]];
        a2 += rowb[c + 128] * mf.work[bi[c + 128]];
        a3 += rowb[c + 192] * mf.work[bi[c + 192]];
    }
    for (; c < nbnd; c += 64) a0 += rowb[c] * mf.work[bi[c]];
    double acc = wave_sum((a0 + a1) + (a2 + a3));
    if (lane == 0) mf.work[f.own_start + r] = acc;
}

}  // namespace mfk
}  // namespace sanm_hip
