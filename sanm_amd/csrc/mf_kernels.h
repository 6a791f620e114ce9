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
// on exit T holds the packed factors (synchronised).  W is scratch.
//
// The elimination is a chain of kb barrier-separated rank-1 updates; the two
// triangular inverses are NOT built by another 32-step substitution but by
// recursive blocking: the four 8x8 diagonal blocks by substitution (one lane per
// column, 8 short steps), then two merge levels
//   [[A,0],[C,B]]^-1 = [[A^-1,0],[-B^-1 C A^-1, B^-1]]   (and its transpose form for U)
// as small matrix products over all 256 threads -- five barriers instead of 32.
__device__ __forceinline__ double tf_l(const double (*T)[TPAD], int kb, int r, int c) {
    return (c < r && c < kb) ? T[r][c] : (r == c ? 1.0 : 0.0);
}
__device__ __forceinline__ double tf_u(const double (*T)[TPAD], int kb, int r, int c) {
    if (r >= kb) return r == c ? 1.0 : 0.0;
    if (c < r) return 0.0;
    const double v = T[r][c];
    return (c == r && !(fabs(v) > 1e-290)) ? 1.0 : v;
}
// one merge level: blocks of size H at offsets (o, o+H) of problem `prob` (0: L, 1: U), output (i, j)
template <int H>
__device__ __forceinline__ void tf_merge_mid(const double (*T)[TPAD], const double (*LI)[TPAD],
                                             const double (*UI)[TPAD], double (*W)[TPAD], int kb, int prob,
                                             int o, int i, int j) {
    double acc = 0;
    if (prob == 0) {  // W = C * A^-1,  C = L[o+H.., o..]
#pragma unroll
        for (int q = 0; q < H; ++q) acc += tf_l(T, kb, o + H + i, o + q) * LI[o + q][o + j];
        W[o + H + i][o + j] = acc;
    } else {  // W = A^-1 * C,  C = U[o.., o+H..]
#pragma unroll
        for (int q = 0; q < H; ++q) acc += UI[o + i][o + q] * tf_u(T, kb, o + q, o + H + j);
        W[o + i][o + H + j] = acc;
    }
}
template <int H>
__device__ __forceinline__ void tf_merge_fin(double (*LI)[TPAD], double (*UI)[TPAD], const double (*W)[TPAD],
                                             int prob, int o, int i, int j) {
    double acc = 0;
    if (prob == 0) {  // X21 = -B^-1 * W
#pragma unroll
        for (int q = 0; q < H; ++q) acc += LI[o + H + i][o + H + q] * W[o + H + q][o + j];
        LI[o + H + i][o + j] = -acc;
    } else {  // X12 = -W * B^-1
#pragma unroll
        for (int q = 0; q < H; ++q) acc += W[o + i][o + H + q] * UI[o + H + q][o + H + j];
        UI[o + i][o + H + j] = -acc;
    }
}

__device__ __forceinline__ void tile_factor(double (*T)[TPAD], double (*LI)[TPAD], double (*UI)[TPAD],
                                            double (*W)[TPAD], int kb, int tid, double* D, int32_t* status) {
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
                const int r = tr + 8 * s;
                if (r > j) T[r][tc] -= (T[r][j] * inv) * u;
            }
        }
        __syncthreads();
    }
    // scale the L columns; clear the inverses
    {
        const double d = (tc < kb) ? T[tc][tc] : 1.0;
        const double dinv = 1.0 / ((fabs(d) > 1e-290) ? d : 1.0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = tr + 8 * s;
            if (tc < kb && r > tc) T[r][tc] *= dinv;
            LI[r][tc] = 0.0;
            UI[r][tc] = 0.0;
        }
    }
    __syncthreads();
    // 8x8 diagonal blocks: lanes 0..31 one column of Lext^-1 each, lanes 32..63 one of Uext^-1
    if (tid < 64) {
        const int o = ((tid & 31) / 8) * 8, c = tid & 7;
        double x[8];
        if (tid < 32) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                double v = (r == c) ? 1.0 : 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q >= c && q < r) v -= tf_l(T, kb, o + r, o + q) * x[q];
                x[r] = (r < c) ? 0.0 : v;
                LI[o + r][o + c] = x[r];
            }
        } else {
#pragma unroll
            for (int r = 7; r >= 0; --r) {
                double v = (r == c) ? 1.0 : 0.0;
#pragma unroll
                for (int q = 7; q >= 0; --q)
                    if (q > r && q <= c) v -= tf_u(T, kb, o + r, o + q) * x[q];
                x[r] = (r > c) ? 0.0 : v / tf_u(T, kb, o + r, o + r);
                UI[o + r][o + c] = x[r];
            }
        }
    }
    __syncthreads();
    {  // 8 -> 16: four problems (L / U) x (pair 0 / 1), 64 outputs each
        const int prob = (tid >> 6) & 1, o = (tid >> 7) * 16, i = (tid & 63) >> 3, j = tid & 7;
        tf_merge_mid<8>(T, LI, UI, W, kb, prob, o, i, j);
        __syncthreads();
        tf_merge_fin<8>(LI, UI, W, prob, o, i, j);
        __syncthreads();
    }
    {  // 16 -> 32: two problems, 256 outputs each: every thread one output of both
        const int i = tid >> 4, j = tid & 15;
        tf_merge_mid<16>(T, LI, UI, W, kb, 0, 0, i, j);
        tf_merge_mid<16>(T, LI, UI, W, kb, 1, 0, i, j);
        __syncthreads();
        tf_merge_fin<16>(LI, UI, W, 0, 0, i, j);
        tf_merge_fin<16>(LI, UI, W, 1, 0, i, j);
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int r = tr + 8 * s;
        D[r * NB + tc] = LI[r][tc];
        D[NB * NB + r * NB + tc] = UI[r][tc];
    }
}

// diagonal tile of panel p of every front of a level
__global__ void __launch_bounds__(256) diag_kernel(MfDev mf, int level_begin, int p) {
    const MfFrontDev f = mf.fronts[mf.level_fronts[level_begin + blockIdx.x]];
    const int ld = f.ld, m = 2 * f.k, r0 = p * NB;  // m: extent of the pivot + augmentation block
    const int kb = min(NB, f.k - r0);
    __shared__ double T[NB][TPAD], LI[NB][TPAD], UI[NB][TPAD], W[NB][TPAD];
    double* F = mf.front_store + f.off;
    const int tid = threadIdx.x, tc = tid % NB, tr = tid / NB;
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s, gr = r0 + r, gc = r0 + tc;
        T[r][tc] = (gr < m && gc < m) ? F[(int64_t)gr * ld + gc] : (r == tc ? 1.0 : 0.0);
    }
    __syncthreads();
    tile_factor(T, LI, UI, W, kb, tid, mf.dinv_store + f.dinv_off + (int64_t)p * 2 * NB * NB, mf.status);
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
    __shared__ double L[NB][TPAD], U[NB][TPAD], T[NB][TPAD], W[NB][TPAD];
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
    tile_factor(T, L, U, W, kb1, tid, mf.dinv_store + f.dinv_off + (int64_t)(p + 1) * 2 * NB * NB, mf.status);
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

// Solve sweeps: one launch per level and direction.  With the augmented fronts a
// front's forward step is one mat-vec [z; upd] = [L11^-1; -L21 L11^-1] t and its
// backward step x_own = [U11^-1, -U11^-1 U12] [z; x_bnd].  A workgroup stages the
// front's input vector in LDS once (adding the children's inbox slots on the
// way, so no gather list is walked and no separate gather launch is needed) and
// its 4 waves then take R rows each, one 64-lane dot product per row.
//
// These launches last a few microseconds and every dependent memory round trip
// costs about one, so the kernels are written for a short dependency chain:
// descriptor -> {row chunks, vector, inbox, destination slots} all in flight
// together -> barrier -> FMAs -> wave reduction -> store.  The first U 64-column
// chunks of every row are loaded to registers by unconditional (index-clamped)
// loads BEFORE the staging barrier; U is chosen per level to cover its widest
// front, so the loop over further chunks only runs for fronts beyond 64*U pivots.

template <int R, int U>
struct RowChunks {
    double a[U][R];
};

// issue the loads of chunks c0 + 64u (u < U) of R rows; entries outside [cbeg, cend) read as 0
template <int R, int U>
__device__ __forceinline__ void rows_preload(RowChunks<R, U>& rc, const double* const (&rowp)[R],
                                             const int (&cbeg)[R], const int (&cend)[R], int c0) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int c = c0 + 64 * u;
            const bool ok = c >= cbeg[q] && c < cend[q];
            const double v = rowp[q][ok ? c : 0];
            rc.a[u][q] = ok ? v : 0.0;
        }
}

// acc[q] += sum_c row_q[c] * v[c] over c = c0, c0 + 64, ... inside [cbeg[q], cend[q]); v in LDS,
// v[csafe] is an initialised entry (read, times zero, by lanes outside the range)
template <int R, int U>
__device__ __forceinline__ void rows_consume(const RowChunks<R, U>& rc, const double* const (&rowp)[R],
                                             const int (&cbeg)[R], const int (&cend)[R], int cmax, int c0,
                                             const double* v, int csafe, double (&acc)[R]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = c0 + 64 * u;
        const double tv = v[c < cmax ? c : csafe];
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] += rc.a[u][q] * tv;
    }
    for (int c = c0 + 64 * U; c < cmax; c += 64) {
        const double tv = v[c];
#pragma unroll
        for (int q = 0; q < R; ++q)
            if (c >= cbeg[q] && c < cend[q]) acc[q] += rowp[q][c] * tv;
    }
}

template <int R, int U>
__global__ void __launch_bounds__(256) fwd_level_kernel(MfDev mf, int level_begin) {
    const MfFrontDev f = mf.lfronts[level_begin + blockIdx.y];
    const int m = f.m, k = f.k;
    const int rb = blockIdx.x * (4 * R);
    if (rb >= m) return;
    extern __shared__ double vs[];  // t = w_own + children's contributions (k entries)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double* inbox = mf.inbox_store + f.inbox_off;
    // rows of this wave; boundary rows also pick up the children's entries and forward the sum to the parent
    int r[R], cbeg[R], cend[R], dst[R];
    const double* rowp[R];
    double pre[R], acc[R];
    int cmax = 0;
#pragma unroll
    for (int q = 0; q < R; ++q) {
        r[q] = rb + wv * R + q;
        const bool live = r[q] < m;
        const int pr = r[q] < k ? r[q] : r[q] + k;  // physical row: columns A hold [L11^-1 ; -L21 L11^-1]
        rowp[q] = mf.front_store + f.off + (int64_t)(live ? pr : 0) * f.ld + k;
        cbeg[q] = 0;
        cend[q] = !live ? 0 : (r[q] < k ? r[q] + 1 : k);  // L11^-1 is lower triangular
        cmax = max(cmax, cend[q]);
        acc[q] = 0;
    }
    RowChunks<R, U> rc;
    rows_preload<R, U>(rc, rowp, cbeg, cend, lane);
#pragma unroll
    for (int q = 0; q < R; ++q) {
        pre[q] = 0;
        dst[q] = -1;
        if (r[q] >= k && r[q] < m) {
            dst[q] = mf.upd_dst[f.bnd_off + r[q] - k];
            for (int j = 0; j < f.nch; ++j) pre[q] += inbox[(int64_t)j * m + r[q]];
        }
    }
    const int kneed = min(k, rb + 4 * R);  // own rows read t[0..r] only
    for (int c = tid; c < kneed; c += 256) {
        double v = mf.work[f.own_start + c];
        for (int j = 0; j < f.nch; ++j) v += inbox[(int64_t)j * m + c];
        vs[c] = v;
    }
    __syncthreads();
    rows_consume<R, U>(rc, rowp, cbeg, cend, cmax, lane, vs, 0, acc);
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const double v = wave_sum(acc[q]);
        if (lane == 0 && r[q] < m) {
            if (r[q] < k)
                mf.work2[f.own_start + r[q]] = v;
            else
                mf.inbox_store[dst[q]] = v + pre[q];
        }
    }
}

template <int R, int U>
__global__ void __launch_bounds__(256) bwd_level_kernel(MfDev mf, int level_begin) {
    const MfFrontDev f = mf.lfronts[level_begin + blockIdx.y];
    const int m = f.m, k = f.k;
    const int rb = blockIdx.x * (4 * R);
    if (rb >= k) return;
    extern __shared__ double vs[];  // [z (k) ; x_bnd (m-k)]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // row r of [U11^-1 , -U11^-1 U12] is physical row k + r: columns P (upper triangular: from r on) and,
    // one augmentation block further right, columns B; addressed as one virtual row over [r, m) whose
    // entries c >= k sit at physical column c + k, matching the layout of vs.
    int r[R], cbeg[R], cend[R];
    const double* rowp[R];
    double acc[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        r[q] = rb + wv * R + q;
        const bool live = r[q] < k;
        rowp[q] = mf.front_store + f.off + (int64_t)(k + (live ? r[q] : 0)) * f.ld;
        cbeg[q] = r[q];
        cend[q] = live ? m : 0;
        acc[q] = 0;
    }
    const int c0 = rb + wv * R + lane;
    double a[U][R];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int c = c0 + 64 * u;
            const bool ok = c >= cbeg[q] && c < cend[q];
            const double v = rowp[q][ok ? (c < k ? c : c + k) : 0];
            a[u][q] = ok ? v : 0.0;
        }
    const int32_t* bi = mf.bnd_idx + f.bnd_off;
    for (int c = rb + tid; c < m; c += 256) vs[c] = c < k ? mf.work2[f.own_start + c] : mf.work[bi[c - k]];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = c0 + 64 * u;
        const double tv = vs[c < m ? c : rb];
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] += a[u][q] * tv;
    }
    for (int c = c0 + 64 * U; c < m; c += 64) {
        const double tv = vs[c];
#pragma unroll
        for (int q = 0; q < R; ++q)
            if (c >= cbeg[q] && c < cend[q]) acc[q] += rowp[q][c < k ? c : c + k] * tv;
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const double v = wave_sum(acc[q]);
        if (lane == 0 && r[q] < k) mf.work[f.own_start + r[q]] = v;
    }
}

}  // namespace mfk
}  // namespace sanm_hip
