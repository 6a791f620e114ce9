// HIP kernels of the multifrontal LU (see multifrontal.h for the method).
// Included by backend_hip.hip only.
//
// All fronts of one tree level are processed by the same launches; blocks that
// fall outside a front's extent exit at once.  Tiles are NB x NB (NB = 32),
// one 256-thread workgroup per tile, operands staged through LDS.
//
// Front layout (mf_types.h): rows / columns ordered [P pivot k | A augmentation k |
// B boundary b], leading dimension ld = 2k + b.  Per level:
//   1. panel loop on the leading 2k x 2k block: diag_kernel for the first diagonal tile, then one
//      update_kernel launch per 32-wide panel (panel solves by substitution, tile update, look-ahead LU
//      of the next diagonal tile) and panel_finalize_kernel for the result tiles; fronts of 500+ pivots
//      defer most of the trailing matrix to block_gemm_kernel (second blocking level).  This is the LU
//      of F[P,P] with the identity blocks turning into L11^-1 and U11^-1;
//   2. gemm1_kernel:  tmpU = L11^-1 F[P,B],  tmpL = F[B,P] U11^-1          (K = k)
//   3. gemm2_kernel:  F[B,B] -= tmpL tmpU,  F[B,A] = -tmpL L11^-1,  F[A,B] = -U11^-1 tmpU
// so the Schur complement is read and written once instead of once per panel.
#pragma once
#include <climits>
#include <type_traits>
#include <hip/hip_runtime.h>

#include "mf_types.h"

namespace sanm_hip {
namespace mfk {

constexpr int NB = MF_NB;
constexpr int TPAD = NB + 1;  // LDS row stride (odd: no bank conflicts on column access)
typedef double mfma_f64x4 __attribute__((ext_vector_type(4)));

// The factor kernels take their pointers as leading scalar arguments (kernarg preload, see SolveArgs below); the
// descriptor pointer arrives already advanced to the level's first front.
struct FactorArgs {
    const MfFrontDev* lfronts;
    double* front_store;
    double* tmp_store;
    int32_t* status;
    const double* piv_amax;
};
#define MF_FACTOR_PARAMS                                                                                  \
    const MfFrontDev *__restrict__ lfronts_, double *front_store_, double *tmp_store_, int32_t *status_, \
        const double *__restrict__ piv_amax_
#define MF_FACTOR_INIT                                                           \
    const FactorArgs mf{lfronts_, front_store_, tmp_store_, status_, piv_amax_}; \
    constexpr int level_begin = 0;
#define MF_FACTOR_ARGS(mf, lb) (mf).lfronts + (lb), (mf).front_store, (mf).tmp_store, (mf).status, (mf).piv_amax

__global__ void scatter_kernel(int64_t nnz, const int64_t* __restrict__ a_dst,
                               const double* __restrict__ val, double* __restrict__ store) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nnz && a_dst[p] >= 0) store[a_dst[p]] = val[p];  // (< 0: an entry of another rank's front)
}

// a_dst of every entry of A (mf_types.h, mf_scatter_slot), once per solver: a wavefront per row of A
__global__ void __launch_bounds__(256) scatter_map_kernel(MfDev mf, const uint32_t* __restrict__ rowptr,
                                                          const uint32_t* __restrict__ col) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= mf.n) return;
    const int32_t pi = mf.perm[i];
    for (uint32_t p = rowptr[i] + (threadIdx.x & 63); p < rowptr[i + 1]; p += 64)
        mf.a_dst[p] = mf_scatter_slot(mf.fronts, mf.own_front, mf.bnd_idx, pi, mf.perm[col[p]], mf.front_here);
}

// identity blocks of the augmentation: F[r, k + r] = F[k + r, r] = 1 for r < k
__global__ void aug_identity_kernel(MfDev mf) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= mf.n) return;
    if (mf.front_here && !mf.front_here[mf.own_front[i]]) return;  // (another rank's front)
    const MfFrontDev f = mf.fronts[mf.own_front[i]];
    const int r = i - f.own_start;
    double* F = mf.front_store + f.off;
    F[(int64_t)r * f.ld + f.k + r] = 1.0;
    F[(int64_t)(f.k + r) * f.ld + r] = 1.0;
}

constexpr int EA_ROWS = 4;  // rows of a Schur complement per workgroup of the extend-add kernels

// Prologue of a factorisation: the parts of every front that are accumulated into or read before they are written --
// the rows P in full (matrix entries, children's contributions, the identity block of the augmentation), the columns P
// and A of the rows A, the columns P of the rows B.  F[A,B] and F[B,A] are outputs of the GEMM passes, F[B,B] is
// assigned by round 0 of the extend-add (schur_gather_kernel) or never read (fronts without children): not touched --
// at 338 k tets 1.8 of the 4.1 GB a memset of the whole front storage wrote.  One workgroup per MF_ZERO_ROWS rows.
__global__ void __launch_bounds__(256) zero_kernel(const MfFrontDev* __restrict__ fronts, double* __restrict__ store,
                                                   const int32_t* __restrict__ blocks) {
    const MfFrontDev f = fronts[blocks[2 * blockIdx.x]];
    const int r0 = blocks[2 * blockIdx.x + 1], k = f.k, ld = f.ld;
    double* F = store + f.off;
    for (int r = r0 + (threadIdx.x >> 6); r < min(r0 + MF_ZERO_ROWS, ld); r += 4) {
        const int len = r < k ? ld : (r < 2 * k ? 2 * k : k);
        double* row = F + (int64_t)r * ld;
        for (int c = threadIdx.x & 63; c < len; c += 64) row[c] = 0.0;
    }
}

// Round 0 of the extend-add, the F[B,B] block of the parent: parent[2k + i, 2k + j] = schur of its first child at
// (inv[i], inv[j]), 0 where the child has no such row or column (MfSchedule::ea_inv) -- every entry of the block is
// written, so it needs no zero-fill and no read.  0.0 + v: the bits an addition to a zeroed block gave.
__global__ void __launch_bounds__(256) schur_gather_kernel(const MfFrontDev* __restrict__ fronts_, double* front_store_,
                                                           const int32_t* __restrict__ inv_,
                                                           const int32_t* __restrict__ children, int rows_per_wg) {
    const MfFrontDev c = fronts_[children[blockIdx.y]];
    const MfFrontDev p = fronts_[c.parent];
    const int bp = p.m - p.k;
    const int ib = blockIdx.x * rows_per_wg;
    if (ib >= bp) return;
    const int32_t* __restrict__ inv = inv_ + p.bnd_off;
    const double* src = front_store_ + c.off + (int64_t)(2 * c.k) * c.ld + 2 * c.k;
    double* dst = front_store_ + p.off + (int64_t)(2 * p.k) * p.ld + 2 * p.k;
    // (rows_per_wg: EA_ROWS unless an experiment says otherwise, backend_hip.hip)
    for (int i0 = ib; i0 < min(ib + rows_per_wg, bp); i0 += EA_ROWS) {
        int ci[EA_ROWS];
#pragma unroll
        for (int q = 0; q < EA_ROWS; ++q) ci[q] = i0 + q < bp ? inv[i0 + q] : -1;
        for (int j = threadIdx.x; j < bp; j += 256) {
            const int cj = inv[j];
            double v[EA_ROWS];
#pragma unroll
            for (int q = 0; q < EA_ROWS; ++q) v[q] = (ci[q] >= 0 && cj >= 0) ? src[(int64_t)ci[q] * c.ld + cj] : 0.0;
#pragma unroll
            for (int q = 0; q < EA_ROWS; ++q)
                if (i0 + q < bp) dst[(int64_t)(i0 + q) * p.ld + j] = 0.0 + v[q];
        }
    }
}

// parent[rel[i], rel[j]] += child_schur[i, j]; one child per blockIdx.y, EA_ROWS rows of its Schur complement per
// workgroup, the threads along the columns.  (Round 2 gave every element a thread of its own: a 64-bit division per
// element, one load in flight per lane and a grid sized for the largest child of the round -- 225 GB/s on the
// 0.5 M-tet block, 16 of the 255 ms of a step.  Here a thread keeps EA_ROWS independent read-modify-writes per
// column in flight and rel[j] is read once per column for all of them.)
// skip_bb (round 0): the entries that land in the parent's F[B,B] are left to schur_gather_kernel -- rel is ascending,
// so those are the rows and columns from the first one at or beyond the parent's 2k on: a row block wholly there only
// walks the columns before it.
__global__ void __launch_bounds__(256) extend_add_kernel(const MfFrontDev* __restrict__ fronts_, double* front_store_,
                                                         const int32_t* __restrict__ rel_,
                                                         const int32_t* __restrict__ children, int skip_bb,
                                                         int rows_per_wg) {
    const MfFrontDev c = fronts_[children[blockIdx.y]];
    const int nb = c.m - c.k;
    const int ib = blockIdx.x * rows_per_wg;
    if (ib >= nb) return;
    const MfFrontDev p = fronts_[c.parent];
    const int32_t* __restrict__ rel = rel_ + c.rel_off;
    const double* src = front_store_ + c.off + (int64_t)(2 * c.k) * c.ld + 2 * c.k;
    double* dst = front_store_ + p.off;
    const int pb = skip_bb ? 2 * p.k : INT_MAX;  // first physical row / column of the parent's boundary block
    for (int i0 = ib; i0 < min(ib + rows_per_wg, nb); i0 += EA_ROWS) {
        int64_t drow[EA_ROWS];
        bool rb[EA_ROWS], all_b = true;
#pragma unroll
        for (int q = 0; q < EA_ROWS; ++q) {
            const int rr = rel[min(i0 + q, nb - 1)];
            drow[q] = (int64_t)rr * p.ld;
            rb[q] = rr >= pb;
            all_b = all_b && rb[q];
        }
        for (int j = threadIdx.x; j < nb; j += 256) {
            const int rj = rel[j];
            const bool cb = rj >= pb;
            if (cb && all_b) break;  // (ascending: every later column is in the boundary block as well)
            double v[EA_ROWS], d[EA_ROWS];
#pragma unroll
            for (int q = 0; q < EA_ROWS; ++q) {
                v[q] = src[(int64_t)min(i0 + q, nb - 1) * c.ld + j];
                d[q] = dst[drow[q] + rj];
            }
#pragma unroll
            for (int q = 0; q < EA_ROWS; ++q)
                if (i0 + q < nb && !(cb && rb[q])) dst[drow[q] + rj] = d[q] + v[q];
        }
    }
}

// LU of a diagonal tile (kb pivots, no pivoting) by ONE wavefront, in registers: lane r < 32 holds row r, the
// pivot row travels by v_readlane (the step index is a compile-time constant after unrolling), so a step is
// ~(2 readlanes + 1 fma) per remaining column and no LDS round trip or barrier -- 3.3 us per tile where the
// 256-thread version (elimination in LDS, one barrier per step) took 9.7 us, all of it on the critical path of
// the panel chain.  On entry T holds the tile (synchronised); on exit the packed factors: multipliers below the
// diagonal, U on and above it; the caller synchronises.  A partial last panel (kb < NB) leaves rows / columns
// kb.. of the tile updated by all kb pivots, i.e. holding the part of the trailing matrix that lives in this tile.
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// (a recursive template rather than a loop over the steps: the step index has to be a compile-time constant for
// the row to stay in registers, and the compiler does not fully unroll the 496-iteration loop nest on its own.
// The body is branch-free -- rows above the pivot take a zero multiplier, bad pivots are counted, not branched on
// -- so that the whole LU is one basic block, and the reciprocal of the NEXT pivot (v_rcp_f64 + two Newton steps,
// a chain of dependent instructions) is started as soon as its column is updated and overlaps with the
// remaining column updates of the current step.)
// (thr = MF_PIVOT_EPS * max|a_ij| > 0 normally; a pivot below it -- or not a number -- is replaced by +-thr and
// counted; an all-zero matrix has thr = 0 and takes 1.  The replaced value is handed back: it becomes the
// diagonal entry of U, where the substitutions read it.)
__device__ __forceinline__ double pivot_reciprocal(double& piv, int& nbad, bool used, double thr) {
    const bool bad = !(fabs(piv) > thr);
    nbad += bad && used;
    piv = bad ? (thr > 0 ? copysign(thr, piv) : 1.0) : piv;
    double r = __builtin_amdgcn_rcp(piv);
    r = __builtin_fma(__builtin_fma(-piv, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-piv, r, 1.0), r, r);
    return r;
}
// a[c] -= l * (row J's a[c]) for the 8 columns C0 .. C0 + 7.  Left to itself the compiler funnels every broadcast
// through ONE scalar register pair -- readlane, readlane, two wait states, multiply-add, and the next readlane
// waits for the multiply-add to have read the pair: 26 cycles per column.  The empty asm statement keeps the 16
// scalars of a chunk alive at once, so the 16 readlanes issue back to back and the 8 multiply-adds after them.
template <int J, int C0>
__device__ __forceinline__ void tile_lu_chunk8(double (&a)[NB], double l) {
    int lo[8], hi[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        lo[q] = __builtin_amdgcn_readlane(__double2loint(a[C0 + q]), J);
        hi[q] = __builtin_amdgcn_readlane(__double2hiint(a[C0 + q]), J);
    }
    asm volatile("" : "+s"(lo[0]), "+s"(hi[0]), "+s"(lo[1]), "+s"(hi[1]), "+s"(lo[2]), "+s"(hi[2]), "+s"(lo[3]),
                      "+s"(hi[3]), "+s"(lo[4]), "+s"(hi[4]), "+s"(lo[5]), "+s"(hi[5]), "+s"(lo[6]), "+s"(hi[6]),
                      "+s"(lo[7]), "+s"(hi[7]));
#pragma unroll
    for (int q = 0; q < 8; ++q) a[C0 + q] = __builtin_fma(-l, __hiloint2double(hi[q], lo[q]), a[C0 + q]);
}
template <int J, int C0>
__device__ __forceinline__ void tile_lu_columns(double (&a)[NB], double l) {
    if constexpr (C0 + 8 <= NB) {
        tile_lu_chunk8<J, C0>(a, l);
        tile_lu_columns<J, C0 + 8>(a, l);
    } else {
#pragma unroll
        for (int c = C0; c < NB; ++c) a[c] = __builtin_fma(-l, readlane_f64(a[c], J), a[c]);
    }
}
template <int J>
__device__ __forceinline__ void tile_lu_step(double (&a)[NB], int r, int kb, double inv, double piv, int& nbad,
                                             double thr) {
    // (J < kb is wave-uniform; a partial panel stops here)
    if (J >= kb) return;
    const double l = (r > J) ? a[J] * inv : 0.0;
    a[J] = (r > J) ? l : (r == J ? piv : a[J]);  // row J keeps the pivot actually used
    double inv_next = 1.0, piv_next = 1.0;
    if constexpr (J + 1 < NB) {
        a[J + 1] = __builtin_fma(-l, readlane_f64(a[J + 1], J), a[J + 1]);
        piv_next = readlane_f64(a[J + 1], J + 1);
        inv_next = pivot_reciprocal(piv_next, nbad, J + 1 < kb, thr);
    }
    tile_lu_columns<J, J + 2>(a, l);
    if constexpr (J + 1 < NB) tile_lu_step<J + 1>(a, r, kb, inv_next, piv_next, nbad, thr);
}
__device__ __forceinline__ void tile_factor(double (*T)[TPAD], int kb, int tid, int32_t* status, double thr) {
    if (tid >= 64) return;
    const int r = tid & (NB - 1);  // lanes 32..63 shadow lanes 0..31 and store nothing
    double a[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) a[c] = T[r][c];
    int nbad = 0;
    double piv0 = readlane_f64(a[0], 0);
    const double inv0 = pivot_reciprocal(piv0, nbad, kb > 0, thr);
    tile_lu_step<0>(a, r, kb, inv0, piv0, nbad, thr);
    if (tid == 0 && nbad) atomicAdd(status, nbad);
    if (tid < NB) {
#pragma unroll
        for (int c = 0; c < NB; ++c) T[r][c] = a[c];
    }
}

// diagonal tile of panel p of every front of a level
__global__ void __launch_bounds__(256) diag_kernel(MF_FACTOR_PARAMS, int p) {
    MF_FACTOR_INIT
    const MfFrontDev f = mf.lfronts[level_begin + blockIdx.x];
    const int ld = f.ld, m = 2 * f.k, r0 = p * NB;  // m: extent of the pivot + augmentation block
    const int kb = min(NB, f.k - r0);
    __shared__ double T[NB][TPAD];
    double* F = mf.front_store + f.off;
    const double thr = MF_PIVOT_EPS * *mf.piv_amax;
    const int tid = threadIdx.x, tc = tid % NB, tr = tid / NB;
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s, gr = r0 + r, gc = r0 + tc;
        T[r][tc] = (gr < m && gc < m) ? F[(int64_t)gr * ld + gc] : (r == tc ? 1.0 : 0.0);
    }
    __syncthreads();
    tile_factor(T, kb, tid, mf.status, thr);
    __syncthreads();
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s, gr = r0 + r, gc = r0 + tc;
        if (gr < m && gc < m) F[(int64_t)gr * ld + gc] = T[r][tc];
    }
}

// Panel tiles by substitution with the factored diagonal tile:
//   U panel tile (p, t): one lane per COLUMN, x_r = b_r - sum_{j<r} l_rj x_j;
//   L panel tile (t, p): one lane per ROW,    x_c = (b_c - sum_{j<c} x_j u_jc) / u_cc          (t > p)
// in the right-looking order: as soon as x_j is final every later entry takes its term, x_i -= s_ji x_j.  The 31-j
// updates of a step are independent of each other, so the only chain through the sweep is one multiply-add per
// step (the left-looking form -- a dot product per entry -- chained i/2 dependent multiply-adds at step i and took
// 3.6 us per tile).  The factor is laid out in LDS as S[j][i] (i > j) = l_ij resp. u_ji, zero elsewhere and for
// pivots j >= kb, with 1/u_jj beside it (1 for j >= kb): row j is one contiguous broadcast read that does not
// depend on x, the substitution is branch-free straight-line code over a register array, and the rows / columns
// kb.. of a partial panel receive their trailing update by the same formula.
// (Earlier versions multiplied by explicit inverses of the tile factors; building those inverses cost the
// look-ahead tile LU as much again as the elimination itself, on the critical path of every panel.)
constexpr int SPAD = NB + 2;  // even row stride: 16-byte aligned rows
template <int J>
__device__ __forceinline__ void trsm_row_load(const double (*S)[SPAD], double2 (&buf)[NB / 2]) {
    // entries J+1 .. NB-1 of row J in aligned pairs; a pair that starts at J itself holds a zero there
#pragma unroll
    for (int i = (J + 1) & ~1; i < NB; i += 2) buf[i / 2] = *reinterpret_cast<const double2*>(&S[J][i]);
}
template <bool SCALE, int J>
__device__ __forceinline__ void trsm_step(const double (*S)[SPAD], const double* Dv, double (&x)[NB],
                                          const double2 (&cur)[NB / 2]) {
    // row J + 1 is requested before row J is consumed: the reads do not depend on x, and an LDS round trip per
    // pair of multiply-adds is what the compiler's own schedule exposes
    double2 nxt[NB / 2];
    if constexpr (J + 1 < NB) trsm_row_load<J + 1>(S, nxt);
    if (SCALE) x[J] *= Dv[J];
    const double xj = x[J];
#pragma unroll
    for (int i = (J + 1) & ~1; i < NB; i += 2) {
        if (i > J) x[i] = __builtin_fma(-cur[i / 2].x, xj, x[i]);
        x[i + 1] = __builtin_fma(-cur[i / 2].y, xj, x[i + 1]);
    }
    if constexpr (J + 1 < NB) trsm_step<SCALE, J + 1>(S, Dv, x, nxt);
}
template <bool SCALE>
__device__ __forceinline__ void trsm_sweep(const double (*S)[SPAD], const double* Dv, double (&x)[NB]) {
    double2 row0[NB / 2];
    trsm_row_load<0>(S, row0);
    trsm_step<SCALE, 0>(S, Dv, x, row0);
}

// One launch per panel p: tile(ti,tj) -= L(ti,p)[:, :kb] * U(p,tj)[:kb, :]   (ti, tj > p), where every
// workgroup first SOLVES its two panel tiles itself (wavefront 0 the L tile, wavefront 1 the U tile, 32
// lanes each, in registers) instead of reading them from a separate panel-solve launch: the redundant
// substitutions cost a few microseconds of an otherwise idle chip, the kernel boundary they replace cost
// twelve on the critical path of each of the ~80 panels.  The panel tiles themselves stay as they are (other
// workgroups of the launch read them); the ones that are results -- the augmentation blocks, which become
// L11^-1 and U11^-1 -- are solved in place by panel_finalize_kernel after the last panel.  Look-ahead: the workgroup that owns the next diagonal tile (p+1,p+1)
// factors it right after updating it, so panels p >= 1 need no separate diagonal launch.
// (A second stream for the diagonal tile was tried as well: the cross-stream events cost as much as they
// hid.)
//
// Two blocking levels (large fronts): with t_end < nt only the tiles of block rows / columns p+1 .. t_end-1
// are updated now -- the L-shaped region the remaining panels of the current OUTER block need -- and the
// rest of the trailing matrix waits for one rank-(32 * outer block) update by block_gemm_kernel: rank-32 tile
// updates move 2 flop per byte and are bandwidth-bound on fronts of thousands of pivots.  The grid is the
// L shape: blockIdx.x = position inside the outer block, blockIdx.y < rem: the column part (all rows),
// blockIdx.y >= rem: the row part beyond t_end.  t_end >= nt gives the plain square.
// (-DSANM_MF_PHASES: the workgroup that factors the next diagonal tile -- the critical path of the panel chain --
// accumulates wall-clock ticks per phase; backend_hip.hip prints them after a factorisation.)
#ifdef SANM_MF_PHASES
__device__ unsigned long long g_phase[8];
#define SANM_PHASE_MARK(name) const unsigned long long name = wall_clock64()
#else
#define SANM_PHASE_MARK(name)
#endif
__global__ void __launch_bounds__(256) update_kernel(MF_FACTOR_PARAMS, int p, int t_end) {
    MF_FACTOR_INIT
    SANM_PHASE_MARK(c0);
    const MfFrontDev f = mf.lfronts[level_begin + blockIdx.z];
    const int ld = f.ld, m = 2 * f.k, nt = (m + NB - 1) / NB;
    const int te = min(t_end, nt), rem = nt - p - 1;
    if ((int)blockIdx.x >= te - p - 1) return;
    int ti, tj;
    if ((int)blockIdx.y < rem) {
        ti = p + 1 + blockIdx.y;
        tj = p + 1 + blockIdx.x;
    } else {
        ti = p + 1 + blockIdx.x;
        tj = te + ((int)blockIdx.y - rem);
    }
    if (ti >= nt || tj >= nt) return;
    // the (augmentation x augmentation) corner is never used
    if (ti * NB >= f.k && tj * NB >= f.k) return;
    const int kb = min(NB, f.k - p * NB);
    const double thr = MF_PIVOT_EPS * *mf.piv_amax;  // (requested with the tiles; used by the look-ahead tile LU)
    __shared__ double L[NB][TPAD], U[NB][TPAD], T[NB][TPAD];
    __shared__ __attribute__((aligned(16))) double SL[NB][SPAD], SU[NB][SPAD];
    __shared__ double Dv[NB];
    double* F = mf.front_store + f.off;
    const int tid = threadIdx.x, tc = tid % NB, tr = tid / NB;
    for (int s = 0; s < 4; ++s) {
        const int r = tr + 8 * s;
        int gr = ti * NB + r, gc = p * NB + tc;
        L[r][tc] = (gr < m && gc < m) ? F[(int64_t)gr * ld + gc] : 0.0;
        gr = p * NB + r;
        gc = tj * NB + tc;
        U[r][tc] = (gr < m && gc < m) ? F[(int64_t)gr * ld + gc] : 0.0;
        // diagonal tile entry (r, tc): multipliers by pivot column for the U panel, U by pivot row for the L panel
        gc = p * NB + tc;
        const double v = (gr < m && gc < m) ? F[(int64_t)gr * ld + gc] : 0.0;
        SL[tc][r] = (tc < r && tc < kb) ? v : 0.0;  // SL[j][i] = l_ij
        SU[r][tc] = (r < tc && r < kb) ? v : 0.0;   // SU[j][i] = u_ji
        if (r == tc) Dv[r] = (r < kb && fabs(v) > 1e-290) ? 1.0 / v : 1.0;
    }
    // this thread's four entries of the tile to update (matrix-core C layout, see below): requested now, they
    // arrive while the panel tiles are solved
    const int lane = tid & 63, qi = (tid >> 6) >> 1, qj = (tid >> 6) & 1;
    double cpre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int gr = ti * NB + 16 * qi + (lane >> 4) + 4 * g, gc = tj * NB + 16 * qj + (lane & 15);
        cpre[g] = (gr < m && gc < m) ? F[(int64_t)gr * ld + gc] : 0.0;
    }
    __syncthreads();
    SANM_PHASE_MARK(c1);
    if (tid < 32) {  // row tid of the L tile
        double x[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) x[c] = L[tid][c];
        trsm_sweep<true>(SU, Dv, x);
#pragma unroll
        for (int c = 0; c < NB; ++c) L[tid][c] = x[c];
    } else if (tid >= 64 && tid < 96) {  // column tid - 64 of the U tile
        const int q = tid - 64;
        double x[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) x[r] = U[r][q];
        trsm_sweep<false>(SL, nullptr, x);
#pragma unroll
        for (int r = 0; r < NB; ++r) U[r][q] = x[r];
    }
    __syncthreads();
    SANM_PHASE_MARK(c2);
    const bool next_diag = (ti == tj) && (ti == p + 1) && ((p + 1) * NB < f.k);  // workgroup-uniform
    {
        // rank-kb update of the 32x32 tile on the fp64 matrix cores: wavefront w owns the 16x16 quadrant
        // (w >> 1, w & 1), 8 steps of v_mfma_f64_16x16x4_f64 (operand layout: see gemm_tile below) -- the vector
        // version read two LDS operands per multiply-add and took 3.8 us of the 22 us of a panel step
        static_assert(NB == 32, "the tile update is written for 32x32 tiles");
        mfma_f64x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < NB / 4; ++ks) {
            const int kk = 4 * ks + (lane >> 4);  // pivot columns / rows only
            const double a = kk < kb ? L[16 * qi + (lane & 15)][kk] : 0.0;
            const double b = kk < kb ? U[kk][16 * qj + (lane & 15)] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r = 16 * qi + (lane >> 4) + 4 * g, c = 16 * qj + (lane & 15);
            const int gr = ti * NB + r, gc = tj * NB + c;
            double v = (r == c) ? 1.0 : 0.0;  // identity padding outside the front
            if (gr < m && gc < m) {
                v = cpre[g] - acc[g];
                if (!next_diag) F[(int64_t)gr * ld + gc] = v;
            }
            if (next_diag) T[r][c] = v;
        }
    }
    if (!next_diag) return;
    __syncthreads();
    SANM_PHASE_MARK(c3);
    const int kb1 = min(NB, f.k - (p + 1) * NB);
    tile_factor(T, kb1, tid, mf.status, thr);
    __syncthreads();
    SANM_PHASE_MARK(c4);
    for (int s = 0; s < 4; ++s) {
        int r = tr + 8 * s, gr = ti * NB + r, gc = tj * NB + tc;
        if (gr < m && gc < m) F[(int64_t)gr * ld + gc] = T[r][tc];
    }
#ifdef SANM_MF_PHASES
    if (tid == 0 && blockIdx.z == 0) {
        atomicAdd(&g_phase[0], c1 - c0);
        atomicAdd(&g_phase[1], c2 - c1);
        atomicAdd(&g_phase[2], c3 - c2);
        atomicAdd(&g_phase[3], c4 - c3);
        atomicAdd(&g_phase[4], wall_clock64() - c4);
        atomicAdd(&g_phase[7], 1ull);
    }
#endif
}

// After the panel loop of a level: solve, in place, the panel tiles that hold results -- U panel tiles
// (p, t) and L panel tiles (t, p) with t in the augmentation block (they become L11^-1 resp. U11^-1).  All
// (front, panel, tile) triples are independent: one launch, 8 tiles per workgroup, one lane per column / row.
constexpr int FIN_TILES = 8;
// (Two blocking levels: called per outer block [p_begin, p_begin + nr_panel) with t_min = its end: the tiles
// from t_min on are exactly what block_gemm_kernel multiplies, and they include the augmentation tiles.)
__global__ void __launch_bounds__(256) panel_finalize_kernel(MF_FACTOR_PARAMS, int p_begin, int nr_panel,
                                                             int t_min) {
    MF_FACTOR_INIT
    const MfFrontDev f = mf.lfronts[level_begin + blockIdx.z / nr_panel];
    const int p = p_begin + blockIdx.z % nr_panel;
    if (p * NB >= f.k) return;
    const int ld = f.ld, m = 2 * f.k, nt = (m + NB - 1) / NB;
    // first tile to solve: the first one with augmentation columns / rows (they hold results), or t_min
    // if that comes earlier (the tiles block_gemm_kernel multiplies)
    const int ta = f.k / NB;
    const int t0 = max(p + 1, t_min >= 0 ? min(t_min, ta) : ta);
    if (t0 + (int)blockIdx.x * FIN_TILES >= nt) return;
    const bool upanel = blockIdx.y == 0;
    const int kb = min(NB, f.k - p * NB);
    __shared__ __attribute__((aligned(16))) double S[NB][SPAD];
    __shared__ double Dv[NB];
    double* F = mf.front_store + f.off;
    const int tid = threadIdx.x;
    for (int s = 0; s < 4; ++s) {
        // (a, c) of the diagonal tile
        const int a = tid / NB + 8 * s, c = tid % NB, ga = p * NB + a, gc = p * NB + c;
        const double v = (ga < m && gc < m) ? F[(int64_t)ga * ld + gc] : 0.0;
        if (upanel) {
            S[c][a] = (c < a && c < kb) ? v : 0.0;  // S[j][i] = l_ij
        } else {
            S[a][c] = (a < c && a < kb) ? v : 0.0;  // S[j][i] = u_ji
            if (a == c) Dv[a] = (a < kb && fabs(v) > 1e-290) ? 1.0 / v : 1.0;
        }
    }
    __syncthreads();
    const int t = t0 + blockIdx.x * FIN_TILES + tid / NB, q = tid % NB;
    if (t >= nt) return;
    // rows (U panel) / columns (L panel) of the panel that lie inside the 2k x 2k block: all NB of them
    // except for fronts of fewer than NB - 1 pivots.  Loads clamp their index (a duplicate read is harmless,
    // per-element predicates cost the register allocation of the whole kernel dearly); stores of a partial
    // panel take the predicated path.
    const int vmax = min(NB, m - p * NB) - 1;
    double x[NB];
    if (upanel) {
        const int gc = t * NB + q;
        if (gc >= m) return;
        double* col = F + (int64_t)p * NB * ld + gc;
#pragma unroll
        for (int r = 0; r < NB; ++r) x[r] = col[(int64_t)min(r, vmax) * ld];
        trsm_sweep<false>(S, nullptr, x);
        if (vmax == NB - 1) {
#pragma unroll
            for (int r = 0; r < NB; ++r) col[(int64_t)r * ld] = x[r];
        } else {
#pragma unroll
            for (int r = 0; r < NB; ++r)
                if (r <= vmax) col[(int64_t)r * ld] = x[r];
        }
    } else {
        const int gr = t * NB + q;
        if (gr >= m) return;
        double* row = F + (int64_t)gr * ld + p * NB;
#pragma unroll
        for (int c = 0; c < NB; ++c) x[c] = row[min(c, vmax)];
        trsm_sweep<true>(S, Dv, x);
        if (vmax == NB - 1) {
#pragma unroll
            for (int c = 0; c < NB; ++c) row[c] = x[c];
        } else {
#pragma unroll
            for (int c = 0; c < NB; ++c)
                if (c <= vmax) row[c] = x[c];
        }
    }
}

// C tile (64x64) of a product of two strided matrices on the fp64 matrix cores: 4 wavefronts, each a
// 32x32 quadrant = 2x2 tiles of v_mfma_f64_16x16x4_f64 (A: lane l holds A[l&15][l>>4], B: B[l>>4][l&15],
// C/D: register g of lane l is C[(l>>4) + 4g][l&15]).  K advances in steps of 16 through LDS; the next
// step's global loads are issued before the current step's MFMAs (register prefetch).  Element (i,j) of
// an operand is p[i*ld + j] inside (rows, cols), else 0.  K range [k0, k1) in elements.
constexpr int GT = MF_GT;  // GEMM tile edge
constexpr int GK = 16;   // GEMM K step

struct GemmStage {
    double a[4], b[4];
};
template <bool IDX = false>
__device__ __forceinline__ void gemm_stage_load(GemmStage& st, const MatView& A, const MatView& B, int ti,
                                                int tj, int kk, int k1) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int idx = tid + 256 * s;             // 0..1023
        const int ar = idx / GK, ae = idx % GK;    // consecutive threads along K: contiguous in a row of A
        int gr = ti * GT + ar, gc = kk + ae;
        if (IDX) {  // A's columns through an index map (MatView::cidx)
            const bool ok = gr < A.rows && gc < A.cols && gc < k1;
            const int pc = (ok && A.cidx) ? A.cidx[gc] : gc;
            st.a[s] = ok ? A.p[(int64_t)gr * A.ld + pc] : 0.0;
        } else
            st.a[s] = (gr < A.rows && gc < A.cols && gc < k1) ? A.p[(int64_t)gr * A.ld + gc] : 0.0;
        const int be = idx / GT, bc = idx % GT;    // consecutive threads along the columns of B
        gr = kk + be;
        gc = tj * GT + bc;
        if (IDX) {  // B's rows through an index map (MatView::ridx)
            const bool ok = gr < B.rows && gr < k1 && gc < B.cols;
            const int pr = (ok && B.ridx) ? B.ridx[gr] : gr;
            st.b[s] = ok ? B.p[(int64_t)pr * B.ld + gc] : 0.0;
        } else
            st.b[s] = (gr < B.rows && gr < k1 && gc < B.cols) ? B.p[(int64_t)gr * B.ld + gc] : 0.0;
    }
}
__device__ __forceinline__ void gemm_stage_store(const GemmStage& st, double (*As)[GT + 1],
                                                 double (*Bs)[GT + 4]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int idx = tid + 256 * s;
        As[idx % GK][idx / GK] = st.a[s];  // A tile 64 x 16 stored transposed As[e][row]
        Bs[idx / GT][idx % GT] = st.b[s];  // B tile 16 x 64 as Bs[e][col]
    }
}

// Fast path for interior tiles (round 3).  The guarded loads above put exec-mask branches into the K loop, and with
// them the compiler kept the accumulators in VGPRs and copied all 32 of them to the matrix cores' registers and
// back at EVERY K step (64 v_accvgpr moves and an s_nop 15 that drains the MFMA pipeline: the ISA of round 2's
// gemm2).  A tile that lies inside both operands needs no guard at all for its full K steps: plain 16-byte loads,
// no selects, nothing the compiler could turn into a branch (a first version with clamped addresses and selects
// WAS turned into guarded loads with a wait after each one -- 34 % slower in place although faster in the
// micro-benchmark); edge tiles and the last partial K step take the guarded loop.  With -amdgpu-mfma-vgpr-form
// (build.py) the fast K loop is loads, LDS traffic and MFMAs.  scripts/micro/gemm_bench.hip, 8192^2, K = 128 / 256 /
// 4096: 35.9 / 43.4 / 58.0 TFLOP/s before, 39.0 / 49.1 / 60.5 with this staging, 37.6 / 51.4 / 64.7 with the
// 128 x 64 tile below; every variant gives the same bits (same summation order per element).
typedef double2 __attribute__((aligned(8))) double2_u;  // a pair of doubles at any 8-byte address
template <int TM>
struct GemmStage2 {
    double2 a[TM / 32], b[2];  // A tile TM x 16 and B tile 16 x 64 over 256 threads, two doubles at a time
};
template <int TM>
__device__ __forceinline__ void gemm_stage2_load(GemmStage2<TM>& st, const MatView& A, const MatView& B, int ti, int tj,
                                                 int kk) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < TM / 32; ++s) {
        const int idx = tid + 256 * s;
        const int ar = idx >> 3, ae = (idx & 7) * 2;  // 8 pairs per row of the A tile
        st.a[s] = *reinterpret_cast<const double2_u*>(A.p + (int64_t)(ti * TM + ar) * A.ld + kk + ae);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int idx = tid + 256 * s;
        const int be = idx >> 5, bc = (idx & 31) * 2;  // 32 pairs per row of the B tile
        st.b[s] = *reinterpret_cast<const double2_u*>(B.p + (int64_t)(kk + be) * B.ld + tj * GT + bc);
    }
}
template <int TM>
__device__ __forceinline__ void gemm_stage2_store(const GemmStage2<TM>& st, double (*As)[TM + 1], double (*Bs)[GT + 4]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < TM / 32; ++s) {
        const int idx = tid + 256 * s;
        const int ar = idx >> 3, ae = (idx & 7) * 2;
        As[ae][ar] = st.a[s].x;  // transposed: As[e][row]
        As[ae + 1][ar] = st.a[s].y;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int idx = tid + 256 * s;
        *reinterpret_cast<double2*>(&Bs[idx >> 5][(idx & 31) * 2]) = st.b[s];
    }
}

// acc[mi][ni]: the 16x16 tile at rows 32*(wave>>1) + 16*mi, columns 32*(wave&1) + 16*ni of the 64x64 tile
template <bool IDX = false>
__device__ __forceinline__ void gemm_tile(const MatView& A, const MatView& B, int ti, int tj, int k0,
                                          int k1, double (*As)[GT + 1], double (*Bs)[GT + 4],
                                          mfma_f64x4 acc[2][2], bool zero = true) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = 32 * (wv >> 1) + (lane & 15), c0 = 32 * (wv & 1) + (lane & 15), kq = lane >> 4;
    if (zero) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f64x4{0, 0, 0, 0};
    }
#ifndef SANM_MF_OLD_STAGING  // (A/B switch: round 2's guarded element loads everywhere)
    if (!IDX) {
        // interior tile (workgroup-uniform): its full K steps without a single guard
        const int kin = min(k1, min(A.cols, B.rows));
        const int kfull = k0 + (max(kin - k0, 0) / GK) * GK;
        if (ti * GT + GT <= A.rows && tj * GT + GT <= B.cols && kfull > k0) {
            GemmStage2<GT> st;
            gemm_stage2_load<GT>(st, A, B, ti, tj, k0);
            for (int kk = k0; kk < kfull; kk += GK) {
                __syncthreads();  // the previous step's fragments have been read
                gemm_stage2_store<GT>(st, As, Bs);
                __syncthreads();
                if (kk + GK < kfull) gemm_stage2_load<GT>(st, A, B, ti, tj, kk + GK);
#pragma unroll
                for (int e = 0; e < GK; e += 4) {
                    const double a0 = As[e + kq][r0], a1 = As[e + kq][r0 + 16];
                    const double b0 = Bs[e + kq][c0], b1 = Bs[e + kq][c0 + 16];
                    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
                }
            }
            k0 = kfull;  // what is left (a partial last step) takes the guarded loop
        }
    }
#endif
    GemmStage st;
    if (k0 < k1) gemm_stage_load<IDX>(st, A, B, ti, tj, k0, k1);
    for (int kk = k0; kk < k1; kk += GK) {
        __syncthreads();  // the previous step's fragments have been read
        gemm_stage_store(st, As, Bs);
        __syncthreads();
        if (kk + GK < k1) gemm_stage_load<IDX>(st, A, B, ti, tj, kk + GK, k1);
#pragma unroll
        for (int e = 0; e < GK; e += 4) {
            const double a0 = As[e + kq][r0], a1 = As[e + kq][r0 + 16];
            const double b0 = Bs[e + kq][c0], b1 = Bs[e + kq][c0 + 16];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
}

// The full K steps of an INTERIOR 128 x 64 tile (rows [128 ti, 128 ti + 128), columns [64 tj, 64 tj + 64) inside both
// operands; K range inside them too): each wavefront a 64 x 32 block = 4 x 2 MFMA tiles, 6 LDS reads per 8 MFMAs instead of 4 per 4 and half
// the B traffic per flop.  For the Schur complements of big fronts (long K).  acc[mi][ni]: rows 64*(wave>>1) +
// 16*mi, columns 32*(wave&1) + 16*ni.  K range [k0, k1).
constexpr int GT2 = 128;
constexpr int kTallMinK = MF_TALL_MIN_K, kTallMinB = MF_TALL_MIN_B;  // fronts from this size on take the tall tile for F[B,B] -= L U
__device__ __forceinline__ void gemm_tile_tall(const MatView& A, const MatView& B, int ti, int tj, int k0, int k1,
                                               double (*As)[GT2 + 1], double (*Bs)[GT + 4], mfma_f64x4 acc[4][2]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = 64 * (wv >> 1) + (lane & 15), c0 = 32 * (wv & 1) + (lane & 15), kq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f64x4{0, 0, 0, 0};
    const int kfull = k0 + ((k1 - k0) / GK) * GK;
    auto mfma_step = [&]() {
#pragma unroll
        for (int e = 0; e < GK; e += 4) {
            double a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[e + kq][r0 + 16 * i];
            b[0] = Bs[e + kq][c0];
            b[1] = Bs[e + kq][c0 + 16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    if (kfull > k0) {
        GemmStage2<GT2> st;
        gemm_stage2_load<GT2>(st, A, B, ti, tj, k0);
        for (int kk = k0; kk < kfull; kk += GK) {
            __syncthreads();
            gemm_stage2_store<GT2>(st, As, Bs);
            __syncthreads();
            if (kk + GK < kfull) gemm_stage2_load<GT2>(st, A, B, ti, tj, kk + GK);
            mfma_step();
        }
    }
    if (kfull < k1) {  // the last, partial K step: element loads with the K bound (same accumulation chain)
        __syncthreads();
        const int tid = threadIdx.x;
#pragma unroll
        for (int s = 0; s < 8; ++s) {  // A tile 128 x 16
            const int idx = tid + 256 * s, ar = idx / GK, ae = idx % GK;
            const int gc = kfull + ae;
            As[ae][ar] = gc < k1 ? A.p[(int64_t)(ti * GT2 + ar) * A.ld + gc] : 0.0;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {  // B tile 16 x 64
            const int idx = tid + 256 * s, be = idx / GT, bc = idx % GT;
            const int gr = kfull + be;
            Bs[be][bc] = gr < k1 ? B.p[(int64_t)gr * B.ld + tj * GT + bc] : 0.0;
        }
        __syncthreads();
        mfma_step();
    }
}

// visit every element of the accumulators: f(row in tile, column in tile, value)
template <class F>
__device__ __forceinline__ void gemm_tile_foreach(const mfma_f64x4 acc[2][2], F&& f) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                f(32 * (wv >> 1) + 16 * mi + (lane >> 4) + 4 * g, 32 * (wv & 1) + 16 * ni + (lane & 15),
                  acc[mi][ni][g]);
}

// C[r, c] -= acc over the part of the tile inside (rows, cols): the 16 old values of a lane are requested TOGETHER,
// then subtracted and stored.  (`*dst = *dst - v` element by element made every store wait for its own load and every
// load for the store before it -- the compiler cannot tell that the addresses differ --: 16 dependent round trips per
// lane, the whole run time of a tile with a short K loop, i.e. of every product of the fronts of the lower levels.)
__device__ __forceinline__ void gemm_tile_sub_from(const mfma_f64x4 acc[2][2], double* __restrict__ C, int64_t ldc, int ti,
                                                   int tj, int rows, int cols) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int rb = ti * GT + 32 * (wv >> 1) + (lane >> 4), cb = tj * GT + 32 * (wv & 1) + (lane & 15);
    double old[2][2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = rb + 16 * mi + 4 * g, c = cb + 16 * ni;
                old[mi][ni][g] = (r < rows && c < cols) ? C[(int64_t)r * ldc + c] : 0.0;
            }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = rb + 16 * mi + 4 * g, c = cb + 16 * ni;
                if (r < rows && c < cols) C[(int64_t)r * ldc + c] = old[mi][ni][g] - acc[mi][ni][g];
            }
}

// step 2:  which = 0: tmpU (k x b) = L11^-1 F[P,B]     (L11^-1 lower: K tiles 0..ti)
//          which = 1: tmpL (b x k) = F[B,P] U11^-1     (U11^-1 upper: K tiles 0..tj)
// two_phase (Level::two_phase): the negated results are also the front's boundary operators, -U12 in the F[A,B] slot and
// -L21 in the F[B,A] slot (gemm2_kernel then leaves its products 1 and 2 out)
__device__ __forceinline__ void gemm1_tile(const FactorArgs& mf, const MfFrontDev& f, int which, int ti, int tj,
                                           double (*As)[GT + 1], double (*Bs)[GT + 4], int two_phase) {
    const int k = f.k, b = f.m - f.k, ld = f.ld;
    const int rows = which ? b : k, cols = which ? k : b;
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
    mfma_f64x4 acc[2][2];
    gemm_tile(A, B, ti, tj, 0, k1, As, Bs, acc);
    double* C = which ? tmp + (int64_t)k * b : tmp;  // tmpU: ld b ; tmpL: ld k
    const int cld = which ? k : b;
    double* slot = const_cast<double*>(F) + (which ? (int64_t)2 * k * ld + k : (int64_t)k * ld + 2 * k);  // F[B,A] : F[A,B]
    gemm_tile_foreach(acc, [&](int i, int j, double v) {
        const int r = ti * GT + i, c = tj * GT + j;
        if (r < rows && c < cols) {
            C[(int64_t)r * cld + c] = v;
            if (two_phase) slot[(int64_t)r * ld + c] = -v;
        }
    });
}
__global__ void __launch_bounds__(256) gemm1_kernel(MF_FACTOR_PARAMS, int two_phase) {
    MF_FACTOR_INIT
    const MfFrontDev f = mf.lfronts[level_begin + blockIdx.z / 2];
    const int which = blockIdx.z & 1;
    const int k = f.k, b = f.m - f.k;
    const int rows = which ? b : k, cols = which ? k : b;
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (ti * GT >= rows || tj * GT >= cols) return;
    __shared__ double As[GK][GT + 1], Bs[GK][GT + 4];
    gemm1_tile(mf, f, which, ti, tj, As, Bs, two_phase);
}

// F[B,B] -= tmpL tmpU of a big front: its interior 128 x 64 tiles go to gemm2_tall_kernel (a kernel of its own: the
// eight accumulator tiles per wavefront cost 136 VGPRs, which would take a wavefront per SIMD from every other
// product of gemm2_kernel), the tile rows at the lower edge and everything on smaller fronts stay here
__device__ __forceinline__ bool gemm2_is_tall(int k, int b, int rows, int cols, int ti, int tj) {
    return rows == b && cols == b && mf_gemm2_is_tall(k, b, ti, tj);
}
// one tall tile: rows [64 ti, 64 ti + 128) x columns [64 tj, 64 tj + 64) of F[B,B], ti even
__device__ __forceinline__ void gemm2_tall_tile(const FactorArgs& mf, const MfFrontDev& f, int ti, int tj,
                                                double (*As)[GT2 + 1], double (*Bs)[GT + 4]);
__global__ void __launch_bounds__(256) gemm2_tall_kernel(MF_FACTOR_PARAMS) {  // (the box grid: SANM_MF_NO_TILE_LISTS)
    MF_FACTOR_INIT
    const MfFrontDev f = mf.lfronts[level_begin + blockIdx.z];
    const int k = f.k, b = f.m - f.k;
    const int ti = 2 * blockIdx.y, tj = blockIdx.x;
    if (!gemm2_is_tall(k, b, b, b, ti, tj)) return;
    __shared__ double As[GK][GT2 + 1], Bs[GK][GT + 4];
    gemm2_tall_tile(mf, f, ti, tj, As, Bs);
}
// the same over the level's flat list of tall tiles, ordered for the eight L2s (mf_types.h, MF_ST_R): a launch of the
// tiles that exist -- the box grid (widest front's tiles x fronts of the level) spent 45 of the 73 ms of the tall
// tiles of a 2.7 M-tet factorisation on workgroups that found nothing to do
__global__ void __launch_bounds__(256) gemm2_tall_list_kernel(MF_FACTOR_PARAMS, const uint32_t* __restrict__ tiles) {
    MF_FACTOR_INIT
    const uint32_t w0 = tiles[2 * blockIdx.x], w1 = tiles[2 * blockIdx.x + 1];
    if (w0 == ~0u) return;  // (padding of the shorter queues)
    const MfFrontDev f = mf.lfronts[level_begin + w0];
    __shared__ double As[GK][GT2 + 1], Bs[GK][GT + 4];
    gemm2_tall_tile(mf, f, 2 * (int)(w1 >> 15), (int)(w1 & 32767), As, Bs);
}
__device__ __forceinline__ void gemm2_tall_tile(const FactorArgs& mf, const MfFrontDev& f, int ti, int tj,
                                                double (*As)[GT2 + 1], double (*Bs)[GT + 4]) {
    const int k = f.k, b = f.m - f.k, ld = f.ld;
    double* F = mf.front_store + f.off;
    const double* tmpU = mf.tmp_store + f.tmp_off;
    const double* tmpL = tmpU + (int64_t)k * b;
    const MatView A{tmpL, k, b, k}, B{tmpU, b, k, b};
    double* C = F + (int64_t)2 * k * ld + 2 * k;
    mfma_f64x4 acc4[4][2];
    gemm_tile_tall(A, B, ti >> 1, tj, 0, k, As, Bs, acc4);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // (old values requested together, see gemm_tile_sub_from; the tile is interior: no guards)
    double* __restrict__ Cw = C + (int64_t)((ti >> 1) * GT2 + 64 * (wv >> 1) + (lane >> 4)) * ld + tj * GT + 32 * (wv & 1) + (lane & 15);
    double old[4][2][4];
    const bool leaf = f.nch == 0;  // (no children: F[B,B] holds nothing yet and was not zeroed, see gemm2_tile)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) old[mi][ni][g] = leaf ? 0.0 : Cw[(int64_t)(16 * mi + 4 * g) * ld + 16 * ni];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) Cw[(int64_t)(16 * mi + 4 * g) * ld + 16 * ni] = old[mi][ni][g] - acc4[mi][ni][g];
}

// step 3:  which = 0: F[B,B] -= tmpL tmpU                        (b x b, K = k)
//          which = 1: F[B,A]  = -tmpL L11^-1   (b x k; L11^-1 lower: K tiles tj..)
//          which = 2: F[A,B]  = -U11^-1 tmpU   (k x b; U11^-1 upper: K tiles ti..)
//          nwhich = 1 (two-phase levels): product 0 only
__device__ __forceinline__ void gemm2_tile(const FactorArgs& mf, const MfFrontDev& f, int which, int ti, int tj,
                                           double (*As)[GT + 1], double (*Bs)[GT + 4], int tr = 0) {
    const int k = f.k, b = f.m - f.k, ld = f.ld;
    const int rows = which == 2 ? k : b, cols = which == 1 ? k : b;
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
    mfma_f64x4 acc[2][2];
    gemm_tile(A, B, ti, tj, k0, k, As, Bs, acc);
    if (which == 0) {
        if (f.nch == 0) {
            // a front without children: nothing was added to its F[B,B], which the prologue no longer zeroes either
            // (zero_kernel) -- the block is written, 0.0 - v being the bits `old - v` gave on a zeroed block
            gemm_tile_foreach(acc, [&](int i, int j, double v) {
                const int r = ti * GT + i, c = tj * GT + j;
                if (r < rows && c < cols) C[(int64_t)r * ld + c] = 0.0 - v;
            });
            return;
        }
        gemm_tile_sub_from(acc, C, ld, ti, tj, rows, cols);
        return;
    }
    if (which == 1 && tr) {
        // Level::fwd_t: F[B,A] transposed into the dead F[P,B] slot (k rows of b entries)
        double* CT = F + 2 * k;
        gemm_tile_foreach(acc, [&](int i, int j, double v) {
            const int r = ti * GT + i, c = tj * GT + j;
            if (r < rows && c < cols) CT[(int64_t)c * ld + r] = -v;
        });
        return;
    }
    gemm_tile_foreach(acc, [&](int i, int j, double v) {
        const int r = ti * GT + i, c = tj * GT + j;
        if (r < rows && c < cols) C[(int64_t)r * ld + c] = -v;
    });
}
__global__ void __launch_bounds__(256) gemm2_kernel(MF_FACTOR_PARAMS, int nwhich, int tr) {
    MF_FACTOR_INIT
    const MfFrontDev f = mf.lfronts[level_begin + blockIdx.z / nwhich];
    const int which = blockIdx.z % nwhich;
    const int k = f.k, b = f.m - f.k;
    const int rows = which == 2 ? k : b, cols = which == 1 ? k : b;
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (ti * GT >= rows || tj * GT >= cols) return;
#ifndef SANM_MF_OLD_STAGING
    // (the interior of the Schur complement of a big front belongs to gemm2_tall_kernel)
    if (which == 0 && gemm2_is_tall(k, b, rows, cols, ti, tj)) return;
#endif
    __shared__ double As[GK][GT + 1], Bs[GK][GT + 4];
    gemm2_tile(mf, f, which, ti, tj, As, Bs, tr);
}

// The same two passes over flat tile lists (mf_types.h, Level::g1_tiles / g2_tiles): blockIdx.x is a tile that exists.
__global__ void __launch_bounds__(256) gemm1_list_kernel(MF_FACTOR_PARAMS, const uint32_t* __restrict__ tiles, int two_phase) {
    MF_FACTOR_INIT
    const uint32_t w0 = tiles[2 * blockIdx.x], w1 = tiles[2 * blockIdx.x + 1];
    const MfFrontDev f = mf.lfronts[level_begin + w0];
    __shared__ double As[GK][GT + 1], Bs[GK][GT + 4];
    gemm1_tile(mf, f, (int)(w1 >> 30), (int)((w1 >> 15) & 32767), (int)(w1 & 32767), As, Bs, two_phase);
}
__global__ void __launch_bounds__(256) gemm2_list_kernel(MF_FACTOR_PARAMS, const uint32_t* __restrict__ tiles, int tr) {
    MF_FACTOR_INIT
    const uint32_t w0 = tiles[2 * blockIdx.x], w1 = tiles[2 * blockIdx.x + 1];
    const MfFrontDev f = mf.lfronts[level_begin + w0];
    __shared__ double As[GK][GT + 1], Bs[GK][GT + 4];
    gemm2_tile(mf, f, (int)(w1 >> 30), (int)((w1 >> 15) & 32767), (int)(w1 & 32767), As, Bs, tr);
}

// ---- small fronts: the whole factorisation of a front in ONE workgroup (round 5) --------------------------------
// On the lower levels of a big tree a front has a few dozen pivots and a boundary of a few hundred rows, and there
// are thousands of them: every tile task of the panel / GEMM launches above is then a handful of dependent memory
// round trips around 32 x 32 x 32 flops -- 10-40 us per task, two to four tasks in flight per compute unit (their LDS),
// three quarters of the workgroups of a GEMM launch empty because the grid is sized for the level's largest front.
// Measured on armadillo with every tet cut into 8 (338 k tets): the leaf level's 2572 fronts (54 pivots, 89 boundary
// rows on average) took 3.2 ms of a 14.6 ms factorisation in seven launches, 8.4 GFLOP at 2.6 TFLOP/s.
// Here a workgroup owns a front (k <= 96 pivots):
//   A. its k x k pivot block goes to LDS once; LU without pivoting (pivots below the threshold perturbed and counted
//      like tile_factor does), then L and U are inverted IN PLACE -- column by column, L from its last column
//      backwards, U from its first forwards, both in the same sweep of k barriers --: no identity blocks carried
//      through a panel loop, no global memory between the steps;
//   B. L11^-1 / U11^-1 are stored where the solve kernels read them (F[P,A], F[A,P]) and the products of steps 2 and
//      3 above run as a loop over their 64 x 64 tiles in this workgroup (gemm1_tile / gemm2_tile: the same arithmetic
//      as the separate launches), operands from L2.
// The pivot block's own factors are not written back: nothing reads F[P,P] after the factorisation.
constexpr int SF_KMAX = 96;
// acc + sum_{t = t0}^{t1 - 1} row[t] * v[t], ONE accumulation chain in ascending t (the sums the plain loop formed, bit for
// bit) with the LDS operands of eight terms requested together: the plain loop waited for an LDS round trip per term,
// and these dot products are the k barrier-separated steps of the in-place inverses -- 77 of the 208 us a front of the
// 338 k-tet mesh spends in small_front_kernel
__device__ __forceinline__ double sf_dot(const double* row, const double* v, int t0, int t1, double acc) {
    int t = t0;
    for (; t + 8 <= t1; t += 8) {
        double a[8], b[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            a[q] = row[t + q];
            b[q] = v[t + q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = __builtin_fma(a[q], b[q], acc);
    }
    for (; t < t1; ++t) acc = __builtin_fma(row[t], v[t], acc);
    return acc;
}
#ifdef SANM_SF_PHASES
__device__ unsigned long long g_sf_phase[8];
#endif
__global__ void __launch_bounds__(256) small_front_kernel(MF_FACTOR_PARAMS, int ks, int tr) {
    MF_FACTOR_INIT
    const MfFrontDev f = mf.lfronts[level_begin + blockIdx.x];
    const int k = f.k, b = f.m - f.k, ld = f.ld;
    extern __shared__ __attribute__((aligned(16))) double sdyn[];
    double* S = sdyn;                       // S[r * ks + c], ks odd >= the level's largest k
    double* vbuf = sdyn + (int64_t)ks * ks;  // 2 x 2 x ks: columns saved ahead of their in-place inversion
    double* F = mf.front_store + f.off;
    const int tid = threadIdx.x;
    const double thr = MF_PIVOT_EPS * *mf.piv_amax;
#ifdef SANM_SF_PHASES  // (-DSANM_SF_PHASES: wall clock of this kernel's phases, summed over the workgroups; 10 ns ticks)
    const unsigned long long sf0 = wall_clock64();
#endif
    for (int r = tid >> 5; r < k; r += 8)
        for (int c = tid & 31; c < k; c += 32) S[r * ks + c] = F[(int64_t)r * ld + c];
    __syncthreads();
#ifdef SANM_SF_PHASES
    const unsigned long long sf1 = wall_clock64();
#endif
    // ---- LU, right-looking, one pivot per step
    int nbad = 0;
    const int ty = tid >> 4, tx = tid & 15;
    for (int j = 0; j < k; ++j) {
        double piv = S[j * ks + j];
        const bool bad = !(fabs(piv) > thr);
        if (bad) piv = thr > 0 ? copysign(thr, piv) : 1.0;
        const double inv = 1.0 / piv;
        // (a thread that reads the pivot after thread 0 has replaced it finds +-thr, replaces it by itself again and
        // gets the same reciprocal: no barrier needed around the replacement)
        if (tid == 0 && bad) {
            ++nbad;
            S[j * ks + j] = piv;
        }
        for (int i = j + 1 + tid; i < k; i += 256) S[i * ks + j] *= inv;
        __syncthreads();
        for (int i = j + 1 + ty; i < k; i += 16) {
            const double l = S[i * ks + j];
            for (int c = j + 1 + tx; c < k; c += 16) S[i * ks + c] = __builtin_fma(-l, S[j * ks + c], S[i * ks + c]);
        }
        __syncthreads();
    }
    if (tid == 0 && nbad) atomicAdd(mf.status, nbad);
#ifdef SANM_SF_PHASES
    const unsigned long long sf2 = wall_clock64();
#endif
    // ---- in-place inverses: step s finishes column k-2-s of L^-1 (threads 0..127) and column s of U^-1 (128..255).
    // A column's ORIGINAL entries are needed while its new ones are written: they are copied one step ahead into vbuf.
    double* vl = vbuf;           // [2][ks]
    double* vu = vbuf + 2 * ks;  // [2][ks]
    {
        const int jl0 = k - 2, ju0 = 0;
        if (tid < 128) {
            if (jl0 >= 0)
                for (int i = tid; i < k; i += 128) vl[i] = S[i * ks + jl0];
        } else {
            for (int i = tid - 128; i < k; i += 128) vu[i] = S[i * ks + ju0];
        }
    }
    __syncthreads();
    for (int s2 = 0; s2 < k; ++s2) {
        const int cur = s2 & 1, nxt = cur ^ 1;
        if (tid < 128) {
            const int jl = k - 2 - s2;
            if (jl >= 0) {
                const double* v = vl + cur * ks;
                // (the next step's column first: its copy does not wait for the dot products)
                if (jl >= 1)
                    for (int i = tid; i < k; i += 128) vl[nxt * ks + i] = S[i * ks + jl - 1];
                for (int i = jl + 1 + tid; i < k; i += 128) S[i * ks + jl] = -sf_dot(S + i * ks, v, jl + 1, i, v[i]);
            }
        } else {
            const int u = tid - 128, ju = s2;
            const double* v = vu + cur * ks;
            const double d = 1.0 / v[ju];
            if (ju + 1 < k)
                for (int i = u; i < k; i += 128) vu[nxt * ks + i] = S[i * ks + ju + 1];
            for (int i = u; i < ju; i += 128) S[i * ks + ju] = -sf_dot(S + i * ks, v, i, ju, 0.0) * d;
            if (u == 0) S[ju * ks + ju] = d;
        }
        __syncthreads();
    }
#ifdef SANM_SF_PHASES
    const unsigned long long sf3 = wall_clock64();
#endif
    // ---- L11^-1 (strictly lower part; its unit diagonal is the identity block's) -> F[P,A], U11^-1 -> F[A,P]
    for (int r = tid >> 5; r < k; r += 8)
        for (int c = tid & 31; c < k; c += 32) {
            const double v = S[r * ks + c];
            if (c < r) F[(int64_t)r * ld + k + c] = v;
            else F[(int64_t)(k + r) * ld + c] = v;
        }
    if (b == 0) return;
    __syncthreads();  // (also: S is dead from here on, the GEMM staging below takes its place)
    auto As = reinterpret_cast<double(*)[GT + 1]>(sdyn);
    auto Bs = reinterpret_cast<double(*)[GT + 4]>(sdyn + GK * (GT + 1));
    const int tk = (k + GT - 1) / GT, tb = (b + GT - 1) / GT;
#ifdef SANM_SF_PHASES
    const unsigned long long sf4 = wall_clock64();
#endif
    for (int tj = 0; tj < tb; ++tj)
        for (int ti = 0; ti < tk; ++ti) gemm1_tile(mf, f, 0, ti, tj, As, Bs, 0);  // tmpU (k x b)
    for (int ti = 0; ti < tb; ++ti)
        for (int tj = 0; tj < tk; ++tj) gemm1_tile(mf, f, 1, ti, tj, As, Bs, 0);  // tmpL (b x k)
    __syncthreads();
#ifdef SANM_SF_PHASES
    const unsigned long long sf5 = wall_clock64();
#endif
    for (int ti = 0; ti < tb; ++ti)
        for (int tj = 0; tj < tb; ++tj) gemm2_tile(mf, f, 0, ti, tj, As, Bs);
    for (int ti = 0; ti < tb; ++ti)
        for (int tj = 0; tj < tk; ++tj) gemm2_tile(mf, f, 1, ti, tj, As, Bs, tr);
    for (int ti = 0; ti < tk; ++ti)
        for (int tj = 0; tj < tb; ++tj) gemm2_tile(mf, f, 2, ti, tj, As, Bs);
#ifdef SANM_SF_PHASES
    if (tid == 0) {
        atomicAdd(&g_sf_phase[0], sf1 - sf0);
        atomicAdd(&g_sf_phase[1], sf2 - sf1);
        atomicAdd(&g_sf_phase[2], sf3 - sf2);
        atomicAdd(&g_sf_phase[3], sf4 - sf3);
        atomicAdd(&g_sf_phase[4], sf5 - sf4);
        atomicAdd(&g_sf_phase[5], wall_clock64() - sf5);
        atomicAdd(&g_sf_phase[6], (unsigned long long)(2 * tk * tb));
        atomicAdd(&g_sf_phase[7], 1ull);
    }
#endif
}

// Two blocking levels: after the panels [p0, p1) of an outer block the trailing matrix beyond it,
//   F[r, c] -= sum_{q in pivots of the block} L[r, q] U[q, c],   r, c >= p1 * NB,
// as one rank-(p1 - p0) * NB product of the solved panel tiles (panel_finalize_kernel) on the matrix cores.
// Tiles inside the (augmentation x augmentation) corner are never used and skipped.
__global__ void __launch_bounds__(256) block_gemm_kernel(MF_FACTOR_PARAMS, int p0, int p1) {
    MF_FACTOR_INIT
    const MfFrontDev f = mf.lfronts[level_begin + blockIdx.z];
    const int k = f.k, ld = f.ld, m = 2 * k, r0 = p1 * NB;
    if (p0 * NB >= k || r0 >= m) return;
    const int ti = blockIdx.y, tj = blockIdx.x, ext = m - r0;
    if (ti * GT >= ext || tj * GT >= ext) return;
    if (r0 + ti * GT >= k && r0 + tj * GT >= k) return;
    __shared__ double As[GK][GT + 1], Bs[GK][GT + 4];
    double* F = mf.front_store + f.off;
    const MatView A{F + (int64_t)r0 * ld, ld, ext, m};  // rows r0.., K = absolute column
    const MatView B{F + r0, ld, m, ext};                 // K = absolute row, columns r0..
    mfma_f64x4 acc[2][2];
    gemm_tile(A, B, ti, tj, p0 * NB, min(p1 * NB, k), As, Bs, acc);
    double* C = F + (int64_t)r0 * ld + r0;
    gemm_tile_sub_from(acc, C, ld, ti, tj, ext, ext);
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
        for (int q = 0; q < R; ++q) acc[q] = __builtin_fma(rc.a[u][q], tv, acc[q]);
    }
    // (fronts wider than the preloaded chunks -- thousands of columns on the large meshes, where these kernels are
    // bound by bandwidth: TAIL chunks per trip with unconditional index-clamped loads, so that TAIL * R loads are in
    // flight per lane instead of one; a plain loop with the range test around the load ran at 2.5 TB/s)
    constexpr int TAIL = 8;
    for (int c = c0 + 64 * U; c < cmax; c += 64 * TAIL) {
        double a[TAIL][R], tv[TAIL];
#pragma unroll
        for (int u = 0; u < TAIL; ++u) {
            const int cc = c + 64 * u;
            tv[u] = v[cc < cmax ? cc : csafe];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const bool ok = cc >= cbeg[q] && cc < cend[q];
                const double x = rowp[q][ok ? cc : 0];
                a[u][q] = ok ? x : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < TAIL; ++u)
#pragma unroll
            for (int q = 0; q < R; ++q) acc[q] = __builtin_fma(a[u][q], tv[u], acc[q]);
    }
}

// sum of the children's contributions to entry i of a front; the dissection tree is binary, and with the
// count known the loads go out together instead of one round trip per child
__device__ __forceinline__ double inbox_sum(const double* __restrict__ inbox, int nch, int m, int i) {
    if (nch == 2) return inbox[i] + inbox[(int64_t)m + i];
    double v = 0;
    for (int j = 0; j < nch; ++j) v += inbox[(int64_t)j * m + i];
    return v;
}

// (The solve kernels take their pointers as separate scalar arguments, not the MfDev record by value: built with
// -amdgpu-kernarg-preload-count the first 14 argument dwords are in SGPRs when a wavefront starts, and the
// descriptor load below is the first memory round trip of the launch instead of the second.)
struct SolveArgs {  // what the bodies below call mf.*
    const MfFrontDev* lfronts;
    const double* front_store;
    double* inbox_store;
    double* work;
    double* work2;
    const int32_t* upd_dst;
    const int32_t* bnd_idx;
};
template <int R, int U, bool LIST = false>  // (LIST: as in bwd_level_kernel)
__global__ void __launch_bounds__(256) fwd_level_kernel(const MfFrontDev* __restrict__ lfronts,
                                                        const double* __restrict__ front_store, double* inbox_store,
                                                        double* work, double* work2,
                                                        const int32_t* __restrict__ upd_dst, int phase) {
    // phase 0: the whole sweep, [z; upd] = [L11^-1; F[B,A]] t.  Two-phase levels (Level::two_phase: F[B,A] holds -L21):
    // phase 1: z = L11^-1 t (rows < k), phase 2, a launch later: upd = F[B,A] z (rows >= k, the vector is z)
    const SolveArgs mf{lfronts, front_store, inbox_store, work, work2, upd_dst, nullptr};
    const MfSolveBlock* blocks = reinterpret_cast<const MfSolveBlock*>(lfronts);
    const MfFrontDev f = LIST ? blocks[blockIdx.x].f : mf.lfronts[blockIdx.y];
    const int k = f.k, m = phase == 1 ? k : f.m;  // (m: end of this launch's rows)
    const int rb = (phase == 2 ? k : 0) + (LIST ? blocks[blockIdx.x].bx : (int)blockIdx.x) * (4 * R);
    if (rb >= m) return;
    extern __shared__ double vs[];  // t = w_own + children's contributions (k entries); phase 2: z
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
            pre[q] = inbox_sum(inbox, f.nch, f.m, r[q]);
        }
    }
    const int kneed = min(k, rb + 4 * R);  // own rows read t[0..r] only
    if (phase == 2)
        for (int c = tid; c < k; c += 256) vs[c] = mf.work2[f.own_start + c];
    else
        for (int c = tid; c < kneed; c += 256) vs[c] = mf.work[f.own_start + c] + inbox_sum(inbox, f.nch, f.m, c);
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

// Forward level kernel for levels whose rows are short (k <= 4 G): G lanes per row, 64 / G rows per wavefront,
// instead of a whole wavefront on a row of a few dozen entries.  The lower levels of the tree have hundreds of
// fronts with a few dozen pivots and a boundary several times that: with one row per wavefront they are
// thousands of workgroups of mostly idle lanes and the launch is bound by dispatch.
template <int G, int R>
__global__ void __launch_bounds__(256) fwd_level_sub_kernel(const MfFrontDev* __restrict__ lfronts,
                                                            const double* __restrict__ front_store,
                                                            double* inbox_store, double* work, double* work2,
                                                            const int32_t* __restrict__ upd_dst) {
    const SolveArgs mf{lfronts, front_store, inbox_store, work, work2, upd_dst, nullptr};
    const MfFrontDev f = mf.lfronts[blockIdx.y];
    const int m = f.m, k = f.k;
    constexpr int RPB = 256 / G * R;  // rows per workgroup: R per lane group
    const int rb = blockIdx.x * RPB;
    if (rb >= m) return;
    extern __shared__ double vs[];
    const int tid = threadIdx.x, sub = tid % G, r0 = rb + tid / G * R;
    const double* inbox = mf.inbox_store + f.inbox_off;
    const double* rowp[R];
    int cend[R], dst[R];
    double a[R][4], pre[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {  // unconditional clamped loads, in flight across the staging below
        const int r = r0 + q;
        const bool live = r < m;
        const int pr = r < k ? r : r + k;
        rowp[q] = mf.front_store + f.off + (int64_t)(live ? pr : 0) * f.ld + k;
        cend[q] = !live ? 0 : (r < k ? r + 1 : k);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = sub + G * u;
            const double v = rowp[q][min(c, k - 1)];
            a[q][u] = c < cend[q] ? v : 0.0;
        }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int r = r0 + q;
        pre[q] = 0;
        dst[q] = -1;
        if (r >= k && r < m) {
            dst[q] = mf.upd_dst[f.bnd_off + r - k];
            pre[q] = inbox_sum(inbox, f.nch, m, r);
        }
    }
    const int kneed = min(k, rb + RPB);
    for (int c = tid; c < kneed; c += 256) vs[c] = mf.work[f.own_start + c] + inbox_sum(inbox, f.nch, m, c);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R; ++q) {
        double acc = 0;
#pragma unroll
        // (entries beyond the row's extent carry a zero factor, but the vector entry they meet must still be an
        // initialised one: whatever an earlier kernel left in the LDS may be a NaN, and 0 * NaN is not 0)
        for (int u = 0; u < 4; ++u) acc = __builtin_fma(a[q][u], vs[min(sub + G * u, kneed - 1)], acc);
        for (int c = sub + 4 * G; c < cend[q]; c += G) acc = __builtin_fma(rowp[q][c], vs[c], acc);
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, G);
        const int r = r0 + q;
        if (sub == 0 && r < m) {
            if (r < k) mf.work2[f.own_start + r] = acc;
            else mf.inbox_store[dst[q]] = acc + pre[q];
        }
    }
}

// Forward level kernel for levels whose boundary operator is stored transposed (Level::fwd_t): the first `nzb` blocks
// of a front take its pivot rows, z = L11^-1 t, G lanes per row like fwd_level_sub_kernel; the others take 64 boundary
// rows each, a THREAD per row: upd_j = sum_i FT[i][j] t[i] with FT = (F[B,A])' in the F[P,B] slot -- every load of a
// wavefront is 512 contiguous bytes of one of the k long rows, no reduction across lanes --, the k terms dealt to the
// four wavefronts in contiguous quarters whose partial sums wavefront 0 adds in order.
template <int G, int R, bool LIST = false>
__global__ void __launch_bounds__(256) fwd_level_tr_kernel(const MfFrontDev* __restrict__ lfronts,
                                                           const double* __restrict__ front_store, double* inbox_store,
                                                           double* work, double* work2,
                                                           const int32_t* __restrict__ upd_dst, int nzb) {
    const MfSolveBlock* blocks = reinterpret_cast<const MfSolveBlock*>(lfronts);  // (LIST: bx < nzb pivot rows, else boundary)
    const MfFrontDev f = LIST ? blocks[blockIdx.x].f : lfronts[blockIdx.y];
    const int bx = LIST ? blocks[blockIdx.x].bx : (int)blockIdx.x;
    const int m = f.m, k = f.k, b = m - k;
    extern __shared__ double vs[];  // t (k entries), then the partial sums [4][64]
    const int tid = threadIdx.x;
    const double* inbox = inbox_store + f.inbox_off;
    if (bx < nzb) {
        constexpr int RPB = 256 / G * R;
        const int rb = bx * RPB;
        if (rb >= k) return;
        const int sub = tid % G, r0 = rb + tid / G * R;
        const double* rowp[R];
        int cend[R];
        double a[R][4];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int r = r0 + q;
            const bool live = r < k;
            rowp[q] = front_store + f.off + (int64_t)(live ? r : 0) * f.ld + k;
            cend[q] = live ? r + 1 : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = sub + G * u;
                const double v = rowp[q][min(c, k - 1)];
                a[q][u] = c < cend[q] ? v : 0.0;
            }
        }
        const int kneed = min(k, rb + RPB);
        for (int c = tid; c < kneed; c += 256) vs[c] = work[f.own_start + c] + inbox_sum(inbox, f.nch, m, c);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < R; ++q) {
            double acc = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_fma(a[q][u], vs[min(sub + G * u, kneed - 1)], acc);
            for (int c = sub + 4 * G; c < cend[q]; c += G) acc = __builtin_fma(rowp[q][c], vs[c], acc);
#pragma unroll
            for (int off = G / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, G);
            const int r = r0 + q;
            if (sub == 0 && r < k) work2[f.own_start + r] = acc;
        }
        return;
    }
    const int jb = (bx - nzb) * 64;
    if (jb >= b) return;
    const int lane = tid & 63, w = tid >> 6;
    const int j = jb + lane, jc = min(j, b - 1);
    // the row's own loads first: destination slot and the children's entries of this boundary row
    int dst = -1;
    double pre = 0;
    if (w == 0 && j < b) {
        dst = upd_dst[f.bnd_off + j];
        pre = inbox_sum(inbox, f.nch, m, k + j);
    }
    const int i0 = (int)((int64_t)k * w / 4), i1 = (int)((int64_t)k * (w + 1) / 4);
    const double* col = front_store + f.off + 2 * k + jc;  // FT[i][j] at col[i * ld]
    constexpr int UN = 8;
    double av[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) av[u] = col[(int64_t)min(i0 + u, k - 1) * f.ld];  // (in flight across the staging)
    for (int c = tid; c < k; c += 256) vs[c] = work[f.own_start + c] + inbox_sum(inbox, f.nch, m, c);
    __syncthreads();
    double acc = 0;
    for (int i = i0; i < i1; i += UN) {
        double nx[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) nx[u] = col[(int64_t)min(i + UN + u, k - 1) * f.ld];
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (i + u < i1) acc = __builtin_fma(av[u], vs[i + u], acc);
#pragma unroll
        for (int u = 0; u < UN; ++u) av[u] = nx[u];
    }
    double* part = vs + k;
    part[w * 64 + lane] = acc;
    __syncthreads();
    if (w == 0 && j < b) inbox_store[dst] = (((part[lane] + part[64 + lane]) + part[128 + lane]) + part[192 + lane]) + pre;
}

// LIST: blockIdx.x indexes a flat list of the (front, row block) pairs that exist (MfSolveBlock) instead of the box grid
template <int R, int U, bool LIST = false>
__global__ void __launch_bounds__(256) bwd_level_kernel(const MfFrontDev* __restrict__ lfronts,
                                                        const double* __restrict__ front_store, double* work,
                                                        double* work2,
                                                        const int32_t* __restrict__ bnd_idx, int phase) {
    // phase 0: the whole sweep, x_own = [U11^-1, F[A,B]] [z; x_bnd].  Two-phase levels (F[A,B] holds -U12):
    // phase 1: z += F[A,B] x_bnd (in work2), phase 2, a launch later: x_own = U11^-1 z
    const SolveArgs mf{lfronts, front_store, nullptr, work, work2, nullptr, bnd_idx};
    const MfSolveBlock* blocks = reinterpret_cast<const MfSolveBlock*>(lfronts);
    const MfFrontDev f = LIST ? blocks[blockIdx.x].f : mf.lfronts[blockIdx.y];
    const int k = f.k, m = phase == 2 ? k : f.m;  // (m: end of this launch's columns)
    const int rb = (LIST ? blocks[blockIdx.x].bx : (int)blockIdx.x) * (4 * R);
    if (rb >= k || (phase == 1 && f.m == k)) return;
    const int safe = phase == 1 ? k : rb;  // a staged entry of vs (read, times zero, by lanes outside the range)
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
        cbeg[q] = phase == 1 ? k : r[q];
        cend[q] = live ? m : 0;
        acc[q] = 0;
    }
    const int c0 = (phase == 1 ? k : rb + wv * R) + lane;
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
    for (int c = (phase == 1 ? k : rb) + tid; c < m; c += 256) vs[c] = c < k ? mf.work2[f.own_start + c] : mf.work[bi[c - k]];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = c0 + 64 * u;
        const double tv = vs[c < m ? c : safe];
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] = __builtin_fma(a[u][q], tv, acc[q]);
    }
    {
        constexpr int TAIL = 8;  // as in rows_consume: TAIL * R loads in flight per lane on the wide fronts
        for (int c = c0 + 64 * U; c < m; c += 64 * TAIL) {
            double av[TAIL][R], tv[TAIL];
#pragma unroll
            for (int u = 0; u < TAIL; ++u) {
                const int cc = c + 64 * u;
                tv[u] = vs[cc < m ? cc : safe];
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const bool ok = cc >= cbeg[q] && cc < cend[q];
                    const double x = rowp[q][ok ? (cc < k ? cc : cc + k) : 0];
                    av[u][q] = ok ? x : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < TAIL; ++u)
#pragma unroll
                for (int q = 0; q < R; ++q) acc[q] = __builtin_fma(av[u][q], tv[u], acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const double v = wave_sum(acc[q]);
        if (lane == 0 && r[q] < k) {
            if (phase == 1)
                mf.work2[f.own_start + r[q]] += v;
            else
                mf.work[f.own_start + r[q]] = v;
        }
    }
}

// ---- level kernels for fronts whose vectors do not fit the LDS (tens of thousands of rows) ---------
// Such levels are bandwidth-bound, not latency-bound: plain wave-per-row mat-vecs on operands in HBM, with a
// small pre-pass that folds the children's inbox slots into the front's own part of the work vector.
__global__ void __launch_bounds__(256) fwd_prep_kernel(MfDev mf, int level_begin) {
    const MfFrontDev f = mf.lfronts[level_begin + blockIdx.y];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= f.k) return;
    double v = mf.work[f.own_start + c];
    for (int j = 0; j < f.nch; ++j) v += mf.inbox_store[f.inbox_off + (int64_t)j * f.m + c];
    mf.work[f.own_start + c] = v;
}
__device__ __forceinline__ double wave_dot_global(const double* __restrict__ row, const double* __restrict__ v,
                                                  int cbeg, int cend, int lane) {
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int c = cbeg + lane;
    for (; c + 448 < cend; c += 512) {  // 16 loads in flight per lane
        const double r0 = row[c], r1 = row[c + 64], r2 = row[c + 128], r3 = row[c + 192], r4 = row[c + 256],
                     r5 = row[c + 320], r6 = row[c + 384], r7 = row[c + 448];
        const double v0 = v[c], v1 = v[c + 64], v2 = v[c + 128], v3 = v[c + 192], v4 = v[c + 256], v5 = v[c + 320],
                     v6 = v[c + 384], v7 = v[c + 448];
        a0 = __builtin_fma(r4, v4, __builtin_fma(r0, v0, a0));
        a1 = __builtin_fma(r5, v5, __builtin_fma(r1, v1, a1));
        a2 = __builtin_fma(r6, v6, __builtin_fma(r2, v2, a2));
        a3 = __builtin_fma(r7, v7, __builtin_fma(r3, v3, a3));
    }
    for (; c < cend; c += 64) a0 = __builtin_fma(row[c], v[c], a0);
    return wave_sum((a0 + a1) + (a2 + a3));
}
__global__ void __launch_bounds__(256) fwd_big_kernel(MfDev mf, int level_begin, int phase) {  // (phases: fwd_level_kernel)
    const MfFrontDev f = mf.lfronts[level_begin + blockIdx.y];
    const int m = f.m, k = f.k, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= m || (phase == 1 && r >= k) || (phase == 2 && r < k)) return;
    const int pr = r < k ? r : r + k;
    const double* row = mf.front_store + f.off + (int64_t)pr * f.ld + k;
    double acc = wave_dot_global(row, (phase == 2 ? mf.work2 : mf.work) + f.own_start, 0, r < k ? r + 1 : k, lane);
    if (lane == 0) {
        if (r < k) {
            mf.work2[f.own_start + r] = acc;
        } else {
            for (int j = 0; j < f.nch; ++j) acc += mf.inbox_store[f.inbox_off + (int64_t)j * m + r];
            mf.inbox_store[mf.upd_dst[f.bnd_off + r - k]] = acc;
        }
    }
}
__global__ void __launch_bounds__(256) bwd_big_kernel(MfDev mf, int level_begin, int phase) {  // (phases: bwd_level_kernel)
    const MfFrontDev f = mf.lfronts[level_begin + blockIdx.y];
    const int m = f.m, k = f.k, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= k) return;
    const double* row = mf.front_store + f.off + (int64_t)(k + r) * f.ld;
    double acc = phase == 1 ? 0.0 : wave_dot_global(row, mf.work2 + f.own_start, r, k, lane);
    if (phase != 2) {
        const double* rowb = row + 2 * k;
        const int32_t* bi = mf.bnd_idx + f.bnd_off;
        double a0 = 0, a1 = 0;
        int c = lane;
        for (; c + 64 < m - k; c += 128) {
            a0 = __builtin_fma(rowb[c], mf.work[bi[c]], a0);
            a1 = __builtin_fma(rowb[c + 64], mf.work[bi[c + 64]], a1);
        }
        if (c < m - k) a0 = __builtin_fma(rowb[c], mf.work[bi[c]], a0);
        acc += wave_sum(a0 + a1);
    }
    if (lane == 0) {
        if (phase == 1)
            mf.work2[f.own_start + r] += acc;
        else
            mf.work[f.own_start + r] = acc;
    }
}

// ---- backward level kernel for fronts with long rows (round 4) --------------------------------------------------
// A backward row is a dot product over the whole front width m.  bwd_level_kernel gives a row to one wavefront and
// stages all m entries of the input vector in LDS per workgroup: on fronts of thousands of rows that is 70 KB of LDS
// (two workgroups per CU) and, on the chains of ~1000-pivot fronts at the top of a big tree (multifrontal.cpp,
// split_big_fronts), k / 4 = 240 workgroups for 50 MB of operator -- 2 TB/s where the forward kernel of the same level
// (m rows of k entries, 8 KB of LDS) streams at 4-5.  Here the four wavefronts of a workgroup share R rows: wavefront
// w takes the 64-column chunks w, w + 4, w + 8, ... of every row (the workgroup reads 2 KB of a row at a time),
// the vector comes straight from the work vectors (the boundary part through bnd_idx: L2 hits after the first row),
// no LDS but the 4 x R partial sums, which wavefront 0 adds in the order w = 0..3.  k / R workgroups of 4 wavefronts.
template <int R>
__global__ void __launch_bounds__(256) bwd_wide_kernel(const MfFrontDev* __restrict__ lfronts,
                                                       const double* __restrict__ front_store, double* work,
                                                       double* work2,
                                                       const int32_t* __restrict__ bnd_idx, int phase) {
    // (phases as in bwd_level_kernel: 1 = boundary columns only, added to z in work2; 2 = pivot columns only)
    const MfFrontDev f = lfronts[blockIdx.y];
    const int k = f.k, m = phase == 2 ? k : f.m;
    const int rb = blockIdx.x * R;
    if (rb >= k) return;
    __shared__ double part[4][R];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double* rowp[R];
    double acc[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int r = rb + q;
        rowp[q] = front_store + f.off + (int64_t)(k + (r < k ? r : rb)) * f.ld;
        acc[q] = 0;
    }
    const int32_t* bi = bnd_idx + f.bnd_off;
    const double* z = work2 + f.own_start;
    constexpr int TAIL = 8;
    // virtual column c of a row: physical column c (c < k: U11^-1, from the diagonal on) or c + k (boundary block)
    for (int c = (phase == 1 ? k : rb) + 64 * wv + lane; c < m; c += 256 * TAIL) {
        double av[TAIL][R], tv[TAIL];
#pragma unroll
        for (int u = 0; u < TAIL; ++u) {
            const int cc = c + 256 * u;
            const bool in = cc < m;
            const int cs = in ? cc : rb;
            tv[u] = cs < k ? z[cs] : work[bi[cs - k]];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const bool ok = in && cc >= rb + q && rb + q < k;
                const double x = rowp[q][ok ? (cc < k ? cc : cc + k) : 0];
                av[u][q] = ok ? x : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < TAIL; ++u)
#pragma unroll
            for (int q = 0; q < R; ++q) acc[q] = __builtin_fma(av[u][q], tv[u], acc[q]);
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const double v = wave_sum(acc[q]);
        if (lane == 0) part[wv][q] = v;
    }
    __syncthreads();
    if (tid < R && rb + tid < k) {
        const double v = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
        if (phase == 1)
            work2[f.own_start + rb + tid] += v;
        else
            work[f.own_start + rb + tid] = v;
    }
}

// ---- merged top of the tree (MfSchedule::Top) -----------------------------------------------------------
// The last two levels of the tree -- the root and the fronts below it -- cost four dependent launches per solve
// (two forward, two backward) that move a few megabytes each.  After the factorisation their solve operators are
// multiplied out into ONE dense matrix M (n_T x n_T, n_T = the pivots of those fronts), so that the top of every
// solve is a single mat-vec x_T = M t_T:
//   M_RR  = U_R^-1 L_R^-1                                   (root R)
//   M_Rc  = M_RR[:, rel_c] G_c,   G_c = -L21 L11^-1 of front c  (its boundary rows sit at rel_c in the root)
//   M_cR  = E_c M_RR[rel_c, :],   E_c = -U11^-1 U12
//   M_cc' = E_c M_Rc'[rel_c, :]  (+ U_c^-1 L_c^-1 for c' = c)
// computed in two dependent batches: P_c = E_c U_R^-1[rel_c, :], Q_c = L_R^-1[:, rel_c] G_c and the diagonal
// blocks first, then M_cR = P_c L_R^-1, M_Rc = U_R^-1 Q_c, M_cc' = P_c Q_c' -- two GEMM launches per factorisation
// against 3 x N launches saved in the N solves that follow.
template <bool IDX>
__global__ void __launch_bounds__(256) top_gemm_kernel(const TopGemm* __restrict__ g) {
    const TopGemm d = g[blockIdx.z];
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (ti * GT >= d.M || tj * GT >= d.N) return;
    __shared__ double As[GK][GT + 1], Bs[GK][GT + 4];
    mfma_f64x4 acc[2][2];
    gemm_tile<IDX>(d.A, d.B, ti, tj, 0, d.K, As, Bs, acc);
    gemm_tile_foreach(acc, [&](int i, int j, double v) {
        const int r = ti * GT + i, c = tj * GT + j;
        if (r < d.M && c < d.N) {
            double* dst = d.C + (int64_t)r * d.ldc + c;
            *dst = d.acc ? *dst + v : v;
        }
    });
}


// x_T = M t_T: every workgroup assembles t_T in LDS from the lists of MfSchedule::Top (own right-hand sides plus
// what the fronts below the block left in the inboxes, W slots per entry in a fixed order), its 4 wavefronts then
// take R rows of M each.  Everything the kernel needs comes as arguments: its first memory round trip already
// fetches data.  The solution goes to a place of its own (xout = work + n): other workgroups of the launch may
// still be reading the right-hand side.
template <int R, int W>
__global__ void __launch_bounds__(256) top_solve_kernel(const double* __restrict__ M,
                                                        const int32_t* __restrict__ wsrc,
                                                        const int32_t* __restrict__ ell,
                                                        const double* __restrict__ inbox_store,
                                                        const double* __restrict__ work, double* __restrict__ xout,
                                                        int n) {
    extern __shared__ double t[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row0 = blockIdx.x * (4 * R) + wv * R;
    // the rows' first chunks are requested before the staging (they do not depend on it)
    constexpr int PRE = 4;
    double a[PRE][R];
    const double* rowp[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        rowp[q] = M + (int64_t)min(row0 + q, n - 1) * n;
#pragma unroll
        for (int u = 0; u < PRE; ++u) a[u][q] = rowp[q][min(lane + 64 * u, n - 1)];
    }
    for (int i = tid; i < n; i += 256) {
        int sl[W];
#pragma unroll
        for (int s = 0; s < W; ++s) sl[s] = ell[(int64_t)s * n + i];
        double v = work[wsrc[i]];
#pragma unroll
        for (int s = 0; s < W; ++s) v += inbox_store[sl[s]];
        t[i] = v;
    }
    __syncthreads();
    double acc[R];
#pragma unroll
    for (int q = 0; q < R; ++q) acc[q] = 0;
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const int c = lane + 64 * u;
        const double tv = c < n ? t[c] : 0.0;
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] = __builtin_fma(a[u][q], tv, acc[q]);
    }
    constexpr int TAIL = 8;
    for (int c = lane + 64 * PRE; c < n; c += 64 * TAIL) {
        double av[TAIL][R], tv[TAIL];
#pragma unroll
        for (int u = 0; u < TAIL; ++u) {
            const int cc = c + 64 * u;
            tv[u] = cc < n ? t[cc] : 0.0;
#pragma unroll
            for (int q = 0; q < R; ++q) av[u][q] = rowp[q][min(cc, n - 1)];
        }
#pragma unroll
        for (int u = 0; u < TAIL; ++u)
#pragma unroll
            for (int q = 0; q < R; ++q) acc[q] = __builtin_fma(av[u][q], tv[u], acc[q]);
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const double v = wave_sum(acc[q]);
        const int row = row0 + q;
        if (lane == 0 && row < n) xout[row] = v;  // (work[n + row]: see MfSchedule::Top::bnd_x)
    }
}

}  // namespace mfk
}  // namespace sanm_hip
