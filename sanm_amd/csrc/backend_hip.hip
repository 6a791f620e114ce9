// HIP backend: the product execution path (gfx950 / MI355X).  One in-order stream; DESIGN.md has the
// measurements behind every choice below.
//
//  * taylor_pass_kernel<MODE, W>  the graph interpreter, one lane per tet on SoA state (every load / store of a
//                         wavefront is one contiguous 512-byte segment); one instantiation per pass; BIAS passes
//                         of higher orders run 4 wavefronts per 64 tets (convolution split, tet_ops.h); the GRAD
//                         pass one (tet, Jacobian row) pair per lane.
//  * gather_rows_kernel   remap_out: 16 lanes per unknown, ~45 gathered entries, two index -> value chains in
//                         flight per lane.
//  * assemble_kernel      8 lanes per CSR non-zero over a fixed gather list (incl. the 1e-9 drop rule).
//  * sanity_check_kernel  the per-order check A x_i = -(t_i g_t + b_i) and x_1 . x_i fused with its reductions.
//  * reductions           grid_commit: per-workgroup partials written through, agent-scope ticket, the last
//                         workgroup combines them in a fixed order and writes to pinned host memory (or to device
//                         memory for the kernels of the order loop / Pade sweep that consume scalars on the device).
//  * mf_factor / mf_solve the multifrontal LU (mf_kernels.h): one launch per 32-wide panel (panel solves, tile
//                         update and look-ahead tile LU), fp64-MFMA GEMMs for the boundary blocks and for the
//                         second blocking level of large fronts; solves = one mat-vec launch per level and direction.
//  * graphs               the Pade basis sweep is captured once and replayed (graph_capture_*).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "backend.h"
#include "graph.h"
#include "mf_kernels.h"
#include "red_ops.h"
#include "row_ops.h"
#include "host_parallel.h"
#include "rtc.h"
#include "tet_ops.h"
#include "vecprog.h"

namespace sanm_hip {

#define HIP_CHECK(expr)                                                              \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess)                                                        \
            sanm_throw(SANM_ERR_HIP, "HIP error %s at %s:%d: %s", hipGetErrorName(e_), \
                       __FILE__, __LINE__, hipGetErrorString(e_));                   \
    } while (0)

namespace {

// The current-order values travel from operator to operator through LDS
// (cur_size doubles per lane, lane-interleaved: conflict-free), not through HBM.
// blockDim.x / 64 wavefronts share the workgroup's 64 tets (tet_ops.h: convolution
// split); their exchange buffer follows the scratch.
// One instantiation per pass so that each carries only its own branch of every operator
// (register pressure decides how many wavefronts fit a SIMD); W = wavefronts per SIMD the
// register allocation must leave room for.
template <int MODE, int W>
__global__ void __launch_bounds__(256, W) taylor_pass_kernel(ProgramDev P0, const OpDesc* __restrict__ ops,
                                                             const VarDesc* __restrict__ vars, int order,
                                                             const double* __restrict__ xvec) {
    // the operator / variable records arrive as separate read-only parameters: known not to alias the arena
    // stores, their (wave-uniform) loads go through the scalar cache instead of three dependent vector-memory
    // round trips per operator
    ProgramDev P = P0;
    P.ops = ops;
    P.vars = vars;
    // one batch of independent scalar loads brings every record into the scalar cache; the walk below
    // would otherwise take two dependent misses (operator, then its variables) per operator
    {
        const int* w = reinterpret_cast<const int*>(ops);
        int acc = 0;
        for (int i = 0; i < P.desc_lines; i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc |= w[(i + j) * 16];
        }
        if (acc == 0x7fffffff && order < -1) return;  // never true (type ids are small): keeps the loads
    }
    extern __shared__ double cur_lds[];
    const int lane = threadIdx.x & 63;
    // wave-uniform by construction; tell the compiler so that the slice bounds stay scalar
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nparts = blockDim.x >> 6;
    int64_t tet = (int64_t)blockIdx.x * 64 + lane;
    // the GRAD pass runs one (tet, Jacobian row) pair per lane: the row is blockIdx.y
    if (tet < P.T)
        exec_program_tet(P, MODE, MODE == PASS_GRAD ? (int)blockIdx.y : order, tet, xvec, cur_lds + lane, 64, part,
                         nparts, cur_lds + (int64_t)P.cur_size * 64 + lane);
}

// remap_out: GATHER_LANES lanes per output row (~45 gathered entries each), shuffle reduce.  The index ->
// value loads of an entry are dependent, so each lane keeps the entries of two strides in flight.
constexpr int ROW_LANES = 8;
constexpr int GATHER_LANES = 16;
// This lane's share of sum_p coef[p] * src[idx[p]] over [p0, e), LANES lanes per row: index -> value is a dependent
// pair of loads, so 4 entries per lane go out together (clamped, unconditional loads: nothing for the compiler to
// serialise) -- a row of up to 4 * LANES entries costs two memory round trips.
template <int LANES, typename I>
__device__ __forceinline__ double row_dot(const I* __restrict__ idx, const double* __restrict__ coef,
                                          const double* __restrict__ src, uint32_t p0, uint32_t e, int sub) {
    double s0 = 0, s1 = 0;
    for (uint32_t base = p0; base < e; base += 4 * LANES) {
        I i[4];
        double c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t q = base + sub + u * LANES;
            const uint32_t qq = q < e ? q : p0;
            i[u] = idx[qq];
            const double cv = coef[qq];
            c[u] = q < e ? cv : 0.0;
        }
        s0 += c[0] * src[i[0]];
        s1 += c[1] * src[i[1]];
        s0 += c[2] * src[i[2]];
        s1 += c[3] * src[i[3]];
    }
    return s0 + s1;
}
__global__ void __launch_bounds__(256) gather_rows_kernel(SparseRowsDev R, const double* __restrict__ src,
                                                          double* __restrict__ dst,
                                                          const int32_t* __restrict__ perm,
                                                          double* __restrict__ dst2) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = gid / GATHER_LANES;
    int sub = gid % GATHER_LANES;
    double s = 0;
    if (i < R.nrows) s = row_dot<GATHER_LANES>(R.idx, R.coef, src, R.ptr[i], R.ptr[i + 1], sub);
    for (int off = GATHER_LANES / 2; off > 0; off >>= 1) s += __shfl_down(s, off, GATHER_LANES);
    if (i < R.nrows && sub == 0) {
        dst[i] = s;
        if (perm) dst2[perm[i]] = s;
    }
}

// Gather lists of the assembly, built on the device (backend.h: AssemblyDev): one workgroup per row of A.  The row's
// triples (p, m, q) are enumerated in order from the two remap tables into LDS -- column, index into the Jacobian
// blocks, c_out * c_in --, then every non-zero of the row (one thread each, plus one for the t column) walks the staged
// triples: MODE 0 counts the ones with its column, MODE 1 writes them to its slice of the list, in order.
constexpr int ASM_THREADS = 128;
constexpr int ASM_PAIRS = 1024;  // (entry of the remap_out row, m) pairs per block
constexpr int ASM_CHUNK = 1024;  // staged triples per pass
constexpr int ASM_NZT = 2;       // non-zeros per thread and sweep
template <int MODE>
__global__ void __launch_bounds__(ASM_THREADS) asm_list_kernel(AssemblyDev A, uint32_t* __restrict__ cnt,
                                                               uint32_t* __restrict__ tcnt, uint32_t* __restrict__ out_jidx,
                                                               double* __restrict__ out_coef, uint32_t* __restrict__ out_tjidx,
                                                               double* __restrict__ out_tcoef, int row_step) {
    const int64_t i = (int64_t)blockIdx.x * row_step;  // (3: the first row of every triple only, AssemblyDev::triples)
    __shared__ uint32_t tcol[ASM_CHUNK];
    __shared__ uint32_t tjx[ASM_CHUNK];
    __shared__ double tcf[ASM_CHUNK];
    __shared__ uint32_t poff[ASM_PAIRS + 1];
    const int tid = threadIdx.x;
    const uint32_t p0 = A.ro_ptr[i];
    const int64_t npairs = (int64_t)(A.ro_ptr[i + 1] - p0) * A.idim;
    const uint32_t c0row = A.rowptr[i];
    const int nnz_row = (int)(A.rowptr[i + 1] - c0row);
    const int ntarget = nnz_row + (A.has_t ? 1 : 0);  // (the last target is the t column)
    for (int nz0 = 0; nz0 < ntarget; nz0 += ASM_THREADS * ASM_NZT) {
        uint32_t seen[ASM_NZT], target[ASM_NZT], base[ASM_NZT];
#pragma unroll
        for (int z = 0; z < ASM_NZT; ++z) {
            seen[z] = 0;
            const int t = nz0 + tid + ASM_THREADS * z;
            target[z] = t < nnz_row ? A.col[c0row + t] : (t < ntarget ? (uint32_t)A.n : 0xffffffffu);
            base[z] = 0;
            if (MODE == 1) base[z] = t < nnz_row ? A.aptr[c0row + t] : (t < ntarget ? A.tptr[i] : 0u);
        }
        for (int64_t pb = 0; pb < npairs; pb += ASM_PAIRS) {
            const int np = (int)min((int64_t)ASM_PAIRS, npairs - pb);
            // triples per pair (own batch items only), then their offsets
            for (int q = tid; q < np; q += ASM_THREADS) {
                const int64_t pair = pb + q;
                const uint32_t e = A.ro_idx[p0 + pair / A.idim];
                const int64_t b = e / A.odim;
                const int64_t irow = b * A.idim + pair % A.idim;
                const bool mine = b >= A.tet_begin && b < A.tet_end;
                poff[q] = mine ? A.ri_ptr[irow + 1] - A.ri_ptr[irow] : 0u;
            }
            __syncthreads();
            if (tid < 64) {  // exclusive scan by one wavefront, 64 pairs at a time
                uint32_t carry = 0;
                for (int b0 = 0; b0 < np; b0 += 64) {
                    const int q = b0 + tid;
                    const uint32_t v = q < np ? poff[q] : 0u;
                    uint32_t incl = v;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const uint32_t o = __shfl_up(incl, off, 64);
                        if (tid >= off) incl += o;
                    }
                    if (q < np) poff[q] = carry + incl - v;
                    carry += __shfl(incl, 63, 64);
                }
                if (tid == 0) poff[np] = carry;
            }
            __syncthreads();
            const uint32_t total = poff[np];
            for (uint32_t cb = 0; cb < total; cb += ASM_CHUNK) {
                for (int q = tid; q < np; q += ASM_THREADS) {
                    const uint32_t off = poff[q], cn = poff[q + 1] - off;
                    if (cn == 0 || off + cn <= cb || off >= cb + ASM_CHUNK) continue;
                    const int64_t pair = pb + q;
                    const uint32_t pe = p0 + (uint32_t)(pair / A.idim);
                    const int m = (int)(pair % A.idim);
                    const uint32_t e = A.ro_idx[pe];
                    const int64_t b = e / A.odim;
                    const int o = (int)(e % A.odim);
                    const double c_out = A.ro_coef[pe];
                    const uint32_t jx = (uint32_t)(((b - A.tet_begin) * A.odim + o) * A.idim + m);
                    const uint32_t r0 = A.ri_ptr[b * A.idim + m];
                    for (uint32_t r = 0; r < cn; ++r) {
                        const uint32_t kk = off + r;
                        if (kk < cb || kk >= cb + ASM_CHUNK) continue;
                        tcol[kk - cb] = A.ri_idx[r0 + r];
                        if (MODE == 1) {
                            tjx[kk - cb] = jx;
                            tcf[kk - cb] = c_out * A.ri_coef[r0 + r];
                        }
                    }
                }
                __syncthreads();
                const int nn = (int)min((uint32_t)ASM_CHUNK, total - cb);
#pragma unroll
                for (int z = 0; z < ASM_NZT; ++z) {
                    if (target[z] == 0xffffffffu) continue;
                    const bool is_t = target[z] == (uint32_t)A.n;
                    for (int kk = 0; kk < nn; ++kk)
                        if (tcol[kk] == target[z]) {
                            if (MODE == 1) {
                                const uint32_t w = base[z] + seen[z];
                                (is_t ? out_tjidx : out_jidx)[w] = tjx[kk];
                                (is_t ? out_tcoef : out_coef)[w] = tcf[kk];
                            }
                            ++seen[z];
                        }
                }
                __syncthreads();
            }
        }
        if (MODE == 0) {
#pragma unroll
            for (int z = 0; z < ASM_NZT; ++z) {
                const int t = nz0 + tid + ASM_THREADS * z;
                if (t < nnz_row) cnt[c0row + t] = seen[z];
                else if (t < ntarget) tcnt[i] = seen[z];
            }
        }
    }
}


// CSR assembly: ROW_LANES lanes per non-zero (~25 contributions each)
struct AsmList {
    const uint32_t* ptr;
    const uint32_t* jidx;
    const double* coef;
    int64_t nslots;
};
__global__ void __launch_bounds__(256) assemble_kernel(AsmList A, const double* __restrict__ jac,
                                                       double* __restrict__ val) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t s = gid / ROW_LANES;
    int sub = gid % ROW_LANES;
    double v = 0;
    if (s < A.nslots) {
        const uint32_t p0 = A.ptr[s], e = A.ptr[s + 1];
        for (uint32_t base = p0; base < e; base += 4 * ROW_LANES) {  // 4 index -> value chains in flight
            uint32_t j[4];
            double c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t q = base + sub + u * ROW_LANES;
                const uint32_t qq = q < e ? q : p0;
                j[u] = A.jidx[qq];
                const double cv = A.coef[qq];
                c[u] = q < e ? cv : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double t = c[u] * jac[j[u]];
                if (fabs(t) >= 1e-9) v += t;  // libsanm/sparse_solver.cpp:291-293
            }
        }
    }
    for (int off = ROW_LANES / 2; off > 0; off >>= 1) v += __shfl_down(v, off, ROW_LANES);
    if (s < A.nslots && sub == 0) val[s] = v;
}

// The same for rows in triples (AssemblyDev::triples): a slot is a non-zero of a row 3u; its list serves the same
// non-zero of rows 3u+1 and 3u+2 with the Jacobian index shifted by `shift` and 2 `shift` -- every value is the sum it
// is in assemble_kernel, term by term.
__global__ void __launch_bounds__(256) assemble3_kernel(AsmList A, const uint32_t* __restrict__ tslot_p,
                                                        const uint32_t* __restrict__ tslot_len, int shift,
                                                        const double* __restrict__ jac, double* __restrict__ val) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t s = gid / ROW_LANES;
    int sub = gid % ROW_LANES;
    double v[3] = {0, 0, 0};
    uint32_t p = 0, len = 0;
    if (s < A.nslots) {
        p = tslot_p[s];
        len = tslot_len[s];
        const uint32_t p0 = A.ptr[p], e = A.ptr[p + 1];
        for (uint32_t base = p0; base < e; base += 4 * ROW_LANES) {  // 4 index -> value chains in flight
            uint32_t j[4];
            double c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t q = base + sub + u * ROW_LANES;
                const uint32_t qq = q < e ? q : p0;
                j[u] = A.jidx[qq];
                const double cv = A.coef[qq];
                c[u] = q < e ? cv : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double t = c[u] * jac[j[u] + r * shift];
                    if (fabs(t) >= 1e-9) v[r] += t;  // libsanm/sparse_solver.cpp:291-293
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
        for (int off = ROW_LANES / 2; off > 0; off >>= 1) v[r] += __shfl_down(v[r], off, ROW_LANES);
    if (s < A.nslots && sub == 0) {
        val[p] = v[0];
        val[p + len] = v[1];
        val[p + 2 * len] = v[2];
    }
}

__global__ void gather_kernel(size_t n, const double* __restrict__ src, const uint32_t* __restrict__ idx,
                              double* __restrict__ dst) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
// A'A + lambda I entry by entry (row_ops.h: ata_entry): one thread per entry of the fixed pattern
__global__ void ata_kernel(CsrDev At, CsrDev M, const uint32_t* __restrict__ mrow, double lambda) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < M.nnz) M.val[e] = ata_entry(At, mrow[e], M.col[e], lambda);
}

constexpr int SPMV_LANES = 8;
__global__ void spmv_kernel(CsrDev A, const double* __restrict__ x, double* __restrict__ y) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t row = gid / SPMV_LANES;
    int sub = gid % SPMV_LANES;
    double s = 0;
    if (row < A.n) {
        for (uint32_t p = A.rowptr[row] + sub, e = A.rowptr[row + 1]; p < e; p += SPMV_LANES)
            s += A.val[p] * x[A.col[p]];
    }
    for (int off = SPMV_LANES / 2; off > 0; off >>= 1) s += __shfl_down(s, off, SPMV_LANES);
    if (row < A.n && sub == 0) y[row] = s;
}

// r = b - A x in double-double arithmetic (error-free products by fma, two-sum accumulation): the residual of
// iterative refinement has to be known to more digits than the solution it corrects
struct DD {
    double hi, lo;
};
__device__ __forceinline__ void dd_add(DD& s, double v, double v_lo) {  // s += v (+ v_lo)
    const double t = s.hi + v, z = t - s.hi;
    const double err = (s.hi - (t - z)) + (v - z);
    s.hi = t;
    s.lo += err + v_lo;
}
__global__ void residual_dd_kernel(CsrDev A, const double* __restrict__ b, const double* __restrict__ x,
                                   double* __restrict__ r) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t row = gid / SPMV_LANES;
    int sub = gid % SPMV_LANES;
    DD s{0, 0};
    if (row < A.n) {
        if (sub == 0) s.hi = b[row];
        for (uint32_t p = A.rowptr[row] + sub, e = A.rowptr[row + 1]; p < e; p += SPMV_LANES) {
            const double a = A.val[p], xv = x[A.col[p]];
            const double pr = a * xv, pe = __builtin_fma(a, xv, -pr);  // pr + pe = a * xv exactly
            dd_add(s, -pr, -pe);
        }
    }
    for (int off = SPMV_LANES / 2; off > 0; off >>= 1) {
        const double ohi = __shfl_down(s.hi, off, SPMV_LANES), olo = __shfl_down(s.lo, off, SPMV_LANES);
        dd_add(s, ohi, olo);
    }
    if (row < A.n && sub == 0) r[row] = s.hi + s.lo;
}

// block reduce (256 threads = 4 waves) then one atomic per block
template <bool IS_MAX>
__device__ __forceinline__ void block_reduce_commit(double v, double* out) {
    __shared__ double sh[4];
    v = IS_MAX ? wave_reduce_max(v) : wave_reduce_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sh[0];
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = IS_MAX ? fmax(r, sh[i]) : r + sh[i];
        if (IS_MAX) {
            // atomic max on doubles through CAS on the bit pattern
            unsigned long long* a = reinterpret_cast<unsigned long long*>(out);
            unsigned long long old = *a, assumed;
            do {
                assumed = old;
                if (__longlong_as_double(assumed) >= r) break;
                old = atomicCAS(a, assumed, __double_as_longlong(r));
            } while (assumed != old);
        } else {
            atomicAdd(out, r);
        }
    }
}

__global__ void __launch_bounds__(256) dot_kernel(size_t n, const double* __restrict__ x,
                                                  const double* __restrict__ y, GridRed g) {
    double s[1] = {0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        s[0] += x[i] * y[i];
    grid_commit<1>(s, 1, 0u, g);
}

__global__ void axpby_kernel(size_t n, double a, const double* x, double b, const double* y,
                             double* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = b == 0.0 ? a * x[i] : a * x[i] + b * y[i];
}

__global__ void axpby_tail_kernel(size_t n, double a, const double* x, double b, const double* y,
                                  double* out, double tail) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a * x[i] + b * y[i];
    else if (i == n) out[i] = tail;
}

constexpr int MAX_VEC = GsPhase::kMaxVec;
struct VecList {
    const double* p[MAX_VEC];
    double c[MAX_VEC];
    int n;
};
__global__ void lincomb_kernel(size_t n, VecList v, double* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc = 0;
    for (int j = 0; j < v.n; ++j) acc += v.c[j] * v.p[j][i];
    out[i] = acc;
}
// ---- Gram-Schmidt phases of the Pade basis (pade.cpp:36-70), as bodies over (workgroup bid of nb) -------------
// They run either as kernels of their own or as EXTRA WORKGROUPS of a kernel the order loop launches anyway
// (GsRider below): a phase is ~6 us of pure launch latency on a chip the host kernel leaves mostly idle.
//
// classical Gram-Schmidt update with the projections read from device memory, and the squared norm of
// the result
template <int NVT>
// (x may be `out` itself -- the later chunks of a step over more than MAX_VEC vectors --: no restrict on it)
__device__ __forceinline__ void gs_update_body(size_t n, const double* x, const VecList& q,
                                               const double* __restrict__ coefs, int first, double* out, GridRed g,
                                               unsigned bid, unsigned nb) {
    double s[1] = {0};
    for (size_t i = (size_t)bid * blockDim.x + threadIdx.x; i < n; i += (size_t)nb * blockDim.x) {
        double acc = x[i], qv[NVT];
        if (q.n > 0) {
#pragma unroll
            for (int j = 0; j < NVT; ++j) qv[j] = q.p[j < q.n ? j : q.n - 1][i];  // one batch of loads
#pragma unroll
            for (int j = 0; j < NVT; ++j)
                if (j >= first && j < q.n) acc += -coefs[j] * qv[j];
        }
        out[i] = acc;
        s[0] += acc * acc;
    }
    grid_commit_at<1>(s, 1, 0u, g, bid, nb);
}
template <int NVT>
__global__ void __launch_bounds__(256) gs_update_kernel(size_t n, const double* x, VecList q,
                                                        const double* __restrict__ coefs, int first, double* out,
                                                        GridRed g) {
    gs_update_body<NVT>(n, x, q, coefs, first, out, g, blockIdx.x, gridDim.x);
}
__global__ void __launch_bounds__(256) scale_rsqrt_kernel(size_t n, double* v, const double* __restrict__ norm2,
                                                          double eps, GridRed g) {
    scale_rsqrt_body(n, v, norm2, eps, g, blockIdx.x, gridDim.x);
}

// second normalisation of an underflowed Gram-Schmidt direction; leaves at once otherwise
__global__ void renorm_scale_kernel(size_t n, double* v, const double* __restrict__ norm2, double eps,
                                    const double* __restrict__ nn2) {
    if (sqrt(*norm2) >= eps) return;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] *= 1.0 / sqrt(*nn2);
}

// Pade range test: two linear combinations of the same vectors, their scaled difference and both norms
__global__ void __launch_bounds__(256) lincomb2_diff_norms_kernel(size_t n, VecList v, VecList v2, double scale,
                                                                  GridRed g) {
    double r[2] = {0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        double u = 0, w = 0;
        for (int j = 0; j < v.n; ++j) {
            const double x = v.p[j][i];
            u += v.c[j] * x;
            w += v2.c[j] * x;
        }
        const double d = scale * w - u;
        r[0] += d * d;
        r[1] += u * u;
    }
    grid_commit<2>(r, 2, 0u, g);
}

// ... for up to PROBE_MAX coefficient sets in one pass over the vectors (speculative bisection of the range)
constexpr int PROBE_MAX = 7;
struct ProbeArgs {
    const double* p[MAX_VEC];
    double c1[PROBE_MAX][MAX_VEC], c2[PROBE_MAX][MAX_VEC];
    double scale[PROBE_MAX];
    int nvec, ncand;
};
__global__ void __launch_bounds__(256) lincomb2_diff_norms_multi_kernel(size_t n, ProbeArgs a, GridRed g) {
    double r[2 * PROBE_MAX];
    for (int c = 0; c < 2 * PROBE_MAX; ++c) r[c] = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        double u[PROBE_MAX], w[PROBE_MAX];
#pragma unroll
        for (int c = 0; c < PROBE_MAX; ++c) u[c] = w[c] = 0;
        for (int j = 0; j < a.nvec; ++j) {
            const double x = a.p[j][i];
#pragma unroll
            for (int c = 0; c < PROBE_MAX; ++c) {  // same accumulation order per candidate as the single probe
                u[c] += a.c1[c][j] * x;
                w[c] += a.c2[c][j] * x;
            }
        }
#pragma unroll
        for (int c = 0; c < PROBE_MAX; ++c) {
            const double d = a.scale[c] * w[c] - u[c];
            r[2 * c] += d * d;
            r[2 * c + 1] += u[c] * u[c];
        }
    }
    grid_commit<2 * PROBE_MAX>(r, 2 * a.ncand, 0u, g);
}

// The same for up to PROBE_GROUPS * PROBE_MAX coefficient sets in ONE launch: blockIdx.y takes a group of PROBE_MAX
// sets whose coefficients sit in pinned host memory (a launch's argument block holds one group; 2.7 KB per group are
// read over the host link once per workgroup, from its scalar loads' cache afterwards), every group reduces into
// partials / ticket / result slots of its own.  With it the range estimate of a step needs two host round trips
// (opening probes + three bisection levels on both branches, then the remaining five levels: 16 and 31 probes)
// instead of four.  Each set's arithmetic is the single probe's.
constexpr int PROBE_GROUPS = 8;
struct ProbeGroup {
    double c1[PROBE_MAX][MAX_VEC], c2[PROBE_MAX][MAX_VEC];
    double scale[PROBE_MAX];
    int ncand, pad;
};
struct ProbePtrs {
    const double* p[MAX_VEC];
    int nvec;
};
__global__ void __launch_bounds__(256) lincomb2_diff_norms_grouped_kernel(size_t n, ProbePtrs a,
                                                                          const ProbeGroup* __restrict__ groups,
                                                                          double* partials, unsigned* tickets,
                                                                          double* results) {
    // the group's coefficients: one coalesced read over the host link per workgroup, broadcast reads from LDS after it
    __shared__ ProbeGroup G;
    {
        const double* src = reinterpret_cast<const double*>(groups + blockIdx.y);
        double* dst = reinterpret_cast<double*>(&G);
        for (unsigned i = threadIdx.x; i < sizeof(ProbeGroup) / sizeof(double); i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    double r[2 * PROBE_MAX];
    for (int c = 0; c < 2 * PROBE_MAX; ++c) r[c] = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double u[PROBE_MAX], w[PROBE_MAX];
#pragma unroll
        for (int c = 0; c < PROBE_MAX; ++c) u[c] = w[c] = 0;
        for (int j = 0; j < a.nvec; ++j) {
            const double x = a.p[j][i];
#pragma unroll
            for (int c = 0; c < PROBE_MAX; ++c) {
                u[c] += G.c1[c][j] * x;
                w[c] += G.c2[c][j] * x;
            }
        }
#pragma unroll
        for (int c = 0; c < PROBE_MAX; ++c) {
            const double d = G.scale[c] * w[c] - u[c];
            r[2 * c] += d * d;
            r[2 * c + 1] += u[c] * u[c];
        }
    }
    const GridRed g{partials + (size_t)blockIdx.y * 2 * PROBE_MAX * RED_MAX_GRID, tickets + blockIdx.y,
                    results + (size_t)blockIdx.y * 2 * PROBE_MAX};
    grid_commit_at<2 * PROBE_MAX>(r, 2 * G.ncand, 0u, g, blockIdx.x, gridDim.x);
}

// out[j] = x . ys[j]; x is read once per element.  The newest vector ys[v.n-1] may still await the second
// normalisation of an underflowed Gram-Schmidt direction (last_norm2 != null and sqrt(*last_norm2) < eps):
// every element is read here anyway, so it is rescaled in place on the way.
// (NVT = vector count rounded up to a multiple of 4, a template parameter: the loop over the vectors has to be
// unrolled with unconditional loads -- slots beyond the count re-read the last vector -- or every vector's load
// becomes a memory round trip of its own)
template <int NVT>
__device__ __forceinline__ void multi_dot_body(size_t n, const double* __restrict__ x, const VecList& v,
                                               const double* last_norm2, const double* last_nn2, double eps,
                                               GridRed g, unsigned bid, unsigned nb) {
    double acc[NVT];
#pragma unroll
    for (int j = 0; j < NVT; ++j) acc[j] = 0;
    const bool fix = last_norm2 && v.n > 0 && sqrt(*last_norm2) < eps;
    const double f = fix ? 1.0 / sqrt(*last_nn2) : 1.0;
    double* last = fix ? const_cast<double*>(v.p[v.n - 1]) : nullptr;
    for (size_t i = (size_t)bid * blockDim.x + threadIdx.x; i < n; i += (size_t)nb * blockDim.x) {
        const double xi = x[i];
        double y[NVT];
#pragma unroll
        for (int j = 0; j < NVT; ++j) y[j] = v.p[j < v.n ? j : v.n - 1][i];
#pragma unroll
        for (int j = 0; j < NVT; ++j)
            if (j < v.n) {
                double yj = y[j];
                if (fix && j == v.n - 1) last[i] = yj = yj * f;
                acc[j] += xi * yj;
            }
    }
    grid_commit_at<NVT>(acc, v.n, 0u, g, bid, nb);
}
template <int NVT>
__global__ void __launch_bounds__(256) multi_dot_kernel(size_t n, const double* __restrict__ x, VecList v,
                                                        const double* last_norm2, const double* last_nn2,
                                                        double eps, GridRed g) {
    multi_dot_body<NVT>(n, x, v, last_norm2, last_nn2, eps, g, blockIdx.x, gridDim.x);
}

// ---- a deferred Gram-Schmidt phase riding on another launch ------------------------------------------------
// (Backend::defer_gs_phase).  The host kernel is launched with `nblk` extra workgroups behind its own `own`; those
// run the phase.  Which kernel carries which phase is fixed (template parameter of the carrier): the remap_out
// gather carries the projections (kind 1, NVT = vector count rounded up to 4), the last kernel of a solve the
// update (kind 2), next_coeff the scaling (kind 3); the order loop launches the three in that order once per order.
struct GsRider {
    size_t n;
    const double* x;
    VecList q;
    const double* coefs;
    int first;
    double* out;
    const double* norm2;
    const double* nn2;
    double eps;
    unsigned nblk;
    GridRed g;
};

// Exclusive prefix sum of v[0 .. m) in place with the total in v[m] (prepare_assembly: list offsets from list lengths
// without a round trip through the host): sums of blocks of 1024, their offsets by one workgroup, then each block again.
constexpr int SCAN_BLOCK = 1024;
__global__ void __launch_bounds__(256) scan_block_sums_kernel(const uint32_t* __restrict__ v, size_t m,
                                                              uint64_t* __restrict__ bsum) {
    __shared__ uint64_t red[256];
    const size_t base = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
    uint64_t s = 0;
    for (int k = 0; k < 4; ++k)
        if (base + k < m) s += v[base + k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = red[0];
}
__global__ void __launch_bounds__(256) scan_offsets_kernel(uint64_t* __restrict__ bsum, size_t nb) {
    __shared__ uint64_t part[256];
    const size_t per = (nb + 255) / 256, b0 = min(nb, threadIdx.x * per), b1 = min(nb, b0 + per);
    uint64_t s = 0;
    for (size_t b = b0; b < b1; ++b) s += bsum[b];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int t = 0; t < 256; ++t) {
            const uint64_t c = part[t];
            part[t] = run;
            run += c;
        }
        bsum[nb] = run;
    }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (size_t b = b0; b < b1; ++b) {
        const uint64_t c = bsum[b];
        bsum[b] = run;
        run += c;
    }
}
__global__ void __launch_bounds__(256) scan_apply_kernel(uint32_t* __restrict__ v, size_t m,
                                                         const uint64_t* __restrict__ bsum, size_t nb) {
    __shared__ uint32_t tsum[256];
    const int tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * SCAN_BLOCK + tid * 4;
    uint32_t x[4], s = 0;
    for (int k = 0; k < 4; ++k) {
        x[k] = base + k < m ? v[base + k] : 0u;
        s += x[k];
    }
    tsum[tid] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const uint32_t add = tid >= off ? tsum[tid - off] : 0u;
        __syncthreads();
        tsum[tid] += add;
        __syncthreads();
    }
    uint32_t run = (uint32_t)bsum[blockIdx.x] + tsum[tid] - s;
    for (int k = 0; k < 4; ++k)
        if (base + k < m) {
            v[base + k] = run;
            run += x[k];
        }
    if (blockIdx.x == 0 && tid == 0) v[m] = (uint32_t)bsum[nb];
}

// remap_out with its rows in triples (SparseRowsDev::bptr): GATHER_LANES lanes per vertex, one list entry feeds the
// three components (their values sit 3 doubles apart in the tet's block of the tet-major output buffer)
template <int RIDE_NVT>
__global__ void __launch_bounds__(256) gather_rows3_kernel(const uint32_t* __restrict__ bptr_,
                                                           const uint32_t* __restrict__ bidx_,
                                                           const double* __restrict__ bcoef_,
                                                           const double* __restrict__ src, double* __restrict__ dst,
                                                           const int32_t* __restrict__ perm,
                                                           double* __restrict__ dst2, int64_t nrows_, unsigned own,
                                                           GsRider rd) {
    // (scalar pointer arguments first: kernarg preload, see mf_kernels.h)
    const struct {
        const uint32_t *bptr, *bidx;
        const double* bcoef;
        int64_t nrows;
    } R{bptr_, bidx_, bcoef_, nrows_};
    if constexpr (RIDE_NVT > 0) {
        if (blockIdx.x >= own) {
            multi_dot_body<RIDE_NVT>(rd.n, rd.x, rd.q, rd.norm2, rd.nn2, rd.eps, rd.g, blockIdx.x - own, rd.nblk);
            return;
        }
    }
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t u = gid / GATHER_LANES;
    const int sub = gid % GATHER_LANES;
    double s[3] = {0, 0, 0};
    int pm[3] = {0, 0, 0};
    if (u < R.nrows / 3) {
        const uint32_t p0 = R.bptr[u], e = R.bptr[u + 1];
        if (perm && sub == 0) {  // requested with the list, not after the sums
#pragma unroll
            for (int c = 0; c < 3; ++c) pm[c] = perm[3 * u + c];
        }
        for (uint32_t base = p0; base < e; base += 4 * GATHER_LANES) {  // 4 index -> value chains in flight
            uint32_t i[4];
            double c[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t q = base + sub + k * GATHER_LANES, qq = q < e ? q : p0;
                i[k] = R.bidx[qq];
                const double cv = R.bcoef[qq];
                c[k] = q < e ? cv : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s[0] += c[k] * src[i[k]];
                s[1] += c[k] * src[i[k] + 3];
                s[2] += c[k] * src[i[k] + 6];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        for (int off = GATHER_LANES / 2; off > 0; off >>= 1) s[c] += __shfl_down(s[c], off, GATHER_LANES);
    if (u < R.nrows / 3 && sub == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dst[3 * u + c] = s[c];
            if (perm) dst2[pm[c]] = s[c];
        }
    }
}

// the last kernel of a solve (x[i] = w[perm[i]]) fused with the dot product the order loop takes of its result
template <int RIDE_NVT>
__global__ void __launch_bounds__(256) permute_out_dot_kernel(int64_t n, const int32_t* __restrict__ perm,
                                                              const double* __restrict__ w, double* __restrict__ x,
                                                              const double* __restrict__ y, GridRed g, unsigned own,
                                                              GsRider rd) {
    if constexpr (RIDE_NVT > 0) {
        if (blockIdx.x >= own) {
            gs_update_body<RIDE_NVT>(rd.n, rd.x, rd.q, rd.coefs, rd.first, rd.out, rd.g, blockIdx.x - own, rd.nblk);
            return;
        }
    }
    double s[1] = {0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)own * blockDim.x) {
        const double v = w[perm[i]];
        x[i] = v;
        s[0] += v * y[i];
    }
    grid_commit_at<1>(s, 1, 0u, g, blockIdx.x, own);
}

// x_i of the order loop with t_i taken from the device: t = *num * scale
// order 1 of the expansion with t_1 formed on the device: t_1 = 1 / sqrt(|xg|^2 + 1), x_1 = -t_1 xg - xb
__global__ void __launch_bounds__(256) x1_kernel(size_t n, const double* __restrict__ xgt2, const double* __restrict__ x,
                                                 const double* __restrict__ y, double* out, double* sc, double* t_out) {
    const double t = 1.0 / sqrt(*xgt2 + 1.0);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = -t * x[i] - y[i];
    else if (i == n) {
        out[i] = t;
        sc[0] = t;
        *t_out = t;
    }
}

template <bool RIDE>
__global__ void __launch_bounds__(256) next_coeff_kernel(size_t n, const double* __restrict__ num, double scale,
                                                         const double* __restrict__ sc,
                                                         const double* __restrict__ x, const double* __restrict__ y,
                                                         double* out, double* t_out, unsigned own, GsRider rd) {
    if constexpr (RIDE) {
        if (blockIdx.x >= own) {
            scale_rsqrt_body(rd.n, rd.out, rd.norm2, rd.eps, rd.g, blockIdx.x - own, rd.nblk);
            return;
        }
    }
    if (sc) scale = 1.0 / (sc[0] - sc[1]);
    const double t = *num * scale;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = -t * x[i] - y[i];
    else if (i == n) {
        out[i] = t;
        *t_out = t;
    }
}

__global__ void vmul_kernel(size_t n, const double* x, const double* y, double* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] * y[i];
}

__global__ void inv_diag_kernel(CsrDev A, double scale, double* d) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < A.n) d[i] = 1.0 / (scale * csr_diag(A, i));
}

// a batch of strided block copies (MfCopy2D): the exchanges of the distributed multifrontal schedule
__global__ void __launch_bounds__(256) copy2d_kernel(const MfCopy2D* __restrict__ d, const double* __restrict__ src,
                                                     double* __restrict__ dst) {
    const MfCopy2D c = d[blockIdx.z];
    for (int i = blockIdx.y; i < c.rows; i += gridDim.y)
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < c.cols; j += gridDim.x * blockDim.x)
            dst[c.dst + (int64_t)i * c.ldd + j] = src[c.src + (int64_t)i * c.lds + j];
}

// the pivot counter of a factorisation as a double where the host reads its asynchronous results
__global__ void status_to_double_kernel(const int32_t* status, double* out) { *out = (double)*status; }

// max |x_i| (the scale the factorisation's pivot threshold refers to)
__global__ void __launch_bounds__(256) absmax_kernel(size_t n, const double* x, GridRed g) {
    double m[1] = {0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m[0] = fmax(m[0], fabs(x[i]));
    grid_commit<1>(m, 1, 1u, g);
}

__global__ void __launch_bounds__(256) nonfinite_kernel(size_t n, const double* x, GridRed g) {
    double s[1] = {0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        s[0] += isfinite(x[i]) ? 0.0 : 1.0;
    grid_commit<1>(s, 1, 0u, g);
}

__global__ void __launch_bounds__(256) allclose_kernel(size_t n, const double* a, const double* b,
                                                       double eps, GridRed g) {
    double m[1] = {-1e300};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        m[0] = fmax(m[0], allclose_excess1(a[i], b[i], eps));
    grid_commit<1>(m, 1, 1u, g);
}

// fused reductions of the ANM sanity check: out[0] = max allclose excess, out[1] = dot
__global__ void __launch_bounds__(256) sanity_kernel(size_t n, const double* a, const double* b,
                                                     double eps, size_t n1, const double* x,
                                                     const double* y, GridRed g) {
    double v[2] = {-1e300, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1;
         i += (size_t)gridDim.x * blockDim.x) {
        if (i < n) v[0] = fmax(v[0], allclose_excess1(a[i], b[i], eps));
        v[1] += x[i] * y[i];
    }
    grid_commit<2>(v, 2, 1u, g);
}

// the per-order sanity check without materialising A*xi or the right-hand side: SPMV_LANES lanes per row
// (in the order loop the matrix comes from HBM every time -- 300 MB of Taylor state and factor pass through the
// caches between two checks -- so the kernel runs at 19 us where the same product on a cached matrix takes 8)
__global__ void __launch_bounds__(256) sanity_check_kernel(CsrDev A, const double* __restrict__ xi,
                                                           const double* ti_ptr, double ti_val,
                                                           const double* __restrict__ grad_t,
                                                           const double* __restrict__ bi, double eps, size_t n1,
                                                           const double* __restrict__ x1, GridRed g) {
    double v[2] = {-1e300, 0};
    const double ti = ti_ptr ? *ti_ptr : ti_val;
    const int sub = threadIdx.x % SPMV_LANES;
    for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < (int64_t)A.n * SPMV_LANES;
         gid += (int64_t)gridDim.x * blockDim.x) {  // A.n * SPMV_LANES is a multiple of the group size
        const int64_t row = gid / SPMV_LANES;
        // right-hand side and the x_1 . x_i term: same round trip as the row pointers
        const double rhs = -ti * grad_t[row] - bi[row];
        if ((size_t)gid < n1) v[1] += x1[gid] * xi[gid];
        double s = row_dot<SPMV_LANES>(A.col, A.val, xi, A.rowptr[row], A.rowptr[row + 1], sub);
        for (int off = SPMV_LANES / 2; off > 0; off >>= 1) s += __shfl_down(s, off, SPMV_LANES);
        if (sub == 0) v[0] = fmax(v[0], allclose_excess1(s, rhs, eps));
    }
    grid_commit<2>(v, 2, 1u, g);
}

// ... for up to SANITY_ORDERS orders in one pass over the matrix: the order loop examines the checks after its last
// order, and a pass over the matrix per order re-reads 15 MB from HBM every time (18 us each, 20 per step)
constexpr int SANITY_ORDERS = 10;
struct SanityBatch {
    const double* x[SANITY_ORDERS];
    const double* b[SANITY_ORDERS];
    int n;  // orders in use; the other slots repeat slot 0
};
__global__ void __launch_bounds__(256) sanity_check_multi_kernel(CsrDev A, SanityBatch a,
                                                                 const double* __restrict__ grad_t, double eps,
                                                                 size_t n1, const double* __restrict__ x1, GridRed g) {
    double v[2 * SANITY_ORDERS];
#pragma unroll
    for (int q = 0; q < SANITY_ORDERS; ++q) {
        v[2 * q] = -1e300;
        v[2 * q + 1] = 0;
    }
    const int sub = threadIdx.x % SPMV_LANES;
    for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < (int64_t)A.n * SPMV_LANES;
         gid += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = gid / SPMV_LANES;
        const uint32_t p0 = A.rowptr[row], e = A.rowptr[row + 1];
        double s[SANITY_ORDERS];
#pragma unroll
        for (int q = 0; q < SANITY_ORDERS; ++q) s[q] = 0;
        for (uint32_t base = p0; base < e; base += 2 * SPMV_LANES) {  // 2 entries x SANITY_ORDERS gathers in flight
            uint32_t ci[2];
            double cv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t k = base + sub + u * SPMV_LANES, kk = k < e ? k : p0;
                ci[u] = A.col[kk];
                const double val = A.val[kk];
                cv[u] = k < e ? val : 0.0;
            }
#pragma unroll
            for (int q = 0; q < SANITY_ORDERS; ++q) s[q] += cv[0] * a.x[q][ci[0]] + cv[1] * a.x[q][ci[1]];
        }
        const double gt = grad_t[row];
#pragma unroll
        for (int q = 0; q < SANITY_ORDERS; ++q) {
            double t = s[q];
            for (int off = SPMV_LANES / 2; off > 0; off >>= 1) t += __shfl_down(t, off, SPMV_LANES);
            if (sub == 0) v[2 * q] = fmax(v[2 * q], allclose_excess1(t, -a.x[q][A.n] * gt - a.b[q][row], eps));
            if ((size_t)gid < n1) v[2 * q + 1] += x1[gid] * a.x[q][gid];
        }
    }
    unsigned maxmask = 0;
    for (int q = 0; q < SANITY_ORDERS; ++q) maxmask |= 1u << (2 * q);
    grid_commit<2 * SANITY_ORDERS>(v, 2 * a.n, maxmask, g);
}

__global__ void __launch_bounds__(256) t0v_kernel(size_t n, const double* fx, const double* v,
                                                  double t0, double tol, GridRed g) {
    double mm[1];
    double m = -1e300;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        // anm.cpp:349-353
        double a = fx[i], b = v[i] * t0;
        double me = fmax(fmin(fabs(a), fabs(b)), 1.0) * tol;
        double d = fabs(a + b);
        m = fmax(m, (d == d) ? d - me : 1e300);
    }
    mm[0] = m;
    grid_commit<1>(mm, 1, 1u, g);
}


// ---- Jacobi-PCG with device-resident scalars --------------------------------
// 3 launches per iteration and no host synchronisation inside the loop: the
// host only reads |r|^2 every PCG_CHECK_EVERY iterations.  Scalars live in a
// small device array; slots indexed by iteration parity are zeroed by the
// kernel that runs while they are dead (see pcg() below).
struct PcgScalars {
    double rz[2];  // r.z of the current / next iteration
    double pq[2];  // p.(M p)
    double rr[2];  // |r|^2
    double bb;     // |b|^2
    int breakdown; // set when p.(M p) <= 0
};

__global__ void __launch_bounds__(256) pcg_init_kernel(size_t n, double sign,
                                                       const double* __restrict__ b,
                                                       const double* __restrict__ dinv, double* x,
                                                       double* r, double* z, double* p,
                                                       PcgScalars* sc) {
    double rz = 0, rr = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        double ri = sign * b[i], zi = dinv[i] * ri;
        x[i] = 0;
        r[i] = ri;
        z[i] = zi;
        p[i] = zi;
        rz += ri * zi;
        rr += ri * ri;
    }
    block_reduce_commit<false>(rz, &sc->rz[0]);
    __syncthreads();
    block_reduce_commit<false>(rr, &sc->bb);
}

// q = A p ; pq[it&1] += sign * p.q ; zeroes rz[(it+1)&1] and rr[it&1]
__global__ void __launch_bounds__(256) pcg_spmv_dot_kernel(CsrDev A, double sign,
                                                           const double* __restrict__ p,
                                                           double* __restrict__ q, PcgScalars* sc,
                                                           int it) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid == 0) {
        sc->rz[(it + 1) & 1] = 0;
        sc->rr[it & 1] = 0;
    }
    int64_t row = gid / SPMV_LANES;
    int sub = gid % SPMV_LANES;
    double s = 0;
    if (row < A.n) {
        for (uint32_t k = A.rowptr[row] + sub, e = A.rowptr[row + 1]; k < e; k += SPMV_LANES)
            s += A.val[k] * p[A.col[k]];
    }
    for (int off = SPMV_LANES / 2; off > 0; off >>= 1) s += __shfl_down(s, off, SPMV_LANES);
    double contrib = 0;
    if (row < A.n && sub == 0) {
        q[row] = s;
        contrib = sign * s * p[row];
    }
    block_reduce_commit<false>(contrib, &sc->pq[it & 1]);
}

// alpha = rz/pq ; x += alpha p ; r -= alpha*sign*q ; z = dinv r ;
// rz[(it+1)&1] += r.z ; rr[it&1] += r.r ; zeroes pq[(it+1)&1]
__global__ void __launch_bounds__(256) pcg_update_kernel(size_t n, double sign,
                                                         const double* __restrict__ dinv,
                                                         const double* __restrict__ p,
                                                         const double* __restrict__ q, double* x,
                                                         double* r, double* z, PcgScalars* sc, int it) {
    const double pq = sc->pq[it & 1], rz = sc->rz[it & 1];
    const bool bad = !(pq > 0);
    const double alpha = bad ? 0.0 : rz / pq;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc->pq[(it + 1) & 1] = 0;
        if (bad && rz != 0) sc->breakdown = 1;
    }
    double rzn = 0, rr = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        x[i] += alpha * p[i];
        double ri = r[i] - alpha * sign * q[i];
        double zi = dinv[i] * ri;
        r[i] = ri;
        z[i] = zi;
        rzn += ri * zi;
        rr += ri * ri;
    }
    block_reduce_commit<false>(rzn, &sc->rz[(it + 1) & 1]);
    __syncthreads();
    block_reduce_commit<false>(rr, &sc->rr[it & 1]);
}

// beta = rz_new / rz ; p = z + beta p
__global__ void pcg_dir_kernel(size_t n, const double* __restrict__ z, double* p,
                               const PcgScalars* sc, int it) {
    const double rz = sc->rz[it & 1], rzn = sc->rz[(it + 1) & 1];
    const double beta = rz != 0 ? rzn / rz : 0.0;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = z[i] + beta * p[i];
}
constexpr int PCG_CHECK_EVERY = 64;

inline unsigned nblk(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }
inline unsigned red_grid(size_t n) {
    size_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > RED_MAX_GRID ? RED_MAX_GRID : g));
}

// ---- RCCL, bound at run time ------------------------------------------------------------------------------
// ---- vector graphs (vecprog.h): one workgroup per batch item, one thread per element, a barrier between operators --
__global__ void __launch_bounds__(VEC_MAX_SIZE) vec_pass_kernel(VecProgDev P, int mode, int order, const double* xin) {
    const int64_t b = blockIdx.x;
    const int e = threadIdx.x;
    if (mode == PASS_GRAD) {
        extern __shared__ double vg[];  // gradient rows of all variables for the Jacobian row in hand
        for (int r = 0; r < P.odim; ++r) {
            for (int i = e; i < P.grad_total; i += VEC_MAX_SIZE) vg[i] = 0.0;
            __syncthreads();
            if (e == 0) vg[P.vars[P.out_var].grad + r] = 1.0;  // symbolic.cpp:219-220
            __syncthreads();
            for (int i = P.nops - 1; i >= 0; --i) {
                vec_backward(P, P.ops[i], b, e, vg);
                __syncthreads();
            }
            if (e < P.idim) P.arena[P.jac + (b * P.odim + r) * P.idim + e] = vg[P.vars[P.in_var].grad + e];
            __syncthreads();
        }
        return;
    }
    for (int i = 0; i < P.nops; ++i) {
        vec_forward(P, P.ops[i], mode, order, b, e, xin);
        __syncthreads();
    }
}

// ---- dense LU with partial pivoting (Backend::dense_lu_factor): one workgroup, the matrix in global memory --------
// The systems are those of graphs on the vector interpreter (tests/symbolic.cpp's operator tests: a few hundred
// unknowns); a generality path, not a hot one.
constexpr int DLU_THREADS = 1024;
__global__ void __launch_bounds__(DLU_THREADS) dense_from_csr_kernel(CsrDev A, double* lu) {
    const int64_t n = A.n;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        for (int64_t j = threadIdx.x; j < n; j += blockDim.x) lu[i * n + j] = 0.0;
        __syncthreads();
        if (threadIdx.x == 0)  // (duplicate columns of a row add up, in their order)
            for (uint32_t p = A.rowptr[i]; p < A.rowptr[i + 1]; ++p) lu[i * n + A.col[p]] += A.val[p];
        __syncthreads();
    }
}
__global__ void __launch_bounds__(DLU_THREADS) dense_lu_kernel(int64_t n, double* lu, int32_t* piv, double* status) {
    __shared__ double s_val[DLU_THREADS];
    __shared__ int s_idx[DLU_THREADS];
    __shared__ double s_piv;
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    constexpr int kWaves = DLU_THREADS / 64;
    double pmin = INFINITY, pmax = 0;
    bool bad = false;  // a column without a single comparable entry (all NaN): reported through status[0] = NaN
    for (int64_t c = 0; c < n; ++c) {
        // pivot: largest |lu[r][c]|, r >= c; ties to the lowest row (as a sequential scan does).  NaN entries never
        // win a comparison; a column made of them alone leaves the search empty-handed (bi stays INT_MAX).
        double best = -1.0;
        int bi = 0x7fffffff;
        for (int64_t r = c + t; r < n; r += DLU_THREADS) {
            const double v = fabs(lu[r * n + c]);
            if (v > best) {
                best = v;
                bi = (int)r;
            }
        }
        s_val[t] = best;
        s_idx[t] = bi;
        __syncthreads();
        for (int w = DLU_THREADS / 2; w > 0; w >>= 1) {
            if (t < w) {
                const double ov = s_val[t + w];
                const int oi = s_idx[t + w];
                if (ov > s_val[t] || (ov == s_val[t] && oi < s_idx[t])) {
                    s_val[t] = ov;
                    s_idx[t] = oi;
                }
            }
            __syncthreads();
        }
        int p = s_idx[0];
        double pabs = s_val[0];
        __syncthreads();
        if (p >= n) {  // nothing comparable in this column (non-finite Jacobian): no exchange, no elimination
            p = (int)c;
            pabs = 0.0;
            bad = true;
        }
        if (t == 0) piv[c] = p;
        pmin = fmin(pmin, pabs);
        pmax = fmax(pmax, pabs);
        if (p != (int)c)
            for (int64_t j = t; j < n; j += DLU_THREADS) {
                const double a = lu[c * n + j];
                lu[c * n + j] = lu[(int64_t)p * n + j];
                lu[(int64_t)p * n + j] = a;
            }
        __syncthreads();
        if (pabs == 0.0) continue;
        if (t == 0) s_piv = lu[c * n + c];
        __syncthreads();
        const double pv = s_piv;
        for (int64_t r = c + 1 + t; r < n; r += DLU_THREADS) lu[r * n + c] = lu[r * n + c] / pv;
        __syncthreads();
        // trailing update: a wavefront per row, its lanes along the columns (coalesced; no index division)
        for (int64_t r = c + 1 + wv; r < n; r += kWaves) {
            const double l = lu[r * n + c];
            for (int64_t j = c + 1 + lane; j < n; j += 64) lu[r * n + j] = __builtin_fma(-l, lu[c * n + j], lu[r * n + j]);
        }
        __syncthreads();
    }
    if (t == 0) {
        status[0] = bad ? NAN : pmin;
        status[1] = pmax;
    }
}
// The same factorisation for systems beyond a few hundred unknowns (round 5: the single workgroup above was the only
// factorisation path of vector-graph systems up to 4096 unknowns): 32-column panels -- the panel by one workgroup
// (pivot search over the whole column, interchanges inside the panel), then the interchanges of the rest of the rows and
// the rows of U beside the panel (a thread per column), then the trailing matrix by 64 x 64 tiles over the chip.  Every
// element receives the updates of the unblocked loop in the unblocked order, one fma per pivot, so the factors are
// the single workgroup's bit for bit; columns whose pivot search finds nothing but zeros are skipped like there
// (`flags`).  acc3 = {smallest pivot, largest pivot, saw a column without a comparable entry}.
constexpr int DLU_NB = 32;
__global__ void __launch_bounds__(DLU_THREADS) dense_lu_panel_kernel(int64_t n, double* lu, int32_t* piv, int32_t* flags,
                                                                     double* acc3, double* status, int64_t c0, int w,
                                                                     int first, int last) {
    __shared__ double s_val[DLU_THREADS];
    __shared__ int s_idx[DLU_THREADS];
    __shared__ double s_piv;
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    constexpr int kWaves = DLU_THREADS / 64;
    double pmin = first ? INFINITY : acc3[0], pmax = first ? 0.0 : acc3[1];
    bool bad = first ? false : acc3[2] != 0.0;
    const int64_t ce = c0 + w;
    for (int64_t c = c0; c < ce; ++c) {
        double best = -1.0;
        int bi = 0x7fffffff;
        for (int64_t r = c + t; r < n; r += DLU_THREADS) {
            const double v = fabs(lu[r * n + c]);
            if (v > best) {
                best = v;
                bi = (int)r;
            }
        }
        s_val[t] = best;
        s_idx[t] = bi;
        __syncthreads();
        for (int ww = DLU_THREADS / 2; ww > 0; ww >>= 1) {
            if (t < ww) {
                const double ov = s_val[t + ww];
                const int oi = s_idx[t + ww];
                if (ov > s_val[t] || (ov == s_val[t] && oi < s_idx[t])) {
                    s_val[t] = ov;
                    s_idx[t] = oi;
                }
            }
            __syncthreads();
        }
        int p = s_idx[0];
        double pabs = s_val[0];
        __syncthreads();
        if (p >= n) {
            p = (int)c;
            pabs = 0.0;
            bad = true;
        }
        if (t == 0) {
            piv[c] = p;
            flags[c] = pabs == 0.0;
        }
        pmin = fmin(pmin, pabs);
        pmax = fmax(pmax, pabs);
        if (p != (int)c)
            for (int64_t j = c0 + t; j < ce; j += DLU_THREADS) {  // (the other columns: dense_lu_rowops_kernel)
                const double a = lu[c * n + j];
                lu[c * n + j] = lu[(int64_t)p * n + j];
                lu[(int64_t)p * n + j] = a;
            }
        __syncthreads();
        if (pabs == 0.0) continue;
        if (t == 0) s_piv = lu[c * n + c];
        __syncthreads();
        const double pv = s_piv;
        for (int64_t r = c + 1 + t; r < n; r += DLU_THREADS) lu[r * n + c] = lu[r * n + c] / pv;
        __syncthreads();
        for (int64_t r = c + 1 + wv; r < n; r += kWaves) {
            const double l = lu[r * n + c];
            for (int64_t j = c + 1 + lane; j < ce; j += 64) lu[r * n + j] = __builtin_fma(-l, lu[c * n + j], lu[r * n + j]);
        }
        __syncthreads();
    }
    if (t == 0) {
        acc3[0] = pmin;
        acc3[1] = pmax;
        acc3[2] = bad ? 1.0 : 0.0;
        if (last) {
            status[0] = bad ? NAN : pmin;
            status[1] = pmax;
        }
    }
}
__global__ void __launch_bounds__(256) dense_lu_rowops_kernel(int64_t n, double* lu, const int32_t* __restrict__ piv,
                                                              const int32_t* __restrict__ flags, int64_t c0, int w) {
    // a thread per column outside the panel: the panel's interchanges in order, then -- right of the panel -- the
    // panel rows of U: row r takes the updates of the pivots c < r of the panel
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= c0) j += w;
    if (j >= n) return;
    const int64_t ce = c0 + w;
    for (int64_t c = c0; c < ce; ++c) {
        const int p = piv[c];
        if (p != (int)c) {
            const double a = lu[c * n + j];
            lu[c * n + j] = lu[(int64_t)p * n + j];
            lu[(int64_t)p * n + j] = a;
        }
    }
    if (j < ce) return;
    for (int64_t c = c0; c < ce; ++c) {
        if (flags[c]) continue;
        const double u = lu[c * n + j];
        for (int64_t r = c + 1; r < ce; ++r) lu[r * n + j] = __builtin_fma(-lu[r * n + c], u, lu[r * n + j]);
    }
}
__global__ void __launch_bounds__(256) dense_lu_trailing_kernel(int64_t n, double* lu, const int32_t* __restrict__ flags,
                                                                int64_t c0, int w) {
    // 64 x 64 tile of the trailing matrix, 4 x 4 elements per thread; per element one fma per pivot of the panel, in order
    __shared__ double Lp[64][DLU_NB + 1], Up[DLU_NB][64 + 1];
    __shared__ int fl[DLU_NB];
    const int64_t ce = c0 + w, r0 = ce + (int64_t)blockIdx.y * 64, j0 = ce + (int64_t)blockIdx.x * 64;
    const int t = threadIdx.x;
    for (int q = t; q < 64 * DLU_NB; q += 256) {
        const int rr = q / DLU_NB, cc = q % DLU_NB;
        Lp[rr][cc] = (r0 + rr < n && cc < w) ? lu[(r0 + rr) * n + c0 + cc] : 0.0;
        const int uc = q / 64, uj = q % 64;
        Up[uc][uj] = (uc < w && j0 + uj < n) ? lu[(c0 + uc) * n + j0 + uj] : 0.0;
    }
    if (t < DLU_NB) fl[t] = t < w ? flags[c0 + t] : 1;
    __syncthreads();
    const int ty = t / 16, tx = t % 16;
    double a[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t r = r0 + ty * 4 + i, j = j0 + tx + 16 * k;
            a[i][k] = (r < n && j < n) ? lu[r * n + j] : 0.0;
        }
    for (int c = 0; c < w; ++c) {
        if (fl[c]) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) a[i][k] = __builtin_fma(-Lp[ty * 4 + i][c], Up[c][tx + 16 * k], a[i][k]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t r = r0 + ty * 4 + i, j = j0 + tx + 16 * k;
            if (r < n && j < n) lu[r * n + j] = a[i][k];
        }
}
__global__ void __launch_bounds__(DLU_THREADS) dense_lu_solve_kernel(int64_t n, const double* lu, const int32_t* piv,
                                                                     const double* b, double* x, double* work) {
    const int t = threadIdx.x;
    for (int64_t i = t; i < n; i += DLU_THREADS) work[i] = b[i];
    __syncthreads();
    // (the factor swapped whole rows, the finished part of L included: all interchanges first, as getrs does)
    if (t == 0)
        for (int64_t c = 0; c < n; ++c) {
            const int p = piv[c];
            const double a = work[c];
            work[c] = work[p];
            work[p] = a;
        }
    __syncthreads();
    for (int64_t c = 0; c < n; ++c) {
        const double yc = work[c];
        for (int64_t r = c + 1 + t; r < n; r += DLU_THREADS) work[r] = __builtin_fma(-lu[r * n + c], yc, work[r]);
        __syncthreads();
    }
    for (int64_t c = n - 1; c >= 0; --c) {
        if (t == 0) work[c] = work[c] / lu[c * n + c];
        __syncthreads();
        const double yc = work[c];
        for (int64_t r = t; r < c; r += DLU_THREADS) work[r] = __builtin_fma(-lu[r * n + c], yc, work[r]);
        __syncthreads();
    }
    for (int64_t i = t; i < n; i += DLU_THREADS) x[i] = work[i];
}

// The tet-sharded mode sums nodal vectors over the ranks with ncclAllReduce on the backend's own stream, so the
// order loop stays free of host synchronisation (the callback form of the C ABI has to synchronise on both sides
// of the call).  RCCL is looked up with dlopen: a process that already holds it (torch.distributed's nccl backend
// loads its copy under the same soname) shares that copy, and single-GPU users never load it.  The four
// prototypes are RCCL's public C interface (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclAllReduce,
// ncclCommDestroy; ncclUniqueId is 128 opaque bytes, ncclFloat64 = 8, ncclSum = 0).
struct Rccl {
    struct UniqueId {
        char internal[128];
    };
    using Comm = void*;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    // the exchanges of the distributed direct solver (MfSchedule::Dist): grouped send / receive pairs for the Schur
    // complements and inbox rows, grouped broadcasts in place (a gather of ranges of unequal length) for the solution
    int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommCount)(Comm, int*) = nullptr;
    int (*CommUserRank)(Comm, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    static const Rccl& get() {
        static Rccl r = [] {
            Rccl x;
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char* nm : names) {
                void* h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
                if (!h) continue;
                x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
                x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
                x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(h, "ncclAllReduce"));
                x.Send = reinterpret_cast<decltype(x.Send)>(dlsym(h, "ncclSend"));
                x.Recv = reinterpret_cast<decltype(x.Recv)>(dlsym(h, "ncclRecv"));
                x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(dlsym(h, "ncclBroadcast"));
                x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(h, "ncclGroupStart"));
                x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
                x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
                x.CommCount = reinterpret_cast<decltype(x.CommCount)>(dlsym(h, "ncclCommCount"));
                x.CommUserRank = reinterpret_cast<decltype(x.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
                x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
                if (x.GetUniqueId && x.CommInitRank && x.AllReduce && x.CommDestroy) break;
                x = Rccl{};
            }
            return x;
        }();
        if (!r.AllReduce) sanm_throw(SANM_ERR_HIP, "RCCL (librccl.so.1) could not be loaded: %s", dlerror());
        return r;
    }
    void check(int rc, const char* what) const {
        if (rc != 0)
            sanm_throw(SANM_ERR_HIP, "RCCL %s failed: %s", what, GetErrorString ? GetErrorString(rc) : "error");
    }
};

// every kernel launch of the backend goes through this: the count lets the phase profile report launches per
// family (bench.py: roofline_families[*].launches_per_step) instead of a formula
#define SANM_LAUNCH(...)                    \
    do {                                    \
        ++m_launch_count;                   \
        hipLaunchKernelGGL(__VA_ARGS__);    \
    } while (0)

class HipBackend final : public Backend {
    hipStream_t m_stream = nullptr;  // the queue launches currently go to: m_main, or m_side between side_fork / side_end
    hipStream_t m_main = nullptr, m_side = nullptr;
    hipStream_t m_sf_queue[3] = {nullptr, nullptr, nullptr};  // size classes of small_front_kernel beside each other (mf_factor)
    bool m_side_dirty = false;  // the side queue got work since the last join / sync
    bool m_side_detached = false;  // ... and nothing waits for it before side_wait()
    static constexpr int kForkEvents = 64;
    hipEvent_t m_fork_ev[kForkEvents] = {};
    int m_fork_next = 0;
    Rccl::Comm m_comm = nullptr;
    int64_t m_launch_count = 0;
    double* m_probe_work = nullptr;         // two work vectors of the range probes over more than MAX_VEC vectors
    size_t m_probe_work_n = 0;
    ProbeGroup* m_probe_groups = nullptr;   // pinned: coefficient sets of a grouped range probe
    double* m_probe_results = nullptr;      // pinned
    double* m_probe_partials = nullptr;
    unsigned* m_probe_tickets = nullptr;
    // Replayed launch chains (hipGraph): the level sweeps of a solve and the whole numeric factorisation are fixed
    // sequences of kernels with fixed arguments per solver.  scripts/micro/graph_chain.hip: a dependent chain costs
    // 3.1 us per kernel as stream launches and 2.5 (16 kernels) to 2.05 us (340 kernels) replayed as a graph.
    // A chain is launched directly the first time (lazy allocations, function attributes), captured the second
    // time and replayed from then on.  MEASURED ON THE REAL CHAINS (armadillo_small, round 3): bit-identical
    // results and SLOWER -- 5.86 against 5.64 ms per step (solves 2.30 against 2.19 ms, factorisation 1.91
    // against 1.90): a level kernel lasts 5-6 us because of its own chain of dependent memory round trips, behind
    // which the command processor already hides the packet handling that a graph saves on empty kernels, and every
    // replay adds its own launch.  Off by default; SANM_MF_GRAPH=1 switches it on.
    struct ChainGraph {
        hipGraphExec_t exec = nullptr;
        int uses = 0;
        int64_t launches = 0;          // kernel launches one replay stands for
        const void* arg0 = nullptr;    // the pointer arguments the capture baked in
        const void* arg1 = nullptr;
    };
    std::map<std::pair<const void*, int>, ChainGraph> m_chains;
    const bool m_chain_graphs = std::getenv("SANM_MF_GRAPH") != nullptr;
    bool m_capturing = false;
    //! runs `body` (a fixed sequence of launches on m_stream): directly, or as a replay of its captured form
    template <class F>
    void run_chain(const void* key, int kind, const void* arg0, const void* arg1, F&& body) {
        if (!m_chain_graphs || m_capturing || m_stream != m_main) {
            body();
            return;
        }
        ChainGraph& c = m_chains[{key, kind}];
        if (c.exec && (c.arg0 != arg0 || c.arg1 != arg1)) {  // other buffers than the captured ones: start over
            (void)hipGraphExecDestroy(c.exec);
            c = ChainGraph{};
        }
        if (c.exec) {
            HIP_CHECK(hipGraphLaunch(c.exec, m_stream));
            m_launch_count += c.launches;
            return;
        }
        if (c.uses++ == 0) {
            body();
            return;
        }
        const int64_t l0 = m_launch_count;
        hipGraph_t graph = nullptr;
        m_capturing = true;
        HIP_CHECK(hipStreamBeginCapture(m_stream, hipStreamCaptureModeThreadLocal));
        try {
            body();
        } catch (...) {
            (void)hipStreamEndCapture(m_stream, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            m_capturing = false;
            throw;
        }
        HIP_CHECK(hipStreamEndCapture(m_stream, &graph));
        m_capturing = false;
        HIP_CHECK(hipGraphInstantiate(&c.exec, graph, nullptr, nullptr, 0));
        HIP_CHECK(hipGraphDestroy(graph));
        c.launches = m_launch_count - l0;
        c.arg0 = arg0;
        c.arg1 = arg1;
        m_launch_count = l0;
        HIP_CHECK(hipGraphLaunch(c.exec, m_stream));
        m_launch_count += c.launches;
    }
    int m_comm_rank = 0, m_comm_world = 0;
    // pass kernels compiled at run time for one program each (specialize)
    struct SpecKernels {
        hipModule_t mod = nullptr;
        hipFunction_t pass[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    };
    std::vector<SpecKernels> m_spec;
    GridRed m_red{nullptr, nullptr, nullptr}, m_red_side{nullptr, nullptr, nullptr};
    static constexpr size_t kPoolBlockMax = size_t(64) << 20, kPoolTotalMax = size_t(4) << 30;
    std::unordered_map<void*, size_t> m_live;
    std::unordered_map<size_t, std::vector<void*>> m_pool_free;
    size_t m_pool_cached = 0;
    double* m_scalar = nullptr;  // device scratch for reductions
    double* m_scalar_host = nullptr;  // pinned
    double* m_pcg_w[4] = {nullptr, nullptr, nullptr, nullptr};
    PcgScalars* m_pcg_sc = nullptr;
    static constexpr int kSolveLdsMax = 150 * 1024;
    static constexpr int kWideMinM = 4096;  // backward levels with fronts this wide take bwd_wide_kernel
    // fronts from this many pivots on are factored with two blocking levels (measured: pays from ~400) (outer blocks of kOuterPanels
    // 32-wide panels); SANM_MF_OUTER_MIN_K overrides it (tests force the path on small fronts)
#ifndef SANM_MF_OUTER_PANELS
#define SANM_MF_OUTER_PANELS 4
#endif
    static constexpr int kOuterPanels = SANM_MF_OUTER_PANELS;
    static constexpr int kOuterMinK = 512;
    // levels with at least this many fronts of at most SF_KMAX pivots take small_front_kernel
    static constexpr int kSmallMinFronts = 1024;
    int m_conv_parts = std::getenv("SANM_CONV_PARTS") ? std::atoi(std::getenv("SANM_CONV_PARTS")) : 4;
    int m_conv_split_order = std::getenv("SANM_CONV_SPLIT_ORDER") ? std::atoi(std::getenv("SANM_CONV_SPLIT_ORDER")) : 4;
    PcgScalars* m_pcg_sc_host = nullptr;
    size_t m_pcg_n = 0;
    bool m_time_passes = false;
    size_t m_pass_lds_limit[4] = {48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024};
    static constexpr int kBiasWaves = 3;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> m_pass_events;
    // phase brackets (phase_begin / phase_end): closed ones wait in m_phase_done for phase_collect
    struct PhaseBracket {
        std::string tag;
        hipEvent_t e0, e1;
    };
    std::vector<PhaseBracket> m_phase_open, m_phase_done;
    std::vector<hipEvent_t> m_event_pool;
    hipEvent_t pooled_event() {
        if (!m_event_pool.empty()) {
            hipEvent_t e = m_event_pool.back();
            m_event_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        return e;
    }

public:
    int m_device_id = 0;
    explicit HipBackend(int device) : m_device_id{device} {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count <= 0) {
            sanm_throw(SANM_ERR_HIP,
                       "no HIP device available (%s); the sanm_hip product path has no CPU "
                       "fallback",
                       e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        }
        HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(hipStreamCreate(&m_main));
        m_stream = m_main;
        HIP_CHECK(hipMalloc(&m_scalar, 64));
        HIP_CHECK(hipHostMalloc(&m_scalar_host, 64));
        // The library's device code is loaded when a kernel of it is first touched -- 6-7 ms that the first solver's
        // constructor paid inside the reference's time_solve: here, with the device's other one-time set-up.  (The pass
        // kernels of a graph are a code object of their own, loaded when that graph's program is built.)
        hipFuncAttributes attr;
        (void)hipFuncGetAttributes(&attr, (const void*)mfk::scatter_kernel);
        ensure_stage();
        // ... and the runtime's own first-use set-up of the copy and fill paths (an API trace of a cold solve of
        // armadillo_small, gpurun_out/r6hip1: the first hipMemcpyAsync 8 ms, the first hipMemset 6 ms -- its fill kernel's
        // code object --, against 0.15 and 0.01 ms for every later one)
        {
            double tmp[8] = {};
            HIP_CHECK(hipMemsetAsync(m_scalar, 0, 64, m_main));
            HIP_CHECK(hipMemset(m_scalar, 0, 64));
            h2d(m_scalar, tmp, sizeof(tmp));
            d2h(tmp, m_scalar, sizeof(tmp));
            // (a copy of every kind the solver makes: a MB from pinned memory, device to device, device to pinned memory)
            void* dev = nullptr;
            HIP_CHECK(hipMalloc(&dev, size_t(2) << 20));
            HIP_CHECK(hipMemcpyAsync(dev, m_stage[0], size_t(1) << 20, hipMemcpyHostToDevice, m_main));
            HIP_CHECK(hipMemcpyAsync(static_cast<char*>(dev) + (size_t(1) << 20), dev, size_t(1) << 20, hipMemcpyDeviceToDevice, m_main));
            HIP_CHECK(hipMemcpyAsync(m_scalar_host, m_scalar, 64, hipMemcpyDeviceToHost, m_main));
            HIP_CHECK(hipStreamSynchronize(m_main));
            HIP_CHECK(hipFree(dev));
        }
    }
    void forget_chains(const void* key) override {
        m_cur_sch = nullptr;
        forget_solve_blocks(key);
        for (auto it = m_chains.begin(); it != m_chains.end();) {
            if (it->first.first == key) {
                if (it->second.exec) {
                    (void)hipStreamSynchronize(m_main);
                    (void)hipGraphExecDestroy(it->second.exec);
                }
                it = m_chains.erase(it);
            } else {
                ++it;
            }
        }
    }
    ~HipBackend() override {
        for (auto& kv : m_chains)
            if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (m_mark_ev) (void)hipEventDestroy(m_mark_ev);
        if (m_dlu_work) (void)hipFree(m_dlu_work);
        if (m_probe_work) (void)hipFree(m_probe_work);
        if (m_probe_groups) (void)hipHostFree(m_probe_groups);
        if (m_probe_results) (void)hipHostFree(m_probe_results);
        if (m_probe_partials) (void)hipFree(m_probe_partials);
        if (m_probe_tickets) (void)hipFree(m_probe_tickets);
        comm_destroy();
        (void)hipFree(m_scalar);
        for (auto& kv : m_pool_free)
            for (void* p : kv.second) (void)hipFree(p);
        if (m_red.partials) (void)hipFree(m_red.partials);
        if (m_red.ticket) (void)hipFree(m_red.ticket);
        if (m_red.host) (void)hipHostFree(m_red.host);
        for (double* w : m_pcg_w)
            if (w) (void)hipFree(w);
        if (m_pcg_sc) (void)hipFree(m_pcg_sc);
        if (m_pcg_sc_host) (void)hipHostFree(m_pcg_sc_host);
        (void)hipHostFree(m_scalar_host);
        for (hipEvent_t e : m_event_pool) (void)hipEventDestroy(e);
        for (hipEvent_t e : m_fork_ev)
            if (e) (void)hipEventDestroy(e);
        if (m_red_rider.partials) (void)hipFree(m_red_rider.partials);
        if (m_red_rider.ticket) (void)hipFree(m_red_rider.ticket);
        if (m_red_rider.host) (void)hipHostFree(m_red_rider.host);
        if (m_red_side.partials) (void)hipFree(m_red_side.partials);
        if (m_red_side.ticket) (void)hipFree(m_red_side.ticket);
        if (m_red_side.host) (void)hipHostFree(m_red_side.host);
        if (m_side) (void)hipStreamDestroy(m_side);
        for (int b = 0; b < 2; ++b) {
            if (m_stage[b]) (void)hipHostFree(m_stage[b]);
            if (m_stage_ev[b]) (void)hipEventDestroy(m_stage_ev[b]);
        }
        for (hipStream_t q : m_sf_queue)
            if (q) (void)hipStreamDestroy(q);
        (void)hipStreamDestroy(m_main);
    }
    const char* name() const override { return "hip"; }
    int64_t launch_count() const override { return m_launch_count; }
    void dense_lu_factor(const CsrDev& A, double* lu, int32_t* piv, double* status) override {
        SANM_LAUNCH(dense_from_csr_kernel, dim3((unsigned)std::min<int64_t>(A.n, 1024)), dim3(256), 0, m_stream, A, lu);
        // small systems: the single workgroup; beyond: panels + trailing tiles over the chip (same factors bit for bit).
        // SANM_DENSE_BLOCKED_MIN_N: from how many unknowns (tests: 0)
        const char* env = std::getenv("SANM_DENSE_BLOCKED_MIN_N");
        const int64_t blocked_min = env ? std::atoll(env) : 384;
        const int64_t n = A.n;
        if (n < blocked_min) {
            SANM_LAUNCH(dense_lu_kernel, dim3(1), dim3(DLU_THREADS), 0, m_stream, n, lu, piv, status);
            HIP_CHECK(hipGetLastError());
            return;
        }
        if ((int64_t)m_dlu_flags_n < n) {
            if (m_dlu_flags) HIP_CHECK(hipFree(m_dlu_flags));
            HIP_CHECK(hipMalloc(&m_dlu_flags, n * sizeof(int32_t) + 3 * sizeof(double) + 8));
            m_dlu_flags_n = n;
        }
        int32_t* flags = static_cast<int32_t*>(m_dlu_flags);
        double* acc3 = reinterpret_cast<double*>(static_cast<char*>(m_dlu_flags) + (n * sizeof(int32_t) + 7) / 8 * 8);
        for (int64_t c0 = 0; c0 < n; c0 += DLU_NB) {
            const int w = (int)std::min<int64_t>(DLU_NB, n - c0);
            SANM_LAUNCH(dense_lu_panel_kernel, dim3(1), dim3(DLU_THREADS), 0, m_stream, n, lu, piv, flags, acc3, status, c0, w,
                        (int)(c0 == 0), (int)(c0 + w >= n));
            if (n - w > 0)
                SANM_LAUNCH(dense_lu_rowops_kernel, dim3(nblk(n - w, 256)), dim3(256), 0, m_stream, n, lu, piv, flags, c0, w);
            const int64_t rest = n - c0 - w;
            if (rest > 0) {
                const unsigned tiles = (unsigned)((rest + 63) / 64);
                SANM_LAUNCH(dense_lu_trailing_kernel, dim3(tiles, tiles), dim3(256), 0, m_stream, n, lu, flags, c0, w);
            }
        }
        HIP_CHECK(hipGetLastError());
    }
    void* m_dlu_flags = nullptr;
    size_t m_dlu_flags_n = 0;
    void dense_lu_solve(int64_t n, const double* lu, const int32_t* piv, const double* b, double* x) override {
        if ((int64_t)m_dlu_work_n < n) {
            if (m_dlu_work) HIP_CHECK(hipFree(m_dlu_work));
            HIP_CHECK(hipMalloc(&m_dlu_work, n * sizeof(double)));
            m_dlu_work_n = n;
        }
        SANM_LAUNCH(dense_lu_solve_kernel, dim3(1), dim3(DLU_THREADS), 0, m_stream, n, lu, piv, b, x, m_dlu_work);
        HIP_CHECK(hipGetLastError());
    }
    double* m_dlu_work = nullptr;
    size_t m_dlu_work_n = 0;
    void run_vec_pass(const VecProgDev& P, int mode, int order, const double* xin) override {
        const size_t lds = mode == PASS_GRAD ? (size_t)P.grad_total * sizeof(double) : 0;
        SANM_LAUNCH(vec_pass_kernel, dim3((unsigned)P.B), dim3(VEC_MAX_SIZE), lds, m_stream, P, mode, order, xin);
        HIP_CHECK(hipGetLastError());
    }

    bool comm_available() override {
        try {
            return Rccl::get().AllReduce != nullptr;
        } catch (...) {
            return false;
        }
    }
    void comm_unique_id(void* id128) override {
        const Rccl& r = Rccl::get();
        Rccl::UniqueId id;
        r.check(r.GetUniqueId(&id), "ncclGetUniqueId");
        std::memcpy(id128, id.internal, 128);
    }
    void comm_init(int rank, int world, const void* id128) override {
        const Rccl& r = Rccl::get();
        comm_destroy();
        Rccl::UniqueId id;
        std::memcpy(id.internal, id128, 128);
        r.check(r.CommInitRank(&m_comm, world, id, rank), "ncclCommInitRank");
        m_comm_rank = rank;
        m_comm_world = world;
    }
    void comm_destroy() override {
        if (!m_comm) return;
        (void)hipStreamSynchronize(m_stream);
        (void)Rccl::get().CommDestroy(m_comm);
        m_comm = nullptr;
        m_comm_world = 0;
    }
    int comm_world() const override { return m_comm_world; }
    int comm_rank() const override { return m_comm_rank; }
    void comm_query(int* world, int* rank) override {
        // what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank), not what we passed in
        *world = *rank = 0;
        if (!m_comm) return;
        const Rccl& r = Rccl::get();
        if (!r.CommCount || !r.CommUserRank) sanm_throw(SANM_ERR_UNSUPPORTED, "this RCCL has no ncclCommCount");
        r.check(r.CommCount(m_comm, world), "ncclCommCount");
        r.check(r.CommUserRank(m_comm, rank), "ncclCommUserRank");
    }
    void allreduce_sum(double* buf, int64_t count) override {
        if (!m_comm) sanm_throw(SANM_ERR_ASSERT, "all-reduce without a communicator (sanm_hip_comm_init first)");
        const Rccl& r = Rccl::get();
        r.check(r.AllReduce(buf, buf, (size_t)count, /*ncclFloat64*/ 8, /*ncclSum*/ 0, m_comm, m_stream), "ncclAllReduce");
    }

    bool comm_p2p_available() override {
        if (!m_comm) return false;
        const Rccl& r = Rccl::get();
        return r.Send && r.Recv && r.Broadcast && r.GroupStart && r.GroupEnd;
    }
    void comm_exchange(double* base, const MfSchedule::Xfer* x, int n) override {
        if (!m_comm) sanm_throw(SANM_ERR_ASSERT, "exchange without a communicator (sanm_hip_comm_init first)");
        const Rccl& r = Rccl::get();
        if (!comm_p2p_available()) sanm_throw(SANM_ERR_UNSUPPORTED, "this RCCL has no ncclSend / ncclRecv / ncclBroadcast");
        // (one group: the transfers of a stage are independent of each other and RCCL runs them together; a rank that
        // takes part in none of them opens and closes an empty group)
        r.check(r.GroupStart(), "ncclGroupStart");
        int rc = 0;
        for (int i = 0; i < n && rc == 0; ++i) {
            double* p = base + x[i].off;
            const size_t cnt = (size_t)x[i].cnt;
            if (cnt == 0) continue;
            if (x[i].dst < 0) {
                rc = r.Broadcast(p, p, cnt, /*ncclFloat64*/ 8, x[i].src, m_comm, m_stream);
            } else if (x[i].src != x[i].dst) {
                if (m_comm_rank == x[i].src) rc = r.Send(p, cnt, 8, x[i].dst, m_comm, m_stream);
                else if (m_comm_rank == x[i].dst) rc = r.Recv(p, cnt, 8, x[i].src, m_comm, m_stream);
            }
        }
        const int rc_end = r.GroupEnd();
        r.check(rc, "ncclSend / ncclRecv / ncclBroadcast");
        r.check(rc_end, "ncclGroupEnd");
    }
    // Work vectors come and go every continuation step (Pade basis, range checks); hipMalloc / hipFree
    // cost tens of microseconds each and hipFree synchronises the device, so freed blocks of up to
    // kPoolBlockMax bytes are kept and handed out again by exact size.  Everything runs on one stream, so a
    // recycled block is never written before its previous readers have finished.
    void* alloc(size_t bytes) override {
        if (!bytes) bytes = 8;
        auto it = m_pool_free.find(bytes);
        if (it != m_pool_free.end() && !it->second.empty()) {
            void* p = it->second.back();
            it->second.pop_back();
            m_pool_cached -= bytes;
            m_live[p] = bytes;
            return p;
        }
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, bytes));
        m_live[p] = bytes;
        return p;
    }
    void* alloc_detached(size_t bytes) override {
        static const bool off = std::getenv("SANM_NO_DETACHED_ALLOC") != nullptr;
        if (off || !bytes) return nullptr;
        void* p = nullptr;
        if (hipSetDevice(m_device_id) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        return p;
    }
    void free_detached(void* p) override {
        if (p) (void)hipFree(p);
    }
    void adopt(void* p, size_t bytes) override { m_live[p] = bytes ? bytes : 8; }
    void free(void* p) override {
        if (!p) return;
        auto it = m_live.find(p);
        const size_t bytes = it == m_live.end() ? 0 : it->second;
        if (it != m_live.end()) m_live.erase(it);
        if (bytes && bytes <= kPoolBlockMax && m_pool_cached + bytes <= kPoolTotalMax) {
            m_pool_free[bytes].push_back(p);
            m_pool_cached += bytes;
        } else {
            (void)hipFree(p);
        }
    }
    // Uploads go through two pinned staging buffers of our own.  hipMemcpy from pageable memory pins the caller's pages
    // in place, and the tables a constructor uploads are temporaries: unmapping pages the driver has seen pinned stops
    // the process's queues, ~20 ms at the next submission (scripts/ctor_sync_probe.py: a second solver's constructor
    // 74 -> 53 ms just by keeping the analysis's upload buffers alive).  Staged, nothing of the caller's is ever pinned.
    // SANM_H2D_DIRECT=1: the plain hipMemcpy.
    static constexpr size_t kStageBytes = size_t(8) << 20;
    void* m_stage[2] = {nullptr, nullptr};
    hipEvent_t m_stage_ev[2] = {nullptr, nullptr};
    void ensure_stage() {
        if (m_stage[0]) return;
        for (int b = 0; b < 2; ++b) {
            HIP_CHECK(hipHostMalloc(&m_stage[b], kStageBytes));
            HIP_CHECK(hipEventCreateWithFlags(&m_stage_ev[b], hipEventDisableTiming));
        }
    }
    void h2d(void* dst, const void* src, size_t bytes) override {
        if (!bytes) return;
        static const bool direct = std::getenv("SANM_H2D_DIRECT") != nullptr;
        if (direct) {  // (small copies take the staging buffers too: nothing of the caller's is ever pinned in place)
            HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, m_stream));
            HIP_CHECK(hipStreamSynchronize(m_stream));
            return;
        }
        ensure_stage();
        int b = 0;
        bool used[2] = {false, false};
        for (size_t off = 0; off < bytes; off += kStageBytes, b ^= 1) {
            const size_t len = std::min(kStageBytes, bytes - off);
            if (used[b]) HIP_CHECK(hipEventSynchronize(m_stage_ev[b]));  // (the copy out of this buffer two chunks ago)
            const char* from = static_cast<const char*>(src) + off;
            char* stage = static_cast<char*>(m_stage[b]);
            parallel_ranges((int64_t)len, 1 << 20, [&](int64_t r0, int64_t r1, int) { std::memcpy(stage + r0, from + r0, (size_t)(r1 - r0)); });
            HIP_CHECK(hipMemcpyAsync(static_cast<char*>(dst) + off, stage, len, hipMemcpyHostToDevice, m_stream));
            HIP_CHECK(hipEventRecord(m_stage_ev[b], m_stream));
            used[b] = true;
        }
        HIP_CHECK(hipStreamSynchronize(m_stream));
    }
    void d2h_async(void* dst_pinned, const void* src, size_t bytes) override {
        if (!bytes) return;
        flush_deferred();
        HIP_CHECK(hipMemcpyAsync(dst_pinned, src, bytes, hipMemcpyDeviceToHost, m_stream));
    }
    void d2h(void* dst, const void* src, size_t bytes) override {
        if (!bytes) return;
        flush_deferred();
        if (m_stream == m_main && m_side_dirty && !m_side_detached) {  // what the host reads may come from the side queue
            HIP_CHECK(hipStreamSynchronize(m_side));
            m_side_dirty = false;
        }
        static const bool direct = std::getenv("SANM_H2D_DIRECT") != nullptr;
        if (direct || bytes <= 4096) {  // (scalars: the runtime stages those itself)
            HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, m_stream));
            HIP_CHECK(hipStreamSynchronize(m_stream));
            return;
        }
        // through the staging buffers, like h2d: the caller's pages are never pinned
        ensure_stage();
        for (size_t off = 0; off < bytes; off += kStageBytes) {
            const size_t len = std::min(kStageBytes, bytes - off);
            HIP_CHECK(hipMemcpyAsync(m_stage[0], static_cast<const char*>(src) + off, len, hipMemcpyDeviceToHost, m_stream));
            HIP_CHECK(hipStreamSynchronize(m_stream));
            std::memcpy(static_cast<char*>(dst) + off, m_stage[0], len);
        }
    }
    void d2d(void* dst, const void* src, size_t bytes) override {
        if (!bytes) return;
        HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, m_stream));
    }
    void zero(void* dst, size_t bytes) override {
        if (!bytes) return;
        HIP_CHECK(hipMemsetAsync(dst, 0, bytes, m_stream));
    }
    hipEvent_t m_mark_ev = nullptr;
    void mark() override {
        flush_deferred();
        if (!m_mark_ev) HIP_CHECK(hipEventCreateWithFlags(&m_mark_ev, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(m_mark_ev, m_stream));
    }
    void wait_mark() override {
        if (!m_mark_ev) {
            sync();
            return;
        }
        HIP_CHECK(hipEventSynchronize(m_mark_ev));
    }
    void sync() override {
        flush_deferred();
        HIP_CHECK(hipStreamSynchronize(m_main));
        if (m_side_dirty && !m_side_detached) {
            HIP_CHECK(hipStreamSynchronize(m_side));
            m_side_dirty = false;
        }
    }
    void side_detach() override {
        if (m_stream != m_main) sanm_throw(SANM_ERR_ASSERT, "side_detach inside a fork");
        m_side_detached = m_side_dirty;
    }
    void side_wait() override {
        if (m_side_dirty) HIP_CHECK(hipStreamSynchronize(m_side));
        m_side_dirty = m_side_detached = false;
    }
    hipEvent_t next_fork_event() {
        hipEvent_t& e = m_fork_ev[m_fork_next];
        m_fork_next = (m_fork_next + 1) % kForkEvents;
        if (!e) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return e;
    }
    void side_fork() override {
        if (m_stream != m_main) sanm_throw(SANM_ERR_ASSERT, "side_fork inside a fork");
        if (m_side_detached) sanm_throw(SANM_ERR_ASSERT, "side_fork while the side queue is detached");
        if (!m_side) HIP_CHECK(hipStreamCreateWithFlags(&m_side, hipStreamNonBlocking));
        red_for(m_red_side);  // (allocated outside any launch sequence)
        hipEvent_t e = next_fork_event();
        HIP_CHECK(hipEventRecord(e, m_main));
        HIP_CHECK(hipStreamWaitEvent(m_side, e, 0));
        m_stream = m_side;
        m_side_dirty = true;
    }
    void side_end() override { m_stream = m_main; }
    void side_join() override {
        if (m_stream != m_main) sanm_throw(SANM_ERR_ASSERT, "side_join inside a fork");
        if (!m_side_dirty || m_side_detached) return;
        hipEvent_t e = next_fork_event();
        HIP_CHECK(hipEventRecord(e, m_side));
        HIP_CHECK(hipStreamWaitEvent(m_main, e, 0));
        m_side_dirty = false;
    }
    hipStream_t stream() const { return m_stream; }

    int specialize(const char* source) override {
        std::vector<char> code;
        std::string log;
        const RtcStats before = rtc_stats();
        const bool compiled_ok = rtc_compile(source, code, log);
        const RtcStats after = rtc_stats();
        m_last_spec_source = after.embedded_hits > before.embedded_hits ? 4 : after.compiled > before.compiled ? 3
                             : after.disk_hits > before.disk_hits ? 2 : after.memory_hits > before.memory_hits ? 1 : 0;
        if (!compiled_ok) {
            std::fprintf(stderr, "sanm_hip: run-time compilation of the pass kernels failed, using the interpreter kernels\n%s\n",
                         log.c_str());
            return -1;
        }
        SpecKernels k;
        if (hipModuleLoadData(&k.mod, code.data()) != hipSuccess) {
            std::fprintf(stderr, "sanm_hip: could not load the compiled pass kernels, using the interpreter kernels\n");
            return -1;
        }
        for (int m = 0; m < 5; ++m) {
            const std::string name = "spec_pass" + std::to_string(m);
            if (hipModuleGetFunction(&k.pass[m], k.mod, name.c_str()) != hipSuccess) {
                (void)hipModuleUnload(k.mod);
                return -1;
            }
        }
        for (size_t i = 0; i < m_spec.size(); ++i)
            if (!m_spec[i].mod) {
                m_spec[i] = k;
                return (int)i;
            }
        m_spec.push_back(k);
        return (int)m_spec.size() - 1;
    }
    int m_last_spec_source = 0;
    int last_specialize_source() const override { return m_last_spec_source; }
    void release_specialized(int id) override {
        if (id < 0 || id >= (int)m_spec.size() || !m_spec[id].mod) return;
        (void)hipStreamSynchronize(m_stream);
        (void)hipModuleUnload(m_spec[id].mod);
        m_spec[id] = SpecKernels{};
    }
    //! one launch of a pass kernel compiled for P's graph.  nc (PASS_COEFF_BIAS only): the launch also stands for
    //! next_coeff -- see spec_pass4 in the generated source (graph.cpp)
    void launch_spec(const ProgramDev& P, int mode, int order, const double* xvec, int nparts, size_t lds,
                     const NextCoeff* nc) {
        struct Args {  // SPEC_PARAMS of the generated source (graph.cpp) ...
            double* arena;
            const uint32_t* rin_idx;
            const double* rin_coef;
            const double* xvec;
            const double* lc_params;
            long long T, Tpad;
            int order, max_order, rin_nslot;
        };
        struct Args4 {  // ... and what spec_pass4 takes behind them
            Args a;  // (72 bytes with its tail padding: the first pointer behind it is 8-aligned in the kernel's list too)
            const double* nc_xg;
            const double* nc_num;
            double nc_scale;
            const double* nc_sc;
            double* nc_out;
            double* nc_thost;
            unsigned long long nc_n;
            double* rd_out;
            const double* rd_norm2;
            double rd_eps;
            unsigned long long rd_n;
            double* g_partials;
            unsigned* g_ticket;
            double* g_host;
            unsigned own, nc_blocks, rd_nblk;
        };
        static_assert(offsetof(Args4, nc_xg) == 72, "kernel argument layout");
        const unsigned own = nblk(P.T, 64);
        Args4 a4{};
        a4.a = Args{P.arena, P.rin.idx, P.rin.coef, xvec, P.lc_params, (long long)P.T, (long long)P.Tpad, order, P.max_order,
                    P.rin.nslot};
        a4.own = own;
        unsigned extra = 0;
        if (nc) {
            a4.a.xvec = nc->xb;
            a4.nc_xg = nc->xg;
            a4.nc_num = nc->num;
            a4.nc_scale = nc->scale;
            a4.nc_sc = nc->sc;
            a4.nc_out = nc->out;
            a4.nc_thost = nc->t_out;
            a4.nc_n = nc->n;
            a4.nc_blocks = std::min<unsigned>(nblk(nc->n + 1, 256), 256);
            if (m_pending.kind == 3 && m_stream == m_main) {  // the scaling of a Gram-Schmidt step rides along
                const GsRider rd = take_rider();
                a4.rd_out = rd.out;
                a4.rd_norm2 = rd.norm2;
                a4.rd_eps = rd.eps;
                a4.rd_n = rd.n;
                a4.g_partials = rd.g.partials;
                a4.g_ticket = rd.g.ticket;
                a4.g_host = rd.g.host;
                a4.rd_nblk = rd.nblk;
            }
            extra = a4.nc_blocks + a4.rd_nblk;
        }
        size_t arg_size = mode == PASS_COEFF_BIAS ? sizeof(Args4) : sizeof(Args);
        void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a4, HIP_LAUNCH_PARAM_BUFFER_SIZE, &arg_size,
                          HIP_LAUNCH_PARAM_END};
        ++m_launch_count;
        HIP_CHECK(hipModuleLaunchKernel(m_spec[P.spec_id].pass[mode], own + extra, mode == PASS_GRAD ? P.odim : 1, 1,
                                        64 * nparts, 1, 1, (unsigned)lds, m_stream, nullptr, config));
    }
    bool run_pass_next_coeff(const ProgramDev& P, int order, const NextCoeff& nc) override {
        static const bool off = std::getenv("SANM_NO_NEXT_COEFF_FUSION") != nullptr;
        const int nparts = order + 1 >= m_conv_split_order ? m_conv_parts : 1;
        const size_t lds = (size_t)(P.cur_size + (nparts - 1) * 9) * 64 * sizeof(double);
        if (off || P.spec_id < 0 || lds > 48 * 1024 || m_stream != m_main) return false;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (m_time_passes) {
            HIP_CHECK(hipEventCreate(&e0));
            HIP_CHECK(hipEventCreate(&e1));
            HIP_CHECK(hipEventRecord(e0, m_stream));
        }
        launch_spec(P, PASS_COEFF_BIAS, order, nc.xb, nparts, lds, &nc);
        if (m_time_passes) {
            HIP_CHECK(hipEventRecord(e1, m_stream));
            m_pass_events.emplace_back(e0, e1);
        }
        return true;
    }
    void run_pass(const ProgramDev& P, int mode, int order, const double* xvec) override {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (m_time_passes) {
            HIP_CHECK(hipEventCreate(&e0));
            HIP_CHECK(hipEventCreate(&e1));
            HIP_CHECK(hipEventRecord(e0, m_stream));
        }
        // convolutions of an order-k bias have k-1 terms: worth splitting over wavefronts once they
        // outweigh the two barriers per operator
        // (the fused pass runs BIAS(order + 1))
        int nparts = ((mode == PASS_BIAS && order >= m_conv_split_order) ||
                      (mode == PASS_COEFF_BIAS && order + 1 >= m_conv_split_order))
                             ? m_conv_parts
                             : 1;
        const size_t lds = (size_t)(P.cur_size + (nparts - 1) * 9) * 64 * sizeof(double);
        if (lds > 160 * 1024) sanm_throw(SANM_ERR_UNSUPPORTED, "graph too large for the LDS scratch");
        if (mode == PASS_COEFF_BIAS && (P.spec_id < 0 || lds > 48 * 1024)) {
            // the fused pass exists among the kernels compiled per graph only: two launches otherwise
            if (m_time_passes) {
                (void)hipEventDestroy(e0);
                (void)hipEventDestroy(e1);
            }
            run_pass(P, PASS_COEFF, order, xvec);
            run_pass(P, PASS_BIAS, order + 1, nullptr);
            return;
        }
        if (P.spec_id >= 0 && lds <= 48 * 1024) {
            // this program's own kernels (same grid, same LDS layout as the interpreter's)
            launch_spec(P, mode, order, xvec, nparts, lds, nullptr);
            if (m_time_passes) {
                HIP_CHECK(hipEventRecord(e1, m_stream));
                m_pass_events.emplace_back(e0, e1);
            }
            return;
        }
        void (*kern)(ProgramDev, const OpDesc*, const VarDesc*, int, const double*) = nullptr;
        switch (mode) {
            case PASS_EVAL0: kern = taylor_pass_kernel<PASS_EVAL0, 1>; break;
            case PASS_GRAD: kern = taylor_pass_kernel<PASS_GRAD, 1>; break;
            case PASS_BIAS: kern = taylor_pass_kernel<PASS_BIAS, kBiasWaves>; break;
            case PASS_COEFF: kern = taylor_pass_kernel<PASS_COEFF, 1>; break;
            default: sanm_throw(SANM_ERR_ASSERT, "unknown pass mode %d", mode);
        }
        if (lds > m_pass_lds_limit[mode]) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            m_pass_lds_limit[mode] = lds;
        }
        ProgramDev Pd = P;
        // measurement hook (scripts/time_nops.py): the pass truncated after n operators, interpreter kernels only
        if (const char* e = std::getenv("SANM_DBG_NOPS")) Pd.nops = std::min(P.nops, std::atoi(e));
        SANM_LAUNCH(kern, dim3(nblk(P.T, 64), mode == PASS_GRAD ? P.odim : 1), dim3(64 * nparts), lds,
                           m_stream, Pd, P.ops, P.vars, order, xvec);
        HIP_CHECK(hipGetLastError());
        if (m_time_passes) {
            HIP_CHECK(hipEventRecord(e1, m_stream));
            m_pass_events.emplace_back(e0, e1);
        }
    }
    void phase_begin(const char* tag) override {
        PhaseBracket b{tag, pooled_event(), pooled_event()};
        HIP_CHECK(hipEventRecord(b.e0, m_stream));
        m_phase_open.push_back(std::move(b));
    }
    void phase_end() override {
        if (m_phase_open.empty()) return;
        HIP_CHECK(hipEventRecord(m_phase_open.back().e1, m_stream));
        m_phase_done.push_back(std::move(m_phase_open.back()));
        m_phase_open.pop_back();
    }
    void phase_collect(std::map<std::string, double>& acc, std::map<std::string, double>* cnt) override {
        HIP_CHECK(hipStreamSynchronize(m_main));
        if (m_side) HIP_CHECK(hipStreamSynchronize(m_side));  // brackets recorded on the side queue
        for (auto& b : m_phase_done) {
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, b.e0, b.e1));
            acc[b.tag] += ms * 1e-3;
            if (cnt) (*cnt)[b.tag] += 1;
            m_event_pool.push_back(b.e0);
            m_event_pool.push_back(b.e1);
        }
        m_phase_done.clear();
    }
    void enable_pass_timing(bool on) override {
        for (auto& ev : m_pass_events) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        m_pass_events.clear();
        m_time_passes = on;
    }
    void pass_timing(double* total_ms, int64_t* count) override {
        HIP_CHECK(hipStreamSynchronize(m_stream));
        double tot = 0;
        for (auto& ev : m_pass_events) {
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, ev.first, ev.second));
            tot += ms;
        }
        *total_ms = tot;
        *count = m_pass_events.size();
    }
    // ---- deferred Gram-Schmidt phases (Backend::defer_gs_phase; GsRider above) ------------------------------
    GsPhase m_pending{};  // kind 0: none
    GridRed m_red_rider{nullptr, nullptr, nullptr};
    const bool m_no_riders = std::getenv("SANM_NO_RIDERS") != nullptr;
    void defer_gs_phase(const GsPhase& ph) override {
        flush_deferred();
        if (m_no_riders || m_stream != m_main || (ph.kind == 1 && ph.nvec == 0) || ph.nvec > MAX_VEC) {
            run_gs_phase(ph);
            return;
        }
        red_for(m_red_rider);
        m_pending = ph;
    }
    void flush_deferred() override {
        if (!m_pending.kind) return;
        const GsPhase ph = m_pending;
        m_pending.kind = 0;
        run_gs_phase(ph);
    }
    //! the pending phase as the extra workgroups of a carrier launch; clears it
    GsRider take_rider() {
        const GsPhase& ph = m_pending;
        GsRider r{};
        r.n = ph.n;
        r.x = ph.x;
        r.q.n = ph.nvec;
        for (int j = 0; j < ph.nvec; ++j) r.q.p[j] = ph.vecs[j];
        r.coefs = ph.coefs;
        r.first = ph.first;
        r.out = ph.out;
        r.norm2 = ph.norm2;
        r.nn2 = ph.nn2;
        r.eps = ph.eps;
        r.nblk = red_grid(ph.n);
        r.g = m_red_rider;
        r.g.host = ph.red_out;
        m_pending.kind = 0;
        return r;
    }
    static int nvt_group(int nvec) { return nvec <= 4 ? 4 : (nvec + 3) / 4 * 4; }

    void gather_rows(const SparseRowsDev& R, const double* src, double* dst, const int32_t* perm,
                     double* dst2) override {
        if (R.bptr) {
            const unsigned own = nblk((size_t)R.nrows / 3 * GATHER_LANES, 256);
            if (m_pending.kind == 1 && m_stream == m_main) {  // the projections of a Gram-Schmidt step ride along
                const int nvt = nvt_group(m_pending.nvec);
                const GsRider rd = take_rider();
#define SANM_G3(NVT)                                                                                                \
    case NVT:                                                                                                       \
        SANM_LAUNCH(gather_rows3_kernel<NVT>, dim3(own + rd.nblk), dim3(256), 0, m_stream, R.bptr, R.bidx,   \
                           R.bcoef, src, dst, perm, dst2, R.nrows, own, rd);                                        \
        break;
                switch (nvt) {
                    SANM_G3(4) SANM_G3(8) SANM_G3(12) SANM_G3(16) SANM_G3(20) SANM_G3(24)
                    default: sanm_throw(SANM_ERR_ASSERT, "rider: %d vectors", nvt);
                }
#undef SANM_G3
            } else {
                SANM_LAUNCH(gather_rows3_kernel<0>, dim3(own), dim3(256), 0, m_stream, R.bptr, R.bidx, R.bcoef, src,
                                   dst, perm, dst2, R.nrows, own, GsRider{});
            }
        } else
            SANM_LAUNCH(gather_rows_kernel, dim3(nblk((size_t)R.nrows * GATHER_LANES, 256)), dim3(256), 0,
                               m_stream, R, src, dst, perm, dst2);
        HIP_CHECK(hipGetLastError());
    }
    void assemble(const AssemblyDev& A, const double* jac, double* val, double* grad_t) override {
        sanm_check(A.aptr, "assembly lists were not prepared");
        sanm_check(!A.has_t || grad_t, "assembly with a t column needs grad_t");
        if (A.triples) {
            const AsmList L{A.aptr, A.ajidx, A.acoef, A.ntslot};
            SANM_LAUNCH(assemble3_kernel, dim3(nblk(L.nslots * ROW_LANES, 256)), dim3(256), 0, m_stream, L, A.tslot_p, A.tslot_len,
                        3 * A.idim, jac, val);
            HIP_CHECK(hipGetLastError());
            return;
        }
        const AsmList L{A.aptr, A.ajidx, A.acoef, A.nnz};
        SANM_LAUNCH(assemble_kernel, dim3(nblk(L.nslots * ROW_LANES, 256)), dim3(256), 0, m_stream, L, jac, val);
        if (A.has_t) {
            const AsmList Lt{A.tptr, A.tjidx, A.tcoef, A.n};
            SANM_LAUNCH(assemble_kernel, dim3(nblk(Lt.nslots * ROW_LANES, 256)), dim3(256), 0, m_stream, Lt, jac, grad_t);
        }
        HIP_CHECK(hipGetLastError());
    }
    void prepare_assembly(AssemblyDev& A, std::vector<void*>& owned) override {
        // counts per non-zero (and per row for the t column), offsets on the host, then the lists themselves
        const size_t nnz = (size_t)A.nnz, n = (size_t)A.n;
        SetupLaps laps("assembly lists");
        const bool dbg = std::getenv("SANM_DEBUG_SETUP") != nullptr;
        uint32_t* cnt = static_cast<uint32_t*>(alloc((nnz + 1) * 4));
        uint32_t* tcnt = static_cast<uint32_t*>(alloc((n + 1) * 4));
        owned.push_back(cnt);
        owned.push_back(tcnt);
        zero(tcnt, (n + 1) * 4);
        // (rows in triples: lists of the rows 3u only; the other rows' counts stay zero)
        const int step = A.triples ? 3 : 1;
        const unsigned rows = (unsigned)(n / step);
        if (A.triples) zero(cnt, (nnz + 1) * 4);
        SANM_LAUNCH(asm_list_kernel<0>, dim3(rows), dim3(ASM_THREADS), 0, m_stream, A, cnt, tcnt, nullptr, nullptr,
                    nullptr, nullptr, step);
        HIP_CHECK(hipGetLastError());
        if (dbg) sync();
        laps.lap("count kernel");
        auto scan = [&](uint32_t* v, size_t m, const char* what) {
            const size_t nb = (m + SCAN_BLOCK - 1) / SCAN_BLOCK;
            uint64_t* bsum = static_cast<uint64_t*>(alloc((nb + 1) * 8));
            if (nb) SANM_LAUNCH(scan_block_sums_kernel, dim3((unsigned)nb), dim3(256), 0, m_stream, v, m, bsum);
            SANM_LAUNCH(scan_offsets_kernel, dim3(1), dim3(256), 0, m_stream, bsum, nb);
            SANM_LAUNCH(scan_apply_kernel, dim3((unsigned)std::max<size_t>(nb, 1)), dim3(256), 0, m_stream, v, m, bsum, nb);
            HIP_CHECK(hipGetLastError());
            uint64_t total = 0;
            d2h(&total, bsum + nb, 8);
            free(bsum);
            sanm_check(total < std::numeric_limits<uint32_t>::max(), "%s list too large", what);
            return (size_t)total;
        };
        const size_t tot = scan(cnt, nnz, "assembly"), ttot = scan(tcnt, n, "grad_t");
        laps.lap("offsets");
        uint32_t* jx = static_cast<uint32_t*>(alloc(std::max<size_t>(tot, 1) * 4));
        double* cf = static_cast<double*>(alloc(std::max<size_t>(tot, 1) * 8));
        uint32_t* tjx = static_cast<uint32_t*>(alloc(std::max<size_t>(ttot, 1) * 4));
        double* tcf = static_cast<double*>(alloc(std::max<size_t>(ttot, 1) * 8));
        owned.push_back(jx);
        owned.push_back(cf);
        owned.push_back(tjx);
        owned.push_back(tcf);
        laps.lap("alloc lists");
        A.aptr = cnt;
        A.tptr = tcnt;
        SANM_LAUNCH(asm_list_kernel<1>, dim3(rows), dim3(ASM_THREADS), 0, m_stream, A, nullptr, nullptr, jx, cf, tjx, tcf, step);
        HIP_CHECK(hipGetLastError());
        if (dbg) sync();
        laps.lap("fill kernel");
        A.ajidx = jx;
        A.acoef = cf;
        A.tjidx = tjx;
        A.tcoef = tcf;
    }
    void gather(size_t n, const double* src, const uint32_t* idx, double* dst) override {
        SANM_LAUNCH(gather_kernel, dim3(nblk(n, 256)), dim3(256), 0, m_stream, n, src, idx, dst);
        HIP_CHECK(hipGetLastError());
    }
    void ata(const CsrDev& At, const CsrDev& M, const uint32_t* mrow, double lambda) override {
        SANM_LAUNCH(ata_kernel, dim3(nblk(M.nnz, 256)), dim3(256), 0, m_stream, At, M, mrow, lambda);
        HIP_CHECK(hipGetLastError());
    }
    void residual(const CsrDev& A, const double* b, const double* x, double* r) override {
        SANM_LAUNCH(residual_dd_kernel, dim3(nblk((size_t)A.n * SPMV_LANES, 256)), dim3(256), 0, m_stream, A, b, x,
                           r);
        HIP_CHECK(hipGetLastError());
    }
    void spmv(const CsrDev& A, const double* x, double* y) override {
        SANM_LAUNCH(spmv_kernel, dim3(nblk((size_t)A.n * SPMV_LANES, 256)), dim3(256), 0,
                           m_stream, A, x, y);
        HIP_CHECK(hipGetLastError());
    }


    // Backend::pcg with device-resident scalars (see the kernels above)
    void pcg(const CsrDev& A, double sign, const double* dinv, const double* b, double* x,
             double rtol, int maxit, int* iters, double* relres) override {
        const size_t n = A.n;
        if (m_pcg_n != n) {
            for (double*& w : m_pcg_w) {
                if (w) (void)hipFree(w);
                HIP_CHECK(hipMalloc(&w, n * sizeof(double)));
            }
            if (!m_pcg_sc) HIP_CHECK(hipMalloc(&m_pcg_sc, sizeof(PcgScalars)));
            if (!m_pcg_sc_host) HIP_CHECK(hipHostMalloc(&m_pcg_sc_host, sizeof(PcgScalars)));
            m_pcg_n = n;
        }
        double *r = m_pcg_w[0], *z = m_pcg_w[1], *p = m_pcg_w[2], *q = m_pcg_w[3];
        HIP_CHECK(hipMemsetAsync(m_pcg_sc, 0, sizeof(PcgScalars), m_stream));
        SANM_LAUNCH(pcg_init_kernel, dim3(red_grid(n)), dim3(256), 0, m_stream, n, sign, b,
                           dinv, x, r, z, p, m_pcg_sc);
        auto fetch = [&]() {
            HIP_CHECK(hipMemcpyAsync(m_pcg_sc_host, m_pcg_sc, sizeof(PcgScalars),
                                     hipMemcpyDeviceToHost, m_stream));
            HIP_CHECK(hipStreamSynchronize(m_stream));
            return *m_pcg_sc_host;
        };
        PcgScalars sc = fetch();
        const double bb = sc.bb;
        int it = 0;
        double rr = bb;
        if (bb > 0) {
            const unsigned g_spmv = nblk(n * SPMV_LANES, 256), g_red = red_grid(n), g_n = nblk(n, 256);
            while (it < maxit) {
                int stop = std::min(maxit, it + PCG_CHECK_EVERY);
                for (; it < stop; ++it) {
                    SANM_LAUNCH(pcg_spmv_dot_kernel, dim3(g_spmv), dim3(256), 0, m_stream, A,
                                       sign, p, q, m_pcg_sc, it);
                    SANM_LAUNCH(pcg_update_kernel, dim3(g_red), dim3(256), 0, m_stream, n,
                                       sign, dinv, p, q, x, r, z, m_pcg_sc, it);
                    SANM_LAUNCH(pcg_dir_kernel, dim3(g_n), dim3(256), 0, m_stream, n, z, p,
                                       m_pcg_sc, it);
                }
                HIP_CHECK(hipGetLastError());
                sc = fetch();
                rr = sc.rr[(it - 1) & 1];
                if (sc.breakdown) {
                    it = -it;
                    break;
                }
                if (!(rr == rr) || std::sqrt(rr) <= rtol * std::sqrt(bb)) break;
            }
        }
        *iters = it;
        *relres = bb > 0 ? std::sqrt(rr / bb) : 0.0;
    }


    // ---- multifrontal LU (mf_kernels.h) ---------------------------------------
#ifdef SANM_SF_PHASES
    void sf_print_phases() {
        unsigned long long ph[8];
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(mfk::g_sf_phase), sizeof(ph)));
        if (!ph[7]) return;
        std::fprintf(stderr, "small_front_kernel, per workgroup, 10 ns ticks: load %.0f  LU %.0f  inverses %.0f  store %.0f  gemm1 %.0f  "
                     "gemm2 %.0f  (%.1f gemm1 tiles per front, %llu fronts)\n", (double)ph[0] / ph[7], (double)ph[1] / ph[7],
                     (double)ph[2] / ph[7], (double)ph[3] / ph[7], (double)ph[4] / ph[7], (double)ph[5] / ph[7],
                     (double)ph[6] / ph[7], ph[7]);
        unsigned long long zero[8] = {};
        HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(mfk::g_sf_phase), zero, sizeof(zero)));
    }
#endif
    int mf_factor(const MfDev& mf, const MfSchedule& sch, const CsrDev& A) override {
        mf_factor_launch(mf, sch, A);
        int32_t* hs = reinterpret_cast<int32_t*>(m_scalar_host);
        HIP_CHECK(hipMemcpyAsync(hs, mf.status, sizeof(int32_t), hipMemcpyDeviceToHost, m_stream));
        HIP_CHECK(hipStreamSynchronize(m_stream));
#ifdef SANM_MF_PHASES
        mf_print_phases();
#endif
        return *hs;
    }
    void mf_factor_async(const MfDev& mf, const MfSchedule& sch, const CsrDev& A, double* status) override {
        run_chain(mf.front_store, 1, A.val, status, [&] {
            mf_factor_launch(mf, sch, A);
            SANM_LAUNCH(status_to_double_kernel, dim3(1), dim3(1), 0, m_stream, mf.status, status);
        });
        HIP_CHECK(hipGetLastError());
#ifdef SANM_SF_PHASES
        sf_print_phases();
#endif
    }
#ifdef SANM_MF_PHASES
    void mf_print_phases() {
        unsigned long long ph[8];
        HIP_CHECK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(mfk::g_phase), sizeof(ph)));
        std::fprintf(stderr, "update_kernel, diagonal workgroup, 10 ns ticks: load %.0f  panel solves %.0f  tile update %.0f  "
                     "tile LU %.0f  store %.0f  (%llu launches)\n", (double)ph[0] / ph[7], (double)ph[1] / ph[7],
                     (double)ph[2] / ph[7], (double)ph[3] / ph[7], (double)ph[4] / ph[7], ph[7]);
    }
#endif
    void mf_factor_launch(const MfDev& mf, const MfSchedule& sch, const CsrDev& A) {
        mf_factor_levels(mf, sch, A, 0, (int)sch.levels.size(), true, true);
    }
    void mf_factor_piece(const MfDev& mf, const MfSchedule& sch, const CsrDev& A, int l0, int l1, bool prologue) override {
        mf_factor_levels(mf, sch, A, l0, l1, prologue, false);
    }
    void mf_factor_status(const MfDev& mf, double* out) override {
        SANM_LAUNCH(status_to_double_kernel, dim3(1), dim3(1), 0, m_stream, mf.status, out);
        HIP_CHECK(hipGetLastError());
    }
    void mf_factor_levels(const MfDev& mf, const MfSchedule& sch, const CsrDev& A, int l0, int l1, bool prologue,
                          bool epilogue) {
        using namespace mfk;
        // Selective zero-fill + assigned F[B,B] blocks (mf_kernels.h: zero_kernel, schur_gather_kernel) from 16 GB of
        // front storage on: there the bytes not written and not read back pay (2.7 M tets: factor -1 %, 26 instead of
        // 60 GB of fill per step); below, the memset of everything and one extend-add launch per round are faster
        // (338 k tets: 13.1 against 13.4 ms per factorisation).  SANM_MF_FULL_ZERO=1 / SANM_MF_SELECTIVE_ZERO=1 force
        // one or the other; same bits either way (tests/test_direct_solver.py).
        const bool selective = !std::getenv("SANM_MF_FULL_ZERO") && sch.n_zero_blocks > 0 &&
                               (std::getenv("SANM_MF_SELECTIVE_ZERO") || std::getenv("SANM_MF_POISON") ||
                                mf.front_store_size >= (int64_t(2) << 30));
        if (prologue) {
        // (read per factorisation: tests switch it) SANM_MF_FULL_ZERO=1: the whole storage of the rank's fronts, as
        // rounds 1-5 did
        if (!selective) {
            if (sch.dist.enabled) {  // (only the fronts this rank factors: MfSchedule::Dist::own_store)
                for (const auto& r : sch.dist.own_store)
                    HIP_CHECK(hipMemsetAsync(mf.front_store + r.first, 0, (size_t)(r.second - r.first) * sizeof(double), m_stream));
            } else {
                HIP_CHECK(hipMemsetAsync(mf.front_store, 0, mf.front_store_size * sizeof(double), m_stream));
            }
        } else {
            // (tests: SANM_MF_POISON=1 fills the whole storage with NaNs first -- whatever the factorisation reads
            // without having written or zeroed it shows up in the factors)
            if (std::getenv("SANM_MF_POISON"))
                HIP_CHECK(hipMemsetAsync(mf.front_store, 0xFF, mf.front_store_size * sizeof(double), m_stream));
            // what is accumulated into or read before it is written, of this rank's fronts (mf_kernels.h, zero_kernel)
            SANM_LAUNCH(zero_kernel, dim3(sch.n_zero_blocks), dim3(256), 0, m_stream, mf.fronts, mf.front_store,
                        sch.zero_blocks);
        }
        HIP_CHECK(hipMemsetAsync(mf.status, 0, sizeof(int32_t), m_stream));
        SANM_LAUNCH(absmax_kernel, dim3(red_grid(mf.nnzA)), dim3(256), 0, m_stream, (size_t)mf.nnzA, A.val,
                           red_to(mf.piv_amax));
        if (!sch.a_dst_ready) {  // (the first factorisation of this solver: where the entries of A go)
            sanm_check(A.rowptr && A.col && A.nnz == mf.nnzA, "multifrontal: the matrix to factor is not the analysed pattern");
            SANM_LAUNCH(scatter_map_kernel, dim3(nblk(mf.n, 4)), dim3(256), 0, m_stream, mf, A.rowptr, A.col);
            sch.a_dst_ready = true;
        }
        SANM_LAUNCH(scatter_kernel, dim3(nblk(mf.nnzA, 256)), dim3(256), 0, m_stream, mf.nnzA,
                           mf.a_dst, A.val, mf.front_store);
        SANM_LAUNCH(aug_identity_kernel, dim3(nblk(mf.n, 256)), dim3(256), 0, m_stream, mf);
        }
        const char* env_min_k = std::getenv("SANM_MF_OUTER_MIN_K");
        const int outer_min_k = env_min_k ? std::atoi(env_min_k) : kOuterMinK;
        for (int li = l0; li < l1; ++li) {
            const auto& L = sch.levels[li];
            for (size_t r = 0; r < L.ea_rounds.size(); ++r) {
                int cnt = L.ea_rounds[r].second - L.ea_rounds[r].first;
                int64_t mb = L.ea_max_b[r];
                if (cnt == 0) continue;
                // round 0: the parents' F[B,B] blocks are ASSIGNED from their first children (every entry, so the block
                // needs no zero-fill), what lands in their pivot rows and columns is added as in the later rounds
                const bool assign0 = r == 0 && selective;
                // (rows of a Schur complement per workgroup: EA_ROWS.  More rows per workgroup on the levels of thousands of
                // small fronts -- fewer, longer workgroups -- measured slower, 13.24 against 13.10 ms per factorisation at
                // 338 k tets, 222.8 against 220.7 at 2.7 M: SANM_MF_EA_ROWS for the experiment)
                static const int env_ea_rows = std::getenv("SANM_MF_EA_ROWS") ? std::atoi(std::getenv("SANM_MF_EA_ROWS")) : 0;
                auto rows_per_wg = [&](int64_t) { return env_ea_rows > 0 ? (env_ea_rows + EA_ROWS - 1) / EA_ROWS * EA_ROWS : EA_ROWS; };
                if (assign0 && L.ea0_max_bp > 0) {
                    const int rw = rows_per_wg(L.ea0_max_bp);
                    SANM_LAUNCH(schur_gather_kernel, dim3((unsigned)((L.ea0_max_bp + rw - 1) / rw), cnt), dim3(256), 0, m_stream,
                                mf.fronts, mf.front_store, sch.ea_inv, sch.ea_children + L.ea_rounds[r].first, rw);
                }
                if (mb > 0) {
                    const int rw = rows_per_wg(mb);
                    SANM_LAUNCH(extend_add_kernel, dim3((unsigned)((mb + rw - 1) / rw), cnt), dim3(256), 0, m_stream, mf.fronts,
                                mf.front_store, mf.rel, sch.ea_children + L.ea_rounds[r].first, (int)assign0, rw);
                }
            }
            const int nfront = L.front_end - L.front_begin;
            // levels of many small fronts: the whole factorisation of a front in one workgroup (mf_kernels.h,
            // small_front_kernel) instead of the panel chain and the two GEMM passes.  SANM_MF_SMALL_MIN_FRONTS: from how
            // many fronts a level takes it (tests: 1; 0: never)
            const char* env_small = std::getenv("SANM_MF_SMALL_MIN_FRONTS");
            const int small_min = env_small ? std::atoi(env_small) : kSmallMinFronts;
            if (small_min > 0 && nfront >= small_min && L.max_k <= SF_KMAX && !L.two_phase) {
                // The pivot block's LDS decides how many fronts a compute unit works on at a time: the level's fronts
                // are sorted by decreasing pivot count (MfSchedule), so they go out in up to three launches by size
                // class -- k <= 96 (77 KB: two workgroups per unit), k <= 64 (34 KB: four), k <= 44 (the GEMM
                // staging's 17 KB; five by the kernel's 88 registers).  (Four classes by exact occupancy -- 96 / 80 /
                // 69 / 62 pivots for 2 / 3 / 4 / 5 workgroups per unit -- measured the same beside each other, 12.9 ms,
                // and worse one after the other, 13.5 against 13.2.)
                constexpr int NCLS = 3;
                const int cls[NCLS] = {SF_KMAX, 64, 44};
                // The classes are independent of each other (different fronts of one level) and a unit that holds two
                // 78 KB workgroups has room for nothing else while four of the small ones leave most of its LDS idle:
                // the later classes go to queues of their own and fill the slots the big class leaves (its last round
                // above all).  SANM_MF_SMALL_SERIAL=1: one queue, one class after the other.
                const bool serial = std::getenv("SANM_MF_SMALL_SERIAL") != nullptr;  // (read per factorisation: tests switch it)
                hipStream_t home = m_stream;
                hipEvent_t fork_ev = nullptr;
                int forked = 0;
                int begin = 0;
                for (int c = 0; c < NCLS && begin < nfront; ++c) {
                    const int kmax = std::min(cls[c], c == 0 ? L.max_k : cls[c]);
                    const int lower = c + 1 < NCLS ? cls[c + 1] : 0;
                    int end = begin;
                    while (end < nfront && L.front_k[end] > lower) ++end;
                    if (end == begin) continue;
                    const int ks = std::min(kmax, (int)L.front_k[begin]) | 1;
                    const size_t lds = std::max<size_t>(((size_t)ks * ks + 4 * (size_t)ks) * sizeof(double),
                                                        (size_t)GK * (2 * GT + 5) * sizeof(double));
                    if (lds > 48 * 1024)
                        HIP_CHECK(hipFuncSetAttribute((const void*)small_front_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
                    hipStream_t q = home;
                    if (!serial && end < nfront && forked < NCLS - 1) {  // more classes follow: this one beside them
                        if (!m_sf_queue[forked]) HIP_CHECK(hipStreamCreateWithFlags(&m_sf_queue[forked], hipStreamNonBlocking));
                        if (!fork_ev) {
                            fork_ev = next_fork_event();
                            HIP_CHECK(hipEventRecord(fork_ev, home));
                        }
                        q = m_sf_queue[forked++];
                        HIP_CHECK(hipStreamWaitEvent(q, fork_ev, 0));
                    }
                    SANM_LAUNCH(small_front_kernel, dim3(end - begin), dim3(256), lds, q,
                                MF_FACTOR_ARGS(mf, L.front_begin + begin), ks, (int)L.fwd_t);
                    begin = end;
                }
                for (int f = 0; f < forked; ++f) {
                    hipEvent_t e = next_fork_event();
                    HIP_CHECK(hipEventRecord(e, m_sf_queue[f]));
                    HIP_CHECK(hipStreamWaitEvent(home, e, 0));
                }
                continue;
            }
            const int nt = (2 * L.max_k + NB - 1) / NB;  // pivot + augmentation block
            // augmentation tiles of a panel: at most ceil(k / NB) + 1 per side
            const int atiles = (L.max_k + NB - 1) / NB + 1;
            if (L.max_k < outer_min_k) {
                // one blocking level.  Panel 0's diagonal tile is factored by diag_kernel; every later
                // diagonal tile by the update kernel of the previous panel (look-ahead)
                if (L.nr_panel > 0)
                    SANM_LAUNCH(diag_kernel, dim3(L.panel_cnt[0]), dim3(256), 0, m_stream,
                                       MF_FACTOR_ARGS(mf, L.front_begin), 0);
                for (int p = 0; p < L.nr_panel; ++p) {
                    const int rem = nt - p - 1;
                    if (rem <= 0) continue;
                    SANM_LAUNCH(update_kernel, dim3(rem, rem, L.panel_cnt[p]), dim3(256), 0, m_stream,
                                       MF_FACTOR_ARGS(mf, L.front_begin), p, nt);
                }
                if (L.nr_panel > 0)
                    SANM_LAUNCH(panel_finalize_kernel,
                                       dim3((atiles + FIN_TILES - 1) / FIN_TILES, 2, nfront * L.nr_panel),
                                       dim3(256), 0, m_stream,
                                       MF_FACTOR_ARGS(mf, L.front_begin), 0, L.nr_panel, -1);
            } else {
                // two blocking levels (mf_kernels.h, update_kernel): outer blocks of kOuterPanels panels
                for (int p0 = 0; p0 < L.nr_panel; p0 += kOuterPanels) {
                    const int p1 = std::min(p0 + kOuterPanels, L.nr_panel), cnt0 = L.panel_cnt[p0];
                    SANM_LAUNCH(diag_kernel, dim3(cnt0), dim3(256), 0, m_stream,
                                       MF_FACTOR_ARGS(mf, L.front_begin), p0);
                    for (int p = p0; p < p1; ++p) {
                        const int w = p1 - p - 1, rem = nt - p - 1, rem2 = nt - p1;
                        if (w <= 0 || rem <= 0) continue;
                        SANM_LAUNCH(update_kernel, dim3(w, rem + std::max(rem2, 0), L.panel_cnt[p]),
                                           dim3(256), 0, m_stream,
                                       MF_FACTOR_ARGS(mf, L.front_begin), p, p1);
                    }
                    const int tiles = nt - p1;  // trailing extent beyond the block
                    // (a smaller front of the level may start its augmentation tiles before p1)
                    SANM_LAUNCH(panel_finalize_kernel,
                                       dim3((std::max(tiles, atiles) + FIN_TILES - 1) / FIN_TILES, 2,
                                            cnt0 * (p1 - p0)),
                                       dim3(256), 0, m_stream,
                                       MF_FACTOR_ARGS(mf, L.front_begin), p0, p1 - p0, p1);
                    if (tiles <= 0) continue;
                    const int gt = (tiles * NB + GT - 1) / GT;
                    SANM_LAUNCH(block_gemm_kernel, dim3(gt, gt, cnt0), dim3(256), 0, m_stream,
                                       MF_FACTOR_ARGS(mf, L.front_begin), p0, p1);
                }
            }
            // Schur complement and the boundary blocks of the solve operators: two GEMM passes
            if (L.max_b > 0) {
                const int nfr = L.front_end - L.front_begin;
                const int tb = (L.max_b + GT - 1) / GT, tk = (L.max_k + GT - 1) / GT;
                const int tmax = std::max(tb, tk);
                // (two-phase levels: the triangular products are the boundary operators, products 1 and 2 left out)
                const int nwhich = L.two_phase ? 1 : 3;
                static const bool no_lists = std::getenv("SANM_MF_NO_TILE_LISTS") != nullptr;  // (A/B: the box grids)
                if (!no_lists && L.g1_tiles) {
                    if (L.n_g1 > 0)
                        SANM_LAUNCH(gemm1_list_kernel, dim3(L.n_g1), dim3(256), 0, m_stream, MF_FACTOR_ARGS(mf, L.front_begin),
                                    L.g1_tiles, (int)L.two_phase);
                    if (L.n_g2 > 0)
                        SANM_LAUNCH(gemm2_list_kernel, dim3(L.n_g2), dim3(256), 0, m_stream, MF_FACTOR_ARGS(mf, L.front_begin),
                                    L.g2_tiles, (int)L.fwd_t);
                } else {
                SANM_LAUNCH(gemm1_kernel, dim3(tmax, tmax, 2 * nfr), dim3(256), 0, m_stream,
                                   MF_FACTOR_ARGS(mf, L.front_begin), (int)L.two_phase);
                SANM_LAUNCH(gemm2_kernel, dim3(tmax, tmax, nwhich * nfr), dim3(256), 0, m_stream,
                                   MF_FACTOR_ARGS(mf, L.front_begin), nwhich, (int)L.fwd_t);
                }
#ifndef SANM_MF_OLD_STAGING
                // big fronts: the interior of the Schur complement in 128 x 64 tiles, a flat list of the tiles that exist
                // in an order that keeps an XCD's workgroups on shared panels (mf_types.h, MF_ST_R)
                if (!no_lists && L.gt_tiles) {
                    if (L.n_gt > 0)
                        SANM_LAUNCH(gemm2_tall_list_kernel, dim3(L.n_gt), dim3(256), 0, m_stream,
                                    MF_FACTOR_ARGS(mf, L.front_begin), L.gt_tiles);
                } else if (L.max_k >= mfk::kTallMinK && L.max_b >= mfk::kTallMinB) {
                    SANM_LAUNCH(gemm2_tall_kernel, dim3(tmax, (tmax + 1) / 2, nfr), dim3(256), 0, m_stream,
                                MF_FACTOR_ARGS(mf, L.front_begin));
                }
#endif
            }
        }
        if (epilogue && sch.top.enabled) {  // the top of the tree as one dense operator (mf_kernels.h)
            const auto& T = sch.top;
            for (int st = 0; st < 2; ++st) {
                const int cnt = T.stage_begin[st + 1] - T.stage_begin[st];
                if (!cnt) continue;
                const int tiles = (T.stage_dim[st] + GT - 1) / GT;
                if (T.indexed)
                    SANM_LAUNCH(top_gemm_kernel<true>, dim3(tiles, tiles, cnt), dim3(256), 0, m_stream,
                                       T.gemms + T.stage_begin[st]);
                else
                    SANM_LAUNCH(top_gemm_kernel<false>, dim3(tiles, tiles, cnt), dim3(256), 0, m_stream,
                                       T.gemms + T.stage_begin[st]);
            }
        }
        HIP_CHECK(hipGetLastError());
    }

    // Level solve kernels are instantiated for R rows per wave and U preloaded 64-column chunks per row
    // with R * U == 16: U covers the level's widest row where it can, R takes what is left.
    // ---- flat block lists of the solve launches (mf_types.h, MfSolveBlock): made on first use from the schedule's host
    //      copy of the level's descriptors, kept until the solver goes (forget_chains)
    struct SolveBlocks {
        const void* owner;  // MfDev::front_store of the solver the level belongs to
        MfSolveBlock* dev;
        int count;
    };
    std::map<std::tuple<const void*, int, int>, SolveBlocks> m_solve_blocks;  // (level, kind, parameter)
    const MfSchedule* m_cur_sch = nullptr;  // the schedule whose levels are being swept (mf_solve_piece / mf_solve_fused)
    // levels of at least this many fronts take the lists (SANM_MF_SOLVE_LISTS: the threshold; 0: never)
    int solve_lists_min() const {  // (read per launch: tests switch it)
        const char* e = std::getenv("SANM_MF_SOLVE_LISTS");
        return e ? std::atoi(e) : 64;
    }
    //! kind 0: `param` rows per workgroup, blocks over the k pivot rows (backward sweep; forward phase 1);
    //! kind 1: transposed forward sweep, `param` pivot rows per workgroup (nzb blocks of the level's largest front come
    //!         first in the numbering), then blocks of 64 boundary rows;
    //! kind 2 / 3: `param` rows per workgroup, blocks over all m rows / over the b boundary rows (forward phases 0 / 2)
    const SolveBlocks* solve_blocks(const MfDev& mf, const MfSchedule::Level& L, int kind, int param, int nzb) {
        if (!m_cur_sch || m_cur_sch->h_lfronts.empty()) return nullptr;
        const auto key = std::make_tuple((const void*)&L, kind, param);
        auto it = m_solve_blocks.find(key);
        if (it != m_solve_blocks.end()) return &it->second;
        if (m_capturing) return nullptr;  // (no allocation or copy while a launch chain is being captured: the box grid)
        std::vector<MfSolveBlock> v;
        for (int32_t i = L.front_begin; i < L.front_end; ++i) {
            const MfFrontDev& f = m_cur_sch->h_lfronts[i];
            const int rows_f = kind == 2 ? f.m : (kind == 3 ? f.m - f.k : f.k);
            const int nb = (rows_f + param - 1) / param;
            for (int b = 0; b < nb; ++b) v.push_back({f, b, 0});
            if (kind == 1)
                for (int b = 0; b < (f.m - f.k + 63) / 64; ++b) v.push_back({f, nzb + b, 0});
        }
        SolveBlocks sb{mf.front_store, nullptr, (int)v.size()};
        if (!v.empty()) {
            sb.dev = static_cast<MfSolveBlock*>(alloc(v.size() * sizeof(MfSolveBlock)));
            h2d(sb.dev, v.data(), v.size() * sizeof(MfSolveBlock));
        }
        return &(m_solve_blocks[key] = sb);
    }
    void forget_solve_blocks(const void* owner) {
        for (auto it = m_solve_blocks.begin(); it != m_solve_blocks.end();)
            if (it->second.owner == owner) {
                if (it->second.dev) free(it->second.dev);
                it = m_solve_blocks.erase(it);
            } else {
                ++it;
            }
    }

    template <int R, int U>
    void launch_level_solve(bool fwd, const MfDev& mf, const MfSchedule::Level& L, int phase) {
        using namespace mfk;
        const int cnt = L.front_end - L.front_begin;
        // rows of the launch and entries of the staged vector, by phase (mf_kernels.h: 0 whole sweep; 1 / 2 the halves
        // of a two-phase level)
        const int rows = fwd ? (phase == 0 ? L.max_m : (phase == 1 ? L.max_k : L.max_b)) : L.max_k;
        const size_t lds = (size_t)(fwd || phase == 2 ? L.max_k : L.max_m) * sizeof(double);
        const void* kern = fwd ? (const void*)fwd_level_kernel<R, U> : (const void*)bwd_level_kernel<R, U>;
        if (lds > 48 * 1024) {
            if (lds > (size_t)kSolveLdsMax)
                sanm_throw(SANM_ERR_UNSUPPORTED,
                           "front of %d rows exceeds the LDS staging of the solve kernels", L.max_m);
            HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, kSolveLdsMax));
        }
        const dim3 grid((rows + 4 * R - 1) / (4 * R), cnt);
        const MfFrontDev* lf = mf.lfronts + L.front_begin;
        if (solve_lists_min() > 0 && cnt >= solve_lists_min()) {
            // the (front, row block) pairs that exist instead of the box grid; the rows of a front in this launch: its
            // pivot rows (backward sweep, forward phase 1), all of them (forward phase 0), its boundary rows (phase 2)
            const int kind = !fwd || phase == 1 ? 0 : (phase == 0 ? 2 : 3);
            if (const SolveBlocks* sb = solve_blocks(mf, L, kind, 4 * R, 0)) {
                const MfFrontDev* bl = reinterpret_cast<const MfFrontDev*>(sb->dev);
                if (sb->count > 0 && fwd)
                    SANM_LAUNCH((fwd_level_kernel<R, U, true>), dim3(sb->count), dim3(256), lds, m_stream, bl, mf.front_store,
                                mf.inbox_store, mf.work, mf.work2, mf.upd_dst, phase);
                else if (sb->count > 0)
                    SANM_LAUNCH((bwd_level_kernel<R, U, true>), dim3(sb->count), dim3(256), lds, m_stream, bl, mf.front_store,
                                mf.work, mf.work2, mf.bnd_idx, phase);
                return;
            }
        }
        if (fwd)
            SANM_LAUNCH((fwd_level_kernel<R, U>), grid, dim3(256), lds, m_stream, lf, mf.front_store,
                               mf.inbox_store, mf.work, mf.work2, mf.upd_dst, phase);
        else
            SANM_LAUNCH((bwd_level_kernel<R, U>), grid, dim3(256), lds, m_stream, lf, mf.front_store, mf.work,
                               mf.work2, mf.bnd_idx, phase);
    }
    void level_solve(bool fwd, const MfDev& mf, const MfSchedule::Level& L) {
        if (!L.two_phase) return level_solve_phase(fwd, mf, L, 0);
        // (Level::two_phase: pivot block and boundary block as two dependent launches)
        level_solve_phase(fwd, mf, L, 1);
        if (fwd ? L.max_b > 0 : true) level_solve_phase(fwd, mf, L, 2);
    }
    void level_solve_phase(bool fwd, const MfDev& mf, const MfSchedule::Level& L, int phase) {
        using namespace mfk;
        if (!fwd && phase == 1 && L.max_b == 0) return;  // (no boundary: nothing to add to z)
        const char* env_lds = std::getenv("SANM_MF_LDS_MAX");  // tests force the large-front path
        const size_t lds_max = env_lds ? (size_t)std::atol(env_lds) : (size_t)kSolveLdsMax;
        // longest row and number of rows of this launch
        const int width = fwd ? L.max_k : (phase == 0 ? L.max_m : (phase == 1 ? L.max_b : L.max_k));
        const int64_t rows = fwd ? (phase == 0 ? L.sum_m : (phase == 1 ? L.sum_k : L.sum_m - L.sum_k)) : L.sum_k;
        // backward sweep over fronts with long rows: the workgroup-per-rows kernel without LDS staging (mf_kernels.h,
        // bwd_wide_kernel).  SANM_MF_WIDE_MIN_M: the row length it starts at (tests force it on small fronts;
        // huge: never); SANM_MF_WIDE_R: rows per workgroup
        const char* env_wide = std::getenv("SANM_MF_WIDE_MIN_M");
        const int wide_min_m = env_wide ? std::atoi(env_wide) : kWideMinM;
        if (!fwd && width >= wide_min_m) {
            const char* env_wr = std::getenv("SANM_MF_WIDE_R");
            const int env_r = env_wr ? std::atoi(env_wr) : 0;
            const int cnt = L.front_end - L.front_begin;
            // 4 rows per workgroup once that still leaves every CU two workgroups (block:48, all wide levels at
            // 1 / 2 / 4 rows: solves 51.2 / 48.0 / 47.7 ms per step; the LDS-staged kernel 53.0)
            const int r = env_r ? env_r : (L.sum_k >= 2048 ? 4 : 2);
            if (r == 1)
                SANM_LAUNCH((bwd_wide_kernel<1>), dim3(L.max_k, cnt), dim3(256), 0, m_stream, mf.lfronts + L.front_begin,
                            mf.front_store, mf.work, mf.work2, mf.bnd_idx, phase);
            else if (r == 2)
                SANM_LAUNCH((bwd_wide_kernel<2>), dim3((L.max_k + 1) / 2, cnt), dim3(256), 0, m_stream,
                            mf.lfronts + L.front_begin, mf.front_store, mf.work, mf.work2, mf.bnd_idx, phase);
            else
                SANM_LAUNCH((bwd_wide_kernel<4>), dim3((L.max_k + 3) / 4, cnt), dim3(256), 0, m_stream,
                            mf.lfronts + L.front_begin, mf.front_store, mf.work, mf.work2, mf.bnd_idx, phase);
            return;
        }
        if (fwd && phase == 0 && L.fwd_t) {
            // boundary operator stored transposed (Level::fwd_t): pivot rows by lane groups, boundary rows by threads
            sanm_check(width <= kFwdTMaxK, "transposed forward operator on a level of %d-pivot fronts", width);
            const int cnt = L.front_end - L.front_begin;
            const size_t lds = ((size_t)L.max_k + 256) * sizeof(double);
            const int g = width <= 32 ? 8 : (width <= 64 ? 16 : (width <= 128 ? 32 : 64));
            const int nbb = (L.max_b + 63) / 64;
#define SANM_FT(G)                                                                                                  \
    if (g == G) {                                                                                                   \
        const int nzb = (L.max_k + 256 / G * 2 - 1) / (256 / G * 2);                                                \
        const SolveBlocks* sb = solve_lists_min() > 0 && cnt >= solve_lists_min() ? solve_blocks(mf, L, 1, 256 / G * 2, nzb) : nullptr; \
        if (sb) {                                                                                                   \
            if (sb->count > 0)                                                                                      \
                SANM_LAUNCH((fwd_level_tr_kernel<G, 2, true>), dim3(sb->count), dim3(256), lds, m_stream,           \
                            reinterpret_cast<const MfFrontDev*>(sb->dev), mf.front_store, mf.inbox_store, mf.work, \
                            mf.work2, mf.upd_dst, nzb);                                                             \
            return;                                                                                                 \
        }                                                                                                           \
        SANM_LAUNCH((fwd_level_tr_kernel<G, 2>), dim3(nzb + nbb, cnt), dim3(256), lds, m_stream,                     \
                    mf.lfronts + L.front_begin, mf.front_store, mf.inbox_store, mf.work, mf.work2, mf.upd_dst, nzb); \
        return;                                                                                                     \
    }
            SANM_FT(8) SANM_FT(16) SANM_FT(32) SANM_FT(64)
#undef SANM_FT
        }
        if ((size_t)(fwd || phase == 2 ? L.max_k : L.max_m) * sizeof(double) > lds_max) {
            // vectors beyond the LDS: plain mat-vec kernels on operands in HBM (bandwidth-bound levels)
            const int cnt = L.front_end - L.front_begin;
            if (fwd) {
                if (phase != 2)
                    SANM_LAUNCH(fwd_prep_kernel, dim3((L.max_k + 255) / 256, cnt), dim3(256), 0, m_stream, mf,
                                       L.front_begin);
                SANM_LAUNCH(fwd_big_kernel, dim3((L.max_m + 3) / 4, cnt), dim3(256), 0, m_stream, mf,
                                   L.front_begin, phase);
            } else {
                SANM_LAUNCH(bwd_big_kernel, dim3((L.max_k + 3) / 4, cnt), dim3(256), 0, m_stream, mf,
                                   L.front_begin, phase);
            }
            return;
        }
        static const bool no_sub = std::getenv("SANM_MF_NO_SUB") != nullptr;
        if (fwd && phase == 0 && width <= 128 && !no_sub) {  // short rows: several rows per wavefront (mf_kernels.h)
            const int cnt = L.front_end - L.front_begin;
            const size_t lds = (size_t)L.max_k * sizeof(double);
            // 2 rows per lane group once the level has enough rows to fill the chip several times over
            // (leaf level of the armadillo mesh: 13.4 / 11.5 / 17.5 us with 1 / 2 / 4 rows)
            static const int env_fs_r = std::getenv("SANM_MF_FS_R") ? std::atoi(std::getenv("SANM_MF_FS_R")) : 0;
            const int rr = env_fs_r ? env_fs_r : (L.sum_m >= 16 * 2048 ? 2 : 1);
            const int g = width <= 32 ? 8 : (width <= 64 ? 16 : 32);
#define SANM_FS(G, R)                                                                                          \
    if (g == G && rr == R) {                                                                                   \
        SANM_LAUNCH((fwd_level_sub_kernel<G, R>), dim3((L.max_m + 256 / G * R - 1) / (256 / G * R), cnt), \
                           dim3(256), lds, m_stream, mf.lfronts + L.front_begin, mf.front_store, mf.inbox_store, \
                           mf.work, mf.work2, mf.upd_dst);                                                     \
        return;                                                                                                \
    }
            SANM_FS(8, 1) SANM_FS(8, 2) SANM_FS(16, 1) SANM_FS(16, 2) SANM_FS(32, 1) SANM_FS(32, 2)
            SANM_FS(8, 4) SANM_FS(16, 4) SANM_FS(32, 4)
#undef SANM_FS
        }
        // U preloaded chunks per row: enough for the level's TYPICAL row, not its longest (the chunks beyond run in the
        // kernels' tail loop, same sums in the same order) -- on a leaf level of 2572 fronts whose rows average 143 entries
        // and peak at 303, U = 8 for the longest row issued 16 loads per lane of which 5 were in range and left R = 2 rows
        // per wavefront.  Alternating runs on one box (profiles/r06_ab_solve.md): solves 10.16 -> 9.63 ms per step at 338 k
        // tets, 106.8 -> 100.3 at 2.7 M, 2.12 -> 2.07 on armadillo_small.  SANM_MF_LS_WIDTH: the factor on the average row
        // (default 1.25; 0: the longest row, as rounds 1-5).
        const char* env_wf = std::getenv("SANM_MF_LS_WIDTH");  // (read per launch: tests switch it)
        const double ls_wf = env_wf ? std::atof(env_wf) : 1.25;
        const int cntf = L.front_end - L.front_begin;
        const double avg_w = fwd ? (double)L.sum_k / cntf : (phase == 0 ? (double)L.sum_m / cntf : (phase == 1 ? (double)(L.sum_m - L.sum_k) / cntf : (double)L.sum_k / cntf));
        const int wsel = ls_wf > 0 ? std::min(width, std::max(64, (int)(ls_wf * avg_w))) : width;
        int u = 1;
        while (u < 16 && 64 * u < wsel) u *= 2;
        // R * U <= 16 row chunks in registers, and enough workgroups to fill the chip (measured: more than
        // 4 rows per wavefront never paid, even on the leaf level)
        int r = rows >= 16 * 2048 ? 4 : (rows >= 8 * 2048 ? 2 : 1);
        static const int env_ls_r = std::getenv("SANM_MF_LS_R") ? std::atoi(std::getenv("SANM_MF_LS_R")) : 0;
        if (env_ls_r && rows >= 8 * 2048) r = env_ls_r;
        while (r * u > 16) r /= 2;
#define SANM_LS(R, U)                                \
    if (r == R && u == U) {                          \
        launch_level_solve<R, U>(fwd, mf, L, phase); \
        return;                                      \
    }
        SANM_LS(4, 1) SANM_LS(4, 2) SANM_LS(4, 4)
        SANM_LS(2, 1) SANM_LS(2, 2) SANM_LS(2, 4) SANM_LS(2, 8)
        SANM_LS(1, 1) SANM_LS(1, 2) SANM_LS(1, 4) SANM_LS(1, 8) SANM_LS(1, 16)
#undef SANM_LS
        sanm_throw(SANM_ERR_ASSERT, "no level solve kernel for R=%d U=%d", r, u);
    }

    void mf_solve(const MfDev& mf, const MfSchedule& sch, const double* b, double* x) override {
        mf_solve_fused(mf, sch, b, x, nullptr, nullptr);
    }
    void mf_solve_piece(const MfDev& mf, const MfSchedule& sch, bool fwd, int l0, int l1) override {
        m_cur_sch = &sch;
        if (fwd)
            for (int li = l0; li < l1; ++li) level_solve(true, mf, sch.levels[li]);
        else
            for (int li = l1 - 1; li >= l0; --li) level_solve(false, mf, sch.levels[li]);
        HIP_CHECK(hipGetLastError());
    }
    void mf_permute(const MfDev& mf, const double* b, double* x) override {
        using namespace mfk;
        if (b) SANM_LAUNCH(permute_in_kernel, dim3(nblk(mf.n, 256)), dim3(256), 0, m_stream, mf.n, mf.perm, b, mf.work);
        if (x) SANM_LAUNCH(permute_out_kernel, dim3(nblk(mf.n, 256)), dim3(256), 0, m_stream, mf.n, mf.perm, mf.work, x);
        HIP_CHECK(hipGetLastError());
    }
    void copy2d_batch(const MfCopy2D* d, int count, int max_rows, int max_cols, const double* src_base,
                      double* dst_base) override {
        if (count <= 0 || max_rows <= 0 || max_cols <= 0) return;
        SANM_LAUNCH(copy2d_kernel, dim3((max_cols + 255) / 256, std::min(max_rows, 1024), count), dim3(256), 0, m_stream, d,
                    src_base, dst_base);
        HIP_CHECK(hipGetLastError());
    }
    void mf_solve_fused(const MfDev& mf, const MfSchedule& sch, const double* b, double* x, const double* dot_y,
                        double* dot_out) override {
        using namespace mfk;
        if (b)
            SANM_LAUNCH(permute_in_kernel, dim3(nblk(mf.n, 256)), dim3(256), 0, m_stream, mf.n,
                               mf.perm, b, mf.work);
        const int nl = (int)sch.levels.size(), below = sch.top.enabled ? nl - 2 : nl;
        m_cur_sch = &sch;
        MfDev mb = mf;
        if (sch.top.enabled) mb.bnd_idx = sch.top.bnd_x;
        const int32_t* perm_out = sch.top.enabled ? sch.top.perm_x : mf.perm;
        run_chain(mf.front_store, 0, mf.work, nullptr, [&] {
        for (int li = 0; li < below; ++li) level_solve(true, mf, sch.levels[li]);
        if (sch.top.enabled) {
            const auto& T = sch.top;
            constexpr int R = 1;
            const dim3 grid((T.n + 4 * R - 1) / (4 * R));
            const size_t lds = (size_t)T.n * sizeof(double);
#define SANM_TS(WW)                                                                                              \
    case WW:                                                                                                     \
        SANM_LAUNCH((top_solve_kernel<R, WW>), grid, dim3(256), lds, m_stream, T.M, T.wsrc, T.ell,         \
                           mf.inbox_store, mf.work, mf.work + mf.n, T.n);                                        \
        break;
            switch (T.W) {
                SANM_TS(2) SANM_TS(4) SANM_TS(6) SANM_TS(8)
                default: sanm_throw(SANM_ERR_ASSERT, "merged top block: lists of %d slots", T.W);
            }
#undef SANM_TS
        }
        // (with the merged block its solution sits behind the n entries of work: redirected lists, mf_types.h)
        for (int li = below - 1; li >= 0; --li) level_solve(false, mb, sch.levels[li]);
        });
        if (dot_y) {
            const unsigned own = red_grid(mf.n);
            if (m_pending.kind == 2 && m_stream == m_main) {  // the update of a Gram-Schmidt step rides along
                const int nvt = nvt_group(m_pending.nvec);
                const GsRider rd = take_rider();
#define SANM_POD(NVT)                                                                                              \
    case NVT:                                                                                                      \
        SANM_LAUNCH(permute_out_dot_kernel<NVT>, dim3(own + rd.nblk), dim3(256), 0, m_stream, mf.n, perm_out, \
                           mf.work, x, dot_y, red_to(dot_out), own, rd);                                           \
        break;
                switch (nvt) {
                    SANM_POD(4) SANM_POD(8) SANM_POD(12) SANM_POD(16) SANM_POD(20) SANM_POD(24)
                    default: sanm_throw(SANM_ERR_ASSERT, "rider: %d vectors", nvt);
                }
#undef SANM_POD
            } else {
                SANM_LAUNCH(permute_out_dot_kernel<0>, dim3(own), dim3(256), 0, m_stream, mf.n, perm_out, mf.work, x,
                                   dot_y, red_to(dot_out), own, GsRider{});
            }
        } else
            SANM_LAUNCH(permute_out_kernel, dim3(nblk(mf.n, 256)), dim3(256), 0, m_stream, mf.n,
                               perm_out, mf.work, x);
        HIP_CHECK(hipGetLastError());
    }

    double time_kernel(int kernel, int reps, const ProgramDev* P, int mode, int order,
                       const CsrDev* A, const double* x, double* y) override {
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        if (kernel == 2 && !m_pcg_sc) HIP_CHECK(hipMalloc(&m_pcg_sc, sizeof(PcgScalars)));
        auto launch = [&]() {
            if (kernel == 0) {
                const bool timing = m_time_passes;
                m_time_passes = false;
                run_pass(*P, mode, order, x);
                m_time_passes = timing;
            } else if (kernel == 1) {
                SANM_LAUNCH(spmv_kernel, dim3(nblk((size_t)A->n * SPMV_LANES, 256)),
                                   dim3(256), 0, m_stream, *A, x, y);
            } else {
                SANM_LAUNCH(pcg_spmv_dot_kernel, dim3(nblk((size_t)A->n * SPMV_LANES, 256)),
                                   dim3(256), 0, m_stream, *A, -1.0, x, y, m_pcg_sc, 0);
            }
        };
        for (int i = 0; i < 3; ++i) launch();  // warm up
        HIP_CHECK(hipEventRecord(e0, m_stream));
        for (int i = 0; i < reps; ++i) launch();
        HIP_CHECK(hipEventRecord(e1, m_stream));
        HIP_CHECK(hipEventSynchronize(e1));
        HIP_CHECK(hipGetLastError());
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return (double)ms / reps;
    }

    // reductions read by the host: see grid_commit
    // (each queue has its own partials and ticket: kernels of the two queues run side by side)
    GridRed& red_for(GridRed& r) {
        if (!r.partials) {
            HIP_CHECK(hipMalloc(&r.partials, MAX_VEC * RED_MAX_GRID * sizeof(double)));
            HIP_CHECK(hipMalloc(&r.ticket, sizeof(unsigned)));
            HIP_CHECK(hipMemset(r.ticket, 0, sizeof(unsigned)));
            HIP_CHECK(hipHostMalloc(&r.host, MAX_VEC * sizeof(double)));
        }
        return r;
    }
    GridRed red() { return red_for(m_stream == m_side && m_side ? m_red_side : m_red); }
    GridRed red_to(double* out) {  // same partials / ticket, result to `out` (device or pinned memory)
        GridRed g = red();
        g.host = out;
        return g;
    }
    const double* red_result() {
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(m_stream));
        return (m_stream == m_side && m_side ? m_red_side : m_red).host;
    }
    double dot(size_t n, const double* x, const double* y) override {
        SANM_LAUNCH(dot_kernel, dim3(red_grid(n)), dim3(256), 0, m_stream, n, x, y, red());
        return red_result()[0];
    }
    void axpby(size_t n, double a, const double* x, double b, const double* y,
               double* out) override {
        SANM_LAUNCH(axpby_kernel, dim3(nblk(n, 256)), dim3(256), 0, m_stream, n, a, x, b, y,
                           out);
        HIP_CHECK(hipGetLastError());
    }
    void axpby_tail(size_t n, double a, const double* x, double b, const double* y, double* out,
                    double tail) override {
        SANM_LAUNCH(axpby_tail_kernel, dim3(nblk(n + 1, 256)), dim3(256), 0, m_stream, n, a, x, b, y,
                           out, tail);
        HIP_CHECK(hipGetLastError());
    }
    void lincomb(size_t n, int nvec, const double* const* ptrs, const double* coefs,
                 double* out) override {
        // more than MAX_VEC vectors: in chunks, `out` itself leading every chunk after the first with coefficient 1
        // (0 + 1 * out is out: the sum is accumulated in the order of a single pass)
        for (int j0 = 0; j0 < nvec || j0 == 0;) {
            VecList v{};
            int m = 0;
            if (j0 > 0) {
                v.p[m] = out;
                v.c[m++] = 1.0;
            }
            for (; m < MAX_VEC && j0 < nvec; ++j0) {
                v.p[m] = ptrs[j0];
                v.c[m++] = coefs[j0];
            }
            v.n = m;
            SANM_LAUNCH(lincomb_kernel, dim3(nblk(n, 256)), dim3(256), 0, m_stream, n, v, out);
            if (nvec == 0) break;
        }
        HIP_CHECK(hipGetLastError());
    }
    void lincomb2_diff_norms(size_t n, int nvec, const double* const* ptrs, const double* c1,
                             const double* c2, double scale, double out_host[2]) override {
        if (nvec > MAX_VEC) sanm_throw(SANM_ERR_ASSERT, "lincomb2_diff_norms: too many vectors");
        VecList v{}, v2{};
        v.n = v2.n = nvec;
        for (int j = 0; j < nvec; ++j) {
            v.p[j] = ptrs[j];
            v.c[j] = c1[j];
            v2.c[j] = c2[j];
        }
        SANM_LAUNCH(lincomb2_diff_norms_kernel, dim3(red_grid(n)), dim3(256), 0, m_stream, n, v, v2,
                           scale, red());
        const double* r = red_result();
        out_host[0] = r[0];
        out_host[1] = r[1];
    }
    void lincomb2_diff_norms_multi(size_t n, int nvec, const double* const* ptrs, int ncand, const double* c1,
                                   const double* c2, const double* scale, double* out_host) override {
        if (nvec > MAX_VEC) {
            // orders beyond 25: both combinations of every candidate through two work vectors (chunked lincomb), then
            // the norms kernel on the pair -- u = 1 * u_buf + 0 * w_buf, w likewise: the values of a single pass
            if (m_probe_work_n < n) {
                if (m_probe_work) HIP_CHECK(hipFree(m_probe_work));
                HIP_CHECK(hipMalloc(&m_probe_work, 2 * n * sizeof(double)));
                m_probe_work_n = n;
            }
            double* u = m_probe_work;
            double* w = m_probe_work + n;
            const double* pair[2] = {u, w};
            const double one_zero[2] = {1.0, 0.0}, zero_one[2] = {0.0, 1.0};
            for (int c = 0; c < ncand; ++c) {
                lincomb(n, nvec, ptrs, c1 + (size_t)c * nvec, u);
                lincomb(n, nvec, ptrs, c2 + (size_t)c * nvec, w);
                lincomb2_diff_norms(n, 2, pair, one_zero, zero_one, scale[c], out_host + 2 * c);
            }
            return;
        }
        if (ncand > PROBE_GROUPS * PROBE_MAX || ncand < 1)
            sanm_throw(SANM_ERR_ASSERT, "lincomb2_diff_norms_multi: %d vectors, %d candidates", nvec, ncand);
        static const bool split = std::getenv("SANM_PROBE_SPLIT") != nullptr;  // (debug: groups as launches of their own)
        if (ncand > PROBE_MAX && split) {
            for (int c0 = 0; c0 < ncand; c0 += PROBE_MAX)
                lincomb2_diff_norms_multi(n, nvec, ptrs, std::min(PROBE_MAX, ncand - c0), c1 + (size_t)c0 * nvec,
                                          c2 + (size_t)c0 * nvec, scale + c0, out_host + 2 * c0);
            return;
        }
        if (ncand > PROBE_MAX) {  // several groups in one launch, coefficients in pinned memory
            if (!m_probe_groups) {
                HIP_CHECK(hipHostMalloc(&m_probe_groups, PROBE_GROUPS * sizeof(ProbeGroup)));
                HIP_CHECK(hipHostMalloc(&m_probe_results, PROBE_GROUPS * 2 * PROBE_MAX * sizeof(double)));
                HIP_CHECK(hipMalloc(&m_probe_partials, (size_t)PROBE_GROUPS * 2 * PROBE_MAX * RED_MAX_GRID * sizeof(double)));
                HIP_CHECK(hipMalloc(&m_probe_tickets, PROBE_GROUPS * sizeof(unsigned)));
                HIP_CHECK(hipMemset(m_probe_tickets, 0, PROBE_GROUPS * sizeof(unsigned)));
            }
            const int ngroup = (ncand + PROBE_MAX - 1) / PROBE_MAX;
            for (int gi = 0; gi < ngroup; ++gi) {
                ProbeGroup& G = m_probe_groups[gi];
                G.ncand = std::min(PROBE_MAX, ncand - gi * PROBE_MAX);
                for (int c = 0; c < PROBE_MAX; ++c) {
                    const int src = std::min(gi * PROBE_MAX + c, ncand - 1);  // (unused slots repeat the last set)
                    G.scale[c] = scale[src];
                    for (int j = 0; j < MAX_VEC; ++j) {
                        G.c1[c][j] = j < nvec ? c1[(size_t)src * nvec + j] : 0.0;
                        G.c2[c][j] = j < nvec ? c2[(size_t)src * nvec + j] : 0.0;
                    }
                }
            }
            ProbePtrs a{};
            a.nvec = nvec;
            for (int j = 0; j < nvec; ++j) a.p[j] = ptrs[j];
            SANM_LAUNCH(lincomb2_diff_norms_grouped_kernel, dim3(red_grid(n), ngroup), dim3(256), 0, m_stream, n, a,
                        m_probe_groups, m_probe_partials, m_probe_tickets, m_probe_results);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipStreamSynchronize(m_stream));
            for (int c = 0; c < ncand; ++c) {
                out_host[2 * c] = m_probe_results[(size_t)(c / PROBE_MAX) * 2 * PROBE_MAX + 2 * (c % PROBE_MAX)];
                out_host[2 * c + 1] = m_probe_results[(size_t)(c / PROBE_MAX) * 2 * PROBE_MAX + 2 * (c % PROBE_MAX) + 1];
            }
            return;
        }
        ProbeArgs a{};
        a.nvec = nvec;
        a.ncand = ncand;
        for (int j = 0; j < nvec; ++j) a.p[j] = ptrs[j];
        for (int c = 0; c < ncand; ++c) {
            a.scale[c] = scale[c];
            for (int j = 0; j < nvec; ++j) {
                a.c1[c][j] = c1[(size_t)c * nvec + j];
                a.c2[c][j] = c2[(size_t)c * nvec + j];
            }
        }
        SANM_LAUNCH(lincomb2_diff_norms_multi_kernel, dim3(red_grid(n)), dim3(256), 0, m_stream, n, a,
                           red());
        const double* r = red_result();
        for (int c = 0; c < 2 * ncand; ++c) out_host[c] = r[c];
    }
    void multi_dot(size_t n, const double* x, int nvec, const double* const* ys,
                   double* out_host) override {
        for (int j0 = 0; j0 < nvec; j0 += MAX_VEC) {
            VecList v{};
            v.n = std::min(MAX_VEC, nvec - j0);
            for (int j = 0; j < v.n; ++j) v.p[j] = ys[j0 + j];
            launch_multi_dot(n, x, v, nullptr, nullptr, 0.0, red());
            const double* r = red_result();
            for (int j = 0; j < v.n; ++j) out_host[j0 + j] = r[j];
        }
    }
    void vmul(size_t n, const double* x, const double* y, double* out) override {
        SANM_LAUNCH(vmul_kernel, dim3(nblk(n, 256)), dim3(256), 0, m_stream, n, x, y, out);
        HIP_CHECK(hipGetLastError());
    }
    void csr_inv_diag(const CsrDev& A, double scale, double* d) override {
        SANM_LAUNCH(inv_diag_kernel, dim3(nblk(A.n, 256)), dim3(256), 0, m_stream, A, scale,
                           d);
        HIP_CHECK(hipGetLastError());
    }
    void count_nonfinite_async(size_t n, const double* x, double* out) override {
        SANM_LAUNCH(nonfinite_kernel, dim3(red_grid(n)), dim3(256), 0, m_stream, n, x, red_to(out));
        HIP_CHECK(hipGetLastError());
    }
    int64_t count_nonfinite(size_t n, const double* x) override {
        SANM_LAUNCH(nonfinite_kernel, dim3(red_grid(n)), dim3(256), 0, m_stream, n, x, red());
        return (int64_t)red_result()[0];
    }
    double allclose_excess(size_t n, const double* a, const double* b, double eps) override {
        SANM_LAUNCH(allclose_kernel, dim3(red_grid(n)), dim3(256), 0, m_stream, n, a, b, eps,
                           red());
        return red_result()[0];
    }
    void sanity_reduce(size_t n, const double* a, const double* b, double eps, size_t n1,
                       const double* x, const double* y, double out[2]) override {
        if (n1 < n) sanm_throw(SANM_ERR_ASSERT, "sanity_reduce: n1 < n");
        SANM_LAUNCH(sanity_kernel, dim3(red_grid(n1)), dim3(256), 0, m_stream, n, a, b, eps, n1, x,
                           y, red());
        const double* r = red_result();
        out[0] = r[0];
        out[1] = r[1];
    }
    void sanity_check(const CsrDev& A, const double* xi, double ti, const double* grad_t, const double* bi,
                      double eps, size_t n1, const double* x1, double*, double*, double out[2]) override {
        if (n1 > (size_t)A.n * SPMV_LANES) sanm_throw(SANM_ERR_ASSERT, "sanity_check: n1 too large");
        SANM_LAUNCH(sanity_check_kernel, dim3(red_grid((size_t)A.n * SPMV_LANES)), dim3(256), 0,
                           m_stream, A, xi, (const double*)nullptr, ti, grad_t, bi, eps, n1, x1, red());
        const double* r = red_result();
        out[0] = r[0];
        out[1] = r[1];
    }
    void sanity_check_async(const CsrDev& A, const double* xi, const double* grad_t, const double* bi,
                            double eps, size_t n1, const double* x1, double*, double*, double* out2) override {
        if (n1 > (size_t)A.n * SPMV_LANES) sanm_throw(SANM_ERR_ASSERT, "sanity_check: n1 too large");
        SANM_LAUNCH(sanity_check_kernel, dim3(red_grid((size_t)A.n * SPMV_LANES)), dim3(256), 0,
                           m_stream, A, xi, xi + A.n, 0.0, grad_t, bi, eps, n1, x1, red_to(out2));
        HIP_CHECK(hipGetLastError());
    }
    void launch_multi_dot(size_t n, const double* x, const VecList& v, const double* last_norm2,
                          const double* last_nn2, double eps, GridRed g) {
        switch ((v.n + 3) / 4) {
            case 1: SANM_LAUNCH(multi_dot_kernel<4>, dim3(red_grid(n)), dim3(256), 0, m_stream, n, x, v, last_norm2, last_nn2, eps, g); break;
            case 2: SANM_LAUNCH(multi_dot_kernel<8>, dim3(red_grid(n)), dim3(256), 0, m_stream, n, x, v, last_norm2, last_nn2, eps, g); break;
            case 3: SANM_LAUNCH(multi_dot_kernel<12>, dim3(red_grid(n)), dim3(256), 0, m_stream, n, x, v, last_norm2, last_nn2, eps, g); break;
            case 4: SANM_LAUNCH(multi_dot_kernel<16>, dim3(red_grid(n)), dim3(256), 0, m_stream, n, x, v, last_norm2, last_nn2, eps, g); break;
            case 5: SANM_LAUNCH(multi_dot_kernel<20>, dim3(red_grid(n)), dim3(256), 0, m_stream, n, x, v, last_norm2, last_nn2, eps, g); break;
            default: SANM_LAUNCH(multi_dot_kernel<24>, dim3(red_grid(n)), dim3(256), 0, m_stream, n, x, v, last_norm2, last_nn2, eps, g); break;
        }
    }
    void sanity_check_batch_async(const CsrDev& A, int nvec, const double* const* xs, const double* grad_t,
                                  const double* const* bs, double eps, size_t n1, const double* x1, double*, double*,
                                  double* out) override {
        if (n1 > (size_t)A.n * SPMV_LANES) sanm_throw(SANM_ERR_ASSERT, "sanity_check: n1 too large");
        if (nvec == 1) {  // one order: the single-vector kernel (the batch kernel gathers for all its slots)
            sanity_check_async(A, xs[0], grad_t, bs[0], eps, n1, x1, nullptr, nullptr, out);
            return;
        }
        for (int q0 = 0; q0 < nvec; q0 += SANITY_ORDERS) {
            SanityBatch a{};
            a.n = std::min(SANITY_ORDERS, nvec - q0);
            for (int q = 0; q < SANITY_ORDERS; ++q) {
                a.x[q] = xs[q0 + (q < a.n ? q : 0)];
                a.b[q] = bs[q0 + (q < a.n ? q : 0)];
            }
            SANM_LAUNCH(sanity_check_multi_kernel, dim3(red_grid((size_t)A.n * SPMV_LANES)), dim3(256), 0,
                               m_stream, A, a, grad_t, eps, n1, x1, red_to(out + 2 * q0));
        }
        HIP_CHECK(hipGetLastError());
    }
    void multi_dot_async(size_t n, const double* x, int nvec, double* const* ys, double* out,
                         const double* last_norm2, const double* last_nn2, double eps) override {
        if (nvec == 0) return;
        // (chunks of MAX_VEC vectors; the second normalisation of the last vector belongs to the chunk that holds it)
        for (int j0 = 0; j0 < nvec; j0 += MAX_VEC) {
            VecList v{};
            v.n = std::min(MAX_VEC, nvec - j0);
            for (int j = 0; j < v.n; ++j) v.p[j] = ys[j0 + j];
            const bool last = j0 + v.n == nvec;
            launch_multi_dot(n, x, v, last ? last_norm2 : nullptr, last ? last_nn2 : nullptr, eps, red_to(out + j0));
        }
        HIP_CHECK(hipGetLastError());
    }
    void gs_update_async(size_t n, const double* x, int nvec, const double* const* qs, const double* coefs,
                         int first, double* out, double* norm2) override {
        if (nvec > MAX_VEC) {
            // chunks of MAX_VEC vectors: the first from x, the others in place on `out` (same order of subtractions as a
            // single pass; every chunk leaves the norm of what it wrote, the last one's is the result's)
            for (int j0 = 0; j0 < nvec; j0 += MAX_VEC)
                gs_update_async(n, j0 == 0 ? x : out, std::min(MAX_VEC, nvec - j0), qs + j0, coefs + j0,
                                std::max(first - j0, 0), out, norm2);
            return;
        }
        VecList v{};
        v.n = nvec;
        for (int j = 0; j < nvec; ++j) v.p[j] = qs[j];
        switch ((nvec + 3) / 4) {
            case 0: case 1: SANM_LAUNCH(gs_update_kernel<4>, dim3(red_grid(n)), dim3(256), 0, m_stream, n, x, v, coefs, first, out, red_to(norm2)); break;
            case 2: SANM_LAUNCH(gs_update_kernel<8>, dim3(red_grid(n)), dim3(256), 0, m_stream, n, x, v, coefs, first, out, red_to(norm2)); break;
            case 3: SANM_LAUNCH(gs_update_kernel<12>, dim3(red_grid(n)), dim3(256), 0, m_stream, n, x, v, coefs, first, out, red_to(norm2)); break;
            case 4: SANM_LAUNCH(gs_update_kernel<16>, dim3(red_grid(n)), dim3(256), 0, m_stream, n, x, v, coefs, first, out, red_to(norm2)); break;
            case 5: SANM_LAUNCH(gs_update_kernel<20>, dim3(red_grid(n)), dim3(256), 0, m_stream, n, x, v, coefs, first, out, red_to(norm2)); break;
            default: SANM_LAUNCH(gs_update_kernel<24>, dim3(red_grid(n)), dim3(256), 0, m_stream, n, x, v, coefs, first, out, red_to(norm2)); break;
        }
        HIP_CHECK(hipGetLastError());
    }
    void scale_rsqrt_async(size_t n, double* v, const double* norm2, double eps, double* nn2) override {
        SANM_LAUNCH(scale_rsqrt_kernel, dim3(red_grid(n)), dim3(256), 0, m_stream, n, v, norm2, eps,
                           red_to(nn2));
        HIP_CHECK(hipGetLastError());
    }
    void gs_renorm_async(size_t n, double* v, const double* norm2, const double* nn2, double eps) override {
        SANM_LAUNCH(renorm_scale_kernel, dim3(nblk(n, 256)), dim3(256), 0, m_stream, n, v, norm2, eps, nn2);
        HIP_CHECK(hipGetLastError());
    }
    bool graph_capture_begin() override {
        flush_deferred();
        red();  // no allocation while capturing
        HIP_CHECK(hipStreamBeginCapture(m_stream, hipStreamCaptureModeThreadLocal));
        m_capturing = true;
        return true;
    }
    void* graph_capture_end() override {
        hipGraph_t graph = nullptr;
        HIP_CHECK(hipStreamEndCapture(m_stream, &graph));
        m_capturing = false;
        hipGraphExec_t exec = nullptr;
        HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        HIP_CHECK(hipGraphDestroy(graph));
        return exec;
    }
    void graph_launch(void* g) override { HIP_CHECK(hipGraphLaunch(static_cast<hipGraphExec_t>(g), m_stream)); }
    void graph_destroy(void* g) override {
        if (g) (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(g));
    }
    void dot_async(size_t n, const double* x, const double* y, double* out) override {
        SANM_LAUNCH(dot_kernel, dim3(red_grid(n)), dim3(256), 0, m_stream, n, x, y, red_to(out));
        HIP_CHECK(hipGetLastError());
    }
    void next_coeff_async(const NextCoeff& nc) override {
        const unsigned own = nblk(nc.n + 1, 256);
        if (m_pending.kind == 3 && m_stream == m_main) {  // the scaling of a Gram-Schmidt step rides along
            const GsRider rd = take_rider();
            SANM_LAUNCH(next_coeff_kernel<true>, dim3(own + rd.nblk), dim3(256), 0, m_stream, nc.n, nc.num, nc.scale, nc.sc,
                        nc.xg, nc.xb, nc.out, nc.t_out, own, rd);
        } else {
            SANM_LAUNCH(next_coeff_kernel<false>, dim3(own), dim3(256), 0, m_stream, nc.n, nc.num, nc.scale, nc.sc, nc.xg,
                        nc.xb, nc.out, nc.t_out, own, GsRider{});
        }
        HIP_CHECK(hipGetLastError());
    }
    bool x1_async(size_t n, const double* xgt2, const double* xg, const double* xb, double* out, double* sc,
                  double* t_host) override {
        SANM_LAUNCH(x1_kernel, dim3(nblk(n + 1, 256)), dim3(256), 0, m_stream, n, xgt2, xg, xb, out, sc, t_host);
        HIP_CHECK(hipGetLastError());
        return true;
    }
    double* alloc_host(size_t n) override {
        double* p = nullptr;
        HIP_CHECK(hipHostMalloc(&p, (n ? n : 1) * sizeof(double)));
        return p;
    }
    void free_host(double* p) override {
        if (p) (void)hipHostFree(p);
    }
    double t0v_excess(size_t n, const double* fx, const double* v, double t0,
                      double tol) override {
        SANM_LAUNCH(t0v_kernel, dim3(red_grid(n)), dim3(256), 0, m_stream, n, fx, v, t0, tol,
                           red());
        return red_result()[0];
    }
};

}  // namespace

Backend* make_backend(int device) { return new HipBackend(device); }

}  // namespace sanm_hip
