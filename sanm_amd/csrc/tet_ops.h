// Per-tet bodies of every graph operator, for all four passes.
//
// One lane handles one tet; every operator is tet-local, so a lane only ever
// re-reads values it wrote itself.  A batch of a few 10^4 tets gives the chip
// less than one wavefront per SIMD, and the order-k bias of the bilinear
// operators is a convolution over k-1 earlier orders whose loads then sit on
// the critical path one round trip after the other.  At higher orders a
// workgroup therefore runs `nparts` wavefronts over the SAME 64 tets: each takes
// a contiguous slice of every convolution, the partial sums meet in LDS
// (conv_reduce) and wavefront 0 alone finishes the operator.  The functions are __host__ __device__ so that the
// GPU-less authoring container can run the very same bodies in a test-only
// host harness (tests/hostsim); the product path is the HIP kernel in
// backend_hip.hip.
//
// Math follows the reference operator by operator (file:line on each block);
// the 3x3 kernels are closed forms instead of Eigen calls.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <cmath>
#endif
#if !defined(__HIPCC_RTC__)  // (run-time compilation has no standard headers; its built-ins cover what is used)
#include <cstdint>
#endif

#include "program.h"

#if defined(__HIPCC_RTC__)
#define SANM_HD __device__ __forceinline__
#define SANM_HD_NOINLINE __device__ inline
#elif defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SANM_HD __host__ __device__ __forceinline__
#define SANM_HD_NOINLINE __host__ __device__ inline
#else
#define SANM_HD inline
#define SANM_HD_NOINLINE inline
#endif

#ifndef SANM_CONV_MAX
#define SANM_CONV_MAX 1
#endif

namespace sanm_hip {

struct TetCtx {
    double* arena;
    const VarDesc* vars;
    int64_t Tpad;
    int64_t tet;
    int32_t order;  // current order k (BIAS / COEFF passes)
    int32_t odim;
    // "current" values of the pass being run (the order-k bias in a BIAS pass, the
    // order-k coefficient in a COEFF pass) are forwarded from operator to operator
    // through a per-lane scratch (LDS on the device) instead of a store->load round
    // trip through HBM: element e of variable v at cur[(vars[v].cur + e) * cur_stride]
    double* cur;
    int64_t cur_stride;
    int32_t out_var;
    // convolution split (device, BIAS pass): this wavefront's part, the number of parts and the
    // LDS exchange buffer of this lane ([part-1][element][64 lanes])
    int32_t part = 0, nparts = 1;
    double* red = nullptr;
    // GRAD pass: the row of the Jacobian d(out)/d(.) this lane propagates.  Rows are independent, so a
    // (tet, row) pair per lane gives 9x the wavefronts, and the gradients of the intermediate variables (one
    // size-vector per variable and row) fit the same LDS slots the forward passes use for current values
    // instead of 9 x size doubles per variable in HBM.
    int32_t grow = 0;
    double* out = nullptr;  // arena + ProgramDev::out_aos
    int32_t max_order = 0;
    // BIAS pass of the kernels compiled per graph: the convolution sums of ALL operators, accumulated by one loop
    // over the history (conv_term below) before the operators run; an operator then takes its sums from
    // conv[OpDesc::conv_off ..] instead of walking the history itself.  false: every operator runs its own loop.
    // (A member array, sized by the generated source through SANM_CONV_MAX, rather than a pointer to a local one:
    // the accumulators must end up in registers, and a pointer kept in the context defeats that.)
    bool has_conv = false;
    double conv[SANM_CONV_MAX];
    // kernels compiled per graph (SANM_SPEC_UNITS): coefficients and bias of the linear combinations, which carry the
    // material constants -- read at run time so that the generated source depends on the graph's structure only
    const double* params = nullptr;
};

// Arena offsets.  The interpreter kernels take them from the operator / variable records as absolute offsets in
// doubles.  The kernels compiled per graph (SANM_SPEC_UNITS, graph.cpp: Program::spec_source) carry them in UNITS OF
// Tpad -- every region of the arena is a whole number of [component][Tpad] planes -- and multiply by the run-time
// Tpad: their source, and with it the compiled code object, then does not depend on the size of the mesh (a scalar
// multiplication per address).
#ifdef SANM_SPEC_UNITS
#define SANM_OFF(c, off) ((int64_t)(off) * (c).Tpad)
#define SANM_LC_PARAM(c, o, k) ((c).params[(o).aux[3] + (k)])
#else
#define SANM_OFF(c, off) ((int64_t)(off))
#define SANM_LC_PARAM(c, o, k) ((o).p[k])
#endif

// slice [lo, hi) of the convolution index range 1 .. order-1 taken by this part
SANM_HD void conv_range(const TetCtx& c, int& lo, int& hi) {
    const int n = c.order - 1;
    lo = 1 + n * c.part / c.nparts;
    hi = 1 + n * (c.part + 1) / c.nparts;
}

// sum the n partial values of all parts into part 0 (no-op when the pass runs unsplit)
SANM_HD void conv_reduce(const TetCtx& c, double* v, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (c.nparts == 1) return;
    __syncthreads();  // part 0 is done reading the previous exchange
    if (c.part)
        for (int e = 0; e < n; ++e) c.red[((c.part - 1) * 9 + e) * 64] = v[e];
    __syncthreads();
    if (!c.part) {
        // (compile-time trip counts: v stays in registers; a workgroup has at most 4 wavefronts)
#pragma unroll
        for (int p = 1; p < 4; ++p)
            if (p < c.nparts)
                for (int e = 0; e < n; ++e) v[e] += c.red[((p - 1) * 9 + e) * 64];
    }
#else
    (void)c; (void)v; (void)n;
#endif
}

// ---------------------------------------------------------------- access --
SANM_HD double* p_coef(const TetCtx& c, int v, int k) {
    const VarDesc& d = c.vars[v];
    return c.arena + SANM_OFF(c, d.coef) + (int64_t)k * d.size * c.Tpad + c.tet;
}
SANM_HD double* p_bias(const TetCtx& c, int v) { return c.arena + SANM_OFF(c, c.vars[v].bias) + c.tet; }
SANM_HD double* p_aux(const TetCtx& c, int64_t off) { return c.arena + SANM_OFF(c, off) + c.tet; }
// value pointer of the "current" term: bias buffer (BIAS pass) or coef[k]
SANM_HD double* p_cur(const TetCtx& c, int v, bool in_coeff) {
    return in_coeff ? p_coef(c, v, c.order) : p_bias(c, v);
}

SANM_HD void ld(const double* p, int64_t s, int n, double* m) {
    for (int i = 0; i < n; ++i) m[i] = p[i * s];
}
SANM_HD void st(double* p, int64_t s, int n, const double* m) {
    for (int i = 0; i < n; ++i) p[i * s] = m[i];
}
SANM_HD void ld9(const double* p, int64_t s, double* m) { ld(p, s, 9, m); }
SANM_HD void st9(double* p, int64_t s, const double* m) { st(p, s, 9, m); }

// ------------------------------------------------------------- 3x3 math --
// c (+)= op(a) * op(b), row-major 3x3  (reference: as_batched_mm,
// libsanm/tensor_linalg.cpp:107-210, static 3x3 path :194)
template <bool TA, bool TB, bool ACC>
SANM_HD void mm3(double* c, const double* a, const double* b) {
    double r[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = ACC ? c[i * 3 + j] : 0.0;
            for (int k = 0; k < 3; ++k) {
                double av = TA ? a[k * 3 + i] : a[i * 3 + k];
                double bv = TB ? b[j * 3 + k] : b[k * 3 + j];
                s = __builtin_fma(av, bv, s);  // (explicit: the library is built with -ffp-contract=off)
            }
            r[i * 3 + j] = s;
        }
    for (int i = 0; i < 9; ++i) c[i] = r[i];
}

// a*b - c*d with one rounding less (what the compiler's contraction made of it before -ffp-contract=off)
SANM_HD double dif2(double a, double b, double c, double d) { return __builtin_fma(a, b, -(c * d)); }
SANM_HD double dot3v(const double* a, const double* b) {
    return __builtin_fma(a[2], b[2], __builtin_fma(a[1], b[1], a[0] * b[0]));
}

SANM_HD double det3(const double* a) {  // tensor_linalg.cpp:319-353
    const double m0 = dif2(a[4], a[8], a[5], a[7]), m1 = dif2(a[3], a[8], a[5], a[6]), m2 = dif2(a[3], a[7], a[4], a[6]);
    return __builtin_fma(a[2], m2, __builtin_fma(-a[1], m1, a[0] * m0));
}

// cofactor matrix by minors.  The reference goes through an SVD with a rank
// test (tensor_linalg.cpp:18-59); for 3x3 the minors are the same matrix up
// to round-off, including rank 2, and are exactly what rank <= 1 collapses to.
SANM_HD void cof3(const double* a, double* c) {
    c[0] = dif2(a[4], a[8], a[5], a[7]);
    c[1] = dif2(a[5], a[6], a[3], a[8]);
    c[2] = dif2(a[3], a[7], a[4], a[6]);
    c[3] = dif2(a[2], a[7], a[1], a[8]);
    c[4] = dif2(a[0], a[8], a[2], a[6]);
    c[5] = dif2(a[1], a[6], a[0], a[7]);
    c[6] = dif2(a[1], a[5], a[2], a[4]);
    c[7] = dif2(a[2], a[3], a[0], a[5]);
    c[8] = dif2(a[0], a[4], a[1], a[3]);
}

SANM_HD void inv3(const double* a, double* r) {  // tensor_linalg.cpp:285-317
    double c[9];
    cof3(a, c);
    double d = dot3v(a, c);
    double id = 1.0 / d;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r[i * 3 + j] = c[j * 3 + i] * id;
}

SANM_HD double clip_div(double x, double y) {  // tensor_svd.cpp:28-31
    return x * y / (y * y + 1e-12);
}

// One-sided Jacobi SVD of a 3x3: a = U diag(s) V', s sorted descending,
// s >= 0.  (Reference: Eigen JacobiSVD, tensor_svd.cpp:66-87.)
SANM_HD_NOINLINE void svd3(const double* a, double* U, double* S, double* V) {
    double B[9];  // working copy, columns get orthogonalised: B = A V
    for (int i = 0; i < 9; ++i) B[i] = a[i];
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; ++i) {
                    alpha += B[i * 3 + p] * B[i * 3 + p];
                    beta += B[i * 3 + q] * B[i * 3 + q];
                    gamma += B[i * 3 + p] * B[i * 3 + q];
                }
                double lim = 1e-32 + 1e-30 * alpha * beta;
                if (gamma * gamma <= lim) continue;
                double rel = fabs(gamma) / sqrt(alpha * beta);
                if (rel > off) off = rel;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < 3; ++i) {
                    double bp = B[i * 3 + p], bq = B[i * 3 + q];
                    B[i * 3 + p] = cs * bp - sn * bq;
                    B[i * 3 + q] = sn * bp + cs * bq;
                    double vp = V[i * 3 + p], vq = V[i * 3 + q];
                    V[i * 3 + p] = cs * vp - sn * vq;
                    V[i * 3 + q] = sn * vp + cs * vq;
                }
            }
        if (off < 1e-15) break;
    }
    double s[3];
    for (int j = 0; j < 3; ++j)
        s[j] = sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
    // sort descending (3 elements)
    int idx[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (s[idx[j]] < s[idx[j + 1]]) {
                int t = idx[j];
                idx[j] = idx[j + 1];
                idx[j + 1] = t;
            }
    double Vs[9];
    for (int j = 0; j < 3; ++j) {
        int k = idx[j];
        S[j] = s[k];
        for (int i = 0; i < 3; ++i) {
            Vs[i * 3 + j] = V[i * 3 + k];
            U[i * 3 + j] = (s[k] > 0) ? B[i * 3 + k] / s[k] : 0.0;
        }
    }
    for (int i = 0; i < 9; ++i) V[i] = Vs[i];
    // complete U for (numerically) zero singular values
    double smax = S[0];
    if (!(S[2] > 1e-300 + 1e-15 * smax)) {
        if (!(S[1] > 1e-300 + 1e-15 * smax)) {
            if (!(S[0] > 0)) {
                U[0] = 1; U[3] = 0; U[6] = 0;
            }
            // pick u1 orthogonal to u0
            double u0[3] = {U[0], U[3], U[6]};
            int m = (fabs(u0[0]) <= fabs(u0[1]) && fabs(u0[0]) <= fabs(u0[2])) ? 0
                    : (fabs(u0[1]) <= fabs(u0[2]) ? 1 : 2);
            double e[3] = {0, 0, 0};
            e[m] = 1;
            double d = u0[m];
            double w[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
            double nw = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
            U[1] = w[0] / nw; U[4] = w[1] / nw; U[7] = w[2] / nw;
        }
        // u2 = u0 x u1
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
}

// Which singular values get negated so that det(W)=+1: literal restatement of
// the selection loop at libsanm/tensor_svd.cpp:88-128 for n = 3 (note the
// reference's `i = j` inside a `for(...; ++i)`).  Returns a 3-bit mask.
SANM_HD int svdw_rotation_fix_mask(const double* ms) {
    const int n = 3;
    const double EPS = 1e-3;
    int best_idx = -1, best_idx_nr = n + 1;
    for (int i = 0; i < n; ++i) {
        int j = i + 1;
        while (j < n && fabs(ms[i] - ms[j]) < EPS) ++j;
        int nr = j - i;
        if (nr <= best_idx_nr || (nr == best_idx_nr + 1 && nr % 2 == 1)) {
            best_idx = i;
            best_idx_nr = nr;
            if (nr == 1) break;
        }
        i = j;
    }
    int mask = 0;
    if (best_idx_nr == 1 || best_idx_nr % 2 == 0) {
        mask = 1 << best_idx;
    } else {
        for (int i = best_idx; i < best_idx + best_idx_nr; ++i) mask |= 1 << i;
    }
    return mask;
}

// M = U S U' W (tensor_svd.cpp:48-145)
SANM_HD_NOINLINE void svdw3(const double* m, bool require_rotation, double* U, double* S,
                            double* W) {
    double V[9];
    svd3(m, U, S, V);
    if (require_rotation) {
        double du = det3(U), dv = det3(V);
        if ((du < 0) != (dv < 0)) {
            int mask = svdw_rotation_fix_mask(S);
            for (int j = 0; j < 3; ++j)
                if (mask & (1 << j)) {
                    S[j] = -S[j];
                    U[j] = -U[j]; U[3 + j] = -U[3 + j]; U[6 + j] = -U[6 + j];
                }
        }
    }
    mm3<false, true, false>(W, U, V);  // W = U V'
}

// order-k terms of the polar decomposition (tensor_svd.cpp:389-475), written
// on the logical (row-major) matrices -- see SURVEY.md appendix A.3.
SANM_HD_NOINLINE void svdw_fwd_p3(const double* Mk, const double* U0, const double* S0,
                                  const double* W0, const double* Bm, const double* Bp,
                                  const double* Bpw, double* Pk, double* Wk) {
    double V0[9], D[9], E[9], Q[9], T1[9], X[9];
    mm3<true, false, false>(V0, W0, U0);  // V0 = W0' U0
    for (int i = 0; i < 9; ++i) D[i] = Bm[i] - Bp[i];
    mm3<true, true, false>(T1, U0, D);    // U0' (Bm-Bp)'
    mm3<false, false, false>(E, T1, U0);  // ... U0
    mm3<true, true, false>(T1, V0, Mk);   // V0' Mk'
    mm3<false, false, false>(Q, T1, U0);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Q[i * 3 + j] *= S0[i];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double e = E[i * 3 + j] + Q[i * 3 + j] + Q[j * 3 + i];
            X[i * 3 + j] = clip_div(e, S0[i] + S0[j]);
        }
    mm3<false, false, false>(T1, U0, X);
    mm3<false, true, false>(D, T1, U0);  // Pk' = U0 X U0'
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Pk[i * 3 + j] = D[j * 3 + i];
    // Wk = U0 diag(1/s) U0' (Mk - Bpw - Pk W0)
    mm3<false, false, false>(T1, Pk, W0);
    for (int i = 0; i < 9; ++i) T1[i] = Mk[i] - Bpw[i] - T1[i];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) D[i * 3 + j] = U0[i * 3 + j] * clip_div(1.0, S0[j]);
    mm3<false, true, false>(X, D, U0);
    mm3<false, false, false>(Wk, X, T1);
}

// order-k terms U_k, S_k, W_k of M = U S U' W (tensor_svd.cpp:275-387), on the logical (row-major) matrices.
//   E = V0' (Mk - Mb)' U0;  X_ij = clip_div(E_ji - E_ij - (V0' Bw V0)_ij s_j, s_i + s_j);  Wk = U0 X V0'
//   E_ij += (Bu_ji - X_ji) s_j;  Sk = diag E;  K_ij = clip_div(E_ij, s_i - s_j) (i < j),
//   K_ji = -Bu_ij - K_ij,  K_jj = -Bu_jj / 2;  Uk = U0 K'
SANM_HD_NOINLINE void svdw_fwd_full3(const double* Mk, const double* Mb, const double* U0, const double* S0,
                                     const double* W0, const double* Bu, const double* Bw, double* Uk, double* Sk,
                                     double* Wk) {
    double V0[9], D[9], T1[9], E[9], R[9], X[9];
    mm3<true, false, false>(V0, W0, U0);
    for (int i = 0; i < 9; ++i) D[i] = Mk[i] - Mb[i];
    mm3<true, true, false>(T1, V0, D);
    mm3<false, false, false>(E, T1, U0);
    mm3<true, false, false>(T1, V0, Bw);
    mm3<false, false, false>(R, T1, V0);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            X[i * 3 + j] = clip_div(E[j * 3 + i] - E[i * 3 + j] - R[i * 3 + j] * S0[j], S0[i] + S0[j]);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) D[i * 3 + j] = E[i * 3 + j] - X[j * 3 + i] * S0[j] + Bu[j * 3 + i] * S0[j];
    mm3<false, false, false>(T1, U0, X);
    mm3<false, true, false>(Wk, T1, V0);
    double K[9];
    for (int j = 0; j < 3; ++j) {
        Sk[j] = D[j * 4];
        for (int i = 0; i < j; ++i) {
            double v = clip_div(D[i * 3 + j], S0[i] - S0[j]);
            K[i * 3 + j] = v;
            K[j * 3 + i] = -Bu[i * 3 + j] - v;
        }
        K[j * 4] = -0.5 * Bu[j * 4];
    }
    mm3<false, true, false>(Uk, U0, K);
}

// ----------------------------------------------------- operator bodies --
// broadcast helper: value c of a var of size sz (scalar -> all elements)
SANM_HD double bval(const double* p, int64_t s, int sz, int c) { return sz == 1 ? p[0] : p[c * s]; }

// ---- access to the current-order values (see TetCtx::cur)
SANM_HD double* p_curv(const TetCtx& c, int v) { return c.cur + (int64_t)c.vars[v].cur * c.cur_stride; }
SANM_HD void ld_cur(const TetCtx& c, int v, int n, double* m) { ld(p_curv(c, v), c.cur_stride, n, m); }
SANM_HD double cur_bval(const TetCtx& c, int v, int sz, int e) {
    return bval(p_curv(c, v), c.cur_stride, sz, e);
}
// the graph output where remap_out gathers it (ProgramDev::out_aos)
SANM_HD void st_out(const TetCtx& c, int n, const double* m) {
    double* p = c.out + c.tet * 9;
    for (int e = 0; e < n; ++e) p[e] = m[e];
}
// publish the current value of v: scratch always; HBM when it is history for later
// orders (COEFF pass) or the graph output read by the remap_out kernel (BIAS pass)
SANM_HD void st_cur(const TetCtx& c, int v, int n, const double* m, bool in_coeff) {
    st(p_curv(c, v), c.cur_stride, n, m);
    if (in_coeff) {
        if (c.vars[v].hist) st(p_coef(c, v, c.order), c.Tpad, n, m);
    }
    else if (v == c.out_var) {
        st_out(c, n, m);
        if (c.vars[v].bias >= 0) st(p_bias(c, v), c.Tpad, n, m);  // kept for the operator-level API only
    }
}

// jac accumulate:  in.jac[r][ci] += v
// gradient slot of variable v for this lane's row (r is that row: c.grow)
SANM_HD void jadd(const TetCtx& c, int v, int, int ci, double val) {
    c.cur[(int64_t)(c.vars[v].cur + ci) * c.cur_stride] += val;
}
SANM_HD void jfma(const TetCtx& c, int v, int, int ci, double a, double b) {  // += a * b, fused
    double& d = c.cur[(int64_t)(c.vars[v].cur + ci) * c.cur_stride];
    d = __builtin_fma(a, b, d);
}
SANM_HD double jget(const TetCtx& c, int v, int, int ci) {
    return c.cur[(int64_t)(c.vars[v].cur + ci) * c.cur_stride];
}

// ---- LINCOMB: elem_arith.cpp:42-124
// (The element loops of the elementwise operators must have compile-time bounds: with a run-time size they stay
// rolled, and every element's load -> use becomes a memory round trip of its own -- 9 per term instead of one.
// Hence the *_t templates on the operand sizes; the program compiler (graph.cpp) admits elementwise operators on
// 3x3 matrices and batched scalars only.)
template <int OSZ>
SANM_HD void op_lincomb_t(const TetCtx& c, const OpDesc& o, int mode) {
    const int64_t s = c.Tpad;
    const int ov = o.out[0], osz = OSZ;
    if (mode == PASS_GRAD) {
        for (int k = 0; k < o.nin; ++k) {
            int iv = o.in[k], isz = c.vars[iv].size;
            if (c.vars[iv].is_const) continue;
            double ck = SANM_LC_PARAM(c, o, k);
            for (int r = c.grow; r <= c.grow; ++r) {
                if (isz == osz) {
                    for (int e = 0; e < osz; ++e) jfma(c, iv, r, e, ck, jget(c, ov, r, e));
                } else {
                    double sum = 0;
                    for (int e = 0; e < osz; ++e) sum += jget(c, ov, r, e);
                    jfma(c, iv, r, 0, ck, sum);
                }
            }
        }
        return;
    }
    double acc[9];
    if (mode == PASS_EVAL0) {
        for (int e = 0; e < osz; ++e) acc[e] = SANM_LC_PARAM(c, o, MAX_OP_IN);
        for (int k = 0; k < o.nin; ++k) {
            int iv = o.in[k], isz = c.vars[iv].size;
            const double* p = p_coef(c, iv, 0);
            for (int e = 0; e < osz; ++e) acc[e] = __builtin_fma(SANM_LC_PARAM(c, o, k), bval(p, s, isz, e), acc[e]);
        }
        st(p_coef(c, ov, 0), s, osz, acc);
        return;
    }
    for (int e = 0; e < osz; ++e) acc[e] = 0.0;
    for (int k = 0; k < o.nin; ++k) {
        int iv = o.in[k], isz = c.vars[iv].size;
        if (c.vars[iv].is_const) continue;
        for (int e = 0; e < osz; ++e) acc[e] = __builtin_fma(SANM_LC_PARAM(c, o, k), cur_bval(c, iv, isz, e), acc[e]);
    }
    st_cur(c, ov, osz, acc, mode == PASS_COEFF);
}

SANM_HD void op_lincomb(const TetCtx& c, const OpDesc& o, int mode) {
    const int osz = c.vars[o.out[0]].size;
    if (osz == 9) op_lincomb_t<9>(c, o, mode);
    else if (osz == 3) op_lincomb_t<3>(c, o, mode);
    else op_lincomb_t<1>(c, o, mode);  // graph.cpp admits sizes 1, 3 and 9 only
}

// ---- MULTIPLY: elem_arith.cpp:128-217   aux0 = self_bias[osz]
template <int ASZ, int BSZ>
SANM_HD void op_multiply_t(const TetCtx& c, const OpDesc& o, int mode) {
    const int64_t s = c.Tpad;
    const int a = o.in[0], b = o.in[1], ov = o.out[0];
    constexpr int asz = ASZ, bsz = BSZ, osz = ASZ > BSZ ? ASZ : BSZ;
    if (mode == PASS_EVAL0) {
        const double *pa = p_coef(c, a, 0), *pb = p_coef(c, b, 0);
        double* po = p_coef(c, ov, 0);
        for (int e = 0; e < osz; ++e) po[e * s] = bval(pa, s, asz, e) * bval(pb, s, bsz, e);
        return;
    }
    if (mode == PASS_GRAD) {
        for (int k = 0; k < 2; ++k) {
            int iv = k ? b : a, other = k ? a : b;
            if (c.vars[iv].is_const) continue;
            int isz = c.vars[iv].size, otsz = c.vars[other].size;
            const double* po = p_coef(c, other, 0);
            for (int r = c.grow; r <= c.grow; ++r) {
                if (isz == osz) {
                    for (int e = 0; e < osz; ++e)
                        jfma(c, iv, r, e, jget(c, ov, r, e), bval(po, s, otsz, e));
                } else {
                    double sum = 0;
                    for (int e = 0; e < osz; ++e) sum = __builtin_fma(jget(c, ov, r, e), bval(po, s, otsz, e), sum);
                    jadd(c, iv, r, 0, sum);
                }
            }
        }
        return;
    }
    const bool in_coeff = mode == PASS_COEFF;
    double sb[9];
    double* psb = p_aux(c, o.aux[0]);
    if (!in_coeff) {
        double sb2[9];
        for (int e = 0; e < osz; ++e) sb[e] = sb2[e] = 0;
        if (c.has_conv) {
            for (int e = 0; e < osz; ++e) sb[e] = o.conv_n ? c.conv[o.conv_off + e] : 0.0;
        } else if (!c.vars[a].is_const && !c.vars[b].is_const) {
            int i, hi;
            conv_range(c, i, hi);
            for (; i + 1 < hi; i += 2) {  // two independent terms in flight
                const double *pa = p_coef(c, a, i), *pb = p_coef(c, b, c.order - i);
                const double *pa2 = p_coef(c, a, i + 1), *pb2 = p_coef(c, b, c.order - i - 1);
                for (int e = 0; e < osz; ++e) {
                    sb[e] = __builtin_fma(bval(pa, s, asz, e), bval(pb, s, bsz, e), sb[e]);
                    sb2[e] = __builtin_fma(bval(pa2, s, asz, e), bval(pb2, s, bsz, e), sb2[e]);
                }
            }
            for (; i < hi; ++i) {
                const double *pa = p_coef(c, a, i), *pb = p_coef(c, b, c.order - i);
                for (int e = 0; e < osz; ++e) sb[e] = __builtin_fma(bval(pa, s, asz, e), bval(pb, s, bsz, e), sb[e]);
            }
            for (int e = 0; e < osz; ++e) sb[e] += sb2[e];
            conv_reduce(c, sb, osz);
        }
        if (c.part) return;
        st(psb, s, osz, sb);
    }
    const double *a0 = p_coef(c, a, 0), *b0 = p_coef(c, b, 0);
    double A0[9], B0[9];  // one batch of loads
    for (int e = 0; e < osz; ++e) {
        A0[e] = bval(a0, s, asz, e);
        B0[e] = bval(b0, s, bsz, e);
        if (in_coeff) sb[e] = psb[e * s];
    }
    if (!c.vars[b].is_const)
        for (int e = 0; e < osz; ++e) sb[e] = __builtin_fma(A0[e], cur_bval(c, b, bsz, e), sb[e]);
    if (!c.vars[a].is_const)
        for (int e = 0; e < osz; ++e) sb[e] = __builtin_fma(cur_bval(c, a, asz, e), B0[e], sb[e]);
    st_cur(c, ov, osz, sb, in_coeff);
}

SANM_HD void op_multiply(const TetCtx& c, const OpDesc& o, int mode) {
    const int asz = c.vars[o.in[0]].size, bsz = c.vars[o.in[1]].size;
    if (asz == 9 && bsz == 9) op_multiply_t<9, 9>(c, o, mode);
    else if (asz == 1 && bsz == 9) op_multiply_t<1, 9>(c, o, mode);
    else if (asz == 9 && bsz == 1) op_multiply_t<9, 1>(c, o, mode);
    else if (asz == 3 && bsz == 3) op_multiply_t<3, 3>(c, o, mode);
    else if (asz == 1 && bsz == 3) op_multiply_t<1, 3>(c, o, mode);
    else if (asz == 3 && bsz == 1) op_multiply_t<3, 1>(c, o, mode);
    else op_multiply_t<1, 1>(c, o, mode);
}

// ---- LOG / POW: oprs/analytic_unary.cpp:113-158, analytic_unary.cpp:13-139
//      aux0 = k = f'(x0) [sz], aux1 = self_bias [sz], aux2 = zero flags of a pow with exponent != 2 (two doubles, raise-only)
constexpr int POW_INT_MAX_ORDER = 32;
SANM_HD bool pow_is_zero(double x) { return fabs(x) < 1e-3; }  // analytic_unary.cpp:43
// [a^k] of (x_0 + x_1 a + ... + x_{k-1} a^{k-1})^p for an integer p >= 2 by repeated multiplication of truncated
// series: what prop_taylor_coeff_int (analytic_unary.cpp:46-92) computes by repeated squaring, for the elements
// whose x_0 is a zero the division recurrence cannot start from.  O(p k^2) per element, only ever run for those.
SANM_HD double pow_int_bias(const TetCtx& c, int x, int e, int k, int p) {
    double y[POW_INT_MAX_ORDER + 1], acc[POW_INT_MAX_ORDER + 1], nxt[POW_INT_MAX_ORDER + 1];
    for (int i = 0; i < k; ++i) y[i] = p_coef(c, x, i)[e * c.Tpad];
    y[k] = 0;
    for (int i = 0; i <= k; ++i) acc[i] = y[i];
    for (int m = 2; m <= p; ++m) {
        for (int d = 0; d <= k; ++d) {
            double sum = 0;
            for (int i = 0; i <= d; ++i) sum = __builtin_fma(acc[i], y[d - i], sum);
            nxt[d] = sum;
        }
        for (int d = 0; d <= k; ++d) acc[d] = nxt[d];
    }
    return acc[k];
}
template <int SZ>
SANM_HD void op_unary_t(const TetCtx& c, const OpDesc& o, int mode) {
    const int64_t s = c.Tpad;
    const int x = o.in[0], ov = o.out[0], sz = SZ;
    const bool is_log = o.type == OP_LOG;
    const double pw = o.p[0];
    double* pk = p_aux(c, o.aux[0]);
    double* psb = p_aux(c, o.aux[1]);
    if (mode == PASS_EVAL0) {
        const double* px = p_coef(c, x, 0);
        double* po = p_coef(c, ov, 0);
        for (int e = 0; e < sz; ++e) {
            double v = px[e * s];
            if (is_log) {
                po[e * s] = log(v);
                pk[e * s] = 1.0 / v;
            } else if (pw == 2.0) {
                po[e * s] = v * v;
                pk[e * s] = 2.0 * v;
            } else {
                po[e * s] = pow(v, pw);
                pk[e * s] = pw * pow(v, pw - 1.0);
                if (pow_is_zero(v) && !c.vars[x].is_const) {
                    // analytic_unary.cpp:112-131: an integer exponent continues on the convolution path, anything
                    // else is SANMNumericalError{"0^p when p is not integer"} (reported by the driver)
                    const bool integer = pw > 0.5 && floor(pw) == pw;
                    // The flag words are only ever RAISED by the kernels (the host clears them): word 0 = "0^p with a
                    // non-integer p", word 1 = "integer p beyond POW_INT_MAX_ORDER".  All pow operators of a graph
                    // share them, and every lane that stores a word stores the same value, so no operator and no
                    // tet can hide another one's report.
                    if (!integer) c.arena[SANM_OFF(c, o.aux[2])] = 1.0;
                    else if (c.max_order > POW_INT_MAX_ORDER) c.arena[SANM_OFF(c, o.aux[2]) + 1] = 2.0;
                }
            }
        }
        return;
    }
    if (mode == PASS_GRAD) {
        if (c.vars[x].is_const) return;
        for (int r = c.grow; r <= c.grow; ++r)
            for (int e = 0; e < sz; ++e) jfma(c, x, r, e, jget(c, ov, r, e), pk[e * s]);
        return;
    }
    const bool in_coeff = mode == PASS_COEFF;
    const int k = c.order;
    double sb[9];
    if (!in_coeff) {
        for (int e = 0; e < sz; ++e) sb[e] = 0;
        if (!c.vars[x].is_const) {
            const bool int2 = (!is_log && pw == 2.0);
            if (c.has_conv) {
                for (int e = 0; e < sz; ++e) sb[e] = c.conv[o.conv_off + e];
            } else {
                int lo, hi;
                conv_range(c, lo, hi);
                for (int i = lo; i < hi; ++i) {
                    // log: x[k-i]*f[i]*(-i/k); pow: f[k-i]*x[i]*((i/k)(p+1)-1); pow 2: x[i]*x[k-i]
                    const double* p1 = int2 ? p_coef(c, x, i) : (is_log ? p_coef(c, x, k - i) : p_coef(c, ov, k - i));
                    const double* p2 = int2 ? p_coef(c, x, k - i) : (is_log ? p_coef(c, ov, i) : p_coef(c, x, i));
                    double w = int2 ? 1.0 : (is_log ? -(double)i / (double)k
                                                    : __builtin_fma((double)i / (double)k, pw + 1.0, -1.0));
                    if (int2) {
                        for (int e = 0; e < sz; ++e) sb[e] = __builtin_fma(p1[e * s], p2[e * s], sb[e]);
                    } else {
                        for (int e = 0; e < sz; ++e) sb[e] = __builtin_fma(p1[e * s] * p2[e * s], w, sb[e]);
                    }
                }
                conv_reduce(c, sb, sz);
            }
            if (c.part) return;
            if (!int2) {
                const double* x0 = p_coef(c, x, 0);
                const bool int_pow = !is_log && pw > 2.5 && floor(pw) == pw && k <= POW_INT_MAX_ORDER;
                for (int e = 0; e < sz; ++e) {
                    const double x0e = x0[e * s];
                    if (int_pow && pow_is_zero(x0e)) sb[e] = pow_int_bias(c, x, e, k, (int)pw);
                    else sb[e] /= x0e;
                }
            }
        }
        if (c.part) return;
        st(psb, s, sz, sb);
    }
    double K[9];
    for (int e = 0; e < sz; ++e) {  // one batch of loads
        K[e] = pk[e * s];
        if (in_coeff) sb[e] = psb[e * s];
    }
    if (!c.vars[x].is_const)
        for (int e = 0; e < sz; ++e) sb[e] = __builtin_fma(K[e], cur_bval(c, x, sz, e), sb[e]);
    st_cur(c, ov, sz, sb, in_coeff);
}

SANM_HD void op_unary(const TetCtx& c, const OpDesc& o, int mode) {
    const int sz = c.vars[o.out[0]].size;
    if (sz == 1) op_unary_t<1>(c, o, mode);
    else if (sz == 3) op_unary_t<3>(c, o, mode);
    else op_unary_t<9>(c, o, mode);
}

// ---- REDUCE_SUM axis=-1: oprs/reduce.cpp:11-102
template <int ISZ>
SANM_HD void op_reduce_t(const TetCtx& c, const OpDesc& o, int mode) {
    const int64_t s = c.Tpad;
    const int x = o.in[0], ov = o.out[0], isz = ISZ;
    if (mode == PASS_GRAD) {
        if (c.vars[x].is_const) return;
        for (int r = c.grow; r <= c.grow; ++r) {
            double g = jget(c, ov, r, 0);
            for (int e = 0; e < isz; ++e) jadd(c, x, r, e, g);
        }
        return;
    }
    double sum = 0;
    if (mode == PASS_EVAL0) {
        const double* p = p_coef(c, x, 0);
        for (int e = 0; e < isz; ++e) sum += p[e * s];
        *p_coef(c, ov, 0) = sum;
        return;
    }
    if (!c.vars[x].is_const)
        for (int e = 0; e < isz; ++e) sum += cur_bval(c, x, isz, e);
    st_cur(c, ov, 1, &sum, mode == PASS_COEFF);
}

SANM_HD void op_reduce(const TetCtx& c, const OpDesc& o, int mode) {
    if (c.vars[o.in[0]].size == 9) op_reduce_t<9>(c, o, mode);
    else if (c.vars[o.in[0]].size == 3) op_reduce_t<3>(c, o, mode);
    else op_reduce_t<1>(c, o, mode);
}

// ---- MATMUL: oprs/linalg.cpp:339-418   aux0 = self_bias[9]
SANM_HD void op_matmul(const TetCtx& c, const OpDesc& o, int mode) {
    const int64_t s = c.Tpad;
    const int a = o.in[0], b = o.in[1], ov = o.out[0];
    double A[9], B[9], R[9];
    if (mode == PASS_EVAL0) {
        ld9(p_coef(c, a, 0), s, A);
        ld9(p_coef(c, b, 0), s, B);
        mm3<false, false, false>(R, A, B);
        st9(p_coef(c, ov, 0), s, R);
        return;
    }
    if (mode == PASS_GRAD) {
        ld9(p_coef(c, a, 0), s, A);
        ld9(p_coef(c, b, 0), s, B);
        for (int r = c.grow; r <= c.grow; ++r) {
            double G[9];
            for (int e = 0; e < 9; ++e) G[e] = jget(c, ov, r, e);
            if (!c.vars[a].is_const) {  // ga[m,k] = sum_n g[m,n] b[k,n]
                mm3<false, true, false>(R, G, B);
                for (int e = 0; e < 9; ++e) jadd(c, a, r, e, R[e]);
            }
            if (!c.vars[b].is_const) {  // gb[k,n] = sum_m g[m,n] a[m,k]
                mm3<true, false, false>(R, A, G);
                for (int e = 0; e < 9; ++e) jadd(c, b, r, e, R[e]);
            }
        }
        return;
    }
    const bool in_coeff = mode == PASS_COEFF;
    double* psb = p_aux(c, o.aux[0]);
    if (!in_coeff) {
        double R2[9], A2[9], B2[9];
        for (int e = 0; e < 9; ++e) R[e] = R2[e] = 0;
        if (c.has_conv) {
            for (int e = 0; e < 9; ++e) R[e] = o.conv_n ? c.conv[o.conv_off + e] : 0.0;
        } else if (!c.vars[a].is_const && !c.vars[b].is_const) {
            int i, hi;
            conv_range(c, i, hi);
            for (; i + 1 < hi; i += 2) {
                ld9(p_coef(c, a, i), s, A);
                ld9(p_coef(c, b, c.order - i), s, B);
                ld9(p_coef(c, a, i + 1), s, A2);
                ld9(p_coef(c, b, c.order - i - 1), s, B2);
                mm3<false, false, true>(R, A, B);
                mm3<false, false, true>(R2, A2, B2);
            }
            for (; i < hi; ++i) {
                ld9(p_coef(c, a, i), s, A);
                ld9(p_coef(c, b, c.order - i), s, B);
                mm3<false, false, true>(R, A, B);
            }
            for (int e = 0; e < 9; ++e) R[e] += R2[e];
            conv_reduce(c, R, 9);
        }
        if (c.part) return;
        st9(psb, s, R);
    }
    double A0[9], B0[9];  // one batch of loads
    ld9(p_coef(c, a, 0), s, A0);
    ld9(p_coef(c, b, 0), s, B0);
    if (in_coeff) ld9(psb, s, R);
    if (!c.vars[a].is_const) {
        ld_cur(c, a, 9, A);
        mm3<false, false, true>(R, A, B0);
    }
    if (!c.vars[b].is_const) {
        ld_cur(c, b, 9, B);
        mm3<false, false, true>(R, A0, B);
    }
    st_cur(c, ov, 9, R, in_coeff);
}

// ---- MATINVMUL: oprs/linalg.cpp:67-217   aux0 = xinv[9], aux1 = self_bias[9]
SANM_HD void op_matinvmul(const TetCtx& c, const OpDesc& o, int mode) {
    const int64_t s = c.Tpad;
    const bool is_left = o.flags & OP_FLAG_IS_LEFT, ident = o.flags & OP_FLAG_USE_IDENTITY;
    const int x = o.in[0], av = ident ? -1 : o.in[1], ov = o.out[0];
    double X[9], Y[9], R[9], Tm[9];
    double* pxinv = p_aux(c, o.aux[0]);
    double* psb = p_aux(c, o.aux[1]);
    if (mode == PASS_EVAL0) {
        ld9(p_coef(c, x, 0), s, X);
        inv3(X, Y);
        st9(pxinv, s, Y);
        if (!ident) {
            ld9(p_coef(c, av, 0), s, X);
            if (is_left) mm3<false, false, false>(R, X, Y);
            else mm3<false, false, false>(R, Y, X);
            st9(p_coef(c, ov, 0), s, R);
        } else {
            st9(p_coef(c, ov, 0), s, Y);
        }
        return;
    }
    if (mode == PASS_GRAD) {
        double XI[9], Y0[9];
        ld9(pxinv, s, XI);
        ld9(p_coef(c, ov, 0), s, Y0);
        for (int e = 0; e < 9; ++e) Y0[e] = -Y0[e];
        const double *m0 = is_left ? Y0 : XI, *m1 = is_left ? XI : Y0;
        for (int r = c.grow; r <= c.grow; ++r) {
            double G[9];
            for (int e = 0; e < 9; ++e) G[e] = jget(c, ov, r, e);
            if (!c.vars[x].is_const) {
                // gx[i,j] = sum_pq g[p,q] m0[p,i] m1[j,q] = (m0' G m1')[i,j]
                mm3<true, false, false>(Tm, m0, G);
                mm3<false, true, false>(R, Tm, m1);
                for (int e = 0; e < 9; ++e) jadd(c, x, r, e, R[e]);
            }
            if (!ident && !c.vars[av].is_const) {
                if (is_left) mm3<false, true, false>(R, G, XI);  // ga[i,j] = sum_q g[i,q] xinv[j,q]
                else mm3<true, false, false>(R, XI, G);           // ga[i,j] = sum_p g[p,j] xinv[p,i]
                for (int e = 0; e < 9; ++e) jadd(c, av, r, e, R[e]);
            }
        }
        return;
    }
    const bool in_coeff = mode == PASS_COEFF;
    if (!in_coeff) {
        double R2[9], X2[9], Y2[9];
        for (int e = 0; e < 9; ++e) R[e] = R2[e] = 0;
        if (c.has_conv) {
            for (int e = 0; e < 9; ++e) R[e] = o.conv_n ? c.conv[o.conv_off + e] : 0.0;
        } else if (!c.vars[x].is_const) {
            // sum_{i=1}^{k-1} Y_i X_{k-i} (left) or X_i Y_{k-i}; two terms in flight
            int i, hi;
            conv_range(c, i, hi);
            for (; i + 1 < hi; i += 2) {
                if (is_left) {
                    ld9(p_coef(c, ov, i), s, Y);
                    ld9(p_coef(c, x, c.order - i), s, X);
                    ld9(p_coef(c, ov, i + 1), s, Y2);
                    ld9(p_coef(c, x, c.order - i - 1), s, X2);
                    mm3<false, false, true>(R, Y, X);
                    mm3<false, false, true>(R2, Y2, X2);
                } else {
                    ld9(p_coef(c, x, i), s, X);
                    ld9(p_coef(c, ov, c.order - i), s, Y);
                    ld9(p_coef(c, x, i + 1), s, X2);
                    ld9(p_coef(c, ov, c.order - i - 1), s, Y2);
                    mm3<false, false, true>(R, X, Y);
                    mm3<false, false, true>(R2, X2, Y2);
                }
            }
            for (; i < hi; ++i) {
                if (is_left) {
                    ld9(p_coef(c, ov, i), s, Y);
                    ld9(p_coef(c, x, c.order - i), s, X);
                    mm3<false, false, true>(R, Y, X);
                } else {
                    ld9(p_coef(c, x, i), s, X);
                    ld9(p_coef(c, ov, c.order - i), s, Y);
                    mm3<false, false, true>(R, X, Y);
                }
            }
        }
        for (int e = 0; e < 9; ++e) R[e] = -(R[e] + R2[e]);
        if (!c.has_conv && !c.vars[x].is_const) conv_reduce(c, R, 9);
        if (c.part) return;
        st9(psb, s, R);
    }
    // one batch of loads (a memory round trip for the operator, not one per conditional block below)
    double XI[9];
    ld9(p_coef(c, ov, 0), s, Y);
    ld9(pxinv, s, XI);
    if (in_coeff) ld9(psb, s, R);
    if (!ident && !c.vars[av].is_const) {
        ld_cur(c, av, 9, X);
        for (int e = 0; e < 9; ++e) R[e] += X[e];
    }
    if (!c.vars[x].is_const) {
        ld_cur(c, x, 9, X);
        if (is_left) mm3<false, false, false>(Tm, Y, X);
        else mm3<false, false, false>(Tm, X, Y);
        for (int e = 0; e < 9; ++e) R[e] -= Tm[e];
    }
    if (is_left) mm3<false, false, false>(Tm, R, XI);
    else mm3<false, false, false>(Tm, XI, R);
    st_cur(c, ov, 9, Tm, in_coeff);
}

// ---- DET: oprs/linalg.cpp:221-282, tensor_polymat.cpp:344-379
//      aux0 = cof(x0)[9], aux1 = self_bias[1],
//      aux2 = c series [(N+1)][3]: c_m = sum_{j+l=m} r1_j x r2_l (rows 1,2 of x),
//      aux3 = partial c_k [3] (terms with j,l <= k-1).
// The order-k coefficient of det(sum_{i<k} x_i a^i) equals
//   sum_{i=1}^{k-1} r0_i . c_{k-i}  +  r0_0 . c_k^partial
// which is the Leibniz expansion of the reference regrouped so that each
// order costs O(k) instead of O(k^2).
SANM_HD void cross3(const double* a, const double* b, double* r) {
    r[0] = dif2(a[1], b[2], a[2], b[1]);
    r[1] = dif2(a[2], b[0], a[0], b[2]);
    r[2] = dif2(a[0], b[1], a[1], b[0]);
}
SANM_HD void op_det(const TetCtx& c, const OpDesc& o, int mode) {
    const int64_t s = c.Tpad;
    const int x = o.in[0], ov = o.out[0];
    double* pcof = p_aux(c, o.aux[0]);
    double* psb = p_aux(c, o.aux[1]);
    double* pcs = p_aux(c, o.aux[2]);
    double* pck = p_aux(c, o.aux[3]);
    double X[9], C[9];
    if (mode == PASS_EVAL0) {
        ld9(p_coef(c, x, 0), s, X);
        cof3(X, C);
        st9(pcof, s, C);
        *p_coef(c, ov, 0) = dot3v(X, C);
        st(pcs, s, 3, C);  // c_0 = r1_0 x r2_0 = first cofactor row
        return;
    }
    if (mode == PASS_GRAD) {
        if (c.vars[x].is_const) return;
        ld9(pcof, s, C);
        for (int r = c.grow; r <= c.grow; ++r) {
            double g = jget(c, ov, r, 0);
            for (int e = 0; e < 9; ++e) jfma(c, x, r, e, g, C[e]);
        }
        return;
    }
    const bool in_coeff = mode == PASS_COEFF;
    const int k = c.order;
    double sb = 0;
    if (c.vars[x].is_const) {
        if (c.part) return;
        st_cur(c, ov, 1, &sb, in_coeff);
        return;
    }
    if (!in_coeff) {
        int lo, hi;
        conv_range(c, lo, hi);
        double ck[4] = {0, 0, 0, 0}, t[3];  // ck[3]: this part's share of sum_i r0_i . c_{k-i}
        if (c.has_conv) {
            for (int e = 0; e < 4; ++e) ck[e] = c.conv[o.conv_off + e];
            lo = hi = 0;
        }
        for (int j = lo; j < hi; ++j) {  // both sums in one sweep: their loads share a memory round trip
            double r0[3], r1[3], r2[3], cm[3];
            ld(p_coef(c, x, j), s, 3, r0);
            ld(p_coef(c, x, j) + 3 * s, s, 3, r1);
            ld(p_coef(c, x, k - j) + 6 * s, s, 3, r2);
            ld(pcs + (int64_t)(k - j) * 3 * s, s, 3, cm);
            cross3(r1, r2, t);
            ck[0] += t[0]; ck[1] += t[1]; ck[2] += t[2];
            ck[3] += __builtin_fma(r0[2], cm[2], __builtin_fma(r0[1], cm[1], r0[0] * cm[0]));
        }
        double x0[3];
        ld(p_coef(c, x, 0), s, 3, x0);
        if (!c.has_conv) conv_reduce(c, ck, 4);
        if (c.part) return;
        st(pck, s, 3, ck);
        sb = __builtin_fma(x0[2], ck[2], __builtin_fma(x0[1], ck[1], x0[0] * ck[0])) + ck[3];
        *psb = sb;
    }
    ld9(pcof, s, C);  // one batch of loads
    double ck[3] = {0, 0, 0}, r1[3] = {0, 0, 0}, r2[3] = {0, 0, 0};
    if (in_coeff) {
        sb = *psb;
        ld(pck, s, 3, ck);
        ld(p_coef(c, x, 0) + 3 * s, s, 3, r1);
        ld(p_coef(c, x, 0) + 6 * s, s, 3, r2);
    }
    ld_cur(c, x, 9, X);
    if (in_coeff) {
        // finish c_k now that x_k is known
        double t[3];
        cross3(r1, X + 6, t);
        ck[0] += t[0]; ck[1] += t[1]; ck[2] += t[2];
        cross3(X + 3, r2, t);
        ck[0] += t[0]; ck[1] += t[1]; ck[2] += t[2];
        st(pcs + (int64_t)k * 3 * s, s, 3, ck);
    }
    double d = sb;
    for (int e = 0; e < 9; ++e) d = __builtin_fma(C[e], X[e], d);
    st_cur(c, ov, 1, &d, in_coeff);
}

// ---- TRANSPOSE: oprs/linalg.cpp:286-335
SANM_HD void op_transpose(const TetCtx& c, const OpDesc& o, int mode) {
    const int64_t s = c.Tpad;
    const int x = o.in[0], ov = o.out[0];
    if (mode == PASS_GRAD) {
        if (c.vars[x].is_const) return;
        for (int r = c.grow; r <= c.grow; ++r)
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) jadd(c, x, r, i * 3 + j, jget(c, ov, r, j * 3 + i));
        return;
    }
    double X[9], Y[9];
    if (mode == PASS_EVAL0) ld9(p_coef(c, x, 0), s, X);
    else ld_cur(c, x, 9, X);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Y[i * 3 + j] = X[j * 3 + i];
    if (mode == PASS_EVAL0) st9(p_coef(c, ov, 0), s, Y);
    else st_cur(c, ov, 9, Y, mode == PASS_COEFF);
}

// ---- MULEYE: oprs/linalg.cpp:422-479
SANM_HD void op_muleye(const TetCtx& c, const OpDesc& o, int mode) {
    const int64_t s = c.Tpad;
    const int x = o.in[0], ov = o.out[0];
    if (mode == PASS_GRAD) {
        if (c.vars[x].is_const) return;
        for (int r = c.grow; r <= c.grow; ++r)
            jadd(c, x, r, 0, jget(c, ov, r, 0) + jget(c, ov, r, 4) + jget(c, ov, r, 8));
        return;
    }
    double v = (mode == PASS_EVAL0) ? *p_coef(c, x, 0) : *p_curv(c, x);
    double Y[9];
    for (int e = 0; e < 9; ++e) Y[e] = (e % 4 == 0) ? v : 0.0;
    if (mode == PASS_EVAL0) st9(p_coef(c, ov, 0), s, Y);
    else st_cur(c, ov, 9, Y, mode == PASS_COEFF);
}

// ---- SVDW: oprs/linalg.cpp:483-615, tensor_svd.cpp.  out = {U, S, W}
//      pw_mode (U and S unread, the ARAP graph): aux0 = P series [(N+1)][9] (P_0 unused), aux1 = Bm[9],
//      aux2 = Bp[9], aux3 = Bpw[9].
//      full mode (OP_FLAG_SVDW_FULL): aux0 = T0 series, T0_i = sum_j U_j diag(S_{i-j}); aux1 = T1 series,
//      T1_i = sum_j T0_j U_{i-j}' (the reference rebuilds both from scratch at every order, linalg.cpp:42-62 and
//      :570-590; they are kept here, one order added per COEFF pass); aux2 = {Bu, Bw, Mbias_k, t0k, t1k} with
//      t0k / t1k the order-k terms of T0 / T1 without U_k, S_k.
SANM_HD void op_svdw_full(const TetCtx& c, const OpDesc& o, int mode) {
    const int64_t s = c.Tpad;
    const int x = o.in[0], uv = o.out[0], sv = o.out[1], wv = o.out[2];
    double U[9], S[3], W[9];
    double* pT0 = p_aux(c, o.aux[0]);
    double* pT1 = p_aux(c, o.aux[1]);
    double* pB = p_aux(c, o.aux[2]);
    ld9(p_coef(c, uv, 0), s, U);
    ld(p_coef(c, sv, 0), s, 3, S);
    ld9(p_coef(c, wv, 0), s, W);
    if (mode == PASS_GRAD) {
        // dU/dM, dS/dM, dW/dM chained with the upstream Jacobians (tensor_svd.cpp:147-273): with V = W' U,
        //   gM_r += U Z V',  Z = Z_W + diag(gS_r) + Z_U,
        //   Z_W[i,j] = clip_div(G_ij - G_ji, s_i + s_j),            G = U' gW_r V
        //   Z_U[i,j] = clip_div((H_ij - H_ji) s_j, s_j^2 - s_i^2),  H = U' gU_r          (i != j)
        if (c.vars[x].is_const) return;
        double V[9], G[9], Tm[9], Z[9];
        mm3<true, false, false>(V, W, U);
        for (int e = 0; e < 9; ++e) Z[e] = 0;
        if (o.flags & OP_FLAG_SVDW_GW) {
            for (int e = 0; e < 9; ++e) G[e] = jget(c, wv, c.grow, e);
            mm3<true, false, false>(Tm, U, G);
            mm3<false, false, false>(G, Tm, V);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    if (i != j) Z[i * 3 + j] += clip_div(G[i * 3 + j] - G[j * 3 + i], S[i] + S[j]);
        }
        if (o.flags & OP_FLAG_SVDW_GS)
            for (int i = 0; i < 3; ++i) Z[i * 4] += jget(c, sv, c.grow, i);
        if (o.flags & OP_FLAG_SVDW_GU) {
            for (int e = 0; e < 9; ++e) Tm[e] = jget(c, uv, c.grow, e);
            mm3<true, false, false>(G, U, Tm);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    if (i != j)
                        Z[i * 3 + j] += clip_div((G[i * 3 + j] - G[j * 3 + i]) * S[j], S[j] * S[j] - S[i] * S[i]);
        }
        mm3<false, false, false>(Tm, U, Z);
        mm3<false, true, false>(G, Tm, V);
        for (int e = 0; e < 9; ++e) jadd(c, x, c.grow, e, G[e]);
        return;
    }
    const bool in_coeff = mode == PASS_COEFF;
    const int k = c.order;
    double Bu[9], Bw[9], Mb[9], t0k[9], t1k[9], A[9], B[9], D[3];
    if (!in_coeff) {
        for (int e = 0; e < 9; ++e) Bu[e] = Bw[e] = Mb[e] = t0k[e] = t1k[e] = 0;
        int lo, hi;
        conv_range(c, lo, hi);
        if (c.has_conv) {
            const double* q = c.conv + o.conv_off;
            for (int e = 0; e < 9; ++e) {
                Bu[e] = q[e];
                Bw[e] = q[9 + e];
                Mb[e] = q[18 + e];
                t0k[e] = q[27 + e];
                t1k[e] = q[36 + e];
            }
            lo = hi = 0;
        }
        for (int i = lo; i < hi; ++i) {
            ld9(p_coef(c, uv, i), s, A);
            ld9(p_coef(c, uv, k - i), s, B);
            mm3<true, false, true>(Bu, A, B);  // U_i' U_{k-i}
            ld(p_coef(c, sv, k - i), s, 3, D);
            for (int r = 0; r < 3; ++r)
                for (int q = 0; q < 3; ++q) t0k[r * 3 + q] = __builtin_fma(A[r * 3 + q], D[q], t0k[r * 3 + q]);  // U_i diag(S_{k-i})
            ld9(pT0 + (int64_t)i * 9 * s, s, A);
            mm3<false, true, true>(t1k, A, B);  // T0_i U_{k-i}'
            ld9(p_coef(c, wv, i), s, A);
            ld9(p_coef(c, wv, k - i), s, B);
            mm3<true, false, true>(Bw, A, B);  // W_i' W_{k-i}
            ld9(pT1 + (int64_t)i * 9 * s, s, A);
            mm3<false, false, true>(Mb, A, B);  // T1_i W_{k-i}
        }
        if (!c.has_conv) {
            conv_reduce(c, Bu, 9);
            conv_reduce(c, Bw, 9);
            conv_reduce(c, t0k, 9);
            conv_reduce(c, t1k, 9);
            conv_reduce(c, Mb, 9);
        }
        if (c.part) return;
        if (k > 1) {  // (order 1: every bias term is zero, linalg.cpp:556-568)
            mm3<false, true, true>(t1k, t0k, U);  // the known part of T0_k times U_0'
            mm3<false, false, true>(Mb, t1k, W);  // the known part of T1_k times W_0
        }
        st9(pB, s, Bu);
        st9(pB + 9 * s, s, Bw);
        st9(pB + 18 * s, s, Mb);
        st9(pB + 27 * s, s, t0k);
        st9(pB + 36 * s, s, t1k);
    } else {
        ld9(pB, s, Bu);
        ld9(pB + 9 * s, s, Bw);
        ld9(pB + 18 * s, s, Mb);
    }
    double M[9], Uk[9], Sk[3], Wk[9];
    ld_cur(c, x, 9, M);
    svdw_fwd_full3(M, Mb, U, S, W, Bu, Bw, Uk, Sk, Wk);
    st_cur(c, uv, 9, Uk, in_coeff);
    st_cur(c, sv, 3, Sk, in_coeff);
    st_cur(c, wv, 9, Wk, in_coeff);
    if (in_coeff && k < c.max_order) {
        // T0_k = t0k + U_0 diag(S_k) + U_k diag(S_0);  T1_k = t1k + (T0_k - t0k) U_0' + T0_0 U_k'
        ld9(pB + 27 * s, s, t0k);
        ld9(pB + 36 * s, s, t1k);
        for (int r = 0; r < 3; ++r)
            for (int q = 0; q < 3; ++q) A[r * 3 + q] = U[r * 3 + q] * Sk[q] + Uk[r * 3 + q] * S[q];
        mm3<false, true, true>(t1k, A, U);
        for (int r = 0; r < 3; ++r)
            for (int q = 0; q < 3; ++q) B[r * 3 + q] = U[r * 3 + q] * S[q];  // T0_0
        mm3<false, true, true>(t1k, B, Uk);
        for (int e = 0; e < 9; ++e) A[e] += t0k[e];
        st9(pT0 + (int64_t)k * 9 * s, s, A);
        st9(pT1 + (int64_t)k * 9 * s, s, t1k);
    }
}

SANM_HD void op_svdw(const TetCtx& c, const OpDesc& o, int mode) {
    const int64_t s = c.Tpad;
    const int x = o.in[0], uv = o.out[0], sv = o.out[1], wv = o.out[2];
    double M[9], U[9], S[3], W[9];
    if (mode == PASS_EVAL0) {
        ld9(p_coef(c, x, 0), s, M);
        svdw3(M, o.flags & OP_FLAG_REQUIRE_ROT, U, S, W);
        st9(p_coef(c, uv, 0), s, U);
        st(p_coef(c, sv, 0), s, 3, S);
        st9(p_coef(c, wv, 0), s, W);
        if (o.flags & OP_FLAG_SVDW_FULL) {
            double T0[9], T1[9];
            for (int r = 0; r < 3; ++r)
                for (int q = 0; q < 3; ++q) T0[r * 3 + q] = U[r * 3 + q] * S[q];
            mm3<false, true, false>(T1, T0, U);
            st9(p_aux(c, o.aux[0]), s, T0);
            st9(p_aux(c, o.aux[1]), s, T1);
        }
        return;
    }
    if (o.flags & OP_FLAG_SVDW_FULL) {
        op_svdw_full(c, o, mode);
        return;
    }
    ld9(p_coef(c, uv, 0), s, U);
    ld(p_coef(c, sv, 0), s, 3, S);
    ld9(p_coef(c, wv, 0), s, W);
    if (mode == PASS_GRAD) {
        // dW/dM chained with the upstream Jacobian (tensor_svd.cpp:147-273):
        //   gM_r += U Z V',  Z_ij = (G_ij - G_ji) d_ij,  G = U' gW_r V,
        //   d_ij = clip_div(1, s_i + s_j), V = W' U.
        if (c.vars[x].is_const) return;
        double V[9], G[9], Tm[9], Z[9];
        mm3<true, false, false>(V, W, U);
        for (int r = c.grow; r <= c.grow; ++r) {
            for (int e = 0; e < 9; ++e) G[e] = jget(c, wv, r, e);
            mm3<true, false, false>(Tm, U, G);
            mm3<false, false, false>(G, Tm, V);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    Z[i * 3 + j] = (i == j) ? 0.0 : clip_div(G[i * 3 + j] - G[j * 3 + i], S[i] + S[j]);
            mm3<false, false, false>(Tm, U, Z);
            mm3<false, true, false>(G, Tm, V);
            for (int e = 0; e < 9; ++e) jadd(c, x, r, e, G[e]);
        }
        return;
    }
    const bool in_coeff = mode == PASS_COEFF;
    const int k = c.order;
    double* pP = p_aux(c, o.aux[0]);
    double Bm[9], Bp[9], Bpw[9], A[9], B[9];
    if (!in_coeff) {
        for (int e = 0; e < 9; ++e) Bm[e] = Bp[e] = Bpw[e] = 0;
        int lo, hi;
        conv_range(c, lo, hi);
        if (c.has_conv) {
            const double* q = c.conv + o.conv_off;
            for (int e = 0; e < 9; ++e) {
                Bm[e] = q[e];
                Bp[e] = q[9 + e];
                Bpw[e] = q[18 + e];
            }
            lo = hi = 0;
        }
        for (int i = lo; i < hi; ++i) {
            ld9(p_coef(c, x, i), s, A);
            ld9(p_coef(c, x, k - i), s, B);
            mm3<false, true, true>(Bm, A, B);  // M_i M_{k-i}'
            ld9(pP + (int64_t)i * 9 * s, s, A);
            ld9(pP + (int64_t)(k - i) * 9 * s, s, B);
            mm3<true, false, true>(Bp, A, B);  // P_i' P_{k-i}
            ld9(p_coef(c, wv, k - i), s, B);
            mm3<false, false, true>(Bpw, A, B);  // P_i W_{k-i}
        }
        if (!c.has_conv) {
            conv_reduce(c, Bm, 9);
            conv_reduce(c, Bp, 9);
            conv_reduce(c, Bpw, 9);
        }
        if (c.part) return;
        st9(p_aux(c, o.aux[1]), s, Bm);
        st9(p_aux(c, o.aux[2]), s, Bp);
        st9(p_aux(c, o.aux[3]), s, Bpw);
    } else {
        ld9(p_aux(c, o.aux[1]), s, Bm);
        ld9(p_aux(c, o.aux[2]), s, Bp);
        ld9(p_aux(c, o.aux[3]), s, Bpw);
    }
    ld_cur(c, x, 9, M);
    double Pk[9], Wk[9];
    svdw_fwd_p3(M, U, S, W, Bm, Bp, Bpw, Pk, Wk);
    st_cur(c, wv, 9, Wk, in_coeff);
    if (in_coeff) st9(pP + (int64_t)k * 9 * s, s, Pk);
}

// ---- fused convolution loop (kernels compiled per graph) ----------------------------------------------------
// Every operator above walks the history for its own sum_{i=1}^{k-1} A_i (x) B_{k-i}; between them the operators of
// a graph read the same series several times (Neo-Hookean: F by the inverse, by the determinant's rows and by its
// cross-product series; F^-1 by the inverse and, transposed, by the product with log J): 42 doubles per term where
// the union is 23.  conv_term(o, i, j) adds operator o's term A_i (x) B_j to its accumulators; the generated pass
// calls it for ALL operators with (i, k-i) and (k-i, i) in one loop body over i < k/2, so that the loads of both
// orientations of every series sit in one basic block and the compiler merges the identical ones.
template <int SZ>
SANM_HD void ldh(const TetCtx& c, int v, int i, double* m) {
    if constexpr (SZ == 9) {
        const int a = c.vars[v].alias;
        if (a >= 0) {  // the transpose of a kept series: read the source
            const double* p = p_coef(c, a, i);
            for (int r = 0; r < 3; ++r)
                for (int q = 0; q < 3; ++q) m[r * 3 + q] = p[(q * 3 + r) * c.Tpad];
            return;
        }
    }
    ld(p_coef(c, v, i), c.Tpad, SZ, m);
}
template <int ASZ, int BSZ>
SANM_HD void conv_term_multiply(const TetCtx& c, const OpDesc& o, int i, int j, double* acc) {
    constexpr int osz = ASZ > BSZ ? ASZ : BSZ;
    double A[ASZ], B[BSZ];
    ldh<ASZ>(c, o.in[0], i, A);
    ldh<BSZ>(c, o.in[1], j, B);
    for (int e = 0; e < osz; ++e) acc[e] = __builtin_fma(A[ASZ == 1 ? 0 : e], B[BSZ == 1 ? 0 : e], acc[e]);
}
template <int SZ>
SANM_HD void conv_term_unary(const TetCtx& c, const OpDesc& o, int i, int j, double* acc) {
    const int x = o.in[0], ov = o.out[0];
    const double k = (double)c.order, pw = o.p[0];
    double P1[SZ], P2[SZ];
    double w = 1.0;
    if (o.type == OP_LOG) {  // x_j f_i (-i/k)
        ldh<SZ>(c, x, j, P1);
        ldh<SZ>(c, ov, i, P2);
        w = -(double)i / k;
    } else if (pw == 2.0) {  // x_i x_j
        ldh<SZ>(c, x, i, P1);
        ldh<SZ>(c, x, j, P2);
    } else {  // f_j x_i ((i/k)(p+1) - 1)
        ldh<SZ>(c, ov, j, P1);
        ldh<SZ>(c, x, i, P2);
        w = __builtin_fma((double)i / k, pw + 1.0, -1.0);
    }
    if (o.type != OP_LOG && pw == 2.0) {  // x_i x_j: one fused multiply-add per element
        for (int e = 0; e < SZ; ++e) acc[e] = __builtin_fma(P1[e], P2[e], acc[e]);
        return;
    }
    for (int e = 0; e < SZ; ++e) acc[e] = __builtin_fma(P1[e] * P2[e], w, acc[e]);
}
SANM_HD void conv_term(const TetCtx& c, const OpDesc& o, int i, int j, double* acc) {
    if (!o.conv_n) return;
    acc += o.conv_off;
    const int64_t s = c.Tpad;
    double A[9], B[9];
    switch (o.type) {
        case OP_MULTIPLY: {
            const int asz = c.vars[o.in[0]].size, bsz = c.vars[o.in[1]].size;
            if (asz == 9 && bsz == 9) conv_term_multiply<9, 9>(c, o, i, j, acc);
            else if (asz == 1 && bsz == 9) conv_term_multiply<1, 9>(c, o, i, j, acc);
            else if (asz == 9 && bsz == 1) conv_term_multiply<9, 1>(c, o, i, j, acc);
            else if (asz == 3 && bsz == 3) conv_term_multiply<3, 3>(c, o, i, j, acc);
            else if (asz == 1 && bsz == 3) conv_term_multiply<1, 3>(c, o, i, j, acc);
            else if (asz == 3 && bsz == 1) conv_term_multiply<3, 1>(c, o, i, j, acc);
            else conv_term_multiply<1, 1>(c, o, i, j, acc);
            break;
        }
        case OP_LOG:
        case OP_POW: {
            const int sz = c.vars[o.out[0]].size;
            if (sz == 1) conv_term_unary<1>(c, o, i, j, acc);
            else if (sz == 3) conv_term_unary<3>(c, o, i, j, acc);
            else conv_term_unary<9>(c, o, i, j, acc);
            break;
        }
        case OP_MATMUL:
            ldh<9>(c, o.in[0], i, A);
            ldh<9>(c, o.in[1], j, B);
            mm3<false, false, true>(acc, A, B);
            break;
        case OP_MATINVMUL:  // Y_i X_j (left) or X_i Y_j; op_matinvmul negates the sum
            if (o.flags & OP_FLAG_IS_LEFT) {
                ldh<9>(c, o.out[0], i, A);
                ldh<9>(c, o.in[0], j, B);
            } else {
                ldh<9>(c, o.in[0], i, A);
                ldh<9>(c, o.out[0], j, B);
            }
            mm3<false, false, true>(acc, A, B);
            break;
        case OP_DET: {  // r1_i x r2_j and r0_i . c_j
            double t[3], cm[3];
            ldh<9>(c, o.in[0], i, A);  // (rows 0 and 1 are used; the unused loads fold away)
            ldh<9>(c, o.in[0], j, B);  // row 2
            ld(p_aux(c, o.aux[2]) + (int64_t)j * 3 * s, s, 3, cm);
            cross3(A + 3, B + 6, t);
            acc[0] += t[0]; acc[1] += t[1]; acc[2] += t[2];
            acc[3] += __builtin_fma(A[2], cm[2], __builtin_fma(A[1], cm[1], A[0] * cm[0]));
            break;
        }
        case OP_SVDW: {
            const int x = o.in[0], uv = o.out[0], sv = o.out[1], wv = o.out[2];
            if (o.flags & OP_FLAG_SVDW_FULL) {  // {Bu, Bw, Mb, t0k, t1k}: see op_svdw_full
                double D[3];
                ldh<9>(c, uv, i, A);
                ldh<9>(c, uv, j, B);
                mm3<true, false, true>(acc, A, B);
                ldh<3>(c, sv, j, D);
                for (int r = 0; r < 3; ++r)
                    for (int q = 0; q < 3; ++q) acc[27 + r * 3 + q] = __builtin_fma(A[r * 3 + q], D[q], acc[27 + r * 3 + q]);
                ld9(p_aux(c, o.aux[0]) + (int64_t)i * 9 * s, s, A);
                mm3<false, true, true>(acc + 36, A, B);
                ldh<9>(c, wv, i, A);
                ldh<9>(c, wv, j, B);
                mm3<true, false, true>(acc + 9, A, B);
                ld9(p_aux(c, o.aux[1]) + (int64_t)i * 9 * s, s, A);
                mm3<false, false, true>(acc + 18, A, B);
            } else {  // {Bm, Bp, Bpw}: see op_svdw
                const double* pP = p_aux(c, o.aux[0]);
                ldh<9>(c, x, i, A);
                ldh<9>(c, x, j, B);
                mm3<false, true, true>(acc, A, B);
                ld9(pP + (int64_t)i * 9 * s, s, A);
                ld9(pP + (int64_t)j * 9 * s, s, B);
                mm3<true, false, true>(acc + 9, A, B);
                ldh<9>(c, wv, j, B);
                mm3<false, false, true>(acc + 18, A, B);
            }
            break;
        }
        default: break;
    }
}
// ---- PLACEHOLDER: oprs/misc.cpp:13-44 fused with remap_in
//      (SparseLinearDesc::apply, anm.cpp:55-75): coef[k] = gather of x_k.
template <int NSLOT>
SANM_HD void gather_remap_in(const TetCtx& c, const RemapInDev& rin, const double* xvec, double* X) {
    const int64_t s = c.Tpad;
    if constexpr (NSLOT > 0) {
        uint32_t idx[9 * NSLOT];
        double coef[9 * NSLOT];
        if (rin.coef) {
            for (int i = 0; i < 9 * NSLOT; ++i) {  // slot sl of element e at i = sl * 9 + e
                idx[i] = rin.idx[(int64_t)i * s + c.tet];
                coef[i] = rin.coef[(int64_t)i * s + c.tet];
            }
        } else {  // (uniform) coefficients +-1 / 0 inside the index words: the same products and sums
            for (int i = 0; i < 9 * NSLOT; ++i) {
                const uint32_t w = rin.idx[(int64_t)i * s + c.tet];
                idx[i] = SANM_RIN_INDEX(w);
                coef[i] = SANM_RIN_COEF(w);
            }
        }
        double v[9 * NSLOT];
        for (int i = 0; i < 9 * NSLOT; ++i) v[i] = xvec[idx[i]];
        if (rin.xg) {  // (uniform) x_i formed on the way: both vectors requested together
            double g[9 * NSLOT];
            for (int i = 0; i < 9 * NSLOT; ++i) g[i] = rin.xg[idx[i]];
            const double mt = -rin.t;
            for (int i = 0; i < 9 * NSLOT; ++i) v[i] = mt * g[i] - v[i];
        }
        for (int e = 0; e < 9; ++e) {
            double acc = 0;
            for (int sl = 0; sl < NSLOT; ++sl) acc = __builtin_fma(coef[sl * 9 + e], v[sl * 9 + e], acc);
            X[e] = acc;
        }
    } else {
        for (int e = 0; e < 9; ++e) {
            double acc = 0;
            for (int sl = 0; sl < rin.nslot; ++sl) {
                int64_t off = ((int64_t)sl * 9 + e) * s + c.tet;
                const uint32_t w = rin.idx[off];
                const uint32_t ix = rin.coef ? w : SANM_RIN_INDEX(w);
                const double cf = rin.coef ? rin.coef[off] : SANM_RIN_COEF(w);
                double v = xvec[ix];
                if (rin.xg) v = -rin.t * rin.xg[ix] - v;
                acc = __builtin_fma(cf, v, acc);
            }
            X[e] = acc;
        }
    }
}
SANM_HD void op_placeholder(const TetCtx& c, const OpDesc& o, int mode, const RemapInDev& rin,
                            const double* xvec) {
    const int64_t s = c.Tpad;
    const int ov = o.out[0];
    if (mode == PASS_GRAD) {
        // the end of the reverse sweep: row grow of d(out)/d(placeholder), what the assembly gathers
        // (tet-major [T][9][9]: the contributions to one Jacobian entry come from a few tets and from up to 9
        // entries of each; tet-major those share cache lines)
        double* j = c.arena + SANM_OFF(c, c.vars[ov].jac) + c.tet * (c.odim * 9) + c.grow * 9;
        for (int e = 0; e < 9; ++e) j[e] = jget(c, ov, c.grow, e);
        return;
    }
    double X[9];
    if (mode == PASS_BIAS) {
        for (int e = 0; e < 9; ++e) X[e] = 0.0;
        st_cur(c, ov, 9, X, false);
        return;
    }
    // index -> value is a dependent pair of loads; with the slot count known at compile time all 9 x NSLOT
    // index / coefficient loads go out together and all gathers after them (two round trips instead of 18 x NSLOT)
    if (rin.nslot == 2) gather_remap_in<2>(c, rin, xvec, X);
    else if (rin.nslot == 3) gather_remap_in<3>(c, rin, xvec, X);
    else if (rin.nslot == 1) gather_remap_in<1>(c, rin, xvec, X);
    else gather_remap_in<0>(c, rin, xvec, X);
    if (mode == PASS_EVAL0) st9(p_coef(c, ov, 0), s, X);
    else st_cur(c, ov, 9, X, true);
}

SANM_HD void exec_op(const TetCtx& c, const OpDesc& o, int mode, const RemapInDev& rin,
                     const double* xvec) {
    // operators fed by constants only are evaluated once (order 0); their
    // higher-order terms are identically zero, never stored and never read
    if (mode != PASS_EVAL0 && c.vars[o.out[0]].is_const) return;
    if (mode == PASS_GRAD)
        for (int i = 0; i < o.nin; ++i)
            if ((o.grad_zero >> i) & 1) {
                const VarDesc& d = c.vars[o.in[i]];
                for (int e = 0; e < d.size; ++e) c.cur[(int64_t)(d.cur + e) * c.cur_stride] = 0.0;
            }
    if (c.part) {
        // helper wavefronts of a split BIAS pass only join the convolutions
        switch (o.type) {
            case OP_MULTIPLY: case OP_LOG: case OP_POW: case OP_MATMUL: case OP_MATINVMUL: case OP_DET:
            case OP_SVDW: break;
            default: return;
        }
    }
    switch (o.type) {
        case OP_PLACEHOLDER: op_placeholder(c, o, mode, rin, xvec); break;
        case OP_CONSTANT: break;  // uploaded at compile time
        case OP_LINCOMB: op_lincomb(c, o, mode); break;
        case OP_MULTIPLY: op_multiply(c, o, mode); break;
        case OP_LOG:
        case OP_POW: op_unary(c, o, mode); break;
        case OP_REDUCE_SUM: op_reduce(c, o, mode); break;
        case OP_MATMUL: op_matmul(c, o, mode); break;
        case OP_MATINVMUL: op_matinvmul(c, o, mode); break;
        case OP_DET: op_det(c, o, mode); break;
        case OP_TRANSPOSE: op_transpose(c, o, mode); break;
        case OP_MULEYE: op_muleye(c, o, mode); break;
        case OP_SVDW: op_svdw(c, o, mode); break;
        default: break;
    }
}

// One whole pass for one tet.  `xvec` is the (n[+1]) coefficient vector the
// placeholder gathers from (EVAL0 / COEFF passes); `cur` the per-lane scratch
// for the current-order values (P.cur_size doubles at stride cur_stride).
SANM_HD void exec_program_tet(const ProgramDev& P, int mode, int order, int64_t tet,
                              const double* xvec, double* cur, int64_t cur_stride, int part = 0,
                              int nparts = 1, double* red = nullptr) {
    TetCtx c{P.arena, P.vars, P.Tpad, tet, order, P.odim, cur, cur_stride, P.out_var, part, nparts, red};
    c.out = P.arena + P.out_aos;
    c.max_order = P.max_order;
    if (mode == PASS_GRAD) {
        // seed: row `order` of d(out)/d(out) = I  (symbolic.cpp:219-220); the launch passes the row in `order`
        c.grow = order;
        const VarDesc& d = P.vars[P.out_var];
        for (int e = 0; e < P.odim; ++e) cur[(int64_t)(d.cur + e) * cur_stride] = (e == order) ? 1.0 : 0.0;
        for (int i = P.nops - 1; i >= 0; --i) exec_op(c, P.ops[i], mode, P.rin, xvec);
    } else {
        for (int i = 0; i < P.nops; ++i) exec_op(c, P.ops[i], mode, P.rin, xvec);
        if (mode == PASS_EVAL0) {  // f(x0) for remap_out
            double Y[9];
            ld(p_coef(c, P.out_var, 0), P.Tpad, P.odim, Y);
            st_out(c, P.odim, Y);
        }
    }
}

}  // namespace sanm_hip
