// Vector graphs: Taylor propagation over batched VECTORS of arbitrary length with Slice / Concat.
//
// The reference's Slice and Concat operators (libsanm/oprs/misc.cpp:104-331) exist for graphs over (batch, n)
// tensors with elementwise arithmetic -- the Rosenbrock gradient of tests/symbolic.cpp:722-763 is the one user -- and
// no FEA graph contains them.  The per-tet machinery of program.h / tet_ops.h is built around 1, 3 or 9 doubles per
// tet; graphs with other sizes, or with Slice / Concat, are compiled into a VecProgram instead and run by the small
// interpreter below: one workgroup (256 threads) per batch item, one thread per vector element, the operators of the graph in
// sequence with a workgroup barrier between them.  Same pass structure as the tet programs (EVAL0 / GRAD / BIAS(k) /
// COEFF(k): TaylorCoeffProp::push_xi / ensure_jacobian / compute_next_order_bias, symbolic.cpp:162-289), same
// operator recurrences (elem_arith.cpp:42-217, analytic_unary.cpp:13-139, reduce.cpp:11-102).
//
// Matrices of any size up to 8 x 8 (the reference's tests run its linear-algebra operators at 4 x 4, 4 x 6, 5 x 5, 7 x 7:
// tests/symbolic.cpp:179-360, :389-424, :640-656) are vectors of rows * cols elements here with their shape kept in
// VecVar: batched_matmul / transpose / mat_inv_mul / det / mul_eye (oprs/linalg.cpp:67-479) at run-time sizes, the
// determinant's self-bias by the expansion for dim <= 4 and by the DFT of the polynomial matrix above
// (tensor_polymat.cpp:30-136, :325-379).  An operator whose recurrence needs a workgroup-wide intermediate
// (X0^-1, the inner product of mat_inv_mul, the partial sums of the determinant) is compiled into two or three
// records, which puts the interpreter's barrier between its phases.
//
// The operator bodies are shared between the HIP kernel (backend_hip.hip: vec_pass_kernel) and the test-only host
// harness (tests/hostsim/backend_host.cpp), which runs them in a loop over the elements.
#pragma once
#include <cstdint>

#include "program.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VEC_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define VEC_HD inline
#endif

namespace sanm_hip {

constexpr int VEC_MAX_SIZE = 256;  // longest vector (threads of the workgroup)
constexpr int VEC_MAX_IN = 8;      // inputs of a concat / linear combination
constexpr int VEC_MAX_DIM = 8;     // largest matrix: 8 x 8
constexpr int VEC_MAX_ORDER = 32;  // highest expansion order of a graph with a determinant (per-thread series buffers)

// records the compiler adds around the reference's operators (not part of the C ABI's operator numbering)
constexpr int VOP_INV_PREP = 32;    // X0^-1 of the mat_inv_mul that follows, order 0 only (thread 0)
constexpr int VOP_MATINV_FIN = 33;  // second product of mat_inv_mul's recurrence
constexpr int VOP_DET_FIN = 34;     // determinant: partial sums -> self-bias; cof(X0) : X_k + self-bias
constexpr int VOP_SVDW_FIN = 35;    // SVD-W (W only): the dense algebra of the polar recurrence (thread 0)
constexpr int VOP_SVDWF_B = 36;     // SVD-W with U or S read (full recurrences): (U S)(i) U' for i <= k
constexpr int VOP_SVDWF_C = 37;     //   ... Bu, Bw and the known part of coefficient k of U S U' W
constexpr int VOP_SVDWF_FIN = 38;   //   ... U_k, S_k, W_k (thread 0)

struct VecVar {
    int64_t coef;   // arena offset of coefficient 0 of batch 0; order k, batch b at coef + (k * B + b) * size
    int64_t bias;   // cur_order_bias, [B][size]
    int32_t size;
    int32_t is_const;  // orders >= 1 are zero
    int32_t grad;      // offset of the variable's gradient row in the workgroup's scratch (GRAD pass)
    int32_t const_batch;  // CONSTANT: 1 = one row broadcast over the batch
    int32_t rows, cols;   // (batch, rows, cols) tensors; cols = 0: a (batch, rows) vector
};

struct VecOp {
    int32_t type, nin, flags;
    int32_t nact;             // threads that take part in the forward passes (0: as many as the output has elements)
    int32_t in[VEC_MAX_IN];
    int32_t out;
    int32_t out_u, out_s;     // SVDW with U or S read: their variables (out is W)
    int32_t begin;            // SLICE: first element taken; CONCAT: unused
    double p[VEC_MAX_IN + 1]; // LINCOMB: coefficients, bias at p[VEC_MAX_IN]; POW: p[0] = exponent
    int64_t aux0, aux1;       // POW / LOG: K = f'(x0) [B][size], self-bias [B][size]; MULTIPLY: self-bias at aux1
    // MATMUL: self-bias at aux1.  MATINVMUL (+ its PREP / FIN): X0^-1 at aux0, self-bias at aux1, the inner
    // product at aux2 (all [B][m*m]).  DET (+ FIN): cof(X0) at aux0 [B][m*m], self-bias at aux1 [B], the threads'
    // partial sums at aux2 [B][VEC_MAX_SIZE]
    int64_t aux2;
    // SVDW (+ FIN; only W is read: the polar recurrences of tensor_svd.cpp:389-475): U0 at aux0 [B][n*n], S0 at aux1
    // [B][n], (Bm - Bp, Bpw) at aux2 [B][2][n*n], the polar factors P_k at aux3 [order][B][n*n].
    // SVDW with U or S read (tensor_svd.cpp:275-387): (Bu, Bw, Mbias) at aux2 [B][3][n*n], the two product series
    // (U S)(i), ((U S) U')(i) at aux3 [B][2][order + 1][n*n]
    int64_t aux3;
};

struct VecProgDev {
    const VecOp* ops;
    const VecVar* vars;
    double* arena;
    int32_t nops, nvars;
    int32_t in_var, out_var;  // the placeholder and the output
    int32_t idim, odim;
    int32_t max_order;
    int32_t grad_total;  // doubles of gradient scratch per workgroup
    int64_t B;
    int64_t jac;         // [B][odim][idim]
    int64_t flag;        // raise-only error words (as Program::pow_flags): [0] 0^p with p not an integer
};

// value of element e of variable v (batch b) at order k, with the scalar broadcast of elementwise operators
VEC_HD double vec_coef(const VecProgDev& P, int v, int k, int64_t b, int e) {
    const VecVar& d = P.vars[v];
    if (k > 0 && d.is_const) return 0.0;
    const int64_t bb = (d.const_batch == 1) ? 0 : b;
    const int64_t BB = (d.const_batch == 1) ? 1 : P.B;
    return P.arena[d.coef + ((int64_t)k * BB + bb) * d.size + (d.size == 1 ? 0 : e)];
}
VEC_HD double vec_bias(const VecProgDev& P, int v, int64_t b, int e) {
    const VecVar& d = P.vars[v];
    if (d.is_const) return 0.0;
    return P.arena[d.bias + b * d.size + (d.size == 1 ? 0 : e)];
}
// "current" value of the pass: the order-k coefficient (COEFF / EVAL0) or the order-k bias (BIAS)
VEC_HD double vec_cur(const VecProgDev& P, int v, int k, bool in_coeff, int64_t b, int e) {
    return in_coeff ? vec_coef(P, v, k, b, e) : vec_bias(P, v, b, e);
}
VEC_HD void vec_store(const VecProgDev& P, int v, int k, bool in_coeff, int64_t b, int e, double val) {
    const VecVar& d = P.vars[v];
    if (in_coeff) P.arena[d.coef + ((int64_t)k * P.B + b) * d.size + e] = val;
    else P.arena[d.bias + b * d.size + e] = val;
}

// ---- small dense helpers of the matrix operators (run-time sizes, per-thread local arrays) ----------------------
// determinant of the n x n row-major matrix a (destroyed): LU with partial pivoting
VEC_HD double vec_lu_det(double* a, int n) {
    double det = 1.0;
    for (int c = 0; c < n; ++c) {
        int p = c;
        double best = fabs(a[c * n + c]);
        for (int r = c + 1; r < n; ++r) {
            const double v = fabs(a[r * n + c]);
            if (v > best) {
                best = v;
                p = r;
            }
        }
        if (best == 0.0) return 0.0;
        if (p != c) {
            for (int j = 0; j < n; ++j) {
                const double t = a[c * n + j];
                a[c * n + j] = a[p * n + j];
                a[p * n + j] = t;
            }
            det = -det;
        }
        const double piv = a[c * n + c];
        det *= piv;
        for (int r = c + 1; r < n; ++r) {
            const double f = a[r * n + c] / piv;
            for (int j = c + 1; j < n; ++j) a[r * n + j] = __builtin_fma(-f, a[c * n + j], a[r * n + j]);
        }
    }
    return det;
}
// the same for a complex matrix (re, im), the determinant in (dr, di)
VEC_HD void vec_lu_det_complex(double* re, double* im, int n, double& dr, double& di) {
    dr = 1.0;
    di = 0.0;
    for (int c = 0; c < n; ++c) {
        int p = c;
        double best = re[c * n + c] * re[c * n + c] + im[c * n + c] * im[c * n + c];
        for (int r = c + 1; r < n; ++r) {
            const double v = re[r * n + c] * re[r * n + c] + im[r * n + c] * im[r * n + c];
            if (v > best) {
                best = v;
                p = r;
            }
        }
        if (best == 0.0) {
            dr = di = 0.0;
            return;
        }
        if (p != c) {
            for (int j = 0; j < n; ++j) {
                double t = re[c * n + j];
                re[c * n + j] = re[p * n + j];
                re[p * n + j] = t;
                t = im[c * n + j];
                im[c * n + j] = im[p * n + j];
                im[p * n + j] = t;
            }
            dr = -dr;
            di = -di;
        }
        const double pr = re[c * n + c], pi = im[c * n + c];
        const double ndr = dr * pr - di * pi;
        di = dr * pi + di * pr;
        dr = ndr;
        for (int r = c + 1; r < n; ++r) {
            // f = a[r][c] / pivot
            const double ar = re[r * n + c], ai = im[r * n + c];
            const double fr = (ar * pr + ai * pi) / best, fi = (ai * pr - ar * pi) / best;
            for (int j = c + 1; j < n; ++j) {
                const double br = re[c * n + j], bi = im[c * n + j];
                re[r * n + j] -= fr * br - fi * bi;
                im[r * n + j] -= fr * bi + fi * br;
            }
        }
    }
}
// inverse of the n x n matrix a into inv (Gauss-Jordan with partial pivoting; a is destroyed).  A singular matrix
// leaves non-finite entries, as Eigen's inverse() does for the reference (tensor_linalg.cpp:285-317).
VEC_HD void vec_inverse(double* a, double* inv, int n) {
    for (int i = 0; i < n * n; ++i) inv[i] = 0.0;
    for (int i = 0; i < n; ++i) inv[i * n + i] = 1.0;
    for (int c = 0; c < n; ++c) {
        int p = c;
        double best = fabs(a[c * n + c]);
        for (int r = c + 1; r < n; ++r) {
            const double v = fabs(a[r * n + c]);
            if (v > best) {
                best = v;
                p = r;
            }
        }
        if (p != c) {
            for (int j = 0; j < n; ++j) {
                double t = a[c * n + j];
                a[c * n + j] = a[p * n + j];
                a[p * n + j] = t;
                t = inv[c * n + j];
                inv[c * n + j] = inv[p * n + j];
                inv[p * n + j] = t;
            }
        }
        const double piv = a[c * n + c];
        for (int j = 0; j < n; ++j) {
            a[c * n + j] /= piv;
            inv[c * n + j] /= piv;
        }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = a[r * n + c];
            if (f == 0.0) continue;
            for (int j = 0; j < n; ++j) {
                a[r * n + j] = __builtin_fma(-f, a[c * n + j], a[r * n + j]);
                inv[r * n + j] = __builtin_fma(-f, inv[c * n + j], inv[r * n + j]);
            }
        }
    }
}

// ---- SVD-W of an n x n matrix (tensor_svd.cpp:48-145) --------------------------------------------------------------
VEC_HD double vec_clip_div(double x, double y) { return x * y / (y * y + 1e-12); }  // tensor_svd.cpp:28-31
// One-sided Jacobi SVD: a = U diag(s) V', s descending and >= 0 (svd3 of tet_ops.h at a run-time size; the reference:
// Eigen JacobiSVD, tensor_svd.cpp:66-87)
VEC_HD void vec_svd_n(const double* a, int n, double* U, double* S, double* V) {
    double B[VEC_MAX_DIM * VEC_MAX_DIM];  // working copy, columns get orthogonalised: B = A V
    for (int i = 0; i < n * n; ++i) B[i] = a[i];
    for (int i = 0; i < n * n; ++i) V[i] = (i / n == i % n) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < n; ++i) {
                    alpha += B[i * n + p] * B[i * n + p];
                    beta += B[i * n + q] * B[i * n + q];
                    gamma += B[i * n + p] * B[i * n + q];
                }
                const double lim = 1e-32 + 1e-30 * alpha * beta;
                if (gamma * gamma <= lim) continue;
                const double rel = fabs(gamma) / sqrt(alpha * beta);
                if (rel > off) off = rel;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < n; ++i) {
                    const double bp = B[i * n + p], bq = B[i * n + q];
                    B[i * n + p] = cs * bp - sn * bq;
                    B[i * n + q] = sn * bp + cs * bq;
                    const double vp = V[i * n + p], vq = V[i * n + q];
                    V[i * n + p] = cs * vp - sn * vq;
                    V[i * n + q] = sn * vp + cs * vq;
                }
            }
        if (off < 1e-15) break;
    }
    double sv[VEC_MAX_DIM];
    int idx[VEC_MAX_DIM];
    for (int j = 0; j < n; ++j) {
        double q = 0;
        for (int i = 0; i < n; ++i) q += B[i * n + j] * B[i * n + j];
        sv[j] = sqrt(q);
        idx[j] = j;
    }
    for (int i = 1; i < n; ++i) {  // descending, stable
        const int key = idx[i];
        int j = i - 1;
        while (j >= 0 && sv[idx[j]] < sv[key]) {
            idx[j + 1] = idx[j];
            --j;
        }
        idx[j + 1] = key;
    }
    double Vs[VEC_MAX_DIM * VEC_MAX_DIM];
    const double smax = sv[idx[0]];
    for (int j = 0; j < n; ++j) {
        const int k = idx[j];
        S[j] = sv[k];
        const bool tiny = !(sv[k] > 1e-300 + 1e-15 * smax);
        for (int i = 0; i < n; ++i) {
            Vs[i * n + j] = V[i * n + k];
            U[i * n + j] = tiny ? 0.0 : B[i * n + k] / sv[k];
        }
    }
    for (int i = 0; i < n * n; ++i) V[i] = Vs[i];
    // complete U for (numerically) zero singular values: the unit vector with the largest part outside the span so far
    for (int j = 0; j < n; ++j) {
        if (S[j] > 1e-300 + 1e-15 * smax) continue;
        double best = -1, w[VEC_MAX_DIM];
        for (int m = 0; m < n; ++m) {
            double c[VEC_MAX_DIM];
            for (int i = 0; i < n; ++i) c[i] = (i == m) ? 1.0 : 0.0;
            for (int q = 0; q < n; ++q) {  // columns set so far: the non-zero ones and the completed ones before j
                if (q == j || (q > j && !(S[q] > 1e-300 + 1e-15 * smax))) continue;
                double d = 0;
                for (int i = 0; i < n; ++i) d += c[i] * U[i * n + q];
                for (int i = 0; i < n; ++i) c[i] -= d * U[i * n + q];
            }
            double nn = 0;
            for (int i = 0; i < n; ++i) nn += c[i] * c[i];
            if (nn > best) {
                best = nn;
                for (int i = 0; i < n; ++i) w[i] = c[i];
            }
        }
        const double inv = 1.0 / sqrt(best);
        for (int i = 0; i < n; ++i) U[i * n + j] = w[i] * inv;
    }
}
// Which singular values get negated so that det(W) = +1: the selection loop of tensor_svd.cpp:88-128 (note the
// reference's `i = j` inside a `for(...; ++i)`); a bit mask
VEC_HD int vec_rotation_fix_mask(const double* ms, int n) {
    const double EPS = 1e-3;
    int best_idx = -1, best_idx_nr = n + 1;
    for (int i = 0; i < n; ++i) {
        int j = i + 1;
        while (j < n && fabs(ms[i] - ms[j]) < EPS) ++j;
        const int nr = j - i;
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
// M = U S U' W with W = U V' (tensor_svd.cpp:48-145)
VEC_HD void vec_svdw_n(const double* m, int n, bool require_rotation, double* U, double* S, double* W) {
    double V[VEC_MAX_DIM * VEC_MAX_DIM];
    vec_svd_n(m, n, U, S, V);
    if (require_rotation) {
        double t[VEC_MAX_DIM * VEC_MAX_DIM];
        for (int i = 0; i < n * n; ++i) t[i] = U[i];
        const double du = vec_lu_det(t, n);
        for (int i = 0; i < n * n; ++i) t[i] = V[i];
        const double dv = vec_lu_det(t, n);
        if ((du < 0) != (dv < 0)) {
            const int mask = vec_rotation_fix_mask(S, n);
            for (int j = 0; j < n; ++j)
                if (mask >> j & 1) {
                    S[j] = -S[j];
                    for (int i = 0; i < n; ++i) U[i * n + j] = -U[i * n + j];
                }
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double acc = 0;
            for (int q = 0; q < n; ++q) acc = __builtin_fma(U[i * n + q], V[j * n + q], acc);
            W[i * n + j] = acc;
        }
}

// Thread e's share of the determinant's self-bias at order k: the coefficient of a^k in det(sum_{i<k} X_i a^i)
// (compute_polymat_det_coeff, tensor_polymat.cpp:344-379, called with the k known coefficients by
// BatchDeterminantOprMeta::compute_order_bias, oprs/linalg.cpp:248-262).  dim <= 4: the Leibniz expansion
// (:201-264, :325-341), one permutation per thread, truncated Cauchy products along the rows; dim > 4: the DFT of
// the polynomial matrix at P = 2^ceil(log2((k-1) dim + 1)) roots of unity, a complex determinant at each, the
// inverse transform's k-th output (:30-136), the points dealt to the threads.  The shares are summed by VOP_DET_FIN.
VEC_HD double vec_det_selfbias_part(const VecProgDev& P, int x, int k, int64_t b, int e, int m) {
    if (k < 2) return 0.0;  // order >= nr_term = (k-1) dim + 1: identically zero
    const int nc = k;       // known coefficients X_0 .. X_{k-1}
    if (m <= 4) {
        int nperm = 1;
        for (int i = 2; i <= m; ++i) nperm *= i;
        if (e >= nperm) return 0.0;
        // e-th permutation (factoradic digits pick from the columns left); the digit sum's parity is the sign
        int perm[4], avail[4] = {0, 1, 2, 3};
        int rest = e, inv_count = 0, fact = nperm;
        for (int i = 0; i < m; ++i) {
            fact /= (m - i);
            const int d = rest / fact;
            rest -= d * fact;
            inv_count += d;
            perm[i] = avail[d];
            for (int j = d; j + 1 < m - i; ++j) avail[j] = avail[j + 1];
        }
        const double sign = (inv_count & 1) ? -1.0 : 1.0;
        auto X = [&](int i, int r) { return vec_coef(P, x, i, b, r * m + perm[r]); };
        if (m == 2) {
            double acc = 0;
            for (int i = 1; i < nc; ++i)  // conv_k: i + j = k with both below nc
                acc = __builtin_fma(X(i, 0), X(k - i, 1), acc);
            return sign * acc;
        }
        double pa[VEC_MAX_ORDER + 1], pb[VEC_MAX_ORDER + 1];
        for (int d = 0; d <= k; ++d) pa[d] = 0.0;
        for (int i = 0; i < nc; ++i)
            for (int j = 0; j < nc && i + j <= k; ++j) pa[i + j] = __builtin_fma(X(i, 0), X(j, 1), pa[i + j]);
        double* cur = pa;
        double* nxt = pb;
        for (int r = 2; r + 1 < m; ++r) {
            for (int d = 0; d <= k; ++d) nxt[d] = 0.0;
            for (int i = 0; i <= k; ++i)
                for (int j = 0; j < nc && i + j <= k; ++j) nxt[i + j] = __builtin_fma(cur[i], X(j, r), nxt[i + j]);
            double* t = cur;
            cur = nxt;
            nxt = t;
        }
        double acc = 0;
        for (int i = 1; i <= k; ++i)  // the last row's coefficient k - i must be a known one: k - i <= nc - 1
            acc = __builtin_fma(cur[i], X(k - i, m - 1), acc);
        return sign * acc;
    }
    int Pn = 1;
    while (Pn < (k - 1) * m + 1) Pn <<= 1;
    const double two_pi = 6.283185307179586476925286766559;
    double acc = 0;
    for (int j = e; j < Pn; j += VEC_MAX_SIZE) {
        const double ang = two_pi * (double)j / (double)Pn;
        const double wr = cos(ang), wi = sin(ang);
        double re[VEC_MAX_DIM * VEC_MAX_DIM], im[VEC_MAX_DIM * VEC_MAX_DIM];
        for (int q = 0; q < m * m; ++q) {  // Horner from X_{k-1}
            re[q] = vec_coef(P, x, nc - 1, b, q);
            im[q] = 0.0;
        }
        for (int i = nc - 2; i >= 0; --i) {
            for (int q = 0; q < m * m; ++q) {
                const double nr = re[q] * wr - im[q] * wi + vec_coef(P, x, i, b, q);
                im[q] = re[q] * wi + im[q] * wr;
                re[q] = nr;
            }
        }
        double dr, di;
        vec_lu_det_complex(re, im, m, dr, di);
        const double oang = -two_pi * (double)((j * k) % Pn) / (double)Pn;
        acc += dr * cos(oang) - di * sin(oang);
    }
    return acc / (double)Pn;
}

// [a^k] of (x_0 + x_1 a + ... + x_{k-1} a^{k-1})^p for an integer p >= 3 by repeated multiplication of truncated
// series: what prop_taylor_coeff_int (analytic_unary.cpp:46-92) computes by repeated squaring, for the elements
// whose x_0 is a zero the division recurrence cannot start from (as pow_int_bias of tet_ops.h)
VEC_HD double vec_pow_int_bias(const VecProgDev& P, int x, int k, int64_t b, int e, int p) {
    double y[VEC_MAX_ORDER + 1], acc[VEC_MAX_ORDER + 1], nxt[VEC_MAX_ORDER + 1];
    for (int i = 0; i < k; ++i) y[i] = vec_coef(P, x, i, b, e);
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

// One operator, one element (thread) e of batch item b, forward passes.  Elements beyond the output's size do
// nothing; reductions (reduce_sum) are done by element 0.  xin: the placeholder's values of this order, [B][idim].
VEC_HD void vec_forward(const VecProgDev& P, const VecOp& o, int mode, int k, int64_t b, int e, const double* xin) {
    const VecVar& ov = P.vars[o.out];
    const int osz = ov.size;
    if (e >= (o.nact > 0 ? o.nact : osz)) return;
    const bool in_coeff = mode != PASS_BIAS;
    if (mode != PASS_EVAL0 && ov.is_const) return;
    switch (o.type) {
        case OP_PLACEHOLDER:
            // misc.cpp:13-44: coefficient k is the caller's x_k; its bias is zero
            vec_store(P, o.out, k, in_coeff, b, e, in_coeff ? xin[b * osz + e] : 0.0);
            break;
        case OP_CONSTANT: break;  // uploaded at compile time
        case OP_LINCOMB: {  // elem_arith.cpp:42-124
            double acc = (mode == PASS_EVAL0) ? o.p[VEC_MAX_IN] : 0.0;
            for (int i = 0; i < o.nin; ++i) acc = __builtin_fma(o.p[i], vec_cur(P, o.in[i], k, in_coeff, b, e), acc);
            vec_store(P, o.out, k, in_coeff, b, e, acc);
            break;
        }
        case OP_MULTIPLY: {  // elem_arith.cpp:128-217
            const int a = o.in[0], c = o.in[1];
            if (mode == PASS_EVAL0) {
                vec_store(P, o.out, 0, true, b, e, vec_coef(P, a, 0, b, e) * vec_coef(P, c, 0, b, e));
                break;
            }
            double* psb = P.arena + o.aux1 + b * osz + e;
            double sb;
            if (!in_coeff) {
                sb = 0;
                for (int i = 1; i < k; ++i) sb = __builtin_fma(vec_coef(P, a, i, b, e), vec_coef(P, c, k - i, b, e), sb);
                *psb = sb;
            } else {
                sb = *psb;
            }
            sb = __builtin_fma(vec_coef(P, a, 0, b, e), vec_cur(P, c, k, in_coeff, b, e), sb);
            sb = __builtin_fma(vec_cur(P, a, k, in_coeff, b, e), vec_coef(P, c, 0, b, e), sb);
            vec_store(P, o.out, k, in_coeff, b, e, sb);
            break;
        }
        case OP_LOG:
        case OP_POW: {  // oprs/analytic_unary.cpp:113-158, analytic_unary.cpp:13-139
            const int x = o.in[0];
            const bool is_log = o.type == OP_LOG;
            const double pw = o.p[0];
            double* pk = P.arena + o.aux0 + b * osz + e;
            double* psb = P.arena + o.aux1 + b * osz + e;
            if (mode == PASS_EVAL0) {
                const double v = vec_coef(P, x, 0, b, e);
                double f, kk;
                if (is_log) {
                    f = log(v);
                    kk = 1.0 / v;
                } else if (pw == 2.0) {
                    f = v * v;
                    kk = 2.0 * v;
                } else {
                    f = pow(v, pw);
                    kk = pw * pow(v, pw - 1.0);
                    // analytic_unary.cpp:112-131: the division recurrence cannot start from a zero; integer
                    // exponents continue on the convolution path (vec_pow_int_bias), up to VEC_MAX_ORDER
                    // (word 0 <- 1: the reference's error; word 1 <- 2: beyond that order -- the values of Program's flags)
                    if (fabs(v) < 1e-3 && !P.vars[x].is_const) {
                        const bool integer = pw > 0.5 && floor(pw) == pw;
                        if (!integer) P.arena[P.flag] = 1.0;
                        else if (P.max_order > VEC_MAX_ORDER) P.arena[P.flag + 1] = 2.0;
                    }
                }
                vec_store(P, o.out, 0, true, b, e, f);
                *pk = kk;
                break;
            }
            double sb;
            if (!in_coeff) {
                sb = 0;
                if (!P.vars[x].is_const) {
                    if (!is_log && pw == 2.0) {
                        for (int i = 1; i < k; ++i) sb = __builtin_fma(vec_coef(P, x, i, b, e), vec_coef(P, x, k - i, b, e), sb);
                    } else {
                        for (int i = 1; i < k; ++i) {
                            // log: x[k-i] f[i] (-i/k); pow: f[k-i] x[i] ((i/k)(p+1) - 1)
                            const double p1 = is_log ? vec_coef(P, x, k - i, b, e) : vec_coef(P, o.out, k - i, b, e);
                            const double p2 = is_log ? vec_coef(P, o.out, i, b, e) : vec_coef(P, x, i, b, e);
                            const double w = is_log ? -(double)i / (double)k
                                                    : __builtin_fma((double)i / (double)k, pw + 1.0, -1.0);
                            sb = __builtin_fma(p1 * p2, w, sb);
                        }
                        const double x0 = vec_coef(P, x, 0, b, e);
                        if (!is_log && pw > 2.5 && floor(pw) == pw && fabs(x0) < 1e-3 && k <= VEC_MAX_ORDER)
                            sb = vec_pow_int_bias(P, x, k, b, e, (int)pw);
                        else
                            sb /= x0;
                    }
                }
                *psb = sb;
            } else {
                sb = *psb;
            }
            if (!P.vars[x].is_const) sb = __builtin_fma(*pk, vec_cur(P, x, k, in_coeff, b, e), sb);
            vec_store(P, o.out, k, in_coeff, b, e, sb);
            break;
        }
        case OP_REDUCE_SUM: {  // reduce.cpp:11-102, axis -1: element 0 sums
            if (o.begin == 1 || o.begin == 2) {  // one axis of a matrix: (1 x cols) or (rows x 1)
                const int ir = P.vars[o.in[0]].rows, ic = P.vars[o.in[0]].cols;
                double sum = 0;
                if (o.begin == 1)
                    for (int i = 0; i < ir; ++i) sum += vec_cur(P, o.in[0], k, in_coeff, b, i * ic + e);
                else
                    for (int j = 0; j < ic; ++j) sum += vec_cur(P, o.in[0], k, in_coeff, b, e * ic + j);
                vec_store(P, o.out, k, in_coeff, b, e, sum);
                break;
            }
            const int isz = P.vars[o.in[0]].size;
            double sum = 0;
            for (int i = 0; i < isz; ++i) sum += vec_cur(P, o.in[0], k, in_coeff, b, i);
            vec_store(P, o.out, k, in_coeff, b, 0, sum);
            break;
        }
        case OP_SLICE:  // misc.cpp:142-164, :199-216 (the order-1 bias is zero because its input's is)
            vec_store(P, o.out, k, in_coeff, b, e, vec_cur(P, o.in[0], k, in_coeff, b, o.begin + e));
            break;
        case OP_CONCAT: {  // misc.cpp:291-318
            int off = 0;
            for (int i = 0; i < o.nin; ++i) {
                const int n = P.vars[o.in[i]].size;
                if (e < off + n) {
                    vec_store(P, o.out, k, in_coeff, b, e, vec_cur(P, o.in[i], k, in_coeff, b, e - off));
                    break;
                }
                off += n;
            }
            break;
        }
        case OP_TRANSPOSE: {  // oprs/linalg.cpp:286-335; out is (cols x rows) of the input
            const int orows = ov.rows, ocols = ov.cols;
            const int r = e / ocols, c = e % ocols;
            vec_store(P, o.out, k, in_coeff, b, e, vec_cur(P, o.in[0], k, in_coeff, b, c * orows + r));
            break;
        }
        case OP_MULEYE: {  // oprs/linalg.cpp:422-479
            const int d = ov.rows;
            vec_store(P, o.out, k, in_coeff, b, e, (e / d == e % d) ? vec_cur(P, o.in[0], k, in_coeff, b, 0) : 0.0);
            break;
        }
        case OP_MATMUL: {  // oprs/linalg.cpp:339-418: Y_k = A_0 B_k + A_k B_0 + sum_{0<i<k} A_i B_{k-i}
            const int a = o.in[0], c = o.in[1];
            const int K = P.vars[a].cols, n = ov.cols;
            const int r = e / n, cc = e % n;
            if (mode == PASS_EVAL0) {
                double s = 0;
                for (int j = 0; j < K; ++j) s = __builtin_fma(vec_coef(P, a, 0, b, r * K + j), vec_coef(P, c, 0, b, j * n + cc), s);
                vec_store(P, o.out, 0, true, b, e, s);
                break;
            }
            double* psb = P.arena + o.aux1 + b * osz + e;
            double sb;
            if (!in_coeff) {
                sb = 0;
                for (int i = 1; i < k; ++i)
                    for (int j = 0; j < K; ++j)
                        sb = __builtin_fma(vec_coef(P, a, i, b, r * K + j), vec_coef(P, c, k - i, b, j * n + cc), sb);
                *psb = sb;
            } else {
                sb = *psb;
            }
            for (int j = 0; j < K; ++j) {
                sb = __builtin_fma(vec_cur(P, a, k, in_coeff, b, r * K + j), vec_coef(P, c, 0, b, j * n + cc), sb);
                sb = __builtin_fma(vec_coef(P, a, 0, b, r * K + j), vec_cur(P, c, k, in_coeff, b, j * n + cc), sb);
            }
            vec_store(P, o.out, k, in_coeff, b, e, sb);
            break;
        }
        case VOP_INV_PREP: {  // X0^-1 for the two records that follow (oprs/linalg.cpp:82-96)
            if (mode != PASS_EVAL0 || e != 0) break;
            const int m = P.vars[o.in[0]].rows;
            double a[VEC_MAX_DIM * VEC_MAX_DIM], inv[VEC_MAX_DIM * VEC_MAX_DIM];
            for (int q = 0; q < m * m; ++q) a[q] = vec_coef(P, o.in[0], 0, b, q);
            vec_inverse(a, inv, m);
            for (int q = 0; q < m * m; ++q) P.arena[o.aux0 + b * m * m + q] = inv[q];
            break;
        }
        case OP_MATINVMUL: {
            // Y X = A (is_left) or X Y = A, oprs/linalg.cpp:67-217: order 0 directly; order k in two products, the
            // inner one here:  tmp = A_k - sum_{0<i<k} (Y_i X_{k-i} | X_i Y_{k-i}) - (Y_0 X_k | X_k Y_0)
            const int x = o.in[0];
            const int m = ov.rows;
            const int r = e / m, c = e % m;
            const bool left = o.flags & OP_FLAG_IS_LEFT, ident = o.flags & OP_FLAG_USE_IDENTITY;
            const double* xinv = P.arena + o.aux0 + b * m * m;
            if (mode == PASS_EVAL0) {
                double s;
                if (ident) {
                    s = xinv[e];
                } else {
                    s = 0;
                    for (int j = 0; j < m; ++j)
                        s = left ? __builtin_fma(vec_coef(P, o.in[1], 0, b, r * m + j), xinv[j * m + c], s)
                                 : __builtin_fma(xinv[r * m + j], vec_coef(P, o.in[1], 0, b, j * m + c), s);
                }
                vec_store(P, o.out, 0, true, b, e, s);
                break;
            }
            double* psb = P.arena + o.aux1 + b * osz + e;
            double sb;
            if (!in_coeff) {
                sb = 0;
                for (int i = 1; i < k; ++i)
                    for (int j = 0; j < m; ++j)
                        sb = left ? __builtin_fma(-vec_coef(P, o.out, i, b, r * m + j), vec_coef(P, x, k - i, b, j * m + c), sb)
                                  : __builtin_fma(-vec_coef(P, x, i, b, r * m + j), vec_coef(P, o.out, k - i, b, j * m + c), sb);
                *psb = sb;
            } else {
                sb = *psb;
            }
            if (!ident) sb += vec_cur(P, o.in[1], k, in_coeff, b, e);
            for (int j = 0; j < m; ++j)
                sb = left ? __builtin_fma(-vec_coef(P, o.out, 0, b, r * m + j), vec_cur(P, x, k, in_coeff, b, j * m + c), sb)
                          : __builtin_fma(-vec_cur(P, x, k, in_coeff, b, r * m + j), vec_coef(P, o.out, 0, b, j * m + c), sb);
            P.arena[o.aux2 + b * osz + e] = sb;
            break;
        }
        case VOP_MATINV_FIN: {  // ... and the outer one: (tmp X0^-1 | X0^-1 tmp)
            if (mode == PASS_EVAL0) break;
            const int m = ov.rows;
            const int r = e / m, c = e % m;
            const bool left = o.flags & OP_FLAG_IS_LEFT;
            const double* xinv = P.arena + o.aux0 + b * m * m;
            const double* tmp = P.arena + o.aux2 + b * osz;
            double s = 0;
            for (int j = 0; j < m; ++j)
                s = left ? __builtin_fma(tmp[r * m + j], xinv[j * m + c], s) : __builtin_fma(xinv[r * m + j], tmp[j * m + c], s);
            vec_store(P, o.out, k, in_coeff, b, e, s);
            break;
        }
        case OP_DET: {  // oprs/linalg.cpp:221-282, first phase: cof(X0) at order 0, the self-bias's shares at BIAS(k)
            const int x = o.in[0];
            const int m = P.vars[x].rows;
            if (mode == PASS_EVAL0) {
                if (e >= m * m) break;
                const int r = e / m, c = e % m;
                double minor[(VEC_MAX_DIM - 1) * (VEC_MAX_DIM - 1)];
                int q = 0;
                for (int i = 0; i < m; ++i) {
                    if (i == r) continue;
                    for (int j = 0; j < m; ++j)
                        if (j != c) minor[q++] = vec_coef(P, x, 0, b, i * m + j);
                }
                const double d = m == 1 ? 1.0 : vec_lu_det(minor, m - 1);
                P.arena[o.aux0 + b * m * m + e] = ((r + c) & 1) ? -d : d;
            } else if (mode == PASS_BIAS) {
                P.arena[o.aux2 + b * VEC_MAX_SIZE + e] = vec_det_selfbias_part(P, x, k, b, e, m);
            }
            break;
        }
        case VOP_DET_FIN: {  // second phase (thread 0): det(X0) along row 0 of the cofactors; cof : X_k + self-bias
            const int x = o.in[0];
            const int m = P.vars[x].rows;
            const double* cof = P.arena + o.aux0 + b * m * m;
            if (mode == PASS_EVAL0) {
                double d = 0;
                for (int c = 0; c < m; ++c) d = __builtin_fma(vec_coef(P, x, 0, b, c), cof[c], d);
                vec_store(P, o.out, 0, true, b, 0, d);
                break;
            }
            double sb;
            if (!in_coeff) {
                sb = 0;
                for (int t = 0; t < VEC_MAX_SIZE; ++t) sb += P.arena[o.aux2 + b * VEC_MAX_SIZE + t];
                P.arena[o.aux1 + b] = sb;
            } else {
                sb = P.arena[o.aux1 + b];
            }
            for (int q = 0; q < m * m; ++q) sb = __builtin_fma(cof[q], vec_cur(P, x, k, in_coeff, b, q), sb);
            vec_store(P, o.out, k, in_coeff, b, 0, sb);
            break;
        }
        case OP_SVDW: {
            // oprs/linalg.cpp:483-615 with only W read (pw_mode, :533-560): order 0 by thread 0; at BIAS(k) the three
            // convolutions Bm - Bp = sum M_i M_{k-i}' - P_i' P_{k-i} and Bpw = sum P_i W_{k-i} (0 < i < k), an
            // element per thread
            const int x = o.in[0];
            const int n = ov.rows, nn = n * n;
            if (mode == PASS_EVAL0) {
                if (e != 0) break;
                double m[VEC_MAX_DIM * VEC_MAX_DIM], U[VEC_MAX_DIM * VEC_MAX_DIM], S[VEC_MAX_DIM], W[VEC_MAX_DIM * VEC_MAX_DIM];
                for (int q = 0; q < nn; ++q) m[q] = vec_coef(P, x, 0, b, q);
                vec_svdw_n(m, n, o.flags & OP_FLAG_REQUIRE_ROT, U, S, W);
                for (int q = 0; q < nn; ++q) {
                    P.arena[o.aux0 + b * nn + q] = U[q];
                    vec_store(P, o.out, 0, true, b, q, W[q]);
                    if (o.flags & OP_FLAG_SVDW_FULL) vec_store(P, o.out_u, 0, true, b, q, U[q]);
                }
                for (int q = 0; q < n; ++q) {
                    P.arena[o.aux1 + b * n + q] = S[q];
                    if (o.flags & OP_FLAG_SVDW_FULL) vec_store(P, o.out_s, 0, true, b, q, S[q]);
                }
                break;
            }
            if (mode != PASS_BIAS || e >= nn) break;
            const int r = e / n, c = e % n;
            if (o.flags & OP_FLAG_SVDW_FULL) {
                // first of the four phases of the full recurrences (VOP_SVDWF_*): T0_i = sum_j U_j diag(S_{i-j}) over
                // the known terms (both indices below k), i <= k  (oprs/linalg.cpp:42-62, :570-590)
                if (k < 2) break;
                double* T0 = P.arena + o.aux3 + (b * 2 + 0) * (int64_t)(P.max_order + 1) * nn;
                for (int i = 0; i <= k; ++i) {
                    double acc = 0;
                    for (int j = (i >= k ? i - k + 1 : 0); j <= i && j < k; ++j)
                        acc = __builtin_fma(vec_coef(P, o.out_u, j, b, e), vec_coef(P, o.out_s, i - j, b, c), acc);
                    T0[i * nn + e] = acc;
                }
                break;
            }
            double d = 0, bpw = 0;
            for (int i = 1; i < k; ++i) {
                const double* Pi = P.arena + o.aux3 + ((int64_t)i * P.B + b) * nn;
                const double* Pk = P.arena + o.aux3 + ((int64_t)(k - i) * P.B + b) * nn;
                for (int j = 0; j < n; ++j) {
                    d = __builtin_fma(vec_coef(P, x, i, b, r * n + j), vec_coef(P, x, k - i, b, c * n + j), d);
                    d = __builtin_fma(-Pi[j * n + r], Pk[j * n + c], d);
                    bpw = __builtin_fma(Pi[r * n + j], vec_coef(P, o.out, k - i, b, j * n + c), bpw);
                }
            }
            P.arena[o.aux2 + (b * 2 + 0) * nn + e] = d;
            P.arena[o.aux2 + (b * 2 + 1) * nn + e] = bpw;
            break;
        }
        case VOP_SVDW_FIN: {
            // svd_w_taylor_fwd_p, tensor_svd.cpp:389-475 (thread 0): with V0 = W0' U0,
            //   Q = S0 V0' Mk' U0;  E = U0' (Bm - Bp)' U0 + Q + Q';  X = clip_div(E_ij, s_i + s_j)
            //   Pk = (U0 X U0')';  Wk = U0 diag(clip_div(1, s_i)) U0' (Mk - Bpw - Pk W0)
            if (mode == PASS_EVAL0 || e != 0) break;
            const int x = o.in[0];
            const int n = ov.rows, nn = n * n;
            const double* U0 = P.arena + o.aux0 + b * nn;
            const double* S0 = P.arena + o.aux1 + b * n;
            const double* D = P.arena + o.aux2 + (b * 2 + 0) * nn;
            const double* Bpw = P.arena + o.aux2 + (b * 2 + 1) * nn;
            double W0[VEC_MAX_DIM * VEC_MAX_DIM], Mk[VEC_MAX_DIM * VEC_MAX_DIM], V0[VEC_MAX_DIM * VEC_MAX_DIM];
            double T0[VEC_MAX_DIM * VEC_MAX_DIM], T1[VEC_MAX_DIM * VEC_MAX_DIM], X[VEC_MAX_DIM * VEC_MAX_DIM];
            for (int q = 0; q < nn; ++q) {
                W0[q] = vec_coef(P, o.out, 0, b, q);
                Mk[q] = vec_cur(P, x, k, in_coeff, b, q);
            }
            auto at = [n](const double* a, int i, int j) { return a[i * n + j]; };
            for (int i = 0; i < n; ++i)  // V0 = W0' U0
                for (int j = 0; j < n; ++j) {
                    double acc = 0;
                    for (int q = 0; q < n; ++q) acc = __builtin_fma(at(W0, q, i), at(U0, q, j), acc);
                    V0[i * n + j] = acc;
                }
            // T0 = D' U0 ; T1 = Mk' U0
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double a0 = 0, a1 = 0;
                    for (int q = 0; q < n; ++q) {
                        a0 = __builtin_fma(at(D, q, i), at(U0, q, j), a0);
                        a1 = __builtin_fma(at(Mk, q, i), at(U0, q, j), a1);
                    }
                    T0[i * n + j] = a0;
                    T1[i * n + j] = a1;
                }
            // E = U0' T0 + Q + Q' with Q_ij = s_i (V0' T1)_ij ; held in X, then divided
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double acc = 0, qij = 0, qji = 0;
                    for (int q = 0; q < n; ++q) {
                        acc = __builtin_fma(at(U0, q, i), at(T0, q, j), acc);
                        qij = __builtin_fma(at(V0, q, i), at(T1, q, j), qij);
                        qji = __builtin_fma(at(V0, q, j), at(T1, q, i), qji);
                    }
                    X[i * n + j] = vec_clip_div(acc + S0[i] * qij + S0[j] * qji, S0[i] + S0[j]);
                }
            // Pk' = U0 X U0'  (T0 = U0 X, then Pk[r][c] = sum_j T0[c][j] U0[r][j])
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double acc = 0;
                    for (int q = 0; q < n; ++q) acc = __builtin_fma(at(U0, i, q), at(X, q, j), acc);
                    T0[i * n + j] = acc;
                }
            double* Pk = T1;
            for (int r = 0; r < n; ++r)
                for (int c = 0; c < n; ++c) {
                    double acc = 0;
                    for (int j = 0; j < n; ++j) acc = __builtin_fma(at(T0, c, j), at(U0, r, j), acc);
                    Pk[r * n + c] = acc;
                }
            if (in_coeff)
                for (int q = 0; q < nn; ++q) P.arena[o.aux3 + ((int64_t)k * P.B + b) * nn + q] = Pk[q];
            // R = Mk - Bpw - Pk W0 (into X); Z = diag(1/s) U0' R (into T0); Wk = U0 Z
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double acc = Mk[i * n + j] - Bpw[i * n + j];
                    for (int q = 0; q < n; ++q) acc = __builtin_fma(-at(Pk, i, q), at(W0, q, j), acc);
                    X[i * n + j] = acc;
                }
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double acc = 0;
                    for (int q = 0; q < n; ++q) acc = __builtin_fma(at(U0, q, i), at(X, q, j), acc);
                    T0[i * n + j] = acc * vec_clip_div(1.0, S0[i]);
                }
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double acc = 0;
                    for (int q = 0; q < n; ++q) acc = __builtin_fma(at(U0, i, q), at(T0, q, j), acc);
                    vec_store(P, o.out, k, in_coeff, b, i * n + j, acc);
                }
            break;
        }
        case VOP_SVDWF_B:
        case VOP_SVDWF_C:
        case VOP_SVDWF_FIN: {
            // Full SVD-W recurrences (U or S is read; oprs/linalg.cpp:561-600, tensor_svd.cpp:275-387).  At BIAS(k):
            // OP_SVDW has left T0_i = sum_j U_j diag(S_{i-j}) (known terms, i <= k); B: T1_i = sum_j T0_j U_{i-j}';
            // C: Mbias = sum_{0<i<=k} T1_i W_{k-i}, Bu = sum U_i' U_{k-i}, Bw = sum W_i' W_{k-i} (0 < i < k);
            // FIN (BIAS and COEFF, thread 0): U_k, S_k, W_k from M_k, Mbias, Bu, Bw.
            if (mode == PASS_EVAL0) break;
            const int x = o.in[0];
            const int n = ov.rows, nn = n * n;
            const int64_t No = P.max_order + 1;
            double* T0 = P.arena + o.aux3 + (b * 2 + 0) * No * nn;
            double* T1 = P.arena + o.aux3 + (b * 2 + 1) * No * nn;
            double* Bu = P.arena + o.aux2 + (b * 3 + 0) * nn;
            double* Bw = P.arena + o.aux2 + (b * 3 + 1) * nn;
            double* Mb = P.arena + o.aux2 + (b * 3 + 2) * nn;
            if (o.type == VOP_SVDWF_B) {
                if (mode != PASS_BIAS || e >= nn || k < 2) break;
                const int r = e / n, c = e % n;
                for (int i = 0; i <= k; ++i) {
                    double acc = 0;
                    for (int j = (i >= k ? i - k + 1 : 0); j <= i; ++j)  // T0_j known for j <= k, U_{i-j} for i - j < k
                        for (int q = 0; q < n; ++q)
                            acc = __builtin_fma(T0[j * nn + r * n + q], vec_coef(P, o.out_u, i - j, b, c * n + q), acc);
                    T1[i * nn + e] = acc;
                }
                break;
            }
            if (o.type == VOP_SVDWF_C) {
                if (mode != PASS_BIAS || e >= nn) break;
                const int r = e / n, c = e % n;
                double mb = 0, bu = 0, bw = 0;
                if (k >= 2) {
                    for (int i = 1; i <= k; ++i)
                        for (int q = 0; q < n; ++q)
                            mb = __builtin_fma(T1[i * nn + r * n + q], vec_coef(P, o.out, k - i, b, q * n + c), mb);
                    for (int i = 1; i < k; ++i)
                        for (int q = 0; q < n; ++q) {
                            bu = __builtin_fma(vec_coef(P, o.out_u, i, b, q * n + r), vec_coef(P, o.out_u, k - i, b, q * n + c), bu);
                            bw = __builtin_fma(vec_coef(P, o.out, i, b, q * n + r), vec_coef(P, o.out, k - i, b, q * n + c), bw);
                        }
                }
                Mb[e] = mb;
                Bu[e] = bu;
                Bw[e] = bw;
                break;
            }
            if (e != 0) break;
            // svd_w_taylor_fwd (tensor_svd.cpp:275-387), on plain (untransposed) matrices:
            //   Et = V0' (M_k - Mb)' U0;  R = Et' - Et - (V0' Bw V0) diag(s);  X_ij = clip_div(R_ij, s_i + s_j)
            //   W_k = U0 X V0';  Et -= X' diag(s);  Et += Bu' diag(s);  S_k = diag Et
            //   K_ij = clip_div(Et_ij, s_i - s_j), K_ji = -Bu_ij - K_ij (i < j), K_jj = -Bu_jj / 2;  U_k = U0 K'
            double U0[VEC_MAX_DIM * VEC_MAX_DIM], V0[VEC_MAX_DIM * VEC_MAX_DIM], S0[VEC_MAX_DIM];
            double A[VEC_MAX_DIM * VEC_MAX_DIM], Et[VEC_MAX_DIM * VEC_MAX_DIM], X[VEC_MAX_DIM * VEC_MAX_DIM];
            for (int q = 0; q < nn; ++q) U0[q] = vec_coef(P, o.out_u, 0, b, q);
            for (int q = 0; q < n; ++q) S0[q] = vec_coef(P, o.out_s, 0, b, q);
            for (int i = 0; i < n; ++i)  // V0 = W0' U0
                for (int j = 0; j < n; ++j) {
                    double acc = 0;
                    for (int q = 0; q < n; ++q) acc = __builtin_fma(vec_coef(P, o.out, 0, b, q * n + i), U0[q * n + j], acc);
                    V0[i * n + j] = acc;
                }
            for (int i = 0; i < n; ++i)  // A = (M_k - Mb)' U0
                for (int j = 0; j < n; ++j) {
                    double acc = 0;
                    for (int q = 0; q < n; ++q)
                        acc = __builtin_fma(vec_cur(P, x, k, in_coeff, b, q * n + i) - Mb[q * n + i], U0[q * n + j], acc);
                    A[i * n + j] = acc;
                }
            for (int i = 0; i < n; ++i)  // Et = V0' A
                for (int j = 0; j < n; ++j) {
                    double acc = 0;
                    for (int q = 0; q < n; ++q) acc = __builtin_fma(V0[q * n + i], A[q * n + j], acc);
                    Et[i * n + j] = acc;
                }
            for (int i = 0; i < n; ++i)  // A = Bw V0
                for (int j = 0; j < n; ++j) {
                    double acc = 0;
                    for (int q = 0; q < n; ++q) acc = __builtin_fma(Bw[i * n + q], V0[q * n + j], acc);
                    A[i * n + j] = acc;
                }
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double acc = 0;  // (V0' Bw V0)_ij
                    for (int q = 0; q < n; ++q) acc = __builtin_fma(V0[q * n + i], A[q * n + j], acc);
                    X[i * n + j] = vec_clip_div(Et[j * n + i] - Et[i * n + j] - acc * S0[j], S0[i] + S0[j]);
                }
            for (int i = 0; i < n; ++i)  // A = X V0' ; W_k = U0 A
                for (int j = 0; j < n; ++j) {
                    double acc = 0;
                    for (int q = 0; q < n; ++q) acc = __builtin_fma(X[i * n + q], V0[j * n + q], acc);
                    A[i * n + j] = acc;
                }
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double acc = 0;
                    for (int q = 0; q < n; ++q) acc = __builtin_fma(U0[i * n + q], A[q * n + j], acc);
                    vec_store(P, o.out, k, in_coeff, b, i * n + j, acc);
                }
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) Et[i * n + j] += (Bu[j * n + i] - X[j * n + i]) * S0[j];
            for (int i = 0; i < n; ++i) vec_store(P, o.out_s, k, in_coeff, b, i, Et[i * n + i]);
            for (int j = 0; j < n; ++j) {  // K (into A)
                for (int i = 0; i < j; ++i) {
                    const double vv = vec_clip_div(Et[i * n + j], S0[i] - S0[j]);
                    A[i * n + j] = vv;
                    A[j * n + i] = -Bu[i * n + j] - vv;
                }
                A[j * n + j] = -Bu[j * n + j] / 2;
            }
            for (int i = 0; i < n; ++i)  // U_k = U0 K'
                for (int j = 0; j < n; ++j) {
                    double acc = 0;
                    for (int q = 0; q < n; ++q) acc = __builtin_fma(U0[i * n + q], A[j * n + q], acc);
                    vec_store(P, o.out_u, k, in_coeff, b, i * n + j, acc);
                }
            break;
        }
        default: break;
    }
}

// Reverse sweep of one operator for the Jacobian row held in `g` (gradient rows of all variables, VecVar::grad):
// element e of every INPUT accumulates what this operator passes back.  Inputs that are batched scalars read by a
// vector operator receive the sum over the output's elements (done by element 0).  accum_inp_grad of the metas.
VEC_HD void vec_backward(const VecProgDev& P, const VecOp& o, int64_t b, int e, double* g) {
    const VecVar& ov = P.vars[o.out];
    const int osz = ov.size;
    const double* go = g + ov.grad;
    auto add = [&](int v, int idx, double val) { g[P.vars[v].grad + idx] += val; };
    // contribution to input v (size isz) of per-element factor fac(e') * go[e']
    switch (o.type) {
        case OP_LINCOMB:
            for (int i = 0; i < o.nin; ++i) {
                const int v = o.in[i], isz = P.vars[v].size;
                if (P.vars[v].is_const) continue;
                if (isz == osz) {
                    if (e < osz) add(v, e, o.p[i] * go[e]);
                } else if (e == 0) {
                    double s = 0;
                    for (int q = 0; q < osz; ++q) s += go[q];
                    add(v, 0, o.p[i] * s);
                }
            }
            break;
        case OP_MULTIPLY:
            for (int i = 0; i < 2; ++i) {
                const int v = o.in[i], other = o.in[1 - i], isz = P.vars[v].size;
                if (P.vars[v].is_const) continue;
                if (isz == osz) {
                    if (e < osz) add(v, e, go[e] * vec_coef(P, other, 0, b, e));
                } else if (e == 0) {
                    double s = 0;
                    for (int q = 0; q < osz; ++q) s = __builtin_fma(go[q], vec_coef(P, other, 0, b, q), s);
                    add(v, 0, s);
                }
            }
            break;
        case OP_LOG:
        case OP_POW:
            if (e < osz && !P.vars[o.in[0]].is_const) add(o.in[0], e, go[e] * P.arena[o.aux0 + b * osz + e]);
            break;
        case OP_REDUCE_SUM:
            if (e < P.vars[o.in[0]].size && !P.vars[o.in[0]].is_const) {
                const int ic = P.vars[o.in[0]].cols;
                add(o.in[0], e, o.begin == 1 ? go[e % ic] : (o.begin == 2 ? go[e / ic] : go[0]));
            }
            break;
        case OP_SLICE:
            if (e < osz && !P.vars[o.in[0]].is_const) add(o.in[0], o.begin + e, go[e]);
            break;
        case OP_CONCAT: {
            // Thread e owns element e of every DISTINCT operand: a variable listed twice (concat([x, x])) receives
            // the sum of its occurrences in one add -- one thread per output element adding into the operand's
            // gradient would race on such a variable (the adds into the LDS scratch are not atomic).
            int off_i = 0;
            for (int i = 0; i < o.nin; ++i) {
                const int v = o.in[i], n = P.vars[v].size;
                bool first = true;
                for (int j = 0; j < i; ++j) first = first && o.in[j] != v;
                if (first && e < n && !P.vars[v].is_const) {
                    double s = 0;
                    int off_j = off_i;
                    for (int j = i; j < o.nin; ++j) {
                        if (o.in[j] == v) s += go[off_j + e];
                        off_j += P.vars[o.in[j]].size;
                    }
                    add(v, e, s);
                }
                off_i += n;
            }
            break;
        }
        case OP_TRANSPOSE: {  // input (ir x ic), output (ic x ir)
            const int v = o.in[0], ir = P.vars[v].rows, ic = P.vars[v].cols;
            if (e < ir * ic && !P.vars[v].is_const) add(v, e, go[(e % ic) * ir + e / ic]);
            break;
        }
        case OP_MULEYE:
            if (e == 0 && !P.vars[o.in[0]].is_const) {
                const int d = ov.rows;
                double s = 0;
                for (int i = 0; i < d; ++i) s += go[i * d + i];
                add(o.in[0], 0, s);
            }
            break;
        case OP_MATMUL: {  // g_A = g_Y B0^T, g_B = A0^T g_Y
            const int a = o.in[0], c = o.in[1];
            const int m = ov.rows, n = ov.cols, K = P.vars[a].cols;
            if (e < m * K && !P.vars[a].is_const) {
                const int r = e / K, j = e % K;
                double s = 0;
                for (int q = 0; q < n; ++q) s = __builtin_fma(go[r * n + q], vec_coef(P, c, 0, b, j * n + q), s);
                add(a, e, s);
            }
            if (e < K * n && !P.vars[c].is_const) {
                const int j = e / n, q = e % n;
                double s = 0;
                for (int r = 0; r < m; ++r) s = __builtin_fma(go[r * n + q], vec_coef(P, a, 0, b, r * K + j), s);
                add(c, e, s);
            }
            break;
        }
        case OP_MATINVMUL: {
            // oprs/linalg.cpp:98-150: g_X[i,j] = sum_pq g_Y[p,q] m0[p,i] m1[j,q] with (m0, m1) = (-Y0, X0^-1) for
            // Y X = A and (X0^-1, -Y0) for X Y = A; g_A = g_Y X0^-T (left) or X0^-T g_Y (right)
            const int x = o.in[0];
            const int m = ov.rows;
            if (e >= m * m) break;
            const int i = e / m, j = e % m;
            const bool left = o.flags & OP_FLAG_IS_LEFT, ident = o.flags & OP_FLAG_USE_IDENTITY;
            const double* xinv = P.arena + o.aux0 + b * m * m;
            if (!P.vars[x].is_const) {
                double s = 0;
                for (int p = 0; p < m; ++p) {
                    const double m0 = left ? -vec_coef(P, o.out, 0, b, p * m + i) : xinv[p * m + i];
                    double t = 0;
                    for (int q = 0; q < m; ++q) {
                        const double m1 = left ? xinv[j * m + q] : -vec_coef(P, o.out, 0, b, j * m + q);
                        t = __builtin_fma(go[p * m + q], m1, t);
                    }
                    s = __builtin_fma(m0, t, s);
                }
                add(x, e, s);
            }
            if (!ident && !P.vars[o.in[1]].is_const) {
                double s = 0;
                for (int q = 0; q < m; ++q)
                    s = left ? __builtin_fma(go[i * m + q], xinv[j * m + q], s) : __builtin_fma(go[q * m + j], xinv[q * m + i], s);
                add(o.in[1], e, s);
            }
            break;
        }
        case OP_DET: {  // oprs/linalg.cpp:234-246: g_X = g_y cof(X0)
            const int x = o.in[0], m = P.vars[x].rows;
            if (e < m * m && !P.vars[x].is_const) add(x, e, go[0] * P.arena[o.aux0 + b * m * m + e]);
            break;
        }
        case OP_SVDW: {
            // dW/dM of tensor_svd.cpp:147-273 applied to the gradient row: with G' = U0' g_W V0 (V0 = W0' U0),
            // g_M[k,l] = sum_{a != b} G'[a,b] clip_div(U0[k,a] V0[l,b] - U0[k,b] V0[l,a], s_a + s_b)
            const int x = o.in[0];
            const int n = ov.rows, nn = n * n;
            if (e >= nn || P.vars[x].is_const) break;
            const int kk = e / n, l = e % n;
            const double* U0 = P.arena + o.aux0 + b * nn;
            const double* S0 = P.arena + o.aux1 + b * n;
            double V0[VEC_MAX_DIM * VEC_MAX_DIM], T[VEC_MAX_DIM * VEC_MAX_DIM];
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double acc = 0;
                    for (int q = 0; q < n; ++q) acc = __builtin_fma(vec_coef(P, o.out, 0, b, q * n + i), U0[q * n + j], acc);
                    V0[i * n + j] = acc;
                }
            for (int i = 0; i < n; ++i)  // T = g_W V0
                for (int j = 0; j < n; ++j) {
                    double acc = 0;
                    for (int q = 0; q < n; ++q) acc = __builtin_fma(go[i * n + q], V0[q * n + j], acc);
                    T[i * n + j] = acc;
                }
            double s = 0;
            for (int a = 0; a < n; ++a)
                for (int c = 0; c < n; ++c) {
                    if (a == c) continue;
                    double gp = 0;  // G'[a,c] = sum_i U0[i,a] T[i,c]
                    for (int i = 0; i < n; ++i) gp = __builtin_fma(U0[i * n + a], T[i * n + c], gp);
                    const double num = U0[kk * n + a] * V0[l * n + c] - U0[kk * n + c] * V0[l * n + a];
                    s = __builtin_fma(gp, vec_clip_div(num, S0[a] + S0[c]), s);
                }
            if (o.flags & OP_FLAG_SVDW_FULL) {
                // dS/dM and dU/dM (tensor_svd.cpp:147-273): dS_i/dM_kl = U0[k,i] V0[l,i];
                // g_M[k,l] += sum_{a != j} (U0' g_U)[a,j] clip_div(U0[k,a] V0[l,j] s_j + U0[k,j] V0[l,a] s_a, s_j^2 - s_a^2)
                const double* gs = g + P.vars[o.out_s].grad;
                const double* gu = g + P.vars[o.out_u].grad;
                for (int i = 0; i < n; ++i) s = __builtin_fma(gs[i], U0[kk * n + i] * V0[l * n + i], s);
                for (int a = 0; a < n; ++a)
                    for (int j = 0; j < n; ++j) {
                        if (a == j) continue;
                        double gp = 0;
                        for (int i = 0; i < n; ++i) gp = __builtin_fma(U0[i * n + a], gu[i * n + j], gp);
                        const double num = U0[kk * n + a] * V0[l * n + j] * S0[j] + U0[kk * n + j] * V0[l * n + a] * S0[a];
                        s = __builtin_fma(gp, vec_clip_div(num, S0[j] * S0[j] - S0[a] * S0[a]), s);
                    }
            }
            add(x, e, s);
            break;
        }
        default: break;
    }
}

}  // namespace sanm_hip
