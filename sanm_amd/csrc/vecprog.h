// Vector graphs: Taylor propagation over batched VECTORS of arbitrary length with Slice / Concat.
//
// The reference's Slice and Concat operators (libsanm/oprs/misc.cpp:104-331) exist for graphs over (batch, n)
// tensors with elementwise arithmetic -- the Rosenbrock gradient of tests/symbolic.cpp:722-763 is the one user -- and
// no FEA graph contains them.  The per-tet machinery of program.h / tet_ops.h is built around 1, 3 or 9 doubles per
// tet; graphs with other sizes, or with Slice / Concat, are compiled into a VecProgram instead and run by the small
// interpreter below: one workgroup per batch item, one thread per vector element, the operators of the graph in
// sequence with a workgroup barrier between them.  Same pass structure as the tet programs (EVAL0 / GRAD / BIAS(k) /
// COEFF(k): TaylorCoeffProp::push_xi / ensure_jacobian / compute_next_order_bias, symbolic.cpp:162-289), same
// operator recurrences (elem_arith.cpp:42-217, analytic_unary.cpp:13-139, reduce.cpp:11-102).
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

constexpr int VEC_MAX_SIZE = 64;   // longest vector (threads of the workgroup)
constexpr int VEC_MAX_IN = 8;      // inputs of a concat / linear combination

struct VecVar {
    int64_t coef;   // arena offset of coefficient 0 of batch 0; order k, batch b at coef + (k * B + b) * size
    int64_t bias;   // cur_order_bias, [B][size]
    int32_t size;
    int32_t is_const;  // orders >= 1 are zero
    int32_t grad;      // offset of the variable's gradient row in the workgroup's scratch (GRAD pass)
    int32_t const_batch;  // CONSTANT: 1 = one row broadcast over the batch
};

struct VecOp {
    int32_t type, nin, flags, pad;
    int32_t in[VEC_MAX_IN];
    int32_t out;
    int32_t begin;            // SLICE: first element taken; CONCAT: unused
    double p[VEC_MAX_IN + 1]; // LINCOMB: coefficients, bias at p[VEC_MAX_IN]; POW: p[0] = exponent
    int64_t aux0, aux1;       // POW / LOG: K = f'(x0) [B][size], self-bias [B][size]; MULTIPLY: self-bias at aux1
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

// One operator, one element (thread) e of batch item b, forward passes.  Elements beyond the output's size do
// nothing; reductions (reduce_sum) are done by element 0.  xin: the placeholder's values of this order, [B][idim].
VEC_HD void vec_forward(const VecProgDev& P, const VecOp& o, int mode, int k, int64_t b, int e, const double* xin) {
    const VecVar& ov = P.vars[o.out];
    const int osz = ov.size;
    if (e >= osz) return;
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
                    // analytic_unary.cpp:112-131: the division recurrence cannot start from a zero; the reference
                    // continues integer exponents on a convolution path, this interpreter carries the square only
                    if (fabs(v) < 1e-3 && !P.vars[x].is_const) P.arena[P.flag + ((pw > 0.5 && floor(pw) == pw) ? 1 : 0)] = 1.0;
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
                        sb /= vec_coef(P, x, 0, b, e);
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
            if (e < P.vars[o.in[0]].size && !P.vars[o.in[0]].is_const) add(o.in[0], e, go[0]);
            break;
        case OP_SLICE:
            if (e < osz && !P.vars[o.in[0]].is_const) add(o.in[0], o.begin + e, go[e]);
            break;
        case OP_CONCAT: {
            int off = 0;
            for (int i = 0; i < o.nin; ++i) {
                const int n = P.vars[o.in[i]].size;
                if (e >= off && e < off + n && !P.vars[o.in[i]].is_const) add(o.in[i], e - off, go[e]);
                off += n;
            }
            break;
        }
        default: break;
    }
}

}  // namespace sanm_hip
