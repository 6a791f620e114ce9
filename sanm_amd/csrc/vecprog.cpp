#include "vecprog_host.h"

#include <algorithm>
#include <cmath>

namespace sanm_hip {

namespace {
// operators `out_var` depends on, in creation (= topological) order
std::vector<int> needed_ops(const Graph& g, int out_var) {
    std::vector<char> need(g.ops.size(), 0);
    std::vector<int> stack{g.vars.at(out_var).producer};
    while (!stack.empty()) {
        const int oi = stack.back();
        stack.pop_back();
        if (need[oi]) continue;
        need[oi] = 1;
        for (int v : g.ops[oi].in) stack.push_back(g.vars[v].producer);
    }
    std::vector<int> r;
    for (size_t i = 0; i < g.ops.size(); ++i)
        if (need[i]) r.push_back((int)i);
    return r;
}
}  // namespace

bool graph_is_vector(const Graph& g, int out_var) {
    for (int oi : needed_ops(g, out_var)) {
        const GraphOp& op = g.ops[oi];
        if (op.type == OP_SLICE || op.type == OP_CONCAT) return true;
        if (op.type == OP_PLACEHOLDER && (op.flags & OP_FLAG_VECTOR)) return true;
        // the per-tet programs carry (T,3,3) matrices, batched scalars and the (T,3) singular values of SVD-W
        for (int v : op.out) {
            const GraphVar& gv = g.vars[v];
            const bool tet_shape = (gv.rows == 3 && gv.cols == 3) || (gv.rows == 1 && gv.cols == 0) ||
                                   (gv.rows == 3 && gv.cols == 0);
            if (!tet_shape) return true;
        }
    }
    return false;
}

VecProgram::VecProgram(Backend* be, const Graph& g, int out_var, int64_t B, int max_order) : m_be{be} {
    sanm_check(out_var >= 0 && out_var < (int)g.vars.size(), "invalid output var");
    sanm_check(B > 0 && max_order >= 0, "invalid batch / order");
    const std::vector<int> order = needed_ops(g, out_var);
    m_var_map.assign(g.vars.size(), -1);
    std::vector<VecOp> ops;
    int64_t off = 0;
    auto take = [&](int64_t n) {
        const int64_t r = off;
        off += n;
        return r;
    };
    int nr_placeholder = 0, grad = 0;
    m_dev.in_var = -1;
    // SVD-W operators whose U or S is read (or is the output): the full recurrences; W alone: the polar ones
    std::vector<char> svdw_full(g.ops.size(), 0);
    auto mark_full = [&](int v) {
        const GraphVar& gv = g.vars[v];
        if (g.ops[gv.producer].type == OP_SVDW && gv.out_idx < 2) svdw_full[gv.producer] = 1;
    };
    mark_full(out_var);
    for (int oi : order)
        for (int v : g.ops[oi].in) mark_full(v);
    const int64_t flag = take(2);
    for (int oi : order) {
        const GraphOp& op = g.ops[oi];
        VecOp o{};
        o.type = op.type;
        o.flags = op.flags;
        o.nin = op.in.size();
        sanm_check(o.nin <= VEC_MAX_IN, "vector graphs: at most %d inputs per operator", VEC_MAX_IN);
        switch (op.type) {
            case OP_PLACEHOLDER: case OP_CONSTANT: case OP_LINCOMB: case OP_MULTIPLY: case OP_LOG: case OP_POW:
            case OP_REDUCE_SUM: case OP_SLICE: case OP_CONCAT: case OP_MATMUL: case OP_MATINVMUL: case OP_DET:
            case OP_TRANSPOSE: case OP_MULEYE: case OP_SVDW: break;
            default:
                sanm_throw(SANM_ERR_UNSUPPORTED, "operator %d in a graph on the vector interpreter", (int)op.type);
        }
        bool all_const = op.type != OP_PLACEHOLDER;
        for (int i = 0; i < o.nin; ++i) {
            o.in[i] = m_var_map[op.in[i]];
            sanm_check(o.in[i] >= 0, "operand of operator %d is not computed", oi);
            all_const = all_const && m_vars[o.in[i]].is_const;
        }
        // (SVD-W: W is the output this interpreter carries; U and S stay unmapped)
        sanm_check(op.out.size() == 1 || op.type == OP_SVDW, "vector graphs: single-output operators only");
        const int gv = op.type == OP_SVDW ? op.out[2] : op.out[0], sz = g.vars[gv].size;
        sanm_check(sz >= 1 && sz <= VEC_MAX_SIZE, "vector of %d elements: at most %d", sz, VEC_MAX_SIZE);
        VecVar v{};
        v.size = sz;
        v.rows = g.vars[gv].rows;
        v.cols = g.vars[gv].cols;
        v.is_const = all_const ? 1 : 0;
        v.const_batch = 0;
        if (op.type == OP_CONSTANT) {
            sanm_check(op.batch == 1 || op.batch == B, "constant of batch %ld in a graph of batch %ld", (long)op.batch, (long)B);
            v.const_batch = op.batch == 1 ? 1 : 0;
            v.coef = take((op.batch == 1 ? 1 : B) * sz);
            v.bias = -1;
        } else {
            v.coef = take((int64_t)(v.is_const ? 1 : max_order + 1) * B * sz);
            v.bias = v.is_const ? -1 : take(B * sz);
        }
        v.grad = grad;
        grad += sz;
        const int lv = m_vars.size();
        m_vars.push_back(v);
        m_var_map[gv] = lv;
        o.out = lv;
        if (op.type == OP_PLACEHOLDER) {
            ++nr_placeholder;
            m_dev.in_var = lv;
            m_dev.idim = sz;
        } else if (op.type == OP_LINCOMB) {
            for (int i = 0; i < o.nin; ++i) o.p[i] = op.coeffs[i];
            o.p[VEC_MAX_IN] = op.bias;
        } else if (op.type == OP_POW || op.type == OP_LOG) {
            o.p[0] = op.exponent;
            if (op.type == OP_POW && op.exponent != 2.0) m_pow_exponents.push_back(op.exponent);
            o.aux0 = take(B * sz);
            o.aux1 = take(B * sz);
        } else if (op.type == OP_MULTIPLY) {
            o.aux1 = take(B * sz);
        } else if (op.type == OP_SLICE || op.type == OP_REDUCE_SUM) {
            o.begin = op.begin;  // (REDUCE_SUM: 0 = everything, 1 / 2 = one axis of a matrix)
        } else if (op.type == OP_MATMUL) {
            o.aux1 = take(B * sz);
        } else if (op.type == OP_MATINVMUL) {
            // three records: X0^-1 (order 0), the inner product, the outer product
            const int m = v.rows;
            sanm_check(m <= VEC_MAX_DIM, "mat_inv_mul of a %d x %d matrix: at most %d", m, m, VEC_MAX_DIM);
            o.aux0 = take(B * sz);
            o.aux1 = take(B * sz);
            o.aux2 = take(B * sz);
            VecOp prep = o;
            prep.type = VOP_INV_PREP;
            prep.nact = 1;
            ops.push_back(prep);
            ops.push_back(o);
            o.type = VOP_MATINV_FIN;
        } else if (op.type == OP_SVDW) {
            const int n = v.rows;
            sanm_check(n <= VEC_MAX_DIM, "SVD-W of a %d x %d matrix: at most %d", n, n, VEC_MAX_DIM);
            o.aux0 = take(B * sz);
            o.aux1 = take(B * n);
            o.nact = sz;
            if (svdw_full[oi]) {
                // U and S as variables of their own beside W; four records per order
                o.flags |= OP_FLAG_SVDW_FULL;
                for (int q = 0; q < 2; ++q) {
                    VecVar w{};
                    w.rows = n;
                    w.cols = q == 0 ? n : 0;
                    w.size = q == 0 ? sz : n;
                    w.is_const = v.is_const;
                    w.coef = take((int64_t)(w.is_const ? 1 : max_order + 1) * B * w.size);
                    w.bias = w.is_const ? -1 : take(B * w.size);
                    w.grad = grad;
                    grad += w.size;
                    m_var_map[op.out[q]] = m_vars.size();
                    (q == 0 ? o.out_u : o.out_s) = m_vars.size();
                    m_vars.push_back(w);
                }
                o.aux2 = take(B * 3 * sz);
                o.aux3 = take(B * 2 * (int64_t)(max_order + 1) * sz);
                ops.push_back(o);
                o.type = VOP_SVDWF_B;
                ops.push_back(o);
                o.type = VOP_SVDWF_C;
                ops.push_back(o);
                o.type = VOP_SVDWF_FIN;
                o.nact = 1;
            } else {
                o.aux2 = take(B * 2 * sz);
                o.aux3 = take((int64_t)(max_order + 1) * B * sz);
                ops.push_back(o);
                o.type = VOP_SVDW_FIN;
                o.nact = 1;
            }
        } else if (op.type == OP_DET) {
            const int m = m_vars[o.in[0]].rows;
            sanm_check(m <= VEC_MAX_DIM, "determinant of a %d x %d matrix: at most %d", m, m, VEC_MAX_DIM);
            sanm_check(max_order <= VEC_MAX_ORDER, "graphs with a determinant on the vector interpreter: order <= %d",
                       VEC_MAX_ORDER);
            o.aux0 = take(B * m * m);
            o.aux1 = take(B);
            o.aux2 = take(B * VEC_MAX_SIZE);
            o.nact = VEC_MAX_SIZE;
            ops.push_back(o);
            o.type = VOP_DET_FIN;
            o.nact = 1;
        }
        ops.push_back(o);
    }
    sanm_check(nr_placeholder == 1, "exactly one placeholder input is supported, got %d", nr_placeholder);
    m_out_graph_var = out_var;
    m_dev.out_var = m_var_map[out_var];
    sanm_check(!m_vars[m_dev.out_var].is_const, "the output does not depend on the input");
    m_dev.odim = m_vars[m_dev.out_var].size;
    m_dev.jac = take(B * m_dev.odim * m_dev.idim);
    m_dev.flag = flag;
    m_dev.B = B;
    m_dev.max_order = max_order;
    m_dev.nops = ops.size();
    m_dev.nvars = m_vars.size();
    m_dev.grad_total = grad;
    sanm_check(grad <= 8192, "vector graph too large for the gradient scratch (%d doubles of LDS; at most 8192)", grad);
    // arena, constants
    m_arena_doubles = off;
    std::vector<double> host(off, 0.0);
    for (int oi : order) {
        const GraphOp& op = g.ops[oi];
        if (op.type != OP_CONSTANT) continue;
        const VecVar& v = m_vars[m_var_map[op.out[0]]];
        std::copy(op.value.begin(), op.value.end(), host.begin() + v.coef);
    }
    m_dev.arena = static_cast<double*>(be->alloc(std::max<int64_t>(off, 1) * sizeof(double)));
    be->h2d(m_dev.arena, host.data(), off * sizeof(double));
    m_d_ops = be->alloc(ops.size() * sizeof(VecOp));
    be->h2d(m_d_ops, ops.data(), ops.size() * sizeof(VecOp));
    m_d_vars = be->alloc(m_vars.size() * sizeof(VecVar));
    be->h2d(m_d_vars, m_vars.data(), m_vars.size() * sizeof(VecVar));
    m_dev.ops = static_cast<const VecOp*>(m_d_ops);
    m_dev.vars = static_cast<const VecVar*>(m_d_vars);
}

VecProgram::~VecProgram() {
    m_be->free(m_dev.arena);
    m_be->free(m_d_ops);
    m_be->free(m_d_vars);
}

void VecProgram::download_var(int graph_var, int order, double* dst) const {
    sanm_check(graph_var >= 0 && graph_var < (int)m_var_map.size() && m_var_map[graph_var] >= 0,
               "var %d is not part of the compiled program", graph_var);
    const VecVar& v = m_vars[m_var_map[graph_var]];
    const int64_t B = m_dev.B;
    if (order < 0) {
        sanm_check(v.bias >= 0, "var %d has no bias (constant)", graph_var);
        m_be->d2h(dst, m_dev.arena + v.bias, B * v.size * 8);
        return;
    }
    sanm_check(order <= m_dev.max_order, "order %d out of range", order);
    if (v.is_const) {
        if (order > 0) {
            std::fill(dst, dst + B * v.size, 0.0);
            return;
        }
        if (v.const_batch == 1) {
            std::vector<double> row(v.size);
            m_be->d2h(row.data(), m_dev.arena + v.coef, v.size * 8);
            for (int64_t b = 0; b < B; ++b) std::copy(row.begin(), row.end(), dst + b * v.size);
            return;
        }
    }
    m_be->d2h(dst, m_dev.arena + v.coef + (int64_t)order * B * v.size, B * v.size * 8);
}

void VecProgram::download_jacobian(double* dst) const {
    m_be->d2h(dst, m_dev.arena + m_dev.jac, m_dev.B * m_dev.odim * m_dev.idim * 8);
}

void VecProgram::take_flags(double fl[2]) {
    m_be->d2h(fl, m_dev.arena + m_dev.flag, 16);
    if (fl[0] != 0 || fl[1] != 0) {
        const double zero[2] = {0, 0};
        m_be->h2d(m_dev.arena + m_dev.flag, zero, 16);
    }
}

}  // namespace sanm_hip
