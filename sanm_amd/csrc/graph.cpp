#include "graph.h"
#include "vecprog_host.h"
#include "host_parallel.h"

#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

namespace sanm_hip {

void sanm_throw(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw SanmError{code, buf};
}

// ------------------------------------------------------------------ Graph --
void Graph::chk(int v) const {
    sanm_check(v >= 0 && v < (int)vars.size(), "invalid var id %d", v);
}

int Graph::add(GraphOp op, std::initializer_list<Shape> out_shapes) {
    int oi = ops.size();
    int k = 0;
    for (Shape sh : out_shapes) {
        op.out.push_back(vars.size());
        GraphVar v{sh.rows * std::max(sh.cols, 1), oi, k++};
        v.rows = sh.rows;
        v.cols = sh.cols;
        v.soft9 = sh.soft9;
        vars.push_back(v);
    }
    int first = op.out[0];
    ops.push_back(std::move(op));
    return first;
}

int Graph::placeholder() {
    GraphOp op;
    op.type = OP_PLACEHOLDER;
    return add(std::move(op), {{3, 3}});
}

int Graph::placeholder_vector(int size) {
    sanm_check(size >= 1, "placeholder: bad size %d", size);
    GraphOp op;
    op.type = OP_PLACEHOLDER;
    op.flags = OP_FLAG_VECTOR;
    return add(std::move(op), {{size, 0}});
}

int Graph::placeholder_matrix(int rows, int cols) {
    sanm_check(rows >= 1 && cols >= 1, "placeholder: bad shape (%d, %d)", rows, cols);
    GraphOp op;
    op.type = OP_PLACEHOLDER;
    if (rows != 3 || cols != 3) op.flags = OP_FLAG_VECTOR;
    return add(std::move(op), {{rows, cols}});
}

int Graph::slice(int x, int axis, int has_begin, int begin, int has_end, int end, int stride) {
    chk(x);
    // misc.cpp:104-133 (abs_interval), :135-152: the reference implements axis 1 and stride 1
    sanm_check(axis >= 0 && stride != 0, "slice: bad axis / stride");
    if (axis != 1 || stride != 1)
        sanm_throw(SANM_ERR_UNSUPPORTED, "slice: axis %d stride %d unimplemented (as in the reference)", axis, stride);
    if (vars[x].is_matrix())
        sanm_throw(SANM_ERR_UNSUPPORTED, "slice of a (batch, %d, %d) matrix: (batch, n) operands only", vars[x].rows,
                   vars[x].cols);
    const int size = vars[x].size;
    int b = has_begin ? begin : 0, e = has_end ? end : size;
    if (has_begin && b < 0) b += size;
    if (has_end && e < 0) e += size;
    sanm_check(b < e && b >= 0 && e <= size, "slice: bad interval [%d, %d) of %d", b, e, size);
    GraphOp op;
    op.type = OP_SLICE;
    op.in = {x};
    op.begin = b;
    return add(std::move(op), {{e - b, 0}});
}

int Graph::concat(int n, const int* vs, int axis) {
    sanm_check(n >= 1 && axis >= 0, "concat: bad arguments");
    if (axis != 1) sanm_throw(SANM_ERR_UNSUPPORTED, "concat: axis %d unimplemented (as in the reference)", axis);
    GraphOp op;
    op.type = OP_CONCAT;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        chk(vs[i]);
        if (vars[vs[i]].is_matrix())
            sanm_throw(SANM_ERR_UNSUPPORTED, "concat of (batch, rows, cols) matrices: (batch, n) operands only");
        op.in.push_back(vs[i]);
        total += vars[vs[i]].size;
    }
    return add(std::move(op), {{total, 0}});
}

int Graph::constant(const double* val, int64_t batch, int size) {
    sanm_check(size >= 1, "constant: bad size %d", size);
    GraphOp op;
    op.type = OP_CONSTANT;
    op.batch = batch;
    op.value.assign(val, val + batch * size);
    // (nine values: the (T,3,3) constants of the FEA graphs)
    return add(std::move(op), {size == 9 ? Shape{3, 3, true} : Shape{size, 0}});
}

int Graph::constant_matrix(const double* val, int64_t batch, int rows, int cols) {
    sanm_check(rows >= 1 && cols >= 1, "constant: bad shape (%d, %d)", rows, cols);
    GraphOp op;
    op.type = OP_CONSTANT;
    op.batch = batch;
    op.value.assign(val, val + batch * rows * cols);
    return add(std::move(op), {{rows, cols}});
}

Graph::Shape Graph::elemwise_shape(Shape a, Shape b) const {
    // only batched scalars broadcast (oprs/elem_arith.cpp:13-38)
    const int sa = a.rows * std::max(a.cols, 1), sb = b.rows * std::max(b.cols, 1);
    sanm_check(sa == sb || sa == 1 || sb == 1, "invalid shape in elem arith: %d vs %d", sa, sb);
    if (sa == 1 && sb != 1) return b;
    if (sb == 1 && sa != 1) return a;
    // a shapeless nine-value constant follows a flat operand (GraphVar::soft9); two of them stay what they are
    if (a.soft9 && !b.soft9 && b.cols == 0) return b;
    if (b.soft9 && !a.soft9 && a.cols == 0) return a;
    if (a.soft9 != b.soft9) return a.soft9 ? b : a;
    // same element count: two matrices must agree in shape; a matrix and a flat operand keep the matrix's
    if (a.cols > 0 && b.cols > 0)
        sanm_check(a.rows == b.rows && a.cols == b.cols, "invalid shape in elem arith: (%d, %d) vs (%d, %d)", a.rows,
                   a.cols, b.rows, b.cols);
    return a.cols > 0 ? a : b;
}

int Graph::linear_combine(int n, const double* coeffs, const int* vs, double bias) {
    // (per-tet programs take up to MAX_OP_IN operands -- checked when one is compiled --, the vector interpreter 8)
    sanm_check(n >= 1 && n <= 8, "linear_combine: 1..8 inputs supported, got %d", n);
    GraphOp op;
    op.type = OP_LINCOMB;
    op.bias = bias;
    Shape osh{1, 0};
    for (int i = 0; i < n; ++i) {
        chk(vs[i]);
        op.in.push_back(vs[i]);
        op.coeffs.push_back(coeffs[i]);
        osh = i == 0 ? shape(vs[i]) : elemwise_shape(osh, shape(vs[i]));
    }
    return add(std::move(op), {osh});
}

int Graph::multiply(int a, int b) {
    chk(a);
    chk(b);
    const Shape osh = elemwise_shape(shape(a), shape(b));
    GraphOp op;
    op.type = OP_MULTIPLY;
    op.in = {a, b};
    return add(std::move(op), {osh});
}

int Graph::pow(int x, double e) {
    chk(x);
    if (e == 1.0) return x;  // oprs.cpp:30-32
    sanm_check(std::fabs(e) > 1e-9, "zero power not handled");
    GraphOp op;
    op.type = OP_POW;
    op.exponent = e;
    op.in = {x};
    return add(std::move(op), {shape(x)});
}

int Graph::log(int x) {
    chk(x);
    GraphOp op;
    op.type = OP_LOG;
    op.in = {x};
    return add(std::move(op), {shape(x)});
}

int Graph::reduce_sum(int x, int axis) {
    chk(x);
    sanm_check(axis != 0, "can not reduce on batch dim");
    // reduce.cpp:11-102 (keepdim): axis -1 flattens everything behind the batch axis -> (batch, 1); axis 1 of a
    // (batch, n) tensor is the same sum; one axis of a (batch, rows, cols) matrix gives (batch, 1, cols) /
    // (batch, rows, 1).  (axis -2, the sum over the batch as well, has no batched output: not on the device.)
    const bool mat = vars[x].is_matrix();
    if (!(axis == -1 || (axis == 1 && !mat) || (mat && (axis == 1 || axis == 2))))
        sanm_throw(SANM_ERR_UNSUPPORTED, "reduce_sum: axis %d of a (batch, %d, %d) tensor", axis, vars[x].rows,
                   vars[x].cols);
    GraphOp op;
    op.type = OP_REDUCE_SUM;
    op.in = {x};
    if (mat && axis != -1) {
        op.begin = axis;  // 1: over the rows, 2: over the columns
        return add(std::move(op), {axis == 1 ? Shape{1, vars[x].cols} : Shape{vars[x].rows, 1}});
    }
    return add(std::move(op), {{1, 0}});
}

int Graph::batched_matmul(int a, int b) {
    chk(a);
    chk(b);
    sanm_check(vars[a].is_matrix() && vars[b].is_matrix() && vars[a].cols == vars[b].rows,
               "invalid operand shapes for matmul: (%d, %d) x (%d, %d)", vars[a].rows, vars[a].cols, vars[b].rows,
               vars[b].cols);
    GraphOp op;
    op.type = OP_MATMUL;
    op.in = {a, b};
    return add(std::move(op), {{vars[a].rows, vars[b].cols}});
}

int Graph::batched_mat_inv_mul(int x, int a, bool is_left) {
    chk(x);
    sanm_check(vars[x].is_matrix() && vars[x].rows == vars[x].cols, "invalid shape for matinv: (%d, %d)", vars[x].rows,
               vars[x].cols);
    GraphOp op;
    op.type = OP_MATINVMUL;
    op.flags = is_left ? OP_FLAG_IS_LEFT : 0;
    op.in = {x};
    if (a >= 0) {
        chk(a);
        sanm_check(vars[a].rows == vars[x].rows && vars[a].cols == vars[x].cols, "invalid shape for matinv");
        op.in.push_back(a);
    } else {
        op.flags |= OP_FLAG_USE_IDENTITY;
    }
    return add(std::move(op), {shape(x)});
}

int Graph::batched_det(int x) {
    chk(x);
    // (tensor_polymat.cpp:354: dim >= 2)
    sanm_check(vars[x].is_matrix() && vars[x].rows == vars[x].cols && vars[x].rows >= 2,
               "invalid shape for determinant: (%d, %d)", vars[x].rows, vars[x].cols);
    GraphOp op;
    op.type = OP_DET;
    op.in = {x};
    return add(std::move(op), {{1, 0}});
}

int Graph::batched_transpose(int x) {
    chk(x);
    sanm_check(vars[x].is_matrix(), "invalid shape for transpose");
    GraphOp op;
    op.type = OP_TRANSPOSE;
    op.in = {x};
    return add(std::move(op), {{vars[x].cols, vars[x].rows}});
}

int Graph::batched_mul_eye(int x, int dim) {
    chk(x);
    sanm_check(vars[x].size == 1, "the input shape must be a scalar");
    sanm_check(dim >= 1, "bad dim %d", dim);
    GraphOp op;
    op.type = OP_MULEYE;
    op.in = {x};
    return add(std::move(op), {{dim, dim}});
}

void Graph::batched_svd_w(int x, bool require_rotation, int out[3]) {
    chk(x);
    sanm_check(vars[x].is_matrix() && vars[x].rows == vars[x].cols, "invalid shape for SVD-W: (%d, %d)", vars[x].rows,
               vars[x].cols);
    const int n = vars[x].rows;
    GraphOp op;
    op.type = OP_SVDW;
    op.flags = require_rotation ? OP_FLAG_REQUIRE_ROT : 0;
    op.in = {x};
    int first = add(std::move(op), {{n, n}, {n, 0}, {n, n}});
    out[0] = first;
    out[1] = first + 1;
    out[2] = first + 2;
}

// ---------------------------------------------------------------- Program --
Program::Program(Backend* be, const Graph& g, int out_var, int64_t T, int max_order,
                 int64_t tet_begin, int64_t T_global, bool full_history, const int64_t* tet_order)
        : m_be{be}, m_tet_begin{tet_begin} {
    if (T_global < 0) T_global = T;
    if (tet_order) m_tet_order.assign(tet_order + tet_begin, tet_order + tet_begin + T);
    sanm_check(out_var >= 0 && out_var < (int)g.vars.size(), "invalid output var");
    if (graph_is_vector(g, out_var))
        sanm_throw(SANM_ERR_UNSUPPORTED,
                   "a graph over vectors (Slice / Concat, sizes other than 1, 3, 9) runs on the vector interpreter: "
                   "operator-level API (sanm_taylor_*) only");
    sanm_check(T > 0 && max_order >= 1, "invalid T/order");
    const int64_t Tpad = (T + 63) / 64 * 64;
    const int N = max_order;

    // topological order of the operators the output depends on
    // (libsanm/symbolic.cpp:63-118)
    std::vector<int> topo;
    std::vector<char> seen(g.ops.size(), 0);
    std::function<void(int)> visit = [&](int oi) {
        if (seen[oi]) return;
        seen[oi] = 1;
        for (int v : g.ops[oi].in) visit(g.vars[v].producer);
        topo.push_back(oi);
    };
    visit(g.vars[out_var].producer);

    // local variable table, reader counts, constness
    m_var_map.assign(g.vars.size(), -1);
    std::vector<int> nr_reader;
    auto local = [&](int gv) {
        if (m_var_map[gv] < 0) {
            m_var_map[gv] = m_vars.size();
            VarDesc d{};
            d.size = g.vars[gv].size;
            d.jac = -1;
            m_vars.push_back(d);
            nr_reader.push_back(0);
        }
        return m_var_map[gv];
    };
    int nr_placeholder = 0;
    for (int oi : topo) {
        const GraphOp& op = g.ops[oi];
        for (int v : op.in) nr_reader[local(v)]++;
        bool all_const = op.type != OP_PLACEHOLDER;
        for (int v : op.in) all_const = all_const && m_vars[local(v)].is_const;
        for (int v : op.out) m_vars[local(v)].is_const = all_const ? 1 : 0;
        if (op.type == OP_PLACEHOLDER) {
            ++nr_placeholder;
            m_placeholder_var = local(op.out[0]);
        }
    }
    sanm_check(nr_placeholder == 1, "exactly one placeholder input is supported, got %d",
               nr_placeholder);
    const int lout = m_var_map[out_var];
    const int odim = m_vars[lout].size;
    // (3: the singular values of batched_svd_w as the output, for the operator-level API; the ANM drivers need the
    // 3x3 form and check it through their remap_out)
    sanm_check(odim == 9 || odim == 3, "the graph output must be a batched 3x3 matrix (or a batched 3-vector)");
    sanm_check(!m_vars[lout].is_const, "the output does not depend on the input");

    // arena layout
    int64_t off = 0;
    auto take = [&](int64_t n) {
        int64_t r = off;
        off += n;
        return r;
    };
    // Which series are history: an order-k coefficient goes back to HBM only if a convolution of a later order
    // reads it (both factors of a product, the argument and the result of log / pow / inverse, the argument of
    // det, M and W of the polar decomposition).  Everything else -- the placeholder, linear combinations,
    // transposes read by linear operators only, the output -- lives in the LDS scratch of its pass and dies
    // there: for the Neo-Hookean graph 29 of 65 doubles per tet and order are stored.
    for (auto& d : m_vars) d.hist = full_history && !d.is_const;
    for (int oi : topo) {
        const GraphOp& op = g.ops[oi];
        auto lv = [&](int i) -> VarDesc& { return m_vars[m_var_map[op.in[i]]]; };
        VarDesc& ov = m_vars[m_var_map[op.out[0]]];
        if (ov.is_const) continue;
        switch (op.type) {
            case OP_MULTIPLY: case OP_MATMUL:
                if (!lv(0).is_const && !lv(1).is_const) lv(0).hist = lv(1).hist = 1;
                break;
            case OP_LOG: case OP_POW: case OP_MATINVMUL:
                if (!lv(0).is_const) lv(0).hist = ov.hist = 1;
                break;
            case OP_DET:
                if (!lv(0).is_const) lv(0).hist = 1;
                break;
            case OP_SVDW: {
                if (lv(0).is_const) break;
                const int lu = m_var_map[op.out[0]], ls = m_var_map[op.out[1]], lw = m_var_map[op.out[2]];
                const bool full = nr_reader[lu] || nr_reader[ls] || lout == lu || lout == ls;
                if (full) m_vars[lu].hist = m_vars[ls].hist = m_vars[lw].hist = 1;  // the U, S, W recurrences
                else lv(0).hist = m_vars[lw].hist = 1;                               // polar mode: M and W
                break;
            }
            default:
                break;
        }
    }
    // a transpose whose input keeps its series is, for the fused convolution loop, a view of that series
    for (auto& d : m_vars) d.alias = -1, d.pad_ = 0;
    for (int oi : topo) {
        const GraphOp& op = g.ops[oi];
        if (op.type != OP_TRANSPOSE) continue;
        const int li = m_var_map[op.in[0]], lo = m_var_map[op.out[0]];
        if (!m_vars[lo].is_const && m_vars[li].hist && m_vars[li].alias < 0) m_vars[lo].alias = li;
    }
    for (size_t i = 0; i < m_vars.size(); ++i) {
        VarDesc& d = m_vars[i];
        d.coef = take((int64_t)(d.hist ? N + 1 : 1) * d.size * Tpad);
        d.bias = (full_history || (int)i == lout) ? take((int64_t)d.size * Tpad) : -1;
    }
    const int64_t out_aos = take(9 * Tpad);
    // Per-lane scratch slots of the current-order values.  A value lives from its producing
    // operator to its last reader, so slots are handed out by a linear scan over the
    // topological order and reused once dead: the scratch sits in LDS and its size bounds the
    // number of resident wavefronts.  An operator's outputs never share a slot with its inputs.
    int32_t cur_size = 0;
    {
        std::vector<int> last_use(m_vars.size(), -1);
        for (size_t pos = 0; pos < topo.size(); ++pos) {
            const GraphOp& op = g.ops[topo[pos]];
            for (int v : op.in) last_use[m_var_map[v]] = pos;
            for (int v : op.out) last_use[m_var_map[v]] = std::max(last_use[m_var_map[v]], (int)pos);
        }
        last_use[lout] = topo.size();  // the output is read by remap_out
        std::vector<std::pair<int32_t, int32_t>> free_list;  // (offset, size), sorted by offset
        auto alloc_slot = [&](int32_t size) {
            for (size_t i = 0; i < free_list.size(); ++i)
                if (free_list[i].second >= size) {
                    int32_t o = free_list[i].first;
                    free_list[i].first += size;
                    free_list[i].second -= size;
                    if (!free_list[i].second) free_list.erase(free_list.begin() + i);
                    return o;
                }
            int32_t o = cur_size;
            cur_size += size;
            return o;
        };
        auto free_slot = [&](int32_t o, int32_t size) {
            auto it = std::lower_bound(free_list.begin(), free_list.end(), std::make_pair(o, (int32_t)0));
            it = free_list.insert(it, {o, size});
            if (it + 1 != free_list.end() && it->first + it->second == (it + 1)->first) {
                it->second += (it + 1)->second;
                free_list.erase(it + 1);
            }
            if (it != free_list.begin() && (it - 1)->first + (it - 1)->second == it->first) {
                (it - 1)->second += it->second;
                free_list.erase(it);
            }
        };
        for (auto& d : m_vars) d.cur = 0;
        for (size_t pos = 0; pos < topo.size(); ++pos) {
            const GraphOp& op = g.ops[topo[pos]];
            for (int v : op.out) {
                VarDesc& d = m_vars[m_var_map[v]];
                if (!d.is_const) d.cur = alloc_slot(d.size);
            }
            auto release = [&](int v) {
                VarDesc& d = m_vars[m_var_map[v]];
                if (!d.is_const && last_use[m_var_map[v]] == (int)pos) {
                    free_slot(d.cur, d.size);
                    last_use[m_var_map[v]] = -2;  // an input listed twice is released once
                }
            };
            for (int v : op.in) release(v);
            for (int v : op.out) release(v);  // outputs nobody reads (U, S of SVD-W)
        }
        if (std::getenv("SANM_DEBUG"))
            std::fprintf(stderr, "program: %zu ops, %zu vars, %d scratch doubles per tet\n", topo.size(),
                         m_vars.size(), cur_size);
    }
    // only the placeholder's Jacobian d(out)/d(placeholder) lives in HBM (the assembly gathers from it);
    // the reverse sweep keeps the gradients of the intermediate variables in the LDS slots (tet_ops.h)
    m_jac_begin = off;
    m_vars[m_placeholder_var].jac = take((int64_t)odim * m_vars[m_placeholder_var].size * Tpad);
    m_jac_end = off;

    // last reader of every variable: in the reverse sweep of the GRAD pass it is the first operator to
    // accumulate into the variable's gradient slot and clears it first
    std::vector<int> last_reader(m_vars.size(), -1);
    for (size_t pos = 0; pos < topo.size(); ++pos)
        for (int v : g.ops[topo[pos]].in) last_reader[m_var_map[v]] = pos;
    int32_t conv_total = 0;
    for (size_t pos = 0; pos < topo.size(); ++pos) {
        const int oi = topo[pos];
        const GraphOp& op = g.ops[oi];
        OpDesc o{};
        o.type = op.type;
        o.flags = op.flags;
        o.nin = op.in.size();
        o.nout = op.out.size();
        for (int i = 0; i < o.nin; ++i) o.in[i] = m_var_map[op.in[i]];
        for (int i = 0; i < o.nout; ++i) o.out[i] = m_var_map[op.out[i]];
        o.grad_zero = 0;
        for (int i = 0; i < o.nin; ++i) {
            bool first = true;  // an input listed twice is cleared once
            for (int j = 0; j < i; ++j) first = first && o.in[j] != o.in[i];
            // (the graph output's slot holds the seed and is never cleared)
            if (first && last_reader[o.in[i]] == (int)pos && o.in[i] != lout && !m_vars[o.in[i]].is_const)
                o.grad_zero |= 1 << i;
        }
        for (int i = 0; i < 4; ++i) o.aux[i] = -1;
        const int osz = m_vars[o.out[0]].size;
        if (op.type == OP_LINCOMB || op.type == OP_MULTIPLY || op.type == OP_LOG || op.type == OP_POW ||
            op.type == OP_REDUCE_SUM) {
            // the device bodies of the elementwise operators exist for these sizes only (tet_ops.h)
            auto sized = [](int sz) { return sz == 1 || sz == 3 || sz == 9; };
            bool ok = sized(osz);
            for (int i = 0; i < o.nin; ++i) ok = ok && sized(m_vars[o.in[i]].size);
            if (!ok)
                sanm_throw(SANM_ERR_UNSUPPORTED,
                           "elementwise operators take 3x3 matrices, 3-vectors and batched scalars only");
        }
        switch (op.type) {
            case OP_LINCOMB:
                for (int i = 0; i < o.nin; ++i) o.p[i] = op.coeffs[i];
                o.p[MAX_OP_IN] = op.bias;
                break;
            case OP_MULTIPLY:
                o.aux[0] = take((int64_t)osz * Tpad);
                break;
            case OP_POW:
                o.p[0] = op.exponent;
                if (op.exponent != 2.0) {
                    // one double shared by all tets: set when an order-0 value is a zero (|x| < 1e-3) that the power
                    // recurrence cannot divide by (analytic_unary.cpp:43, :112-131; Program::pow_flags)
                    // (a whole [Tpad] plane: every region of the arena is a multiple of Tpad, Program::spec_source)
                    if (m_pow_flags.empty()) m_pow_flag_off = take(Tpad);
                    o.aux[2] = m_pow_flag_off;
                    m_pow_flags.push_back({o.aux[2], op.exponent});
                }
                [[fallthrough]];
            case OP_LOG:
                o.aux[0] = take((int64_t)osz * Tpad);
                o.aux[1] = take((int64_t)osz * Tpad);
                break;
            case OP_MATMUL:
                o.aux[0] = take(9 * Tpad);
                break;
            case OP_MATINVMUL:
                o.aux[0] = take(9 * Tpad);
                o.aux[1] = take(9 * Tpad);
                break;
            case OP_DET:
                o.aux[0] = take(9 * Tpad);
                o.aux[1] = take(Tpad);
                o.aux[2] = take((int64_t)(N + 1) * 3 * Tpad);
                o.aux[3] = take(3 * Tpad);
                break;
            case OP_SVDW: {
                // pw_mode (oprs/linalg.cpp:533) when U and S have no reader; the full recurrences otherwise
                const bool full = nr_reader[o.out[0]] || nr_reader[o.out[1]] || lout == o.out[0] || lout == o.out[1];
                if (nr_reader[o.out[0]] || lout == o.out[0]) o.flags |= OP_FLAG_SVDW_GU;
                if (nr_reader[o.out[1]] || lout == o.out[1]) o.flags |= OP_FLAG_SVDW_GS;
                if (nr_reader[o.out[2]] || lout == o.out[2]) o.flags |= OP_FLAG_SVDW_GW;
                if (full) {
                    o.flags |= OP_FLAG_SVDW_FULL;
                    o.aux[0] = take((int64_t)(N + 1) * 9 * Tpad);  // T0_i = sum_j U_j diag(S_{i-j})
                    o.aux[1] = take((int64_t)(N + 1) * 9 * Tpad);  // T1_i = sum_j T0_j U_{i-j}'
                    o.aux[2] = take(45 * Tpad);                    // Bu, Bw, Mbias_k, partial T0_k, partial T1_k
                } else {
                    o.aux[0] = take((int64_t)(N + 1) * 9 * Tpad);
                    o.aux[1] = take(9 * Tpad);
                    o.aux[2] = take(9 * Tpad);
                    o.aux[3] = take(9 * Tpad);
                }
                break;
            }
            default:
                break;
        }
        // the operator's share of the fused convolution loop (tet_ops.h, conv_term)
        o.conv_off = conv_total;
        o.conv_n = 0;
        if (!m_vars[o.out[0]].is_const) {
            auto var_in = [&](int i) { return !m_vars[o.in[i]].is_const; };
            switch (op.type) {
                case OP_MULTIPLY: if (var_in(0) && var_in(1)) o.conv_n = osz; break;
                case OP_LOG: case OP_POW: if (var_in(0)) o.conv_n = osz; break;
                case OP_MATMUL: if (var_in(0) && var_in(1)) o.conv_n = 9; break;
                case OP_MATINVMUL: if (var_in(0)) o.conv_n = 9; break;
                case OP_DET: if (var_in(0)) o.conv_n = 4; break;
                case OP_SVDW: if (var_in(0)) o.conv_n = (o.flags & OP_FLAG_SVDW_FULL) ? 45 : 27; break;
                default: break;
            }
        }
        conv_total += o.conv_n;
        m_ops.push_back(o);
    }
    m_arena_doubles = off;

    if (!be) {
        // layout only (no device): what Program::spec_source needs -- the build uses it to compile the pass kernels of
        // the fea models' graphs ahead of time (capi.cpp: sanm_fea_spec_source)
        m_dev.nops = m_ops.size();
        m_dev.out_var = lout;
        m_dev.odim = odim;
        m_dev.max_order = N;
        m_dev.cur_size = cur_size;
        m_dev.conv_total = conv_total;
        m_dev.T = T;
        m_dev.Tpad = Tpad;
        m_dev.out_aos = out_aos;
        m_dev.spec_id = -1;
        return;
    }
    // device buffers
    double* arena = static_cast<double*>(be->alloc(off * sizeof(double)));
    be->zero(arena, off * sizeof(double));
    // operator and variable records share one block, padded to whole groups of 8 cache lines: the pass kernel
    // pulls the block into the scalar cache with one batch of loads before it walks the operators
    const size_t ops_bytes = (m_ops.size() * sizeof(OpDesc) + 63) / 64 * 64;
    const size_t desc_bytes = (ops_bytes + m_vars.size() * sizeof(VarDesc) + 511) / 512 * 512;
    m_d_ops = be->alloc(desc_bytes);
    be->zero(m_d_ops, desc_bytes);
    m_d_vars = static_cast<char*>(m_d_ops) + ops_bytes;
    be->h2d(m_d_ops, m_ops.data(), m_ops.size() * sizeof(OpDesc));
    be->h2d(m_d_vars, m_vars.data(), m_vars.size() * sizeof(VarDesc));

    // constants: AoS (T,size) or (1,size) -> SoA coefficient 0
    std::vector<double> soa;
    for (int oi : topo) {
        const GraphOp& op = g.ops[oi];
        if (op.type != OP_CONSTANT) continue;
        const VarDesc& d = m_vars[m_var_map[op.out[0]]];
        sanm_check(op.batch == T_global || op.batch == 1,
                   "ConstantOprMeta shape mismatch in data parallel: tot_batch=%ld value_shape=%ld",
                   (long)T_global, (long)op.batch);
        soa.assign((size_t)d.size * Tpad, 0.0);
        parallel_ranges(T, 4096, [&](int64_t e0, int64_t e1, int) {
            for (int64_t e = e0; e < e1; ++e) {
                const int64_t src = op.batch == 1 ? 0 : (tet_order ? tet_order[tet_begin + e] : tet_begin + e);
                for (int c = 0; c < d.size; ++c) soa[c * Tpad + e] = op.value[src * d.size + c];
            }
        });
        // pad lanes replicate tet 0 so that padded lanes stay finite
        for (int64_t e = T; e < Tpad; ++e)
            for (int c = 0; c < d.size; ++c) soa[c * Tpad + e] = soa[c * Tpad];
        be->h2d(arena + d.coef, soa.data(), soa.size() * sizeof(double));
    }

    // coefficients and bias of the linear combinations as a table of their own, for the kernels compiled per graph
    // (which read them at run time: the material constants stay out of the generated source)
    {
        std::vector<double> lcp;
        for (const OpDesc& o : m_ops)
            if (o.type == OP_LINCOMB) lcp.insert(lcp.end(), o.p, o.p + MAX_OP_IN + 2);
        if (lcp.empty()) lcp.push_back(0.0);
        m_d_lc_params = be->alloc(lcp.size() * sizeof(double));
        be->h2d(m_d_lc_params, lcp.data(), lcp.size() * sizeof(double));
    }
    m_dev.lc_params = static_cast<const double*>(m_d_lc_params);
    m_dev.ops = static_cast<const OpDesc*>(m_d_ops);
    m_dev.vars = static_cast<const VarDesc*>(m_d_vars);
    m_dev.arena = arena;
    m_dev.nops = m_ops.size();
    m_dev.desc_lines = desc_bytes / 64;
    m_dev.out_var = lout;
    m_dev.odim = odim;
    m_dev.max_order = N;
    m_dev.cur_size = cur_size;
    m_dev.conv_total = conv_total;
    m_dev.T = T;
    m_dev.Tpad = Tpad;
    m_dev.out_aos = out_aos;
    m_dev.rin = {nullptr, nullptr, 0};
    m_dev.spec_id = -1;
    // Run-time specialisation: with the records as compile-time constants every branch on them folds, the element
    // loops unroll, the LDS scratch turns into registers and each pass becomes a few basic blocks whose loads the
    // compiler hoists.  It costs a compilation (about a second), so small batches -- the
    // tests -- stay on the interpreter kernels.
    const char* env_min = std::getenv("SANM_JIT_MIN_T");
    const int64_t jit_min_t = env_min ? std::atoll(env_min) : 2048;
    if (!std::getenv("SANM_NO_JIT") && T >= jit_min_t) {
        const auto t0 = std::chrono::steady_clock::now();
        m_dev.spec_id = be->specialize(spec_source().c_str());
        jit_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        jit_source = m_dev.spec_id >= 0 ? be->last_specialize_source() : 0;
    }
}

std::string Program::spec_source() const {
    std::string src;
    char buf[512];
    auto add = [&](const char* fmt, auto... a) {
        std::snprintf(buf, sizeof(buf), fmt, a...);
        src += buf;
    };
    // SANM_NO_CONV_FUSION: every operator walks the history itself, as in the interpreter kernels
    const int conv_total = std::getenv("SANM_NO_CONV_FUSION") ? 0 : m_dev.conv_total;
    // The source depends on the STRUCTURE of the graph and on the order only -- not on the number of tets (arena
    // offsets are written in units of Tpad, a run-time argument: tet_ops.h, SANM_SPEC_UNITS) and not on the material
    // (the linear combinations read their coefficients from Program's lc_params table) --, so one code object serves
    // every mesh and every material of an energy model: it is found in the caches (rtc.cpp), or among the code
    // objects built ahead of time for the fea models' own graphs (sanm_amd/build.py: rtc_embedded.inc).
    const int64_t U = m_dev.Tpad;
    auto units = [&](int64_t off) -> long long {
        if (off < 0) return -1;
        sanm_check(off % U == 0, "arena offset %lld is not a multiple of Tpad", (long long)off);
        return off / U;
    };
    add("#define SANM_CONV_MAX %d\n#define SANM_SPEC_UNITS 1\n", std::max(conv_total, 1));
    src += "#include \"tet_ops.h\"\n#include \"red_ops.h\"\nusing namespace sanm_hip;\nnamespace {\n";
    add("constexpr int kNops = %zu, kCurSize = %d, kOutVar = %d, kConvTotal = %d;\nconstexpr long long "
        "kOutAosUnits = %lld;\nconstexpr bool kNoOverlap = %s;\n",
        m_ops.size(), (int)m_dev.cur_size, (int)m_dev.out_var, conv_total, units(m_dev.out_aos),
        std::getenv("SANM_NO_COEFF_OVERLAP") ? "true" : "false");
    src += "#define SPEC_OPS { \\\n";
    int lc_base = 0;
    for (const OpDesc& o : m_ops) {
        add("  {%d, %d, %d, %d, {%d, %d, %d, %d}, {%d, %d, %d}, %d, {", o.type, o.nin, o.nout, o.flags, o.in[0], o.in[1],
            o.in[2], o.in[3], o.out[0], o.out[1], o.out[2], o.grad_zero);
        // (a linear combination's p[] comes from the table at run time; aux[3] = its first entry there)
        const bool lc = o.type == OP_LINCOMB;
        for (int i = 0; i < MAX_OP_IN + 2; ++i) add("%a%s", lc ? 0.0 : o.p[i], i + 1 < MAX_OP_IN + 2 ? ", " : "}, {");
        add("%lldLL, %lldLL, %lldLL, %lldLL}, %d, %d}, \\\n", units(o.aux[0]), units(o.aux[1]), units(o.aux[2]),
            lc ? (long long)lc_base : units(o.aux[3]), o.conv_off, o.conv_n);
        if (lc) lc_base += MAX_OP_IN + 2;
    }
    src += "}\n#define SPEC_VARS { \\\n";
    for (const VarDesc& d : m_vars)
        add("  {%lldLL, %lldLL, %lldLL, %d, %d, %d, %d, %d, 0}, \\\n", units(d.coef), units(d.bias), units(d.jac), d.size,
            d.is_const, d.cur, d.hist, d.alias);
    src += "}\n#define SPEC_CONV_TERMS(I, J) { \\\n";
    for (size_t i = 0; i < m_ops.size(); ++i)
        if (m_ops[i].conv_n) add("  conv_term(c, kOps[%zu], I, J, conv); \\\n", i);
    src += "}\n#define SPEC_CONV_REDUCE { \\\n";
    for (size_t i = 0; i < m_ops.size(); ++i)
        for (int e = 0; e < m_ops[i].conv_n; e += 9)  // chunks of the exchange buffer's 9 slots
            add("  conv_reduce(c, conv + %d, %d); \\\n", m_ops[i].conv_off + e, std::min(9, m_ops[i].conv_n - e));
    src += "}\n";
    src += R"SRC(
// AFTER_COEFF: BIAS(order) in the launch that ran COEFF(order - 1) on wavefront 0 (spec_pass4)
template <int MODE, bool AFTER_COEFF = false>
__device__ __forceinline__ void spec_body(const ProgramDev& P, int order, const double* __restrict__ xvec) {
    extern __shared__ double cur_lds[];
    static constexpr OpDesc kOps[] = SPEC_OPS;
    static constexpr VarDesc kVars[] = SPEC_VARS;
    const int lane = threadIdx.x & 63;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nparts = blockDim.x >> 6;
    const int64_t tet = (int64_t)blockIdx.x * 64 + lane;
    if (tet >= P.T) return;
    double* cur = cur_lds + lane;
    TetCtx c{P.arena, kVars, P.Tpad, tet, MODE == PASS_GRAD ? (int)blockIdx.y : order, kVars[kOutVar].size, cur, 64, kOutVar, part, nparts,
             cur_lds + (int64_t)kCurSize * 64 + lane};
    c.out = P.arena + kOutAosUnits * P.Tpad;
    c.max_order = P.max_order;
    c.params = P.lc_params;
    if (MODE == PASS_GRAD) {
        c.grow = blockIdx.y;
        for (int e = 0; e < kVars[kOutVar].size; ++e) cur[(int64_t)(kVars[kOutVar].cur + e) * 64] = (e == c.grow) ? 1.0 : 0.0;
)SRC";
    for (int i = (int)m_ops.size() - 1; i >= 0; --i) add("        exec_op(c, kOps[%d], MODE, P.rin, xvec);\n", i);
    src += R"SRC(    } else {
        // BIAS: the convolution sums of all operators in one loop over the history, both orientations of a pair
        // (i, k - i) in one body (tet_ops.h, conv_term); the middle term of an even order goes to the last wavefront
        if (MODE == PASS_BIAS && kConvTotal > 0) {
            double* const conv = c.conv;
            for (int e = 0; e < kConvTotal; ++e) conv[e] = 0.0;
            const int k = order, npair = (k - 1) / 2;
            const bool mid = k >= 2 && !(k & 1);
            if (AFTER_COEFF && nparts > 1) {
                // Only the pair (1, k - 1) (and the middle term of order 2) reads the coefficient wavefront 0 has
                // just computed: it takes that pair when it gets here, the helper wavefronts share the pairs
                // 2 .. npair -- history only -- and run them WHILE wavefront 0 is busy with COEFF(k - 1).
                if (part == 0) {
                    if (npair >= 1) {
                        SPEC_CONV_TERMS(1, k - 1)
                        SPEC_CONV_TERMS(k - 1, 1)
                    }
                    if (k == 2) SPEC_CONV_TERMS(1, 1)
                } else {
                    const int n2 = npair > 1 ? npair - 1 : 0, np = nparts - 1, q = part - 1;
                    const int lo = 2 + n2 * q / np, hi = 2 + n2 * (q + 1) / np;
                    for (int i = lo; i < hi; ++i) {
                        SPEC_CONV_TERMS(i, k - i)
                        SPEC_CONV_TERMS(k - i, i)
                    }
                    if (mid && k >= 4 && part == nparts - 1) SPEC_CONV_TERMS(k / 2, k / 2)
                }
            } else {
                const int lo = 1 + npair * part / nparts, hi = 1 + npair * (part + 1) / nparts;
                for (int i = lo; i < hi; ++i) {
                    SPEC_CONV_TERMS(i, k - i)
                    SPEC_CONV_TERMS(k - i, i)
                }
                if (mid && part == nparts - 1) SPEC_CONV_TERMS(k / 2, k / 2)
            }
            SPEC_CONV_REDUCE
            if (part) return;
            c.has_conv = true;
        }
)SRC";
    for (size_t i = 0; i < m_ops.size(); ++i) add("        exec_op(c, kOps[%zu], MODE, P.rin, xvec);\n", i);
    src += R"SRC(        if (MODE == PASS_EVAL0) {
            double Y[9];
            ld(p_coef(c, kOutVar, 0), P.Tpad, kVars[kOutVar].size, Y);
            st_out(c, kVars[kOutVar].size, Y);
        }
    }
}
}  // namespace
// (scalar arguments, pointers first: with -amdgpu-kernarg-preload-count they are in SGPRs when a wavefront starts)
#define SPEC_PARAMS                                                                                              \
    double *arena, const uint32_t *__restrict__ rin_idx, const double *__restrict__ rin_coef, const double *xvec, \
        const double *__restrict__ lc_params, long long T, long long Tpad, int order, int max_order, int rin_nslot
#define SPEC_P             \
    ProgramDev P{};        \
    P.arena = arena;       \
    P.lc_params = lc_params; \
    P.T = T;               \
    P.Tpad = Tpad;         \
    P.max_order = max_order; \
    P.rin = {rin_idx, rin_coef, rin_nslot};
#if !defined(SPEC_ONLY) || SPEC_ONLY == 0
extern "C" __global__ void __launch_bounds__(256, 1) spec_pass0(SPEC_PARAMS) {
    SPEC_P
    spec_body<PASS_EVAL0>(P, order, xvec);
}
#endif
#if !defined(SPEC_ONLY) || SPEC_ONLY == 1
extern "C" __global__ void __launch_bounds__(256, 1) spec_pass1(SPEC_PARAMS) {
    SPEC_P
    spec_body<PASS_GRAD>(P, order, xvec);
}
#endif
#if !defined(SPEC_ONLY) || SPEC_ONLY == 2
extern "C" __global__ void __launch_bounds__(256, 3) spec_pass2(SPEC_PARAMS) {
    SPEC_P
    spec_body<PASS_BIAS>(P, order, xvec);
}
#endif
#if !defined(SPEC_ONLY) || SPEC_ONLY == 4
// COEFF(order) by wavefront 0 of every workgroup, then BIAS(order + 1) by all of them (PASS_COEFF_BIAS).
// With nc_xg set the launch also stands for the order loop's next_coeff (Backend::run_pass_next_coeff): the gather of
// the placeholder forms x_order = -t xg - xvec on the way (t = *nc_num * nc_scale, a device scalar), and the
// workgroups behind the first `own` are riders: nc_blocks of them store x_order (and t, also to pinned memory), rd_nblk
// more run the scaling phase of a deferred Gram-Schmidt step -- what next_coeff_kernel and its rider did in a launch
// of their own.
extern "C" __global__ void __launch_bounds__(256, 3) spec_pass4(SPEC_PARAMS, const double* nc_xg, const double* nc_num,
                                                                double nc_scale, const double* nc_sc, double* nc_out,
                                                                double* nc_thost,
                                                                unsigned long long nc_n, double* rd_out,
                                                                const double* rd_norm2, double rd_eps,
                                                                unsigned long long rd_n, double* g_partials,
                                                                unsigned* g_ticket, double* g_host, unsigned own,
                                                                unsigned nc_blocks, unsigned rd_nblk) {
    // (the scale of an order 1 that stayed on the device: 1 / (t_1 - xg . x_1), Backend::x1_async)
    if (nc_xg && nc_sc) nc_scale = 1.0 / (nc_sc[0] - nc_sc[1]);
    if (blockIdx.x >= own) {
        const unsigned r = blockIdx.x - own;
        if (r < nc_blocks) {
            const double t = *nc_num * nc_scale;
            for (unsigned long long i = (unsigned long long)r * blockDim.x + threadIdx.x; i <= nc_n;
                 i += (unsigned long long)nc_blocks * blockDim.x) {
                if (i < nc_n) nc_out[i] = -t * nc_xg[i] - xvec[i];
                else {
                    nc_out[i] = t;
                    *nc_thost = t;
                }
            }
        } else {
            scale_rsqrt_body(rd_n, rd_out, rd_norm2, rd_eps, GridRed{g_partials, g_ticket, g_host}, r - nc_blocks, rd_nblk);
        }
        return;
    }
    SPEC_P
    if (nc_xg) {
        P.rin.xg = nc_xg;
        P.rin.t = *nc_num * nc_scale;
    }
    if ((threadIdx.x >> 6) == 0) spec_body<PASS_COEFF>(P, order, xvec);
    // the coefficients just stored are history for the convolutions
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (kConvTotal > 0 && !kNoOverlap) {
        // (read back by wavefront 0 alone: no barrier, the helper wavefronts are already in their history-only pairs)
        spec_body<PASS_BIAS, true>(P, order + 1, xvec);
    } else {
        __syncthreads();
        spec_body<PASS_BIAS>(P, order + 1, xvec);
    }
}
#endif
#if !defined(SPEC_ONLY) || SPEC_ONLY == 3
extern "C" __global__ void __launch_bounds__(256, 1) spec_pass3(SPEC_PARAMS) {
    SPEC_P
    spec_body<PASS_COEFF>(P, order, xvec);
}
#endif
)SRC";
    return src;
}

Program::~Program() {
    if (!m_be) return;
    if (m_dev.spec_id >= 0) m_be->release_specialized(m_dev.spec_id);
    m_be->free(m_dev.arena);
    m_be->free(m_d_ops);  // the variable records live in the same block
    if (m_d_lc_params) m_be->free(m_d_lc_params);
    if (m_d_rin_idx) m_be->free(m_d_rin_idx);
    if (m_d_rin_coef) m_be->free(m_d_rin_coef);
}

void Program::set_remap_in(int64_t n_in, const uint64_t* rowptr, const uint64_t* idx,
                           const double* coef) {
    set_remap_in(prepare_remap_in(n_in, rowptr, idx, coef));
}

Program::RemapInHost Program::prepare_remap_in(int64_t n_in, const uint64_t* rowptr, const uint64_t* idx,
                                               const double* coef) const {
    const int64_t T = m_dev.T, Tpad = m_dev.Tpad;
    int nslot = 0;
    const int64_t* ord = m_tet_order.empty() ? nullptr : m_tet_order.data();
    SetupLaps laps("remap_in table");
    {
        std::vector<int> part_max(64, 0);
        std::vector<char> part_bad(64, 0);
        parallel_ranges(T, 1 << 16, [&](int64_t e0, int64_t e1, int t) {
            int mx = 0;
            for (int64_t e = e0; e < e1; ++e) {
                const int64_t o0 = (ord ? ord[e] : m_tet_begin + e) * 9;
                for (int64_t o = o0; o < o0 + 9; ++o) {
                    if (rowptr[o + 1] < rowptr[o]) part_bad[t % 64] = 1;
                    else mx = std::max<int>(mx, (int)std::min<uint64_t>(rowptr[o + 1] - rowptr[o], 1u << 20));
                }
            }
            part_max[t % 64] = std::max(part_max[t % 64], mx);
        });
        for (char b : part_bad) sanm_check(!b, "remap_in: rowptr not monotone");
        for (int m : part_max) nslot = std::max(nslot, m);
    }
    sanm_check(nslot <= 64, "remap_in: %d entries for one output element", nslot);
    nslot = std::max(nslot, 1);
    laps.lap("slots");
    const size_t tab = (size_t)nslot * 9 * Tpad;
    auto hidx = raw_array<uint32_t>(tab);   // every entry written below: the workers touch their own pages
    auto hcoef = raw_array<double>(tab);
    std::vector<int64_t> bad(64, -1);
    parallel_ranges(Tpad, 4096, [&](int64_t e0, int64_t e1, int t) {
        for (int64_t e = e0; e < e1; ++e)
            for (int c = 0; c < 9; ++c) {
                int s = 0;
                if (e < T) {
                    const int64_t o = (ord ? ord[e] : m_tet_begin + e) * 9 + c;
                    for (uint64_t p = rowptr[o]; p < rowptr[o + 1]; ++p, ++s) {
                        const bool ok = (int64_t)idx[p] < n_in;
                        if (!ok) bad[t % 64] = (int64_t)idx[p];
                        hidx[((size_t)s * 9 + c) * Tpad + e] = ok ? (uint32_t)idx[p] : 0u;
                        hcoef[((size_t)s * 9 + c) * Tpad + e] = ok ? coef[p] : 0.0;
                    }
                }
                for (; s < nslot; ++s) {  // unused slots and the pad lanes: index 0, coefficient 0
                    hidx[((size_t)s * 9 + c) * Tpad + e] = 0;
                    hcoef[((size_t)s * 9 + c) * Tpad + e] = 0.0;
                }
            }
    });
    for (int64_t b : bad) sanm_check(b < 0, "remap_in: index %lu out of range", (unsigned long)b);
    // Coefficients that are all +1, -1 or (the empty slots') +0 -- the edge vectors x_j - x_0 of a tet mesh -- go into
    // the two top bits of their index words (program.h: RemapInDev): a third of the table's bytes in every pass, and
    // the kernels form the same products.  SANM_RIN_NO_PACK=1: the table with its coefficients as doubles.
    bool pack = n_in < (int64_t(1) << 30) && !std::getenv("SANM_RIN_NO_PACK");
    if (pack) {
        std::vector<char> ok(64, 1);
        parallel_ranges((int64_t)tab, 1 << 18, [&](int64_t q0, int64_t q1, int t) {
            for (int64_t q = q0; q < q1; ++q) {
                const double cf = hcoef[q];
                if (!(cf == 1.0 || cf == -1.0 || (cf == 0.0 && !std::signbit(cf)))) {
                    ok[t % 64] = 0;
                    return;
                }
            }
        });
        for (char c : ok) pack = pack && c;
    }
    if (pack)
        parallel_ranges((int64_t)tab, 1 << 18, [&](int64_t q0, int64_t q1, int) {
            for (int64_t q = q0; q < q1; ++q) {
                const double cf = hcoef[q];
                hidx[q] |= cf == 1.0 ? 0u : (cf == -1.0 ? 2u << 30 : 3u << 30);
            }
        });
    laps.lap("fill");
    RemapInHost out;
    out.idx = std::move(hidx);
    out.coef = std::move(hcoef);
    out.tab = tab;
    out.nslot = nslot;
    out.packed = pack;
    out.n_in = n_in;
    return out;
}

void Program::set_remap_in(RemapInHost&& t) {
    SetupLaps laps("remap_in table");
    if (m_d_rin_idx) m_be->free(m_d_rin_idx);
    if (m_d_rin_coef) m_be->free(m_d_rin_coef);
    m_d_rin_coef = nullptr;
    m_d_rin_idx = m_be->alloc(t.tab * sizeof(uint32_t));
    m_be->h2d(m_d_rin_idx, t.idx.get(), t.tab * sizeof(uint32_t));
    if (!t.packed) {
        m_d_rin_coef = m_be->alloc(t.tab * sizeof(double));
        m_be->h2d(m_d_rin_coef, t.coef.get(), t.tab * sizeof(double));
    }
    laps.lap("upload");
    m_dev.rin = {static_cast<const uint32_t*>(m_d_rin_idx),
                 static_cast<const double*>(m_d_rin_coef), t.nslot};
    m_n_in = t.n_in;
}

void Program::download_var(int graph_var, int order, double* dst) const {
    sanm_check(graph_var >= 0 && graph_var < (int)m_var_map.size() && m_var_map[graph_var] >= 0,
               "var %d is not part of the compiled program", graph_var);
    const VarDesc& d = m_vars[m_var_map[graph_var]];
    const int64_t T = m_dev.T, Tpad = m_dev.Tpad;
    std::vector<double> soa((size_t)d.size * Tpad);
    if (order >= 1 && d.is_const) {
        std::fill(dst, dst + T * d.size, 0.0);
        return;
    }
    sanm_check(order <= m_dev.max_order, "order %d out of range", order);
    sanm_check(order < 1 || d.hist, "the series of var %d is not kept by this program", graph_var);
    sanm_check(order >= 0 || d.bias >= 0, "the bias of var %d is not kept by this program", graph_var);
    int64_t off = order < 0 ? d.bias : d.coef + (int64_t)order * d.size * Tpad;
    m_be->d2h(soa.data(), m_dev.arena + off, soa.size() * sizeof(double));
    for (int64_t e = 0; e < T; ++e)
        for (int c = 0; c < d.size; ++c) dst[e * d.size + c] = soa[c * Tpad + e];
}

void Program::download_jacobian(double* dst) const {
    const VarDesc& d = m_vars[m_placeholder_var];
    const int64_t T = m_dev.T, Tpad = m_dev.Tpad;
    const int odim = m_dev.odim;
    (void)Tpad;
    m_be->d2h(dst, m_dev.arena + d.jac, (size_t)T * odim * d.size * sizeof(double));  // tet-major already
}

}  // namespace sanm_hip
