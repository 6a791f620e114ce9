/*
 * adapter/anm_hip.h -- the reference-side binding of libsanm_hip.so.
 *
 * A header a maintainer of jia-kai/SANM drops into the reference tree (as libsanm/anm_hip.h) to run the ANM inner
 * loop on an MI355X behind the reference's own C++ interface.  It includes only the reference's public headers
 * and include/sanm_hip.h, and defines
 *
 *   sanm::hip::export_graph(y)        symbolic::VarNode graph  -> sanm_graph      (walk over OperatorNode::meta(),
 *                                                                                   libsanm/symbolic.h:166-296)
 *   sanm::hip::export_desc(desc)      SparseLinearDesc         -> sanm_sparse_desc (libsanm/anm.h:24-73)
 *   sanm::hip::ANMEqnSolver           same public surface as sanm::ANMEqnSolver       (libsanm/anm.h:245-283)
 *   sanm::hip::ANMSolverVecScale      ... sanm::ANMSolverVecScale                     (libsanm/anm.h:209-243)
 *   sanm::hip::ANMImplicitSolver      ... sanm::ANMImplicitSolver                     (libsanm/anm.h:285-305)
 *
 * so that fea/main.cpp:418 (`ANMEqnSolver solver{model->y.node(), model->lt_inp, model->lt_out, x0, f_load_sub,
 * hyper_param}`), :393-399 and :516-520 keep their text with `hip::` in front of the class name.
 *
 * Nothing of the reference is copied here or shipped with this repository.  The file is compiled against the
 * reference's headers by tests/test_adapter.py (`g++ -std=c++20 -fsyntax-only -I/root/reference -Iinclude`)
 * wherever /root/reference exists.
 *
 * Operator parameters.  The reference keeps every operator's parameters in a private `Param` struct behind
 * OperatorNode::storage() (a void*).  The mirrors below restate those layouts (file:line beside each); a
 * maintainer who prefers accessors can replace each `mirror<...>(opr)` by a public `param()`.  The analytic
 * functions (log, pow) hide behind UnaryAnalyticTrait with file-local implementations
 * (libsanm/analytic_unary.cpp:13-139); the adapter recognises them by evaluating the trait at two points.
 */
#pragma once

#include <cmath>
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "libsanm/anm.h"
#include "libsanm/oprs/analytic_unary.h"
#include "libsanm/oprs/elem_arith.h"
#include "libsanm/oprs/linalg.h"
#include "libsanm/oprs/misc.h"
#include "libsanm/oprs/reduce.h"
#include "libsanm/symbolic.h"
#include "sanm_hip.h"

namespace sanm {
namespace hip {

//! error codes of the C ABI back into the reference's exception types (libsanm/utils.h:34-50)
inline void check(int rc) {
    if (rc == SANM_HIP_OK) return;
    std::string msg = sanm_hip_last_error();
    if (rc == SANM_HIP_ERR_NUMERICAL) throw SANMNumericalError{msg};
    if (rc == SANM_HIP_ERR_UNSUPPORTED) throw SANMError{"unsupported on the device path: " + msg};
    throw SANMAssertionError{msg};
}

namespace detail {
// ---- mirrors of the private Param structs ------------------------------------------------------------------
struct ConstantParam {      // libsanm/oprs/misc.h:41-43
    TensorND val;
};
struct LinearCombinationParam {  // libsanm/oprs/elem_arith.h:16-19
    std::vector<fp_t> coeffs;
    fp_t bias;
};
struct AnalyticUnaryParam {  // libsanm/oprs/analytic_unary.h:16-18
    UnaryAnalyticTraitPtr trait;
};
struct ReduceParam {         // libsanm/oprs/reduce.h:15-19
    symbolic::ReduceMode mode;
    int axis;
    bool keepdim;
};
struct BatchMatInvMulParam {  // libsanm/oprs/linalg.h:15-18
    bool use_identity;
    bool is_left;
};
struct BatchMulEyeParam {    // libsanm/oprs/linalg.h:161-163
    size_t dim;
};
struct BatchSVDWParam {      // libsanm/oprs/linalg.h:205-207
    bool require_rotation;
};
struct SliceParam {          // libsanm/oprs/misc.h:80-83
    int axis, stride;
    Maybe<int> begin, end;
};
struct ConcatParam {         // libsanm/oprs/misc.h:136-138
    int nr_input, axis;
};
template <class P>
const P& mirror(symbolic::OperatorNode* opr) {
    return *static_cast<const P*>(opr->storage());
}

struct GraphDeleter {
    void operator()(sanm_graph* g) const { sanm_graph_destroy(g); }
};
struct DescDeleter {
    void operator()(sanm_sparse_desc* d) const { sanm_sparse_desc_destroy(d); }
};
struct SolverDeleter {
    void operator()(sanm_anm_solver* s) const { sanm_anm_solver_destroy(s); }
};
struct TaylorDeleter {
    void operator()(sanm_taylor_prop* p) const { sanm_taylor_destroy(p); }
};

//! which analytic function a trait is: evaluates it at 2 and 3 (log: ln 2, ln 3; pow p: 2^p, 3^p)
inline bool classify_unary(const UnaryAnalyticTrait& trait, double* exponent) {
    TensorND x{TensorShape{2}};
    fp_t* px = x.woptr();
    px[0] = 2;
    px[1] = 3;
    TensorND y = trait.eval(x);
    const double y2 = y.ptr()[0], y3 = y.ptr()[1];
    if (std::fabs(y2 - std::log(2.0)) < 1e-12 && std::fabs(y3 - std::log(3.0)) < 1e-12) return true;  // log
    const double p = std::log(y2) / std::log(2.0);
    if (!(std::fabs(std::pow(3.0, p) - y3) <= 1e-9 * std::fabs(y3)))
        throw SANMError{"hip adapter: analytic unary function is neither log nor pow"};
    // exponents in the reference's graphs are small rationals (2, -2/3, ...): remove the round-off of the probe
    const double snapped = std::round(p * 720720.0) / 720720.0;
    *exponent = std::fabs(snapped - p) < 1e-12 ? snapped : p;
    return false;
}
}  // namespace detail

using GraphPtr = std::unique_ptr<sanm_graph, detail::GraphDeleter>;
using DescPtr = std::unique_ptr<sanm_sparse_desc, detail::DescDeleter>;

/*!
 * Replay the graph that produces \p y through the operator-construction entry points of the C ABI
 * (sanm_graph_* <-> libsanm/oprs.h:14-103), in topological order.  Returns the graph; *out_var is y's id in it.
 * placeholder_shape: the shape the placeholder will be fed with (remap_inp->out_shape() of a solver, the x of a
 * TaylorCoeffProp) -- the reference infers it at the first push, the device graph declares it.  null or (T,3,3): the
 * per-tet path; (batch, n): a vector (Slice / Concat allowed); (batch, rows, cols): a matrix of that size.
 */
inline GraphPtr export_graph(symbolic::VarNode* y, int* out_var, const TensorShape* placeholder_shape = nullptr) {
    using namespace symbolic;
    sanm_graph* raw = nullptr;
    check(sanm_graph_create(&raw));
    GraphPtr g{raw};
    std::unordered_map<VarNode*, int> id;
    auto in = [&](OperatorNode* opr, size_t i) { return id.at(opr->input(i)); };
    for (OperatorNode* opr : topo_sort({y})) {
        int out = -1;
        if (opr->isinstance<PlaceholderOprMeta>()) {
            // (the reference infers a placeholder's shape when it is fed; the device graph declares it: a (T,3,3)
            // matrix, or -- for graphs with Slice / Concat -- a (batch, n) vector of the given length)
            const TensorShape* ps = placeholder_shape;
            if (!ps || (ps->rank == 3 && ps->dim[1] == 3 && ps->dim[2] == 3)) check(sanm_graph_placeholder(g.get(), &out));
            else if (ps->rank == 3) check(sanm_graph_placeholder_matrix(g.get(), (int)ps->dim[1], (int)ps->dim[2], &out));
            else if (ps->rank <= 2) check(sanm_graph_placeholder_vector(g.get(), ps->rank == 2 ? (int)ps->dim[1] : 1, &out));
            else throw SANMError{"hip adapter: placeholder of rank above 3"};
        } else if (opr->isinstance<ConstantOprMeta>()) {
            const TensorND& v = detail::mirror<detail::ConstantParam>(opr).val;
            const TensorShape& s = v.shape();
            // scalar [1], batched scalar (b) / (b,1), or batched matrix (b,r,c): libsanm/tensor.h:20-30
            const int64_t batch = s.rank == 1 && s.dim[0] == 1 ? 1 : (int64_t)s.dim[0];
            const int size = (int)(s.total_nr_elems() / (size_t)batch);
            if (s.rank == 3 && !(s.dim[1] == 3 && s.dim[2] == 3))
                check(sanm_graph_constant_matrix(g.get(), v.ptr(), batch, (int)s.dim[1], (int)s.dim[2], &out));
            else
                check(sanm_graph_constant(g.get(), v.ptr(), batch, size, &out));
        } else if (opr->isinstance<LinearCombinationOprMeta>()) {
            const auto& p = detail::mirror<detail::LinearCombinationParam>(opr);
            std::vector<int> vars(opr->inputs().size());
            for (size_t i = 0; i < vars.size(); ++i) vars[i] = in(opr, i);
            check(sanm_graph_linear_combine(g.get(), (int)vars.size(), p.coeffs.data(), vars.data(), p.bias, &out));
        } else if (opr->isinstance<MultiplyOprMeta>()) {
            check(sanm_graph_multiply(g.get(), in(opr, 0), in(opr, 1), &out));
        } else if (opr->isinstance<AnalyticUnaryOprMeta>()) {
            double e = 0;
            if (detail::classify_unary(*detail::mirror<detail::AnalyticUnaryParam>(opr).trait, &e))
                check(sanm_graph_log(g.get(), in(opr, 0), &out));
            else
                check(sanm_graph_pow(g.get(), in(opr, 0), e, &out));
        } else if (opr->isinstance<ReduceOprMeta>()) {
            const auto& p = detail::mirror<detail::ReduceParam>(opr);
            if (p.mode != ReduceMode::SUM || !p.keepdim) throw SANMError{"hip adapter: reduce mode / keepdim"};
            check(sanm_graph_reduce_sum(g.get(), in(opr, 0), p.axis, &out));
        } else if (opr->isinstance<BatchMatMulOprMeta>()) {
            check(sanm_graph_batched_matmul(g.get(), in(opr, 0), in(opr, 1), &out));
        } else if (opr->isinstance<BatchMatInvMulOprMeta>()) {
            const auto& p = detail::mirror<detail::BatchMatInvMulParam>(opr);
            check(sanm_graph_batched_mat_inv_mul(g.get(), in(opr, 0), p.use_identity ? -1 : in(opr, 1),
                                                 p.is_left ? 1 : 0, &out));
        } else if (opr->isinstance<BatchDeterminantOprMeta>()) {
            check(sanm_graph_batched_det(g.get(), in(opr, 0), &out));
        } else if (opr->isinstance<BatchMatTransposeOprMeta>()) {
            check(sanm_graph_batched_transpose(g.get(), in(opr, 0), &out));
        } else if (opr->isinstance<BatchMulEyeOprMeta>()) {
            check(sanm_graph_batched_mul_eye(g.get(), in(opr, 0), (int)detail::mirror<detail::BatchMulEyeParam>(opr).dim,
                                             &out));
        } else if (opr->isinstance<BatchSVDWOprMeta>()) {
            int usw[3];
            check(sanm_graph_batched_svd_w(g.get(), in(opr, 0),
                                           detail::mirror<detail::BatchSVDWParam>(opr).require_rotation ? 1 : 0, usw));
            for (int i = 0; i < 3; ++i) id[opr->output(i)] = usw[i];
            continue;
        } else if (opr->isinstance<SliceOprMeta>()) {
            const auto& p = detail::mirror<detail::SliceParam>(opr);
            check(sanm_graph_slice(g.get(), in(opr, 0), p.axis, p.begin.valid() ? 1 : 0, p.begin.valid() ? p.begin.val() : 0,
                                   p.end.valid() ? 1 : 0, p.end.valid() ? p.end.val() : 0, p.stride, &out));
        } else if (opr->isinstance<ConcatOprMeta>()) {
            std::vector<int> vars(opr->inputs().size());
            for (size_t i = 0; i < vars.size(); ++i) vars[i] = in(opr, i);
            check(sanm_graph_concat(g.get(), (int)vars.size(), vars.data(), detail::mirror<detail::ConcatParam>(opr).axis,
                                    &out));
        } else {
            throw SANMError{std::string{"hip adapter: operator not on the device path: "} + opr->meta()->name()};
        }
        id[opr->output(0)] = out;
    }
    *out_var = id.at(y);
    return g;
}

/*!
 * SparseLinearDesc -> CSR by output element: row i is desc.get(i, 0) (libsanm/anm.h:46-61); what
 * SparseLinearDesc::apply walks (libsanm/anm.cpp:55-75).
 */
inline DescPtr export_desc(const SparseLinearDesc& desc, const double* out_coords = nullptr) {
    const size_t nout = desc.out_shape().total_nr_elems(), nin = desc.inp_shape().total_nr_elems();
    std::vector<uint64_t> rowptr(nout + 1, 0), idx;
    std::vector<double> coeff;
    for (size_t i = 0; i < nout; ++i) {
        for (const SparseLinearDesc::InputElem& e : desc.get(i, 0)) {
            idx.push_back(e.idx);
            coeff.push_back(e.coeff);
        }
        rowptr[i + 1] = idx.size();
    }
    sanm_sparse_desc* raw = nullptr;
    check(sanm_sparse_desc_create((int64_t)nout, (int64_t)nin, rowptr.data(), idx.data(), coeff.data(), &raw));
    DescPtr d{raw};
    // optional: positions of the unknowns (fea/mesh.h:94) for the nested dissection of the direct solver
    if (out_coords) check(sanm_sparse_desc_set_out_coords(d.get(), out_coords));
    return d;
}

namespace detail {
inline sanm_hyper_param make_hyper(const ANMDriverHelper::HyperParam& hp, bool eqn) {
    sanm_hyper_param p;
    sanm_hyper_param_default(&p, eqn ? 1 : 0);
    p.use_pade = hp.use_pade;
    p.sanity_check = hp.sanity_check;
    p.order = hp.order;
    p.maxr = hp.maxr;
    p.solution_check_tol = hp.solution_check_tol;
    p.xcoeff_l2_penalty = hp.xcoeff_l2_penalty;
    return p;
}

//! what the three drivers share (ANMDriverHelper's public part, libsanm/anm.h:116-139)
class DriverBase : public NonCopyable {
protected:
    std::unique_ptr<sanm_anm_solver, SolverDeleter> m_s;
    TensorShape m_x_shape;
    size_t m_n = 0;
    mutable TensorArray m_xt_coeffs;

    DriverBase(const TensorND& x0) : m_x_shape{x0.shape()}, m_n{x0.shape().total_nr_elems()} {}

public:
    void update_approx() { check(sanm_anm_update_approx(m_s.get())); }
    fp_t get_t_upper() const {
        double t;
        check(sanm_anm_get_t_upper(m_s.get(), &t));
        return t;
    }
    fp_t solve_a(fp_t t) const {
        double a;
        check(sanm_anm_solve_a(m_s.get(), t, &a));
        return a;
    }
    std::pair<TensorND, fp_t> eval(fp_t a) const {
        TensorND x{m_x_shape};
        double t;
        check(sanm_anm_eval(m_s.get(), a, x.woptr(), &t));
        return {x, t};
    }
    //! [x(a); t(a)] coefficients, n+1 entries each, fetched from the device on demand
    std::span<const TensorND> xt_coeffs() const {
        int nr;
        check(sanm_anm_nr_xt_coeffs(m_s.get(), &nr));
        m_xt_coeffs.resize(nr);
        for (int i = 0; i < nr; ++i) {
            m_xt_coeffs[i].set_shape({m_n + 1});
            check(sanm_anm_xt_coeff(m_s.get(), i, m_xt_coeffs[i].woptr()));
        }
        return m_xt_coeffs;
    }
    size_t get_nr_ieter() const {
        int64_t it;
        check(sanm_anm_nr_iter(m_s.get(), &it));
        return (size_t)it;
    }
};
}  // namespace detail

//! symbolic::TaylorCoeffProp for a single-input graph (libsanm/symbolic.h:337-383; what check_taylor_prop of
//! tests/symbolic.cpp:76-137 and ANMDriverHelper::solve_expansion_coeffs drive): push_xi / compute_next_order_bias
//! alternate; the Jacobian comes back as its dense blocks.  The placeholder's shape is given up front (the
//! reference learns it at the first push_xi).
class TaylorCoeffProp : public NonCopyable {
    GraphPtr m_g;
    DescPtr m_ident;
    std::unique_ptr<sanm_taylor_prop, detail::TaylorDeleter> m_p;
    size_t m_batch = 0, m_idim = 0;
    int m_odim = 0;
    TensorND m_out, m_bias;

public:
    TaylorCoeffProp(symbolic::VarNode* output, const TensorShape& x_shape, int max_order) {
        int out;
        m_g = export_graph(output, &out, &x_shape);
        const size_t n = x_shape.total_nr_elems();
        m_batch = x_shape.dim[0];
        m_idim = n / m_batch;
        std::vector<uint64_t> rowptr(n + 1), idx(n);
        std::vector<double> one(n, 1.0);
        for (size_t i = 0; i < n; ++i) rowptr[i] = idx[i] = i;
        rowptr[n] = n;
        sanm_sparse_desc* raw = nullptr;
        check(sanm_sparse_desc_create((int64_t)n, (int64_t)n, rowptr.data(), idx.data(), one.data(), &raw));
        m_ident.reset(raw);
        sanm_taylor_prop* p = nullptr;
        check(sanm_taylor_create(m_g.get(), out, m_ident.get(), max_order, &p));
        m_p.reset(p);
        check(sanm_taylor_output_size(p, &m_odim));
        m_out.set_shape({m_batch, (size_t)m_odim});
        m_bias.set_shape({m_batch, (size_t)m_odim});
    }
    //! the coefficient at the output node, (batch, elements of the output) -- reshape as needed
    const TensorND& push_xi(const TensorND& xi) {
        check(sanm_taylor_push_xi(m_p.get(), xi.ptr(), m_out.woptr()));
        return m_out;
    }
    const TensorND& compute_next_order_bias() {
        check(sanm_taylor_compute_next_order_bias(m_p.get(), m_bias.woptr()));
        return m_bias;
    }
    const TensorND& get_prev_next_order_bias() const { return m_bias; }
    //! d output / d input per batch item: (batch, output elements, input elements), what StSparseLinearTrans holds
    //! in its batched-full form (libsanm/sparse_linear_trans.h)
    TensorND get_jacobian_blocks() {
        TensorND j{TensorShape{m_batch, (size_t)m_odim, m_idim}};
        check(sanm_taylor_get_jacobian(m_p.get(), j.woptr()));
        return j;
    }
};

//! f(x) + t v = 0 (libsanm/anm.h:209-243; constructed at fea/main.cpp:393-399)
class ANMSolverVecScale final : public detail::DriverBase {
public:
    using HyperParam = ::sanm::ANMSolverVecScale::HyperParam;
    ANMSolverVecScale(symbolic::VarNode* f, SparseLinearDescPtr remap_inp, SparseLinearDescPtr remap_out, TensorND x0,
                      fp_t t0, TensorND v, const HyperParam& hyper_param = {}, const double* unknown_coords = nullptr)
            : DriverBase{x0} {
        int out;
        const TensorShape fed = remap_inp->out_shape();  // what the placeholder is fed with
        GraphPtr g = export_graph(f, &out, &fed);
        DescPtr in = export_desc(*remap_inp), ro = export_desc(*remap_out, unknown_coords);
        sanm_hyper_param p = detail::make_hyper(hyper_param, false);
        sanm_anm_solver* s = nullptr;
        check(sanm_anm_vecscale_solver_create(g.get(), out, in.get(), ro.get(), x0.ptr(), t0, v.ptr(), (int64_t)m_n, &p,
                                              &s));
        m_s.reset(s);
    }
};

//! f(x) + y = 0 (libsanm/anm.h:245-283; constructed at fea/main.cpp:418)
class ANMEqnSolver final : public detail::DriverBase {
public:
    using HyperParam = ::sanm::ANMEqnSolver::HyperParam;
    ANMEqnSolver(symbolic::VarNode* f, SparseLinearDescPtr remap_inp, SparseLinearDescPtr remap_out, TensorND x0,
                 TensorND y, const HyperParam& hyper_param = {}, const double* unknown_coords = nullptr)
            : DriverBase{x0} {
        int out;
        const TensorShape fed = remap_inp->out_shape();  // what the placeholder is fed with
        GraphPtr g = export_graph(f, &out, &fed);
        DescPtr in = export_desc(*remap_inp), ro = export_desc(*remap_out, unknown_coords);
        sanm_hyper_param p = detail::make_hyper(hyper_param, true);
        p.converge_rms = hyper_param.converge_rms;
        sanm_anm_solver* s = nullptr;
        check(sanm_anm_eqn_solver_create(g.get(), out, in.get(), ro.get(), x0.ptr(), y.ptr(), (int64_t)m_n, &p, &s));
        m_s.reset(s);
    }
    fp_t residual_rms() const {
        double r;
        check(sanm_anm_residual_rms(m_s.get(), &r));
        return r;
    }
    bool converged() const {
        int f;
        check(sanm_anm_converged(m_s.get(), &f));
        return f != 0;
    }
    ANMEqnSolver& next_iter() {
        check(sanm_anm_next_iter(m_s.get()));
        return *this;
    }
    TensorND get_x() const {
        TensorND x{m_x_shape};
        check(sanm_anm_get_x(m_s.get(), x.woptr()));
        return x;
    }
};

//! F(x, t) = F(x0, t0) (libsanm/anm.h:285-305; constructed at fea/main.cpp:516-520)
class ANMImplicitSolver final : public detail::DriverBase {
public:
    using HyperParam = ::sanm::ANMImplicitSolver::HyperParam;
    ANMImplicitSolver(symbolic::VarNode* f, SparseLinearDescPtr remap_inp, SparseLinearDescPtr remap_out,
                      const TensorND& x0, fp_t t0, const HyperParam& hyper_param = {},
                      const double* unknown_coords = nullptr)
            : DriverBase{x0} {
        int out;
        const TensorShape fed = remap_inp->out_shape();  // what the placeholder is fed with
        GraphPtr g = export_graph(f, &out, &fed);
        DescPtr in = export_desc(*remap_inp), ro = export_desc(*remap_out, unknown_coords);
        sanm_hyper_param p = detail::make_hyper(hyper_param, false);
        sanm_anm_solver* s = nullptr;
        check(sanm_anm_implicit_solver_create(g.get(), out, in.get(), ro.get(), x0.ptr(), t0, (int64_t)m_n, &p, &s));
        m_s.reset(s);
    }
};

}  // namespace hip
}  // namespace sanm
