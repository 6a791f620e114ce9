// Compile check of adapter/anm_hip.h against the reference's own headers (tests/test_adapter.py): the three call
// sites of fea/main.cpp that construct ANM drivers, with `hip::` in front of the class name and otherwise the
// argument lists of the reference (fea/main.cpp:393-399, :418, :516-520), and the loops that consume them
// (run_anm, fea/main.cpp:172-215).  Only ever compiled with -fsyntax-only; nothing here is linked or shipped.
#include "anm_hip.h"

using namespace sanm;

namespace {
struct ModelLike {  // what fea's ElasticForceModel offers the solvers (fea/mesh.h:149-226)
    symbolic::VarNode* y;
    SparseLinearDescPtr lt_inp, lt_out;
};
}  // namespace

TensorND callsite_eqn_solver(const ModelLike& model, const TensorND& x0, const TensorND& f_load_sub,
                             const ANMEqnSolver::HyperParam& hyper_param, const double* vertex_loc) {
    hip::ANMEqnSolver solver{model.y, model.lt_inp, model.lt_out, x0, f_load_sub, hyper_param, vertex_loc};
    while (!solver.converged()) {  // run_anm, fea/main.cpp:172-190
        solver.next_iter();
        (void)solver.residual_rms();
    }
    (void)solver.get_nr_ieter();
    return solver.get_x();
}

TensorND callsite_implicit_solver(const ModelLike& model, const TensorND& x0,
                                  const ANMImplicitSolver::HyperParam& hyper_param) {
    hip::ANMImplicitSolver solver{model.y, model.lt_inp, model.lt_out, x0, 0, hyper_param};
    while (solver.get_t_upper() < 1) solver.update_approx();  // run_anm, fea/main.cpp:193-215
    (void)solver.xt_coeffs();
    return solver.eval(solver.solve_a(1)).first;
}

TensorND callsite_vecscale_solver(const ModelLike& model, const TensorND& x0, const TensorND& v,
                                  const ANMSolverVecScale::HyperParam& hyper_param) {
    hip::ANMSolverVecScale solver{model.y, model.lt_inp, model.lt_out, x0, 0, v, hyper_param};
    solver.update_approx();
    return solver.eval(solver.solve_a(solver.get_t_upper())).first;
}

int callsite_graph_export(symbolic::VarNode* y, const SparseLinearDesc& desc) {
    int out = -1;
    hip::GraphPtr g = hip::export_graph(y, &out);
    hip::DescPtr d = hip::export_desc(desc);
    return out;
}

// check_taylor_prop of tests/symbolic.cpp:76-137 with the device-side propagator: x_i pushed one by one, the
// coefficient compared with bias + Jacobian . x_i
TensorND callsite_taylor_prop(symbolic::VarNode* y, const TensorArray& xarr) {
    hip::TaylorCoeffProp tprop{y, xarr[0].shape(), (int)xarr.size() - 1};
    TensorND y0 = tprop.push_xi(xarr[0]);
    TensorND jac = tprop.get_jacobian_blocks();  // (batch, odim, idim)
    for (size_t i = 1; i < xarr.size(); ++i) {
        const TensorND& bi = tprop.compute_next_order_bias();
        const TensorND& yi = tprop.push_xi(xarr[i]);
        (void)bi;
        (void)yi;
    }
    return jac;
}
