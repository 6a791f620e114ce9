"""The checks the reference makes order by order (libsanm/anm.cpp:271-285 sanity check, sparse_solver.cpp:160-161
finite right-hand side, :288-289 finite coefficients, PARDISO's pivot handling :107-127) are batched on the device
path: queued without waiting and examined after the order loop (or at its first synchronisation).  Each of them is
made to fire here by corrupting one entry through the test hook sanm_anm_debug_inject."""
import numpy as np
import pytest

from sanm_amd import fea as dfea
from sanm_amd.api import SanmAssertionError, SanmNumericalError

CFG = {"material": {"young": 3e3, "poisson": 0.45, "density": 1000.0}, "g": [0, -9.81, 0],
       "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_c", "order": 12}


def _run(api):
    run = dfea.GravityRun(api, dfea.make_cuboid(6, 3, 3, 0.025), dict(CFG), solver_rtol=1e-15)
    run.construct()
    assert not run.solver.converged()
    return run


def test_corrupted_coefficient_fails_the_sanity_check_of_its_order(api):
    run = _run(api)
    # x_5 no longer solves A x_5 = -(t_5 g_t + b_5)
    run.solver.debug_inject(1, 5, 7, 1.5, scale=True)
    with pytest.raises(SanmAssertionError, match=r"ANM check coeff eqn: order 5"):
        run.solver.next_iter()
    # the hook is consumed: the solver is usable again from a fresh start
    run.solver.restart(run.model.x0())
    while not run.solver.converged():
        run.solver.next_iter()
    assert run.solver.residual_rms() < 1e-10


def test_first_coefficient_off_the_unit_sphere_fails_the_xdot_check(api):
    run = _run(api)
    n = run.model.n
    run.solver.debug_inject(1, 1, n, 3.0, scale=True)  # t_1 (entry n of x_1): x_1 . x_1 != 1
    with pytest.raises(SanmAssertionError, match=r"xdot|ANM check"):
        run.solver.next_iter()


def test_nonfinite_right_hand_side_is_reported_with_its_order(api):
    run = _run(api)
    run.solver.debug_inject(2, 3, 0, float("nan"))
    with pytest.raises(SanmNumericalError, match=r"non-finite right-hand side / solution at order 3"):
        run.solver.next_iter()


def test_nonfinite_jacobian_coefficient_is_reported(api):
    run = _run(api)
    run.solver.debug_inject(3, 0, 10, float("inf"))
    with pytest.raises(SanmAssertionError, match=r"non-finite Jacobian coefficient"):
        run.solver.next_iter()


def test_singular_jacobian_perturbs_pivots_and_refinement_reports_it(api):
    """A Jacobian whose first row is wiped out: a zero pivot, which the factorisation perturbs (PARDISO's static
    pivoting, libsanm/sparse_solver.cpp:107-127) instead of dividing by it; the iterative refinement that a
    perturbed factorisation switches on cannot make the residual small and reports the matrix as singular."""
    run = _run(api)
    A = run.solver.jacobian_csr()
    run.solver.debug_inject(3, int(A.indptr[1] - A.indptr[0]), int(A.indptr[0]), 0.0)
    with pytest.raises(SanmNumericalError, match=r"pivots were perturbed.*numerically singular"):
        run.solver.next_iter()


def test_refinement_steps_on_request_leave_the_solution_in_place(api):
    """solver_refine = 1: every solve is followed by one step of iterative refinement with a double-double
    residual; same continuation steps, same equilibrium (the corrections are of the order of 1e-11)."""
    for pade_on in (False, True):
        cfg = dict(CFG, disable_pade=not pade_on)
        ref = dfea.GravityRun(api, dfea.make_cuboid(6, 3, 3, 0.025), dict(cfg), solver_rtol=1e-15).run()
        run = dfea.GravityRun(api, dfea.make_cuboid(6, 3, 3, 0.025), dict(cfg), solver_rtol=1e-15, solver_refine=1).run()
        if not pade_on:  # (with Pade a 1e-11 change of the series may flip an ill-conditioned decision: tests/lockstep.py)
            assert run.solver.get_nr_iter() == ref.solver.get_nr_iter()
        assert np.abs(run.vertices() - ref.vertices()).max() < 1e-9 * np.abs(ref.vertices()).max()
        assert run.rms[-1] < 1e-10


def test_nan_jacobian_of_a_vector_graph_is_reported_not_a_memory_fault(api):
    """Graphs on the vector interpreter solve their small general systems by a dense LU with partial pivoting
    (dense_lu_kernel), queued BEFORE the host has looked at the finiteness count of the Jacobian.  A column of NaNs
    alone used to leave the pivot search without a winner (row index INT_MAX) and the row exchange then wrote far out
    of bounds -- a GPU memory fault instead of the reference's error (sparse_solver.cpp:288-289).  One batch item with
    a dense 6 x 6 block, every coefficient replaced by NaN."""
    import scipy.sparse as sp
    from sanm_amd import api as A
    x0 = np.random.default_rng(5).uniform(1, 2, (1, 6))
    g = api.graph()
    x = g.placeholder_vector(6)
    y = x.pow(2.3) * x.reduce_sum(-1)
    y0 = x0 ** 2.3 * x0.sum(axis=1, keepdims=True)
    eye = sp.identity(6, format="csr")
    hp = api.default_hyper(order=6, use_pade=0)
    sol = A.ANMSolverVecScale(api, y, A.SparseLinearDesc(api, eye), A.SparseLinearDesc(api, eye), x0.ravel(), 1.0,
                              -y0.ravel(), hp)
    sol.debug_inject(3, 36, 0, float("nan"))
    with pytest.raises(SanmAssertionError, match=r"non-finite Jacobian coefficient"):
        sol.update_approx()
    # one NaN among finite coefficients: the pivot search passes it over, the error is the same
    sol2 = A.ANMSolverVecScale(api, y, A.SparseLinearDesc(api, eye), A.SparseLinearDesc(api, eye), x0.ravel(), 1.0,
                               -y0.ravel(), hp)
    sol2.debug_inject(3, 1, 7, float("nan"))
    with pytest.raises(SanmAssertionError, match=r"non-finite Jacobian coefficient"):
        sol2.update_approx()
