"""Solver-level parity of the device path against the oracle and the golden
fixtures: final vertex positions within 1e-6 relative (north_star tolerance),
Jacobian CSR, and the continuation compared STEP BY STEP from common states
(tests/lockstep.py): residuals, series, ranges, and every Pade decision either
identical or certified ill-conditioned -- step counts equal wherever no such
event occurs (always without Pade).

Runs on the HIP library with -m gpu and on the test-only host harness otherwise.
The linear solver tolerance is tightened to 1e-15 here: the discrete Pade /
range decisions are compared step by step and need solves as accurate as the
oracle's LU (see DESIGN.md "Linear solve").
"""
import json
import os

import numpy as np
import pytest

from oracle import fea as ofea
from oracle import symbolic as S
from oracle.anm import build_jacobian_csr
from sanm_amd import api as A
from sanm_amd import fea as dfea
from tests.lockstep import LockStep

GOLD = os.path.join(os.path.dirname(__file__), "golden")
VTX_RTOL = 1e-6  # BASELINE.json north_star: relative vertex-position tolerance


def _run_device(api, dims, spacing, cfg, **over):
    mesh = dfea.make_cuboid(*dims, spacing)
    run = dfea.GravityRun(api, mesh, dict(cfg), solver_rtol=1e-15, **over)
    run.run()
    return run


def _no_pade(cfg):
    return dict(cfg, disable_pade=True)


def _same_continuation(run, ref, pade_on, xtol=1e-9):
    """two device runs of one task that differ in something that must not matter (tet order, kernel flavour, row
    scaling ...): same equilibrium always; same step count without Pade -- with it the summation-order differences
    between the two can flip an ill-conditioned Pade decision (tests/lockstep.py) and with it the count."""
    assert run.solver.converged() and ref.solver.converged()
    x, xr = run.solver.get_x(), ref.solver.get_x()
    assert np.abs(x - xr).max() <= xtol * np.abs(xr).max()
    if not pade_on:
        assert run.solver.get_nr_iter() == ref.solver.get_nr_iter()
        assert np.allclose(run.rms[:-1], ref.rms[:-1], rtol=1e-6)


@pytest.mark.parametrize("name", ["cuboid_nc", "cuboid_ni", "cuboid_arap", "cuboid_nc_nopade_o8", "cuboid_nc_l2"])
def test_gravity_cuboid_vs_golden_and_oracle(api, name):
    gold = json.load(open(os.path.join(GOLD, f"anm_{name}.json")))
    run = dfea.GravityRun(api, dfea.make_cuboid(*gold["dims"], gold["spacing"]), dict(gold["config"]),
                          solver_rtol=1e-15, profile=1).construct()
    _, osolver, _ = ofea.make_gravity_solver(ofea.make_cuboid(*gold["dims"], gold["spacing"]), gold["config"])
    # every step from a common state: rms, series, a_bound, restart point; Pade outcomes equal or certified
    ls = LockStep(run, osolver).run_to_convergence()
    print(name, "steps", ls.nr_steps, "free-running oracle", gold["iter"], "events",
          [(e["step"], e["device"], e["oracle"]) for e in ls.events])
    if gold["config"].get("disable_pade"):
        assert not ls.events
    if not ls.events:
        assert ls.nr_steps == gold["iter"], "continuation-step count differs"
        assert np.allclose(run.rms[:-1], gold["residual_rms"][:-1], rtol=1e-5)
    # the equilibrium of the free-running oracle (the golden file)
    V = run.vertices()
    Vg = np.array(gold["vertices"])
    assert np.abs(V - Vg).max() <= VTX_RTOL * np.abs(Vg).max()
    assert run.rms[-1] < 1e-10
    assert run.model.n == gold["nr_unknown"]


def test_first_step_coefficients_and_jacobian(api):
    """per-order t_k, |x_k| of the first ANM step and the assembled Jacobian
    against the live oracle (same inputs, no restart error yet)."""
    cfg = {"material": {"young": 3e3, "poisson": 0.45, "density": 900.0}, "g": [0, -9.81, 0],
           "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_i",
           "order": 12}
    dims, sp = (5, 3, 3), 0.03
    omesh = ofea.make_cuboid(*dims, sp)
    omodel, osolver, of = ofea.make_gravity_solver(omesh, cfg)
    run = dfea.GravityRun(api, dfea.make_cuboid(*dims, sp), dict(cfg), solver_rtol=1e-15, profile=1).construct()
    tr = run.solver.trace()
    assert np.allclose(tr["t"], osolver.trace[0]["t"], rtol=1e-8)
    assert np.allclose(tr["x_norm"], osolver.trace[0]["x_norm"], rtol=1e-8)
    assert np.allclose(tr["b_norm"], osolver.trace[0]["b_norm"], rtol=1e-7, atol=1e-12)
    # range estimate of the first expansion: a_bound, Pade outcome (identical or certified ill-conditioned)
    LockStep(run, osolver)
    xc = run.solver.xt_coeffs()
    for i in (1, 2, 6, 12):
        assert np.allclose(xc[i], osolver.xt_coeffs[i], rtol=1e-7, atol=1e-9 * np.abs(osolver.xt_coeffs[i]).max())
    # Jacobian CSR (build_sparse_coeff, anm.cpp:362-438) incl. the 1e-9 drop rule
    prop = S.TaylorCoeffProp(omodel.y)
    prop.push_xi([(omodel.lt_inp.mat @ omodel.lt_inp.x0).reshape(-1, 3, 3)])
    Ao, _ = build_jacobian_csr(omodel.lt_out, prop.get_jacobian(), omodel.lt_inp.mat, omodel.lt_inp.n)
    Ad = run.solver.jacobian_csr()
    assert abs(Ad - Ao).max() <= 1e-10 * abs(Ao).max()
    st = run.solver.stats()
    assert st["nr_unknown"] == omodel.lt_inp.n and st["nr_tet"] == omesh.nr_tet


def _verbose_numbers(text):
    import re
    return [float(x) for x in re.findall(r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?|nan|inf", text)]


def test_verbose_printout_matches_the_reference_format(api):
    """SANM_VERBOSE (anm.cpp:200-203, :247-259, :295-309): '=== ANM iter K:', gt / xgt / jacob (= coeff_l2),
    'i:(bi=.. xbi=..)' per order, 'bound=.. t=..', 'x(a): ..', 't(a): ..,' -- same text layout as the oracle's
    restatement, numbers equal to the printed precision."""
    cfg = {"material": {"young": 3e3, "poisson": 0.45, "density": 900.0}, "g": [0, -9.81, 0],
           "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_c", "order": 8,
           "disable_pade": True}
    dims, sp = (5, 3, 3), 0.03
    _, osolver, _ = ofea.make_gravity_solver(ofea.make_cuboid(*dims, sp), cfg)
    run = dfea.GravityRun(api, dfea.make_cuboid(*dims, sp), dict(cfg), solver_rtol=1e-15, profile=1).construct()
    td, to = run.solver.verbose_text(), osolver.verbose_text
    assert td.startswith("=== ANM iter 0:\ngt=") and " 8:(bi=" in td and "\nx(a):" in td and "\nt(a):" in td
    import re
    skeleton = lambda t: re.sub(r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?", "#", t)
    assert skeleton(td) == skeleton(to)
    nd, no = _verbose_numbers(td), _verbose_numbers(to)
    assert len(nd) == len(no)
    # (%g prints 6 digits, %.3g 3: equal up to the last printed digit; the order-1 bias is exactly zero on both)
    for a, b in zip(nd, no):
        assert abs(a - b) <= 2e-3 * abs(b) + 1e-12, (a, b)
    run.step()
    osolver.next_iter()
    assert run.solver.verbose_text().startswith("=== ANM iter 1:\n")
    nd, no = _verbose_numbers(run.solver.verbose_text()), _verbose_numbers(osolver.verbose_text)
    assert len(nd) == len(no) and all(abs(a - b) <= 2e-3 * abs(b) + 1e-12 for a, b in zip(nd, no))


@pytest.mark.parametrize("use_pade", [False, True])
def test_vecscale_solver_path_following(api, use_pade):
    """ANMSolverVecScale: f(x) + t*v = 0 followed with update_approx (the
    save_interm branch of run_and_save, fea/main.cpp:386-414).  Without Pade both sides evaluate the same
    polynomial: tight tolerances.  With it the two rational approximants have denominators that agree to ~1e-4 only
    (tests/lockstep.py) while both meet the range criterion eps = 1e-6: points on the path agree to that accuracy."""
    cfg = {"material": {"young": 5e3, "poisson": 0.4, "density": 1000.0}, "g": [0, -9.81, 0],
           "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_c", "order": 10}
    dims, sp = (5, 3, 3), 0.03
    omesh = ofea.make_cuboid(*dims, sp)
    mat, fixed, fl = ofea.setup_gravity_task(omesh, cfg)
    om = ofea.make_forward(omesh, mat, fixed, "neohookean_c")
    from oracle.anm import ANMSolverVecScale, HyperParam
    ohp = HyperParam(order=10, use_pade=use_pade, solution_check_tol=0.01)
    osol = ANMSolverVecScale(om.y, om.lt_inp.mat, om.lt_out, om.lt_inp.out_shape, om.lt_inp.x0, 0.0,
                             om.lt_inp.copy_vtx_values(fl), ohp)
    dmesh = dfea.make_cuboid(*dims, sp)
    dfixed, dfl = dfea.setup_gravity(api, dmesh, cfg)
    dm = api.fea_model(dmesh.V, dmesh.tets, dfixed, "neohookean_c", 5e3, 0.4)
    hp = api.default_hyper(order=10, use_pade=int(use_pade), solution_check_tol=0.01, solver_rtol=1e-15)
    dsol = A.ANMSolverVecScale(api, dm.y, dm.lt_inp, dm.lt_out, dm.x0(), 0.0, dm.copy_vtx_values(dfl), hp)
    # every expansion from a common state: series, a_bound, Pade outcome identical or certified, restart points
    # (round 3 broke out of the loop silently when a Pade flag differed; LockStepPath either certifies or fails)
    from tests.lockstep import LockStepPath
    lp = LockStepPath(dsol, osol)
    tol = 1e-4 if use_pade else 1e-6
    for _ in range(3):
        forced = lp.steps[-1]["forced"]
        if not forced:
            assert dsol.get_t_upper() == pytest.approx(osol.get_t_upper(), rel=tol)
        t = 0.5 * (osol.t_coeffs[0] + min(osol.get_t_upper(), dsol.get_t_upper()))
        ao, ad = osol.solve_a(t), dsol.solve_a(t)
        if not forced:
            # (Brent's zero stops within its absolute tolerance 1e-6 of the root, wherever its path of iterates ends)
            assert ad == pytest.approx(ao, rel=tol, abs=2.5e-6)
        xo, to = osol.eval(ao)
        xd, td = dsol.eval(ad)  # the point of the path with parameter value t, each side through its own map
        assert td == pytest.approx(t, rel=5e-5) and to == pytest.approx(t, rel=5e-5)
        assert np.abs(xd - xo).max() <= max(VTX_RTOL, tol) * np.abs(xo).max()
        lp.update_approx()
    assert dsol.get_nr_iter() == osol.get_nr_iter() == 4
    print("vecscale path: events", [(e["step"], e["device"], e["oracle"]) for e in lp.events])


def test_cuboid_twist_baseline_config1(api):
    """BASELINE config 1: ANMImplicitSolver (displacement driven, t column ->
    grad_t) followed by the order-6 ANMEqnSolver refinement.  The free-running device run against the golden
    file; then every stage again with the oracle in lock step (tests/lockstep.py: LockStepPath for the implicit
    solver, LockStep for the refinement) -- step counts per stage EQUAL to the free-running oracle's (the golden
    file) unless the lock-step run logged a certified ill-conditioned Pade decision in that stage."""
    from tests.lockstep import lockstep_vtx_delta_stage
    gold = json.load(open(os.path.join(GOLD, "anm_cuboid_twist.json")))
    cfg = dict(gold["config"])
    V, stats = dfea.test_cuboid_twist(api, cfg)
    Vg = np.array(gold["vertices"])
    assert np.abs(V - Vg).max() <= VTX_RTOL * np.abs(Vg).max()
    assert stats[-1]["force_rms_recomp"] < 1e-10
    # the same stages, oracle beside the device
    mc = cfg["material"]
    omesh = ofea.make_cuboid(int(cfg["x"]), int(cfg["y"]), int(cfg["z"]), float(cfg["spacing"]))
    omat = ofea.Material(mc["young"], mc["poisson"], mc.get("density", 0.0))
    recs = []

    def stage(api_, mesh, fixed, config, delta, vtx_cur, require_refine):
        vtx, st = lockstep_vtx_delta_stage(api_, mesh, omesh, omat, fixed, config, delta, vtx_cur, require_refine)
        recs.append(st)
        return vtx, st

    V2, _ = dfea.test_cuboid_twist(api, cfg, stage=stage)
    assert np.abs(V2 - V).max() <= 1e-9 * np.abs(V).max()
    assert len(recs) == len(stats) == len(gold["stats"])
    for s_, r_, g_ in zip(stats, recs, gold["stats"]):
        # the device repeats itself
        assert (s_["iter_deform"], s_["iter_refine"]) == (r_["iter_deform"], r_["iter_refine"])
        print("stage", (s_["iter_deform"], s_["iter_refine"]), "oracle free-running", (g_["iter_deform"], g_["iter_refine"]),
              "events", [(e["step"], e["device"], e["oracle"]) for e in r_["events"]])
        if not r_["events"]:
            assert (s_["iter_deform"], s_["iter_refine"]) == (g_["iter_deform"], g_["iter_refine"])
            assert np.allclose(s_["t_upper"], g_["t_upper"], rtol=1e-6)


def test_inverse_single_tet(api):
    gold = json.load(open(os.path.join(GOLD, "anm_single_tet_inverse.json")))
    cfg = gold["config"]
    sp, ang = cfg["spacing"], np.pi * 2 / 3
    V = np.zeros((4, 3))
    V[:3, 0] = np.cos(ang * np.arange(3)) * sp
    V[:3, 1] = np.sin(ang * np.arange(3)) * sp
    V[3, 2] = sp
    fixed = np.zeros((4, 3), bool)
    fixed[:3] = True
    for load, want in ((-1000.0, np.array(gold["vertices"])[3, 2]), (1000.0, 0.022755286528750494)):
        f = np.zeros((4, 3))
        f[3, 2] = load
        m = api.fea_model(V, np.array([[0, 1, 2, 3]]), fixed, cfg["energy_model"], cfg["material"]["young"],
                          cfg["material"]["poisson"], inverse=True)
        hp = dfea.hyper_from_config(api, cfg, solver_rtol=1e-15)
        s = A.ANMEqnSolver(api, m.y, m.lt_inp, m.lt_out, m.x0(), m.copy_vtx_values(f), hp)
        while not s.converged():
            s.next_iter()
        z = m.full_vertices(s.get_x(), V)[3, 2]
        assert z == pytest.approx(want, rel=2e-8)


def test_jacobi_pcg_solver_against_the_oracle(api):
    """solver_kind 0, the Jacobi-preconditioned conjugate gradients north_star names first (pcg_init / pcg_spmv_dot /
    pcg_update kernels; the reference's direct solve, sparse_solver.cpp:154-180, stands behind the oracle): the first
    expansion's coefficients against the live oracle at 1e-8, then the whole continuation without Pade -- same step
    count, same residuals, the oracle's equilibrium (golden file) at the north-star tolerance.  The iteration stops
    at a relative residual of 1e-13: what the conditioning of the cuboid's Jacobian leaves of that in x is 1e-9."""
    gold = json.load(open(os.path.join(GOLD, "anm_cuboid_nc_nopade_o8.json")))
    cfg = gold["config"]
    assert cfg.get("disable_pade")
    run = dfea.GravityRun(api, dfea.make_cuboid(*gold["dims"], gold["spacing"]), dict(cfg), solver_kind=0,
                          solver_rtol=1e-13, solver_maxit=20000, profile=1).construct()
    _, osolver, _ = ofea.make_gravity_solver(ofea.make_cuboid(*gold["dims"], gold["spacing"]), cfg)
    tr = run.solver.trace()
    assert np.allclose(tr["t"], osolver.trace[0]["t"], rtol=1e-8)
    assert np.allclose(tr["x_norm"], osolver.trace[0]["x_norm"], rtol=1e-8)
    xc = run.solver.xt_coeffs()
    for i in range(1, int(cfg["order"]) + 1):
        assert np.abs(xc[i] - osolver.xt_coeffs[i]).max() <= 1e-8 * np.abs(osolver.xt_coeffs[i]).max(), i
    st = run.solver.stats()
    assert st["nr_linear_solve"] >= int(cfg["order"]) and 0 < st["linear_iters_last"] < 20000
    while not run.solver.converged():
        run.step()
    assert run.solver.get_nr_iter() == gold["iter"]
    assert np.allclose(run.rms[:-1], gold["residual_rms"][:-1], rtol=1e-5)
    V, Vg = run.vertices(), np.array(gold["vertices"])
    assert np.abs(V - Vg).max() <= VTX_RTOL * np.abs(Vg).max()
    assert run.rms[-1] < 1e-10


def test_error_paths(api):
    mesh = dfea.make_cuboid(3, 3, 3, 0.05)
    fixed = np.zeros((mesh.nr_vertices, 3), bool)
    fixed[mesh.V[:, 0] < 0.01] = True
    m = api.fea_model(mesh.V, mesh.tets, fixed, "neohookean_c", 1e4, 0.4)
    v = np.ones(m.n)
    # f(x0) + t0*v != 0 -> SANMNumericalError (anm.cpp:343-360)
    with pytest.raises(A.SanmNumericalError):
        A.ANMSolverVecScale(api, m.y, m.lt_inp, m.lt_out, m.x0(), 1.0, v, api.default_hyper(order=4))
    # order < 2 -> assertion (anm.cpp:108-110)
    with pytest.raises(A.SanmAssertionError):
        A.ANMEqnSolver(api, m.y, m.lt_inp, m.lt_out, m.x0(), v, api.default_hyper(order=1))
    # the Tikhonov path needs the direct solver: with the iterative one it is rejected, not ignored
    with pytest.raises(A.SanmUnsupportedError):
        A.ANMEqnSolver(api, m.y, m.lt_inp, m.lt_out, m.x0(), v,
                       api.default_hyper(order=4, xcoeff_l2_penalty=0.1, solver_kind=0))


@pytest.mark.parametrize("pade_on", [False, True])
def test_tet_renumbering_is_transparent(api, monkeypatch, pade_on):
    """the driver renumbers the tets along a Morton curve when it knows the positions of the
    unknowns (gather locality); nothing it returns may depend on that: same step count and the
    same solution with the renumbering switched off."""
    gold = json.load(open(os.path.join(GOLD, "anm_cuboid_nc.json")))
    cfg = gold["config"] if pade_on else _no_pade(gold["config"])
    run = _run_device(api, gold["dims"], gold["spacing"], cfg)
    monkeypatch.setenv("SANM_NO_TET_ORDER", "1")
    ref = _run_device(api, gold["dims"], gold["spacing"], cfg)
    _same_continuation(run, ref, pade_on)
    # the Jacobian handed out in CSR form lives in the space of the unknowns as well
    J, Jr = run.solver.jacobian_csr(), ref.solver.jacobian_csr()
    assert np.array_equal(J.indices, Jr.indices) and np.allclose(J.data, Jr.data, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("pade_on", [False, True])
@pytest.mark.parametrize("gold_name", ["anm_cuboid_nc.json", "anm_cuboid_ni.json", "anm_cuboid_arap.json"])
def test_specialised_pass_kernels_agree_with_the_interpreter(api, monkeypatch, gold_name, pade_on):
    """batches of SANM_JIT_MIN_T tets or more run pass kernels compiled at run time for their graph (the operator
    records as compile-time constants); forced on a small model here, they must reproduce the interpreter kernels'
    continuation: same step count, same solution.  (On the host harness both runs take the shared bodies.)"""
    gold = json.load(open(os.path.join(GOLD, gold_name)))
    cfg = gold["config"] if pade_on else _no_pade(gold["config"])
    monkeypatch.setenv("SANM_NO_JIT", "1")
    ref = _run_device(api, gold["dims"], gold["spacing"], cfg)
    monkeypatch.delenv("SANM_NO_JIT")
    monkeypatch.setenv("SANM_JIT_MIN_T", "1")
    run = _run_device(api, gold["dims"], gold["spacing"], cfg)
    _same_continuation(run, ref, pade_on)


@pytest.mark.parametrize("pade_on", [False, True])
def test_remap_out_without_the_triple_structure(api, pade_on):
    """the FEA builder's remap_out has its rows in triples (equal coefficients for the three force components of a
    vertex), which the device path exploits with one list per vertex; any other sparse map takes the row-by-row
    gather.  Scaling the rows of one component (and the load with them) keeps the solution and the continuation
    but breaks the structure -- and makes the Jacobian unsymmetric: same step count, same solution."""
    import scipy.sparse as sp
    gold = json.load(open(os.path.join(GOLD, "anm_cuboid_nc.json")))
    cfg = gold["config"] if pade_on else _no_pade(gold["config"])
    ref = _run_device(api, gold["dims"], gold["spacing"], cfg)
    run = dfea.GravityRun(api, dfea.make_cuboid(*gold["dims"], gold["spacing"]), dict(cfg), solver_rtol=1e-15)
    R = run.model.lt_out.to_scipy()
    scale = np.ones(R.shape[0])
    scale[1::3] = 2.0
    lt_out = A.SparseLinearDesc(api, sp.diags(scale) @ R)
    run.solver = A.ANMEqnSolver(api, run.model.y, run.model.lt_inp, lt_out, run.model.x0(), run.f_sub * scale,
                                run.hyper)
    run.rms = [run.solver.residual_rms()]
    run.run()
    # (the scaled rows change the residual norm and with it rms-based quantities: equilibrium and count only)
    assert run.solver.converged() and ref.solver.converged()
    if not pade_on:
        assert run.solver.get_nr_iter() == ref.solver.get_nr_iter()
    x, xr = run.solver.get_x(), ref.solver.get_x()
    assert np.abs(x - xr).max() <= 1e-8 * np.abs(xr).max()


def test_jacobian_of_a_mesh_built_by_several_host_threads(api):
    """the Jacobian pattern / gather lists of more than 4096 unknowns are built by several host threads and
    merged; the assembled matrix must still be the oracle's, entry by entry."""
    cfg = {"material": {"young": 3e4, "poisson": 0.45, "density": 900.0}, "g": [0, -9.81, 0],
           "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_c", "order": 6}
    dims, sp = (13, 13, 13), 0.01
    omesh = ofea.make_cuboid(*dims, sp)
    omodel, osolver, _ = ofea.make_gravity_solver(omesh, cfg)
    assert omodel.lt_inp.n > 4096  # two builder threads
    run = dfea.GravityRun(api, dfea.make_cuboid(*dims, sp), dict(cfg)).construct()
    prop = S.TaylorCoeffProp(omodel.y)
    prop.push_xi([(omodel.lt_inp.mat @ omodel.lt_inp.x0).reshape(-1, 3, 3)])
    Ao, _ = build_jacobian_csr(omodel.lt_out, prop.get_jacobian(), omodel.lt_inp.mat, omodel.lt_inp.n)
    Ad = run.solver.jacobian_csr()
    assert Ad.shape == Ao.shape and abs(Ad - Ao).max() <= 1e-10 * abs(Ao).max()
    assert run.solver.get_nr_iter() == osolver.get_nr_iter() == 1
    assert run.rms[-1] == pytest.approx(osolver.residual_rms, rel=1e-6, abs=1e-14)


def test_setup_loops_on_host_threads_are_deterministic(api, monkeypatch):
    """tet order (piecewise sort + merges), permuted remap tables and ELL tables are filled by several host threads
    above 4096 tets per thread (host_parallel.h; they are inside the reference's time_solve): the Jacobian and the
    first iteration must be bit-identical to the single-threaded build."""
    cfg = {"material": {"young": 3e4, "poisson": 0.45, "density": 900.0}, "g": [0, -9.81, 0],
           "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_c", "order": 4}
    dims, sp = (20, 20, 20), 0.01
    out = []
    for threads in ("1", "8"):
        monkeypatch.setenv("SANM_HOST_THREADS", threads)
        mesh = dfea.make_cuboid(*dims, sp)
        assert mesh.nr_tet >= 8 * 4096
        run = dfea.GravityRun(api, mesh, dict(cfg)).construct()
        J = run.solver.jacobian_csr()
        out.append((J.indptr.copy(), J.indices.copy(), J.data.copy(), run.rms[-1], run.solver.get_x().copy()))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


def test_pattern_and_analysis_from_block_rows(api, monkeypatch):
    """the Jacobian of a tet mesh has a 3 x 3 block per pair of vertices: the pattern is built from its BLOCK rows and the
    direct solver's analysis starts from them while the rows of the unknowns are written out (sparse.cpp, on_blocks;
    multifrontal.h, BlockPattern; round 6).  The routes it replaced -- the analysis from the finished rows
    (SANM_ANALYSIS_FROM_ROWS), the rows one by one (SANM_PATTERN_NO_BLOCKS), everything on one thread
    (SANM_SETUP_SERIAL) -- give the same pattern, the same analysis (statistics of the solver) and the same bits of the
    first step."""
    cfg = {"material": {"young": 3e4, "poisson": 0.45, "density": 900.0}, "g": [0, -9.81, 0],
           "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_c", "order": 5}
    dims, sp = (18, 14, 11), 0.01
    out = []
    for env in (None, "SANM_ANALYSIS_FROM_ROWS", "SANM_PATTERN_NO_BLOCKS", "SANM_SETUP_SERIAL"):
        if env:
            monkeypatch.setenv(env, "1")
        run = dfea.GravityRun(api, dfea.make_cuboid(*dims, sp), dict(cfg)).construct()
        J = run.solver.jacobian_csr()
        st = run.solver.stats()
        out.append((J.indptr.copy(), J.indices.copy(), J.data.copy(), run.rms[-1], run.solver.get_x().copy(),
                    np.array([st[k] for k in ("jacobian_nnz", "assembly_contribs", "factor_nnz", "factor_flops", "nr_front", "nr_level", "max_front") if k in st])))
        if env:
            monkeypatch.delenv(env)
    assert out[0][5].size == 7
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert np.array_equal(a, b)


def test_remap_in_coefficients_packed_into_the_index_words(api, monkeypatch):
    """a remap_in table whose coefficients are all +1 / -1 / 0 (the edge vectors of a tet mesh) is kept as index words
    with the coefficient in their two top bits (program.h, RemapInDev: 4 bytes per entry instead of 12 in every Taylor
    pass); the kernels decode and form the same products: Jacobian, residuals and solution bit-identical with
    SANM_RIN_NO_PACK=1, interpreter and compiled kernels alike."""
    gold = json.load(open(os.path.join(GOLD, "anm_cuboid_nc.json")))
    out = []
    for flag in (None, "1"):
        if flag:
            monkeypatch.setenv("SANM_RIN_NO_PACK", flag)
        run = _run_device(api, gold["dims"], gold["spacing"], gold["config"])
        J = run.solver.jacobian_csr()
        out.append((J.data.copy(), np.array(run.rms), run.solver.get_x().copy(), run.solver.get_nr_iter()))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


def test_assembly_lists_of_row_triples_give_the_same_jacobian(api, monkeypatch):
    """the three rows of a vertex share their columns and, up to a shift of the Jacobian index, their gather lists
    (backend.h, AssemblyDev::triples): the device keeps the lists of every third row only and assemble3_kernel forms the
    three values from one pass over a list.  SANM_ASM_NO_TRIPLES=1 is the list per non-zero: same sums, term by term --
    the Jacobian, the first expansion and the solution must be bit-identical.  (The host harness assembles from the
    remap tables either way.)"""
    cfg = {"material": {"young": 3e4, "poisson": 0.45, "density": 900.0}, "g": [0, -9.81, 0],
           "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_c", "order": 6}
    dims, sp = (11, 7, 6), 0.01
    out = []
    for flag in (None, "1"):
        if flag:
            monkeypatch.setenv("SANM_ASM_NO_TRIPLES", flag)
        run = dfea.GravityRun(api, dfea.make_cuboid(*dims, sp), dict(cfg)).construct()
        J = run.solver.jacobian_csr()
        run.step()
        out.append((J.indptr.copy(), J.indices.copy(), J.data.copy(), np.array(run.rms), run.solver.get_x().copy()))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    assert np.abs(out[0][2]).max() > 0


def test_pade_approx_on_its_own(api):
    """tests/pade.cpp:64-110 (Pade.Approx) through the C ABI: a stand-alone PadeApproximation over nine coefficient
    vectors of 500 entries (the last entry is t), its range accepted from range0 / 10, eval against the plain
    polynomial inside the range and solve_a / eval consistent -- with the reference's tolerances -- and the
    oracle's PadeApproximation on the same series beside it: same t_max_a (the bisection's probes agree), same
    values."""
    from oracle import unary_polynomial as up
    from oracle.pade import PadeApproximation as OPade
    rng = np.random.default_rng(7)
    SIZE, N, eps = 500, 9, 1e-5
    xs = [rng.uniform(-1, 1, SIZE) * 0.5 ** (i + 1) for i in range(N)]
    xs[1][SIZE - 1] = 2.3
    range0 = (eps * np.linalg.norm(xs[1]) / np.linalg.norm(xs[N - 1])) ** (1.0 / (N - 2))
    pade = A.PadeApproximation(api, xs, False)
    assert pade.estimate_valid_range(range0 / 10, eps)
    opade = OPade(xs, False, True)
    assert opade.estimate_valid_range(range0 / 10, eps)
    assert pade.get_t_max_a() == pytest.approx(opade.t_max_a, rel=1e-6)
    assert pade.get_t_max() == pytest.approx(opade.t_max, rel=1e-6)
    tmin, tmax = xs[0][SIZE - 1], pade.get_t_max()
    assert tmax > tmin
    for div in (8.0, 3.0, 1.01):
        a = pade.get_t_max_a() / div
        expect, got = up.eval_tensor(xs, a), pade.eval_xt(a)
        assert np.allclose(expect, got, rtol=1e-4, atol=1e-4)
        assert expect[-1] == pytest.approx(got[-1], rel=1.2e-5)
        assert np.abs(got - opade.eval_xt(a)).max() <= 1e-9 * np.abs(got).max()
    for frac in (1e-3, 0.27, 0.96):
        t = tmin * (1 - frac) + tmax * frac
        a = pade.solve_a(t)
        assert a == pytest.approx(opade.solve_a(t), rel=1e-5, abs=2.5e-6)
        got = pade.eval_xt(a)
        assert got[-1] == pytest.approx(t, rel=1.2e-5)
        assert np.allclose(up.eval_tensor(xs, a), got, rtol=1e-4, atol=1e-4)


def test_pade_approx_beyond_order_25(api):
    """The reference's PadeApproximation takes any order (pade.cpp:13-30); the device's Gram-Schmidt / linear-combination
    kernels take 24 vectors per launch and run longer series in chunks (round 4; rounds 1-3 refused order > 25).  31
    coefficient vectors: same accepted range, same values and same solve_a as the oracle's on the same series."""
    from oracle import unary_polynomial as up
    from oracle.pade import PadeApproximation as OPade
    rng = np.random.default_rng(11)
    SIZE, N, eps = 700, 31, 1e-5
    xs = [rng.uniform(-1, 1, SIZE) * 0.6 ** (i + 1) for i in range(N)]
    xs[1][SIZE - 1] = 2.3
    range0 = (eps * np.linalg.norm(xs[1]) / np.linalg.norm(xs[N - 1])) ** (1.0 / (N - 2))
    pade = A.PadeApproximation(api, xs, False)
    opade = OPade(xs, False, True)
    ok_d, ok_o = pade.estimate_valid_range(range0 / 10, eps), opade.estimate_valid_range(range0 / 10, eps)
    assert ok_d == ok_o
    assert ok_d, "the series was chosen so that the approximant is accepted"
    assert pade.get_t_max_a() == pytest.approx(opade.t_max_a, rel=1e-6)
    assert pade.get_t_max() == pytest.approx(opade.t_max, rel=1e-6)
    for div in (8.0, 3.0, 1.01):
        a = pade.get_t_max_a() / div
        got = pade.eval_xt(a)
        assert np.abs(got - opade.eval_xt(a)).max() <= 1e-9 * np.abs(got).max()
        assert np.allclose(up.eval_tensor(xs, a), got, rtol=1e-4, atol=1e-4)
    tmin, tmax = xs[0][SIZE - 1], pade.get_t_max()
    for frac in (0.27, 0.96):
        t = tmin * (1 - frac) + tmax * frac
        assert pade.solve_a(t) == pytest.approx(opade.solve_a(t), rel=1e-5, abs=2.5e-6)


def test_anm_solver_at_order_30(api):
    """a whole continuation at order 30 with Pade on (orders beyond 25 were refused until round 4): the oracle's
    equilibrium, the oracle's step count unless a certified ill-conditioned decision is met (tests/lockstep.py)"""
    cfg = {"material": {"young": 3e3, "poisson": 0.45, "density": 1000.0}, "g": [0, -9.81, 0],
           "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_c", "order": 30}
    dims, sp = (5, 3, 3), 0.03
    run = dfea.GravityRun(api, dfea.make_cuboid(*dims, sp), dict(cfg), solver_rtol=1e-15).construct()
    _, osolver, _ = ofea.make_gravity_solver(ofea.make_cuboid(*dims, sp), cfg)
    ls = LockStep(run, osolver).run_to_convergence()
    omodel, ofree, _ = ofea.make_gravity_solver(ofea.make_cuboid(*dims, sp), cfg)
    xo, _ = ofea.run_anm(ofree)
    Vo = omodel.lt_inp.full_vertices(xo)
    V = run.vertices()
    assert np.abs(V - Vo).max() <= VTX_RTOL * np.abs(Vo).max()
    print("order 30: steps", ls.nr_steps, "oracle", ofree.get_nr_iter(), "events", len(ls.events))
    if not ls.events:
        assert ls.nr_steps == ofree.get_nr_iter()


@pytest.mark.gpu
def test_kernels_built_ahead_of_time_equal_the_ones_compiled_at_run_time(monkeypatch):
    """The pass kernels of the fea models' graphs are compiled when the library is built (sanm_amd/build.py:
    rtc_embedded.bin, found by the key of the generated source, which depends on the graph's structure and the order
    only) -- by the build's compiler, possibly another version than the process's run-time compiler.  Both must give
    the same bits (everything is built with -ffp-contract=off and says fma where it wants one): the same cuboid
    continuation with the embedded set and with SANM_NO_JIT_EMBEDDED=1 (compiled now), vertex for vertex."""
    import ctypes
    import sanm_amd
    api = sanm_amd.get_api(0)
    gold = json.load(open(os.path.join(GOLD, "anm_cuboid_nc.json")))
    monkeypatch.setenv("SANM_JIT_MIN_T", "1")
    monkeypatch.setenv("SANM_NO_JIT_CACHE", "1")
    hits = ctypes.c_int64()
    api.lib.sanm_rtc_embedded_hits(ctypes.byref(hits))
    h0 = hits.value
    run = _run_device(api, gold["dims"], gold["spacing"], gold["config"])
    api.lib.sanm_rtc_embedded_hits(ctypes.byref(hits))
    assert hits.value == h0 + 1 and run.solver.setup_profile()["jit_source"] == "embedded"
    monkeypatch.setenv("SANM_NO_JIT_EMBEDDED", "1")
    api.lib.sanm_rtc_cache_drop_memory()
    ref = _run_device(api, gold["dims"], gold["spacing"], gold["config"])
    assert ref.solver.setup_profile()["jit_source"] == "compiled"
    assert run.solver.get_nr_iter() == ref.solver.get_nr_iter()
    assert np.array_equal(run.vertices(), ref.vertices())
