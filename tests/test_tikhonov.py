"""Tikhonov path (libsanm/sparse_solver.cpp:366-395, :162-176; config/override_l2_penalty.json): with
`xcoeff_l2_penalty` = lambda the Taylor coefficients solve (A'A + lambda I) x = A'b instead of A x = b; the sanity
check is skipped (anm.cpp:271) and the Pade basis keeps its projection on the first vector (anm.cpp:145).  Device
path against the oracle, which factors the same normal equations with SuperLU."""
import numpy as np
import pytest

from oracle import fea as ofea
from sanm_amd import fea as dfea

CFG = {"material": {"young": 3e3, "poisson": 0.45, "density": 1000.0}, "g": [0, -9.81, 0],
       "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_c", "order": 10}


@pytest.mark.parametrize("lam", [1e-6, 1e-2])
def test_first_step_coefficients_with_l2_penalty(api, lam):
    cfg = dict(CFG, xcoeff_l2_penalty=lam)
    dims, sp = (6, 3, 3), 0.025
    run = dfea.GravityRun(api, dfea.make_cuboid(*dims, sp), dict(cfg)).construct()
    omodel, osolver, _ = ofea.make_gravity_solver(ofea.make_cuboid(*dims, sp), cfg)
    cd, co = run.solver.xt_coeffs(), osolver.xt_coeffs
    assert len(cd) == len(co) == 11
    for k in (1, 2, 5, 10):
        assert np.abs(cd[k] - co[k]).max() <= 1e-7 * np.abs(co[k]).max(), k
    # range estimate of the first expansion: a_bound, Pade outcome identical or certified ill-conditioned
    from tests.lockstep import LockStep
    ls = LockStep(run, osolver)
    if not ls.events:
        assert abs(run.solver.get_t_upper() - osolver.t_max) <= 1e-6 * abs(osolver.t_max)


def test_l2_penalty_run_converges_like_the_oracle(api):
    cfg = dict(CFG, xcoeff_l2_penalty=1e-6)
    dims, sp = (6, 3, 3), 0.025
    from tests.lockstep import LockStep
    run = dfea.GravityRun(api, dfea.make_cuboid(*dims, sp), dict(cfg)).construct()
    omodel, osolver, _ = ofea.make_gravity_solver(ofea.make_cuboid(*dims, sp), cfg)
    xo, _ = ofea.run_anm(osolver)
    _, ostep, _ = ofea.make_gravity_solver(ofea.make_cuboid(*dims, sp), cfg)
    ls = LockStep(run, ostep, series_rtol=1e-5).run_to_convergence()
    if not ls.events:
        assert run.solver.get_nr_iter() == osolver.get_nr_iter()
    Vo = omodel.lt_inp.full_vertices(xo)
    assert np.abs(run.vertices() - Vo).max() <= 1e-6 * np.abs(Vo).max()
    assert run.rms[-1] < 1e-10
