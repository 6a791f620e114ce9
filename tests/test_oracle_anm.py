"""Oracle: Taylor propagation, ANM drivers, Pade / polynomial helpers.

Restates the reference's integration checks (tests/symbolic.cpp:89-137
check_taylor_prop, :640-675; tests/pade.cpp:16-110) and pins the end-to-end
fixtures under tests/golden/.
"""
import json
import os

import numpy as np
import pytest

from oracle import fea, symbolic as S
from oracle import unary_polynomial as up
from oracle.pade import PadeApproximation

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("energy,mode", [("neohookean_c", "pk1"), ("neohookean_i", "pk1"), ("arap", "pk1"),
                                         ("stvk_stretch", "pk1"), ("neohookean_c", "cauchy"),
                                         ("neohookean_i", "cauchy")])
def test_taylor_prop_matches_direct_evaluation(energy, mode):
    rng = np.random.default_rng(0)
    T, N = 5, 10
    mat = fea.Material(1e3, 0.45)
    cg = S.ComputingGraph()
    X = S.placeholder(cg)
    F = X.batched_matmul(S.constant(cg, np.eye(3)[None].repeat(T, 0) + 0.1 * rng.standard_normal((T, 3, 3))))
    y = fea.pk1(energy, mat, F) if mode == "pk1" else fea.cauchy_stress(energy, mat, F)
    xs = [np.eye(3)[None].repeat(T, 0) + 0.2 * rng.standard_normal((T, 3, 3))] + \
         [rng.standard_normal((T, 3, 3)) * 0.5 for _ in range(N)]
    prop = S.TaylorCoeffProp(y)
    ys = [prop.push_xi([xs[0]])]
    for k in range(1, N + 1):
        b = prop.compute_next_order_bias()
        if k == 1:
            assert not np.any(b)  # symbolic.cpp:278-285
        yk = prop.push_xi([xs[k]])
        J = prop.get_jacobian()
        pred = b.reshape(T, 9) + np.einsum("bij,bj->bi", J, xs[k].reshape(T, 9))
        assert np.allclose(pred, yk.reshape(T, 9), rtol=1e-9, atol=1e-9 * np.abs(yk).max())
        ys.append(yk)
    a = 0.02
    direct = S.eval_unary_func(y, sum(x * a ** k for k, x in enumerate(xs)))
    series = sum(v * a ** k for k, v in enumerate(ys))
    assert np.abs(direct - series).max() <= 1e-12 * np.abs(direct).max()


def test_polynomial_roots_kat():
    # tests/pade.cpp:16-62: q(x) * (x-3)(x+4) has the real roots 3 and -4 (bit-exact pins: test_oracle_ref_poly.py)
    rng = np.random.default_rng(5)
    N = 10
    cf0 = rng.uniform(-1, 1, N - 2)
    cf0[N - 3] = 2.3
    coeffs = np.convolve(cf0, [-12, 1, 1])
    allr = up.roots(coeffs, False)
    assert allr is not None and len(allr) == N - 1
    for z in allr:
        s = 0j
        for c in coeffs[::-1]:
            s = s * z + c
        assert abs(s.real) < 2e-4 and abs(s.imag) < 2e-4
    roots = up.real_roots(coeffs)
    assert sorted(roots) == sorted(z.real for z in allr if z.imag == 0) and len(roots) >= 2
    assert min(abs(r - 3) for r in roots) < 1e-8 and min(abs(r + 4) for r in roots) < 1e-8


def test_brent_solve_eqn():
    f = [-2.0, 0.0, 1.0]  # x^2 - 2
    assert up.solve_eqn(f, 0, 2) == pytest.approx(np.sqrt(2), abs=2e-6)
    assert up.solve_eqn([0.0, 1.0, 0.5], 0, 3, b=1.0) == pytest.approx(-1 + np.sqrt(3), abs=2e-6)


def test_pade_approx_invariants():
    # tests/pade.cpp:64-110
    rng = np.random.default_rng(7)
    SIZE, N, eps = 500, 9, 1e-5
    xs = [rng.uniform(-1, 1, SIZE) * 0.5 ** (i + 1) for i in range(N)]
    xs[1][SIZE - 1] = 2.3
    range0 = (eps * np.linalg.norm(xs[1]) / np.linalg.norm(xs[N - 1])) ** (1.0 / (N - 2))
    pade = PadeApproximation(xs, False, True)
    assert pade.estimate_valid_range(range0 / 10, eps)
    tmin, tmax = xs[0][SIZE - 1], pade.t_max
    assert tmax > tmin
    for div in (8.0, 3.0, 1.01):
        a = pade.t_max_a / div
        expect = up.eval_tensor(xs, a)
        got = pade.eval_xt(a)
        assert np.allclose(expect, got, rtol=1e-4, atol=1e-4)
    for frac in (1e-3, 0.27, 0.96):
        t = tmin * (1 - frac) + tmax * frac
        a = pade.solve_a(t)
        got = pade.eval_xt(a)
        assert got[-1] == pytest.approx(t, rel=1e-5)
        assert np.allclose(up.eval_tensor(xs, a), got, rtol=1e-4, atol=1e-4)


def test_single_tet_inverse_known_answer():
    """KAT of the reference: utils/check_single_tet.py:61 holds the rest apex
    height 0.022755286528750494 for the stress state whose elastic force on the
    apex is -1000 N (the sign its printout uses)."""
    cfg = json.load(open(os.path.join(GOLD, "anm_single_tet_inverse.json")))["config"]
    mat = fea.Material(cfg["material"]["young"], cfg["material"]["poisson"])
    sp = cfg["spacing"]
    ang = np.pi * 2 / 3
    V = np.zeros((4, 3))
    V[:3, 0] = np.cos(ang * np.arange(3)) * sp
    V[:3, 1] = np.sin(ang * np.arange(3)) * sp
    V[3, 2] = sp
    fixed = np.zeros((4, 3), bool)
    fixed[:3] = True
    f = np.zeros((4, 3))
    f[3, 2] = 1000
    model, solver, x = fea.solve_static(fea.TetMesh(V, np.array([[0, 1, 2, 3]])), mat, fixed,
                                        cfg["energy_model"], f, cfg, inverse=True)
    z = model.lt_inp.full_vertices(x)[3, 2]
    assert z == pytest.approx(0.022755286528750494, rel=2e-8)
    # and the task exactly as the reference runs it (load -1000): regression fixture
    gold = json.load(open(os.path.join(GOLD, "anm_single_tet_inverse.json")))
    V2, s2 = fea.test_single_tet_inverse(cfg)
    assert s2.get_nr_iter() == gold["iter"]
    assert np.allclose(V2, gold["vertices"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("name", ["cuboid_nc", "cuboid_ni", "cuboid_arap", "cuboid_nc_nopade_o8"])
def test_gravity_cuboid_golden(name):
    gold = json.load(open(os.path.join(GOLD, f"anm_{name}.json")))
    mesh = fea.make_cuboid(*gold["dims"], gold["spacing"])
    model, solver, f = fea.make_gravity_solver(mesh, gold["config"])
    x, rms = fea.run_anm(solver)
    assert solver.get_nr_iter() == gold["iter"]
    assert [r["pade"] for r in solver.trace] == gold["pade"]
    # the range bound of the early steps is reproducible to round-off; near convergence it is the ratio of
    # two norms at noise level and moves by tens of per cent with the linear solver (PARDISO here when the
    # image has MKL, SuperLU otherwise) without changing a step or the solution
    ab = [r["a_bound"] for r in solver.trace]
    assert np.allclose(ab[:3], gold["a_bound"][:3], rtol=1e-6)
    V = model.lt_inp.full_vertices(x)
    assert np.abs(V - np.array(gold["vertices"])).max() <= 1e-9 * np.abs(V).max()
    # force equilibrium (fea/mesh_template.h:232-235)
    y = S.eval_unary_func(model.y, (model.lt_inp.mat @ x).reshape(-1, 3, 3))
    assert np.allclose(model.lt_out @ y.ravel(), -f, atol=1e-8)


def test_cuboid_twist_baseline_config1():
    gold = json.load(open(os.path.join(GOLD, "anm_cuboid_twist.json")))
    V, stats = fea.test_cuboid_twist(gold["config"])
    assert [s["iter_tot"] for s in stats] == [s["iter_tot"] for s in gold["stats"]]
    assert np.allclose(V, gold["vertices"], rtol=1e-8, atol=1e-11)
    assert stats[-1]["force_rms_recomp"] < 1e-10
