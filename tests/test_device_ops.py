"""Operator-level parity: every OperatorMeta of the device program against the
oracle's op-by-op restatement, through the C ABI (TaylorCoeffProp entry points).

Runs on the HIP library with -m gpu and on the test-only host harness otherwise.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import fea as ofea
from oracle import symbolic as S
from sanm_amd import api as A

RTOL = 1e-11


def _identity_remap(api, T):
    return api.sparse_desc(sp.identity(T * 9, format="csr"))


def _series(rng, T, N, near_identity=True, scale=0.3):
    x0 = rng.standard_normal((T, 3, 3)) * 0.2
    if near_identity:
        x0 += np.eye(3)[None]
    return [x0] + [rng.standard_normal((T, 3, 3)) * scale for _ in range(N)]


def _run_both(api, build, T=37, N=6, seed=0, near_identity=True, rtol=RTOL, x0_base=None):
    """build(ops, X, const) -> output var, for ops in {oracle S, device A}."""
    rng = np.random.default_rng(seed)
    cvals = [np.eye(3)[None] + 0.1 * rng.standard_normal((T, 3, 3)), rng.uniform(0.5, 2.0, (T, 1))]
    xs = _series(rng, T, N, near_identity)
    if x0_base is not None:
        xs[0] = xs[0] - np.eye(3)[None] * near_identity + x0_base
    # oracle
    ocg = S.ComputingGraph()
    oy = build(S, S.placeholder(ocg), [S.constant(ocg, c) for c in cvals])
    oprop = S.TaylorCoeffProp(oy)
    # device
    dcg = api.graph()
    dy = build(A, dcg.placeholder(), [dcg.constant(c) for c in cvals])
    dprop = A.TaylorCoeffProp(api, dy, _identity_remap(api, T), N, T)

    def close(a, b, what):
        scale = max(np.abs(b).max(), 1e-300)
        err = np.abs(a - b).max() / scale
        assert err <= rtol, f"{what}: rel err {err:.3e}"

    close(dprop.push_xi(xs[0]), oprop.push_xi([xs[0]]), "order 0")
    for k in range(1, N + 1):
        ob = oprop.compute_next_order_bias()
        db = dprop.compute_next_order_bias()
        if k == 1:
            close(dprop.get_jacobian(), oprop.get_jacobian(), "jacobian")
            assert not np.any(db)
        else:
            close(db, ob, f"bias {k}")
        close(dprop.push_xi(xs[k]), oprop.push_xi([xs[k]]), f"coeff {k}")


def test_matmul_const(api):
    _run_both(api, lambda M, X, c: X.batched_matmul(c[0]))


def test_matmul_var_var(api):
    _run_both(api, lambda M, X, c: X.batched_matmul(X.batched_transpose()).batched_matmul(X))


def test_transpose_lincomb_bias(api):
    _run_both(api, lambda M, X, c: M.linear_combine([(2.0, X), (-0.5, X.batched_transpose()), (1.5, c[0])], 0.25))


def test_matinv_identity_left(api):
    _run_both(api, lambda M, X, c: M.batched_mat_inv_mul(X, None, True))


def test_matinvmul_left_right(api):
    _run_both(api, lambda M, X, c: M.batched_mat_inv_mul(X, X.batched_matmul(c[0]), True)
              + M.batched_mat_inv_mul(X, c[0], False))


def test_det_muleye(api):
    _run_both(api, lambda M, X, c: X.batched_det().batched_mul_eye(3), N=8)


def test_det_log_multiply_broadcast(api):
    _run_both(api, lambda M, X, c: X.batched_det().log() * X)


def test_pow_fractional_and_square(api):
    _run_both(api, lambda M, X, c: (X.batched_det().pow(-2.0 / 3.0) * X.pow(2).reduce_sum(-1)) * X)


def test_scalar_constant_broadcast(api):
    _run_both(api, lambda M, X, c: (c[1] * X.batched_det()) * X + X * c[1])


def test_svdw_polar(api):
    # polar decomposition rotation, incl. inverted inputs (det < 0 -> rotation fix)
    _run_both(api, lambda M, X, c: X - X.batched_svd_w(True)[2], N=6, rtol=1e-9)
    _run_both(api, lambda M, X, c: X.batched_svd_w(True)[2], N=3, near_identity=False, seed=3, rtol=1e-8)


def _spread(T, seed=11):
    """Order-0 matrices R1 diag(3, 2, 1) R2: well separated singular values, so that dU/dM (which divides by
    s_j^2 - s_i^2, tensor_svd.cpp:236-262) is well conditioned and the comparison is not a test of clip_div."""
    rng = np.random.default_rng(seed)
    q1 = np.linalg.qr(rng.standard_normal((T, 3, 3)))[0]
    q2 = np.linalg.qr(rng.standard_normal((T, 3, 3)))[0]
    return q1 @ (np.diag([3.0, 2.0, 1.0])[None] * rng.uniform(0.9, 1.1, (T, 1, 1))) @ q2


@pytest.mark.parametrize("case", ["U", "S", "USW", "S_vector_output", "U_only"])
def test_svdw_full_mode(api, case):
    """batched_svd_w with U or S read by other operators: the full U, S, W recurrences (oprs/linalg.cpp:533-600,
    tensor_svd.cpp:275-387) and dU/dM, dS/dM (tensor_svd.cpp:147-273), which the FEA graphs never reach (the ARAP
    energy reads W only) but the reference's operator tests do (tests/tensor.cpp:841-868, tests/oprs.cpp)."""
    def build(M, X, c):
        u, s, w = X.batched_svd_w(False)
        # The columns of U are defined up to a sign (Eigen's JacobiSVD in the reference, LAPACK in the oracle, a
        # one-sided Jacobi on the device); U enters through its elementwise square, which is not.
        uu = u * u
        if case == "U":  # U and W read
            return uu.batched_matmul(w) + uu.batched_transpose()
        if case == "S":  # S read through the 3-vector elementwise operators
            return (s * s).reduce_sum(-1).batched_mul_eye(3).batched_matmul(w) + X
        if case == "USW":  # all three outputs read
            t = (s.log() * s).reduce_sum(-1)
            return uu.batched_matmul(w) * t + c[1] * uu
        if case == "S_vector_output":  # a (T, 3) output: Jacobian (T, 3, 9)
            return M.linear_combine([(2.0, s), (0.5, s.pow(2))], 0.25)
        return uu.batched_matmul(c[0])  # W unread: its gradient slot is dead (OP_FLAG_SVDW_GW clear)
    _run_both(api, build, T=29, N=6, seed=5, near_identity=False, rtol=1e-9, x0_base=_spread(29))


_FUSED_CASES = {
    "matmul": lambda M, X, c: X.batched_matmul(X.batched_transpose()).batched_matmul(X),
    "matinvmul": lambda M, X, c: M.batched_mat_inv_mul(X, X.batched_matmul(c[0]), True)
    + M.batched_mat_inv_mul(X, c[0], False),
    # F read by the inverse, the determinant and (transposed: an alias of the inverse's series) the product
    "neohookean_like": lambda M, X, c: X.batched_det().log() * M.batched_mat_inv_mul(X, None, True).batched_transpose()
    + X.batched_det().pow(-2.0 / 3.0) * X,
    "pow": lambda M, X, c: (X.batched_det().pow(-2.0 / 3.0) * X.pow(2).reduce_sum(-1)) * X + X.pow(3),
    "svdw_polar": lambda M, X, c: X - X.batched_svd_w(True)[2],
}


@pytest.mark.parametrize("case", sorted(_FUSED_CASES) + ["svdw_full"])
def test_compiled_kernels_with_fused_convolutions(api, monkeypatch, case):
    """The kernels compiled per graph (forced here for a small batch) compute the convolution sums of all operators
    in one loop over the history, pairs (i, k-i) in both orientations, before the operators run (tet_ops.h,
    conv_term); odd and even orders, the middle term and the wavefront split are all covered by orders 1..7.
    (On the host harness there is no run-time compiler: the shared bodies run as in the other tests.)"""
    monkeypatch.setenv("SANM_JIT_MIN_T", "1")
    if case == "svdw_full":
        def build(M, X, c):
            u, s, w = X.batched_svd_w(False)
            return (u * u).batched_matmul(w) * (s.log() * s).reduce_sum(-1) + c[1] * (u * u)
        _run_both(api, build, T=100, N=7, seed=5, near_identity=False, rtol=1e-9, x0_base=_spread(100))
    else:
        _run_both(api, _FUSED_CASES[case], T=100, N=7, seed=2, rtol=1e-9 if case == "svdw_polar" else 1e-10)


@pytest.mark.parametrize("energy", ["neohookean_c", "neohookean_i", "arap", "stvk_stretch"])
def test_pk1_graphs(api, energy):
    mat = ofea.Material(1e3, 0.45)

    def build(M, X, c):
        F = X.batched_matmul(c[0])
        if M is S:
            return ofea.pk1(energy, mat, F)
        # the same graph built with the device operator API
        mu, lam, k = mat.shear, mat.lame_first, mat.bulk
        if energy == "neohookean_c":
            FTinv = M.batched_mat_inv_mul(F, None, True).batched_transpose()
            J = F.batched_det()
            return M.linear_combine([(mu, F), (-mu, FTinv), (lam, J.log() * FTinv)])
        if energy == "neohookean_i":
            FTinv = M.batched_mat_inv_mul(F, None, True).batched_transpose()
            J = F.batched_det()
            Ic = F.pow(2).reduce_sum(-1)
            J23 = J.pow(-2.0 / 3.0)
            t2 = M.linear_combine([(mu / -3.0, J23 * Ic), (k, J * J), (-k, J)], 0) * FTinv
            return M.linear_combine([(mu, J23 * F), (1.0, t2)])
        if energy == "arap":
            return (F - F.batched_svd_w(True)[2]) * mu
        return M.linear_combine([(mu, F.batched_matmul(F.batched_transpose()).batched_matmul(F)), (-mu, F)])

    _run_both(api, build, N=10, rtol=1e-9)


@pytest.mark.parametrize("energy", ["neohookean_c", "neohookean_i"])
def test_cauchy_graphs_via_fea_model(api, energy):
    """inverse model (make_inverse + cauchy_stress) built by the C++ front-end."""
    rng = np.random.default_rng(4)
    mesh = ofea.make_cuboid(3, 3, 2, 0.05)
    fixed = np.zeros((mesh.nr_vertices, 3), bool)
    fixed[mesh.V[:, 0] < 0.01] = True
    mat = ofea.Material(1e4, 0.4)
    om = ofea.make_inverse(mesh, mat, fixed, energy)
    dm = api.fea_model(mesh.V, mesh.tets, fixed, energy, 1e4, 0.4, inverse=True)
    N, T, n = 5, mesh.nr_tet, om.lt_inp.n
    assert dm.n == n
    xs = [om.lt_inp.x0 + 0.002 * rng.standard_normal(n)] + [0.01 * rng.standard_normal(n) for _ in range(N)]
    oprop = S.TaylorCoeffProp(om.y)
    dprop = A.TaylorCoeffProp(api, dm.y, dm.lt_inp, N, T)
    app = lambda x: (om.lt_inp.mat @ x).reshape(T, 3, 3)
    assert np.allclose(dprop.push_xi(xs[0]), oprop.push_xi([app(xs[0])]), rtol=1e-10, atol=1e-6)
    for k in range(1, N + 1):
        ob, db = oprop.compute_next_order_bias(), dprop.compute_next_order_bias()
        assert np.allclose(db, ob, rtol=1e-8, atol=1e-8 * np.abs(ob).max() + 1e-300)
        oy, dy = oprop.push_xi([app(xs[k])]), dprop.push_xi(xs[k])
        assert np.allclose(dy, oy, rtol=1e-8, atol=1e-8 * np.abs(oy).max())


def test_remap_tables_match_oracle(api):
    """MeshShapeMatTrans / MeshForceOutputTrans built in C++ vs the oracle's."""
    mesh = ofea.make_cuboid(4, 3, 3, 0.03)
    rng = np.random.default_rng(9)
    fixed = np.zeros((mesh.nr_vertices, 3), bool)
    fixed[mesh.V[:, 0] < 0.01] = True
    fixed[5, 1] = True  # a partially fixed vertex
    delta = rng.standard_normal(mesh.V.shape) * fixed
    for vd in (None, delta):
        om = ofea.make_forward(mesh, ofea.Material(1e4, 0.4), fixed, "neohookean_c", None, vd)
        dm = api.fea_model(mesh.V, mesh.tets, fixed, "neohookean_c", 1e4, 0.4, vtx_delta=vd)
        assert abs(dm.lt_inp.to_scipy() - om.lt_inp.mat).max() < 1e-15
        assert abs(dm.lt_out.to_scipy() - om.lt_out).max() < 1e-18
        assert np.array_equal(dm.x0(), om.lt_inp.x0)
    fl = api.gravity_load(mesh.V, mesh.tets, 1234.0, [0.1, -9.81, 0.3])
    assert np.allclose(fl, ofea.gravity_load(mesh, ofea.Material(1, 0.3, 1234.0), [0.1, -9.81, 0.3]), rtol=1e-13)
    cfg = {"boundary_thresh": 0.3, "boundary_proj_dir": [1, 1, 0],
           "boundary_filter": {"dir": [0, 0, 1], "min": 0.0, "max": 0.6}}
    got = api.boundary_by_threshold(mesh.V, mesh.surface_vtx, [1, 1, 0], 0.3, [0, 0, 1], 0.0, 0.6)
    assert np.array_equal(got, ofea.boundary_by_config(mesh, None, cfg))


def test_unsupported_and_invalid_graphs(api):
    g = api.graph()
    X = g.placeholder()
    with pytest.raises(A.SanmAssertionError):
        X.batched_det().batched_matmul(X)  # scalar into matmul
    with pytest.raises(A.SanmAssertionError):
        X.pow(0.0)
    with pytest.raises(A.SanmUnsupportedError):  # the sum over the batch as well (no batched output)
        X.reduce_sum(-2)
    assert X.reduce_sum(1).id >= 0  # one axis of a matrix: served by the vector interpreter (tests/test_matrix_dims.py)


def test_host_poly_helpers(api):
    """sanm_poly_roots / sanm_poly_real_roots / sanm_poly_solve_eqn (the product's host code, poly.cpp) against
    tests/golden/ref_poly.json -- outcomes of the reference's own unary_polynomial.cpp + BRENT compiled with g++ -O2
    (tests/golden/make_ref_poly.py): the valid flag (None => the Pade approximant is rejected, pade.cpp:113-116)
    exactly, roots and Brent zeros bit for bit."""
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_poly.json")))
    unhex = lambda h: np.array([float.fromhex(v) for v in h])
    n_invalid = 0
    for rec in fx["roots"]:
        f = unhex(rec["f"])
        got = api.poly_real_roots(f)
        assert (got is not None) == rec["valid"], rec["src"]
        if got is None:
            n_invalid += 1
            continue
        assert np.array_equal(got, unhex(rec["real"]), equal_nan=True), rec["src"]
        full = api.poly_roots(f, only_real=False)
        assert (full is None) == (rec["all"] is None)
        if full is not None:
            want = np.array([complex(float.fromhex(re), float.fromhex(im)) for re, im in rec["all"]])
            assert np.array_equal(full.real, want.real, equal_nan=True) and np.array_equal(full.imag, want.imag)
    assert n_invalid > 0
    for rec in fx["solve_eqn"]:
        a = [float.fromhex(rec[k]) for k in ("xmin", "xmax", "b", "eps")]
        assert api.poly_solve_eqn(unhex(rec["f"]), *a) == float.fromhex(rec["x"])
    kat = fx["kat"]  # tests/pade.cpp:16-62
    roots = api.poly_real_roots(unhex(kat["f"]))
    assert np.array_equal(roots, unhex(kat["real"]))
    assert min(abs(roots - 3)) < 1e-9 and min(abs(roots + 4)) < 1e-9
    # and live against the oracle's restatement
    from oracle import unary_polynomial as up
    rng = np.random.default_rng(5)
    for _ in range(20):
        f = rng.uniform(-1, 1, int(rng.integers(3, 21))) * 0.5 ** np.arange(1)
        want, got = up.roots(f, False), api.poly_roots(f, False)
        assert (want is None) == (got is None)
        if want is not None:
            assert np.array_equal(got, np.array(want, dtype=complex), equal_nan=True)
    f = rng.uniform(-1, 1, 7)
    f[0] = -abs(f[0])
    hi = 1.0
    while up.eval_poly(f, hi) <= 0:
        hi *= 2
        f[-1] = abs(f[-1]) + 0.1
    assert api.poly_solve_eqn(f, 0.0, hi) == up.solve_eqn(f, 0.0, hi)


def test_integer_pow_through_zero_takes_the_convolution_path(api):
    """analytic_unary.cpp:112-131: where the order-0 value is a zero (|x| < 1e-3) an integer exponent continues on
    the convolution path (prop_taylor_coeff_int, :46-92) instead of the recurrence that divides by x_0.  The
    reference switches the whole tensor, the device path each element on its own: same coefficients."""
    T, N = 11, 6
    rng = np.random.default_rng(5)
    xs = _series(rng, T, N, near_identity=True)   # I + noise: the off-diagonal entries are near zero ...
    xs[0][:, 0, 1] = 0.0                          # ... and these are exactly zero
    xs[0][3, 2, 2] = 5e-4                         # a "zero" by the 1e-3 rule that is not 0.0
    for p in (3.0, 4.0):
        ocg = S.ComputingGraph()
        oy = S.placeholder(ocg).pow(p)
        oprop = S.TaylorCoeffProp(oy)
        dcg = api.graph()
        dy = dcg.placeholder().pow(p)
        dprop = A.TaylorCoeffProp(api, dy, _identity_remap(api, T), N, T)
        o0, d0 = oprop.push_xi([xs[0]]), dprop.push_xi(xs[0])
        assert np.allclose(d0, o0, rtol=1e-12, atol=1e-300)
        for k in range(1, N + 1):
            ob, db = oprop.compute_next_order_bias(), dprop.compute_next_order_bias()
            assert np.all(np.isfinite(db))
            assert np.abs(db - ob).max() <= 1e-11 * max(np.abs(ob).max(), 1e-30), (p, k)
            oc, dc = oprop.push_xi([xs[k]]), dprop.push_xi(xs[k])
            assert np.abs(dc - oc).max() <= 1e-11 * max(np.abs(oc).max(), 1e-30), (p, k)


def test_fractional_pow_of_zero_is_a_numerical_error(api):
    """SANMNumericalError{"0^p when p is not integer"} (analytic_unary.cpp:115-120)"""
    T = 5
    x0 = np.tile(np.eye(3), (T, 1, 1))  # off-diagonal zeros
    dcg = api.graph()
    dprop = A.TaylorCoeffProp(api, dcg.placeholder().pow(1.5), _identity_remap(api, T), 3, T)
    with pytest.raises(A.SanmNumericalError, match="0\\^p when p is not integer"):
        dprop.push_xi(x0)
    # the same input is fine for pow(2) (always on the convolution path) ...
    dprop2 = A.TaylorCoeffProp(api, dcg.placeholder().pow(2), _identity_remap(api, T), 3, T)
    assert np.allclose(dprop2.push_xi(x0), x0 ** 2)
