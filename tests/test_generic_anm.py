"""The ANM solvers over graphs of arbitrary (batch, ...) tensors through the C ABI: the reference's own operator
tests (tests/symbolic.cpp:140-560, :803-900 -- ANMSolverVecScale / ANMImplicitSolver / ANMEqnSolver over pow,
mat_inv_mul, elementwise arithmetic, linear_combine, determinants at 4 / 5 / 7, reductions, transposes, random sparse
input / output maps, constants, log, the implicit solver and the slice / concat example of the paper) run on the
device path (vector interpreter as the pass engine of the same order loop: sanm_amd/csrc/anm.cpp,
construct_on_vector_interpreter) with the reference's acceptance criteria -- run_anm's iteration limit, t == Approx(t_dst),
f(solution) == Approx(target), Catch2's default relative epsilon -- and, beside them, the oracle's ANMSolverVecScale on
the same inputs: the first expansion's coefficients and the solution."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import symbolic as S
from oracle.anm import ANMSolverVecScale as OVecScale, HyperParam as OHyper
from sanm_amd import api as A

APPROX = 1.2e-5  # Catch2's Approx: epsilon = 100 * FLT_EPSILON


def _close(a, b, eps=APPROX, margin=0.0):
    a, b = np.asarray(a, dtype=float).ravel(), np.asarray(b, dtype=float).ravel()
    assert a.shape == b.shape
    bad = np.abs(a - b) > margin + eps * np.maximum(np.abs(a), np.abs(b))
    assert not bad.any(), (np.abs(a - b).max(), np.flatnonzero(bad)[:5])


def _placeholder(g, shape):
    return g.placeholder_matrix(*shape) if len(shape) == 2 else g.placeholder_vector(shape[0])


def run_anm(sol, t_dst, maxiter=20):
    """tests/symbolic.cpp:28-54"""
    it = 0
    while True:
        t_upper = sol.get_t_upper()
        it += 1
        assert it <= maxiter
        if t_upper > t_dst:
            break
        sol.update_approx()
    x, t = sol.eval(sol.solve_a(t_dst))
    assert t == pytest.approx(t_dst, rel=APPROX)
    return x, it


def _solve(api, build, x0, y0, t_dst, remap_in=None, remap_out=None, order=8, oracle=True, xshape=None, mid=None):
    """ANMSolverVecScale{y, remap_in, remap_out, x0, 1, -y0} followed to t_dst on the device; the oracle beside it"""
    x0 = np.asarray(x0, dtype=float)
    n = x0.size
    mid = mid or x0.shape                      # shape of the placeholder: (batch, ...)
    batch, pshape = mid[0], tuple(mid[1:])
    g = api.graph()
    y = build(_placeholder(g, pshape), A)
    rin = sp.identity(n, format="csr") if remap_in is None else remap_in
    rout = sp.identity(np.asarray(y0).size, format="csr") if remap_out is None else remap_out
    dri, dro = A.SparseLinearDesc(api, rin), A.SparseLinearDesc(api, rout)
    hp = api.default_hyper(order=order, use_pade=0)
    dsol = A.ANMSolverVecScale(api, y, dri, dro, x0, 1.0, -np.asarray(y0, dtype=float), hp)
    if oracle:
        osol = OVecScale(build(S.placeholder(S.ComputingGraph()), S), rin, rout, mid, x0, 1.0,
                         -np.asarray(y0, dtype=float).ravel(), OHyper(order=order, use_pade=False))
        # first expansion: the same series on both sides
        dco = dsol.xt_coeffs()
        # (absolute floor: the DFT determinant of dim > 4 carries the round-off of O(1) values into every coefficient,
        # on both sides -- tensor_polymat.cpp:100-136)
        floor = 1e-12 * np.abs(osol.xt_coeffs[1]).max()
        for k in range(order + 1):
            co, cd = osol.xt_coeffs[k], dco[k]
            assert np.abs(cd - co).max() <= 1e-8 * np.abs(co).max() + floor, k
        assert dsol.get_t_upper() == pytest.approx(osol.get_t_upper(), rel=1e-6)
    xd, it = run_anm(dsol, t_dst)
    if oracle:
        xo, ito = run_anm(osol, t_dst)
        assert it == ito
        assert np.abs(xd - xo).max() <= 1e-6 * np.abs(xo).max()
    return xd.reshape(x0.shape)


def _rng(seed, lo=-1.0, hi=1.0):
    r = np.random.default_rng(seed)
    return lambda shape, l=lo, h=hi: r.uniform(l, h, shape)


# ------------------------------------------------------------------------------------------ Symbolic.Pow (:140-176)
@pytest.mark.parametrize("exp,lo,hi", [(2.0, 1, 2), (1.5, 1, 2), (-0.75, 1, 2), (3.0, -5, 5), (0.3, 1, 2)])
def test_pow(api, exp, lo, hi):
    """pow(x, e) + t v = 0 with v = -pow(x0, e): followed from t = 1 to 2 (x0 of shape (3, 2))"""
    x0 = _rng(1, lo, hi)((3, 2))
    if lo < 0:
        x0 = np.where(np.abs(x0) < 0.5, x0 + np.sign(x0 + 1e-9) * 0.5, x0)  # (keeps the path off the origin)
    y0 = x0 ** exp
    sol = _solve(api, lambda x, M: x.pow(exp), x0, y0, 2.0)
    _close(sol ** exp, y0 * 2)


# ------------------------------------------------------------------------------------ Symbolic.MatInvMul (:179-233)
def _x0_diag(seed, lo, hi, batch=9, m=4, shift=4.0):
    x0 = _rng(seed, lo, hi)((batch, m, m))
    for i in range(m):
        x0[:, i, i] += shift
    return x0


@pytest.mark.parametrize("is_left", [False, True])
def test_matinv_simple(api, is_left):
    x0 = _x0_diag(2, 1, 4)
    sol = _solve(api, lambda x, M: M.batched_mat_inv_mul(x, None, is_left), x0, np.linalg.inv(x0), 2.0)
    _close(sol, x0 * 0.5)


@pytest.mark.parametrize("is_left,t_dst", [(True, 1.3), (False, 1.2)])
def test_matinv_mul(api, is_left, t_dst):
    x0 = _x0_diag(3, 1, 4)
    xinv = np.linalg.inv(x0)
    y0 = (x0 ** 1.5) @ xinv if is_left else xinv @ (x0 ** 1.5)
    sol = _solve(api, lambda x, M: M.batched_mat_inv_mul(x, x.pow(1.5), is_left), x0, y0, t_dst)
    _close(sol, x0 * t_dst ** 2)


# ------------------------------------------------------------------------------------- Symbolic.ElemArith (:235-299)
ELEM = {
    "add": (lambda x, M: x + x.pow(1.5), lambda x: x + x ** 1.5, 2.0),
    "sub": (lambda x, M: x.pow(5. / 3.) - x.pow(1.5), lambda x: x ** (5. / 3.) - x ** 1.5, 2.0),
    "mul": (lambda x, M: x.pow(-.75) * x.pow(1.5), lambda x: x ** -0.75 * x ** 1.5, 1.42),
    "mul-bcast": (lambda x, M: x.pow(2.3) * x.reduce_sum(-1),
                  lambda x: x ** 2.3 * x.sum(axis=(1, 2), keepdims=True), 2.0),
    "mul-bcast-fullgy": (lambda x, M: M.batched_mat_inv_mul(
        x.pow(1.2) * x.reduce_sum(-1) + x.reduce_sum(-1).batched_mul_eye(4), None, False),
        lambda x: np.linalg.inv(x ** 1.2 * x.sum(axis=(1, 2), keepdims=True)
                                + x.sum(axis=(1, 2), keepdims=True) * np.eye(4)), 2.0),
}


@pytest.mark.parametrize("name", list(ELEM))
def test_elem_arith(api, name):
    build, f, t_dst = ELEM[name]
    x0 = _rng(4, 2, 5)((9, 4, 4))
    y0 = f(x0)
    sol = _solve(api, build, x0, y0, t_dst)
    _close(f(sol), y0 * t_dst)


# ------------------------------------------------------------------------------ Symbolic.LinearCombination (:301-322)
def test_linear_combination(api):
    f = lambda x: x ** 1.5 * 1.4 + x ** (2.0 / 3) * 2.3 + x.sum(axis=(1, 2), keepdims=True) * 1.2 + 2.5
    build = lambda x, M: M.linear_combine([(1.2, x.reduce_sum(-1)), (2.3, x.pow(2. / 3.)), (1.4, x.pow(1.5))], 2.5)
    x0 = _rng(5, 2, 5)((9, 4, 4))
    y0 = f(x0)
    sol = _solve(api, build, x0, y0, 2.0)
    _close(f(sol), y0 * 2)


# ----------------------------------------------------------------------------------- Symbolic.Determinant (:324-360)
@pytest.mark.parametrize("name,batch,m", [("small", 1, 3), ("mid", 10, 4), ("large0", 10, 5), ("large1", 10, 7)])
def test_determinant(api, name, batch, m):
    """det(x) x followed from t = 1 to 2: the expansion (dim <= 4) and the DFT (dim > 4) of the determinant's series
    inside the order loop"""
    x0 = _rng(6 + m)((batch, m, m))
    f = lambda x: np.linalg.det(x)[:, None, None] * x
    y0 = f(x0)
    sol = _solve(api, lambda x, M: x.batched_det() * x, x0, y0, 2.0)
    _close(f(sol), y0 * 2, margin=1e-9)


# ---------------------------------------------------------------------------------------- Symbolic.Reduce (:362-387)
@pytest.mark.parametrize("name,shape,axis", [("axis", (10, 5), 1), ("flatten", (8, 9, 7), -1)])
def test_reduce(api, name, shape, axis):
    x0 = _rng(7)(shape)
    ax = tuple(range(1, len(shape)))
    f = lambda x: x.sum(axis=ax, keepdims=True) * x ** -2.0
    y0 = f(x0)
    sol = _solve(api, lambda x, M: x.reduce_sum(axis) * x.pow(-2), x0, y0, 2.0)
    _close(f(sol), y0 * 2, margin=1e-9)


# ----------------------------------------------------------------- Symbolic.Transpose / TransMul (:389-424)
def test_transpose(api):
    x0 = _rng(8, 1, 2)((5, 4, 6))
    f = lambda x: np.swapaxes(x ** 1.5, 1, 2)
    y0 = f(x0)
    sol = _solve(api, lambda x, M: x.pow(1.5).batched_transpose(), x0, y0, 2.0)
    _close(f(sol), y0 * 2)


def test_trans_mul(api):
    x0 = _rng(9)((5, 4, 6))
    f = lambda x: x * (x @ np.swapaxes(x, 1, 2)).sum(axis=(1, 2), keepdims=True)
    y0 = f(x0)
    sol = _solve(api, lambda x, M: x.batched_matmul(x.batched_transpose()).reduce_sum(-1) * x, x0, y0, 2.0)
    _close(f(sol), y0 * 2, margin=1e-9)


# --------------------------------------------------------------------------------------- Symbolic.IORemap (:426-523)
def _rand_sparse(rng, nr_in, nr_out):
    """RandSparseLinearDesc (tests/symbolic.cpp:428-489): rows of 2 (more outputs than inputs) or nr_in / nr_out + 1
    entries over a shuffled cycle of the inputs, one empty row when there are more outputs than inputs, rows that
    name the same input twice"""
    entry = 2 if nr_out > nr_in else nr_in // nr_out + 1
    perm, pos = rng.permutation(nr_in), 0
    rows, cols, vals = [], [], []
    empty_used = identical_used = False
    for i in range(nr_out):
        if nr_out > nr_in and not empty_used and (rng.integers(9) == 0 or i == nr_out - 1):
            empty_used = True
            continue
        k = 2 if rng.integers(5) == 0 else entry
        same = k == 2 and rng.integers(2) == 0
        for q in range(k):
            if not (same and q == 1):
                if pos == nr_in:
                    perm, pos = rng.permutation(nr_in), 0
                j = perm[pos]
                pos += 1
            identical_used |= same and q == 1
            rows.append(i)
            cols.append(j)
            vals.append(rng.uniform(-1, 1))
    m = sp.coo_matrix((vals, (rows, cols)), shape=(nr_out, nr_in))
    return m, empty_used, identical_used


@pytest.mark.parametrize("name,xshp,midshp", [("small", (2, 2), (4, 4)), ("large", (5, 11), (10, 4, 6))])
def test_io_remap(api, name, xshp, midshp):
    """x -> remap_in -> pow(., 2) -> remap_out with random sparse maps (an empty row, rows naming one input twice);
    the scipy matrices keep duplicate entries apart (COO -> CSR without summing) like the reference's list form"""
    rng = np.random.default_rng(11)
    nx, nmid = int(np.prod(xshp)), int(np.prod(midshp))
    for _ in range(50):  # (draw until the Jacobian at x0 is comfortably regular, as the reference's fixed seed is)
        rin, e_in, i_in = _rand_sparse(rng, nx, nmid)
        rout, e_out, i_out = _rand_sparse(rng, nmid, nx)
        x0 = rng.uniform(-1, 1, xshp)
        J = rout.tocsr() @ sp.diags(2 * (rin.tocsr() @ x0.ravel())) @ rin.tocsr()
        if e_in and i_in and not e_out and np.linalg.cond(J.toarray()) < 1e4:
            break
    else:
        pytest.skip("no well-conditioned draw")
    rin_csr, rout_csr = sp.csr_matrix(rin), sp.csr_matrix(rout)
    f = lambda x: (rout_csr @ (rin_csr @ x.ravel()) ** 2).reshape(xshp)
    y0 = f(x0)
    sol = _solve(api, lambda x, M: x.pow(2), x0, y0, 2.0, remap_in=rin_csr, remap_out=rout_csr, mid=midshp)
    _close(f(sol), y0 * 2, margin=1e-9)


# -------------------------------------------------------------------------------------- Symbolic.Constant (:525-556)
def test_constant(api):
    r = _rng(12)
    x0, cval = r((5, 4, 4)), r((5, 1))
    f = lambda x: (x.sum(axis=(1, 2), keepdims=True) * np.eye(4) + x) * cval[:, :, None]

    def build(x, M):
        c = M.constant(x.graph if M is A else x.var.graph, cval)
        return (x.reduce_sum(-1).batched_mul_eye(4) + x) * c
    y0 = f(x0)
    sol = _solve(api, build, x0, y0, 2.0)
    _close(f(sol), y0 * 2, margin=1e-9)


# -------------------------------------------------------------------------------------- Symbolic.Analytic (:558-581)
def test_analytic_log(api):
    x0 = _rng(13, 0.1, 2.5)((10, 20))
    y0 = np.log(x0)
    sol = _solve(api, lambda x, M: x.log(), x0, y0, 2.0)
    _close(np.log(sol), y0 * 2, margin=1e-9)


def test_long_vectors_and_larger_matrices(api):
    """beyond the reference's own test sizes: (batch, 200) vectors and 12 x 12 matrices through the elementwise
    operators, a transpose and a product (one thread per element: up to 256 elements per batch item)"""
    x0 = _rng(18, 1.2, 2.0)((3, 200))  # (away from the fold of log(x) x^1.5 at x = exp(-2/3))
    y0 = np.log(x0) * x0 ** 1.5
    sol = _solve(api, lambda x, M: x.log() * x.pow(1.5), x0, y0, 2.0)
    _close(np.log(sol) * sol ** 1.5, y0 * 2, margin=1e-9)
    x0 = _rng(19, 0.5, 1.5)((2, 12, 12))
    f = lambda x: x * (x @ np.swapaxes(x, 1, 2)).sum(axis=(1, 2), keepdims=True)  # cubic: the path is x0 t^(1/3)
    y0 = f(x0)
    sol = _solve(api, lambda x, M: x.batched_matmul(x.batched_transpose()).reduce_sum(-1) * x, x0, y0, 2.0)
    _close(f(sol), y0 * 2, margin=1e-9)
    _close(sol, x0 * 2 ** (1 / 3), margin=1e-9)


# --------------------------------------------------------------------------------- Symbolic.GeneralSolve (:583-638)
def _general_solve(api, build, x0, y, maxiter=20):
    """anm_general_solve (tests/symbolic.cpp:56-73): ANMEqnSolver{f, id, id, x0, -y} iterated until converged"""
    x0 = np.asarray(x0, dtype=float)
    g = api.graph()
    f = build(_placeholder(g, x0.shape[1:]), A)
    ident = A.SparseLinearDesc(api, sp.identity(x0.size, format="csr"))
    sol = A.ANMEqnSolver(api, f, ident, ident, x0, -np.asarray(y, dtype=float), api.default_hyper(order=8, use_pade=0))
    it = 0
    while not sol.converged():
        it += 1
        assert it <= maxiter
        sol.next_iter()
    return sol.get_x().reshape(x0.shape)


@pytest.mark.parametrize("name", ["sqr", "pow-log-pow"])
def test_general_solve(api, name):
    r = np.random.default_rng(14)
    if name == "sqr":
        build, f, lo, hi = (lambda x, M: x * x), (lambda x: x * x), 0.2, 1.5
    else:
        build, f, lo, hi = (lambda x, M: x.pow(2.3).log().pow(1.5)), (lambda x: np.log(x ** 2.3) ** 1.5), 1.5, 4.3
    xsol = r.uniform(lo, hi, (10, 20))
    ysol = f(xsol)
    xinit = xsol * r.uniform(0.6, 1.5, xsol.shape)
    xt = _general_solve(api, build, xinit, ysol)
    _close(f(xt), ysol)


@pytest.mark.parametrize("exp", [2, 5, 6, 8, 15])
def test_general_solve_pow_with_zero_gradient(api, exp):
    """pow-zg (tests/symbolic.cpp:611-628): x^1.7 + log(x)^exp from a start with one entry at exactly 1 -- an integer
    power of a series through zero, which continues on the convolution path (analytic_unary.cpp:46-92)"""
    r = np.random.default_rng(16)
    f = lambda x: x ** 1.7 + np.log(x) ** exp
    xsol = r.uniform(0.8, 1.5, (10, 8, 3))
    ysol = f(xsol)
    xinit = xsol * r.uniform(0.8, 1.2, xsol.shape)
    xinit.flat[2] = 1.0
    xt = _general_solve(api, lambda x, M: x.pow(1.7) + x.log().pow(exp), xinit, ysol)
    _close(f(xt), ysol)


# ----------------------------------------------------------------------------- Symbolic.PolarDecompSolve (:677-728)
def _svdw(x, rot):
    from oracle import tensor_ops as T_
    return T_.batched_svd_w(x, rot)


def _polar_case(seed, rot, eq_singular):
    r = np.random.default_rng(seed)
    batch, n = 7, 4
    x0 = r.uniform(-1, 1, (batch, n, n))
    dx = r.uniform(-0.05, 0.05, x0.shape)
    if eq_singular:  # make_eq_singular (:698-709): s_1 := s_0, M = U S U' W
        u, sv, w = _svdw(x0, rot)
        sv = sv.copy()
        sv[:, 1] = sv[:, 0]
        x0 = np.einsum("bij,bj,bkj,bkl->bil", u, sv, u, w)
    xsol = x0 + dx
    return x0, xsol, xsol - _svdw(xsol, rot)[2]


@pytest.mark.parametrize("name,seed,rot,eq_singular", [
    ("simple", 17, False, False), ("simple-rot", 17, True, False),
    ("eqs-x0", 22, False, True), ("eqs-x0-rot", 22, True, True), ("eqs-x0-b", 27, False, True),
    ("eqs-x0-slow", 26, True, True)])
def test_polar_decomp_solve(api, name, seed, rot, eq_singular):
    """x - W(x) = y solved from x0 = xsol - dx for 4 x 4 matrices (SVD-W outside 3 x 3, only W read: the polar
    recurrences on the vector interpreter), also from starts whose two largest singular values coincide.  How fast
    those converge depends on the draw (the reference remarks on it, :719); the oracle's ANMEqnSolver runs beside the
    device on the same input: same number of iterations, same residuals."""
    from oracle.anm import ANMEqnSolver as OEqn
    x0, xsol, ysol = _polar_case(seed, rot, eq_singular)
    ident = sp.identity(x0.size, format="csr")
    build = lambda x, M: x - x.batched_svd_w(rot)[2]
    g = api.graph()
    di = A.SparseLinearDesc(api, ident)
    dsol = A.ANMEqnSolver(api, build(g.placeholder_matrix(4, 4), A), di, di, x0, -ysol,
                          api.default_hyper(order=8, use_pade=0))
    osol = OEqn(build(S.placeholder(S.ComputingGraph()), S), ident, ident, x0.shape, x0, -ysol.ravel(),
                OHyper(order=8, use_pade=False))
    it = 0
    while not dsol.converged():
        assert not osol.converged
        assert dsol.residual_rms() == pytest.approx(osol.residual_rms, rel=1e-5)
        it += 1
        assert it <= 40  # anm_general_solve(..., 40)
        dsol.next_iter()
        osol.next_iter()
    assert osol.converged
    _close(dsol.get_x().reshape(x0.shape), xsol, margin=1e-7)
    assert np.abs(dsol.get_x() - osol.get_x()).max() <= 1e-7


def test_polar_decomp_solve_failing_draw(api):
    """a draw on which the reference algorithm itself gives up (the order-8 coefficient fails the orthogonality check
    of anm.cpp:271-285): the device reports the same assertion"""
    from oracle.anm import ANMEqnSolver as OEqn
    x0, xsol, ysol = _polar_case(23, True, True)
    ident = sp.identity(x0.size, format="csr")
    build = lambda x, M: x - x.batched_svd_w(True)[2]
    with pytest.raises(AssertionError, match="xdot"):
        o = OEqn(build(S.placeholder(S.ComputingGraph()), S), ident, ident, x0.shape, x0, -ysol.ravel(),
                 OHyper(order=8, use_pade=False))
        for _ in range(60):
            o.next_iter()
    g = api.graph()
    di = A.SparseLinearDesc(api, ident)
    with pytest.raises(A.SanmAssertionError, match="xdot"):
        d = A.ANMEqnSolver(api, build(g.placeholder_matrix(4, 4), A), di, di, x0, -ysol,
                           api.default_hyper(order=8, use_pade=0))
        for _ in range(60):
            d.next_iter()


# --------------------------------------------------------------------------- Symbolic.ANMImplicitSolver (:775-833)
def test_implicit_solver(api):
    """pow(x + t dx, 1.5) held at pow(x0, 1.5) while t goes from 0 to 1: the input map carries t as its last column"""
    batch = 5
    r = np.random.default_rng(15)
    x0, dx = r.uniform(1, 2, batch), r.uniform(-2, -1, batch)
    rin = sp.hstack([sp.identity(batch), sp.csr_matrix(dx[:, None])], format="csr")  # x_i + dx_i t
    g = api.graph()
    y = g.placeholder_vector(1).pow(1.5)
    dri = A.SparseLinearDesc(api, rin)
    dro = A.SparseLinearDesc(api, sp.identity(batch, format="csr"))
    sol = A.ANMImplicitSolver(api, y, dri, dro, x0, 0.0, api.default_hyper(order=8, use_pade=0))
    it = 0
    while sol.get_t_upper() < 1:
        it += 1
        assert it < 20
        sol.update_approx()
    xt, t = sol.eval(sol.solve_a(1.0))
    assert t == pytest.approx(1.0, rel=APPROX)
    _close((xt + dx) ** 1.5, x0 ** 1.5)


# ------------------------------------------------------------------------------ Symbolic.PaperGeoExample (:835-900)
def geo_graph(coord, M):
    x, y = coord.slice(1, 0, 1), coord.slice(1, 1, 2)
    f0 = M.linear_combine([(2.0, x.pow(2)), (-5.0, x), (1.0, y.pow(2)), (-4.0, y), (-2.0, x * y)], -5)
    f1 = (x + 1).pow(2) + y.pow(2) - 2
    return M.concat([f0, f1], 1)


def test_paper_geo_example(api):
    """the two-conic example of the paper: f(coord) + t df = 0 from (0, -1) at t = 0 to t = 1 with order 20 -- Slice
    and Concat inside the ANM order loop; the end point satisfies f = (0, 6) (print_err, :884-890)"""
    coord0, df = np.array([[0.0, -1.0]]), np.array([[0.0, -6.0]])
    ident = sp.identity(2, format="csr")
    g = api.graph()
    f_all = geo_graph(g.placeholder_vector(2), A)
    dri, dro = A.SparseLinearDesc(api, ident), A.SparseLinearDesc(api, ident)
    dsol = A.ANMSolverVecScale(api, f_all, dri, dro, coord0, 0.0, df, api.default_hyper(order=20, use_pade=0))
    osol = OVecScale(geo_graph(S.placeholder(S.ComputingGraph()), S), ident, ident, (1, 2), coord0, 0.0, df.ravel(),
                     OHyper(order=20, use_pade=False))
    it = 0
    while True:
        dco = dsol.xt_coeffs()
        for k in range(21):
            co, cd = osol.xt_coeffs[k], dco[k]
            assert np.abs(cd - co).max() <= 1e-7 * max(1e-300, np.abs(co).max()), (it, k)
        assert dsol.get_t_upper() == pytest.approx(osol.get_t_upper(), rel=1e-5)
        if dsol.get_t_upper() >= 1:
            break
        it += 1
        assert it < 20
        dsol.update_approx()
        osol.update_approx()
    sol, t = dsol.eval(dsol.solve_a(1.0))
    assert t == pytest.approx(1.0, rel=APPROX)
    x, y = sol
    err = np.array([2 * x * x - 5 * x + y * y - 4 * y - 2 * x * y - 5, (x + 1) ** 2 + y * y - 2 - 6])
    assert np.sqrt((err ** 2).mean()) < 1e-4
    so, _ = osol.eval(osol.solve_a(1.0))
    assert np.abs(sol - so).max() <= 1e-6


def test_sharded_solver_rejects_vector_graphs(api):
    g = api.graph()
    y = g.placeholder_matrix(4, 4).pow(2)
    ident = A.SparseLinearDesc(api, sp.identity(16, format="csr"))
    with pytest.raises(A.SanmUnsupportedError):
        A.ANMEqnSolver(api, y, ident, ident, np.ones(16), -np.ones(16), api.default_hyper(order=4), shard=(0, 2, None))


@pytest.mark.gpu
def test_dense_lu_by_panels_gives_the_single_workgroups_factors(monkeypatch):
    """systems of graphs on the vector interpreter are solved by a dense LU with partial pivoting; from 384 unknowns on
    it runs by 32-column panels with the trailing matrix over the whole chip (backend_hip.hip, round 5: the single
    workgroup was the only path up to 4096 unknowns).  Every element receives the unblocked loop's updates in the
    unblocked order, so the factors -- and with them the whole continuation -- are the single workgroup's bit for
    bit: SANM_DENSE_BLOCKED_MIN_N = 0 (panels always) against a huge value (never), on a (12, 5, 5) matrix graph (300
    unknowns: ten panels, a partial last one) and on random sparse remaps with zero diagonal entries (row interchanges
    in every panel) of 120 and of 400 unknowns (the latter beyond the default threshold)."""
    import sanm_amd
    api = sanm_amd.get_api(0)

    def remap_case(seed, nx, midshp):
        rng = np.random.default_rng(seed)
        nmid = int(np.prod(midshp))
        for _ in range(50):
            rin, e_in, i_in = _rand_sparse(rng, nx, nmid)
            rout, e_out, i_out = _rand_sparse(rng, nmid, nx)
            x1 = rng.uniform(-1, 1, (nx,))
            J = rout.tocsr() @ sp.diags(2 * (rin.tocsr() @ x1.ravel())) @ rin.tocsr()
            if not e_out and np.linalg.cond(J.toarray()) < 1e5:
                break
        else:
            pytest.skip("no well-conditioned draw")
        rin_csr, rout_csr = sp.csr_matrix(rin), sp.csr_matrix(rout)
        g = lambda x: (rout_csr @ (rin_csr @ x.ravel()) ** 2).reshape((nx,))
        return lambda: _solve(api, lambda x, M: x.pow(2), x1, g(x1), 1.5, remap_in=rin_csr, remap_out=rout_csr, mid=midshp,
                              oracle=False)

    x0 = _x0_diag(5, -1, 1, batch=12, m=5)
    cases = [lambda: _solve(api, lambda x, M: M.batched_mat_inv_mul(x, None, True), x0, np.linalg.inv(x0), 1.5, oracle=False),
             remap_case(11, 120, (1, 150)), remap_case(12, 400, (2, 225))]
    monkeypatch.setenv("SANM_DENSE_BLOCKED_MIN_N", "1000000")
    ref = [c() for c in cases]
    monkeypatch.setenv("SANM_DENSE_BLOCKED_MIN_N", "0")
    got = [c() for c in cases]
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    monkeypatch.delenv("SANM_DENSE_BLOCKED_MIN_N")
    assert np.array_equal(cases[2](), ref[2])
