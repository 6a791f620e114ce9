"""Matrices of other sizes than 3 x 3 through the C ABI (libsanm/tensor_linalg.cpp:107-210 dynamic sizes,
tensor_polymat.cpp:30-136 / :325-379 determinant series): the graphs of the reference's own operator tests
(tests/symbolic.cpp:179-424, :640-656 -- 4 x 4 mat_inv_mul and elementwise arithmetic with batched_mul_eye(4),
determinants at 4, 5 and 7, transposes and products of 4 x 6 matrices, reduce over 9 x 7, log det(X^T X) of 4 x 3)
on the device's vector interpreter (sanm_amd/csrc/vecprog.h) against the oracle's restatement of the operator metas:
order-0 values, Jacobians, biases and coefficients up to order 6, and the series against a direct evaluation."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import symbolic as S
from oracle import tensor_ops as T_
from sanm_amd import api as A


def _mk_device(api, build, shape, batch, order):
    g = api.graph()
    x = g.placeholder_matrix(*shape) if len(shape) == 2 else g.placeholder_vector(shape[0])
    y = build(x, A)
    n = int(np.prod(shape))
    ident = A.SparseLinearDesc(api, sp.identity(batch * n, format="csr"))
    return A.TaylorCoeffProp(api, y, ident, order, batch, in_size=n), (g, ident)


def _mk_oracle(build):
    return S.TaylorCoeffProp(build(S.placeholder(S.ComputingGraph()), S))


# ---- the graphs (one function serves both operator APIs; `M` is the module providing linear_combine etc.) ----------
def g_inv_right(x, M):          # tests/symbolic.cpp:198-207 simple-inv (is_left = false)
    return M.batched_mat_inv_mul(x, None, False)


def g_inv_left(x, M):
    return M.batched_mat_inv_mul(x, None, True)


def g_inv_mul_left(x, M):       # :209-220
    return M.batched_mat_inv_mul(x, x.pow(1.5), True)


def g_inv_mul_right(x, M):      # :222-232
    return M.batched_mat_inv_mul(x, x.pow(1.5), False)


def g_bcast_fullgy(x, M):       # :286-298 (dim taken from the operand)
    xs = x.reduce_sum(-1)
    return M.batched_mat_inv_mul(x.pow(1.2) * xs + xs.batched_mul_eye(4), None, False)


def g_det_x(x, M):              # :331-346: det(x) * x
    return x.batched_det() * x


def g_transpose(x, M):          # :389-406
    return x.pow(1.5).batched_transpose()


def g_trans_mul(x, M):          # :408-424: sum(x x^T) * x
    return x.batched_matmul(x.batched_transpose()).reduce_sum(-1) * x


def g_reduce(x, M):             # :362-387 reduce-flatten
    return x.reduce_sum(-1) * x.pow(-2)


def g_logdet(x, M):             # :640-656
    return x.batched_transpose().batched_matmul(x).batched_det().log()


def g_polar(x, M):              # :658-675 PolarDecompTaylorProp: x - W of the rotation-fixed SVD-W
    return x - x.batched_svd_w(True)[2]


def g_polar_norot(x, M):
    return x - x.batched_svd_w(False)[2]


def g_axis_sums(x, M):          # reduce over one axis of a matrix (reduce.cpp:36-47): (rows x 1) (1 x cols) -> rows x cols
    return x.reduce_sum(2).batched_matmul(x.pow(2).reduce_sum(1)) + x


def g_lincomb(x, M):            # :301-322
    return M.linear_combine([(1.2, x.reduce_sum(-1)), (2.3, x.pow(2. / 3.)), (1.4, x.pow(1.5))], 2.5)


CASES = [
    # name, graph, placeholder shape, batch, value range, diagonal shift
    ("inv_right_4", g_inv_right, (4, 4), 9, (1.0, 4.0), 4.0),
    ("inv_left_4", g_inv_left, (4, 4), 9, (1.0, 4.0), 4.0),
    ("inv_mul_left_4", g_inv_mul_left, (4, 4), 9, (1.0, 4.0), 4.0),
    ("inv_mul_right_4", g_inv_mul_right, (4, 4), 9, (1.0, 4.0), 4.0),
    ("bcast_fullgy_4", g_bcast_fullgy, (4, 4), 9, (2.0, 5.0), 0.0),
    ("det_2", g_det_x, (2, 2), 3, (0.0, 1.0), 1.0),
    ("det_4", g_det_x, (4, 4), 10, (0.0, 1.0), 1.0),
    ("det_5", g_det_x, (5, 5), 10, (0.0, 1.0), 1.0),
    ("det_7", g_det_x, (7, 7), 10, (0.0, 1.0), 1.0),
    ("det_8", g_det_x, (8, 8), 2, (0.0, 1.0), 1.0),
    ("transpose_4x6", g_transpose, (4, 6), 5, (1.0, 2.0), 0.0),
    ("trans_mul_4x6", g_trans_mul, (4, 6), 5, (0.0, 1.0), 0.0),
    ("reduce_9x7", g_reduce, (9, 7), 8, (0.5, 1.5), 0.0),
    ("logdet_4x3", g_logdet, (4, 3), 10, (0.0, 1.0), 0.0),
    ("lincomb_4", g_lincomb, (4, 4), 9, (2.0, 5.0), 0.0),
    ("axis_sums_4x6", g_axis_sums, (4, 6), 5, (0.5, 1.5), 0.0),
    ("polar_4", g_polar, (4, 4), 7, (-1.0, 1.0), 0.0),
    ("polar_norot_4", g_polar_norot, (4, 4), 7, (-1.0, 1.0), 0.0),
    ("polar_5", g_polar, (5, 5), 4, (-1.0, 1.0), 0.0),
    ("polar_2", g_polar, (2, 2), 6, (-1.0, 1.0), 0.0),
]


def _inputs(shape, batch, lo, hi, shift, N, seed):
    rng = np.random.default_rng(seed)
    x0 = rng.uniform(lo, hi, (batch,) + shape)
    if shift:
        for i in range(min(shape)):
            x0[:, i, i] += shift
    return [x0] + [0.1 * (hi - lo + 0.2) * rng.standard_normal((batch,) + shape) for _ in range(N)]


@pytest.mark.parametrize("name,build,shape,batch,rng_,shift", CASES, ids=[c[0] for c in CASES])
def test_matrix_graph_series_against_oracle(api, name, build, shape, batch, rng_, shift):
    N = 6
    xs = _inputs(shape, batch, rng_[0], rng_[1], shift, N, seed=len(name))
    prop, keep = _mk_device(api, build, shape, batch, N)
    oprop = _mk_oracle(build)
    flat = lambda a: np.asarray(a).reshape(batch, -1)
    y = flat(prop.push_xi(xs[0]))
    yo = flat(oprop.push_xi([xs[0]]))
    assert y.shape == yo.shape
    scale = max(1.0, np.abs(yo).max())
    assert np.abs(y - yo).max() <= 1e-11 * scale
    J = prop.get_jacobian()
    Jo = np.asarray(oprop.get_jacobian()).reshape(J.shape)
    assert np.abs(J - Jo).max() <= 1e-10 * max(1.0, np.abs(Jo).max())
    ys = [y]
    n = int(np.prod(shape))
    for k in range(1, N + 1):
        b = flat(prop.compute_next_order_bias())
        bo = flat(oprop.compute_next_order_bias())
        if k == 1:
            assert not np.any(b)  # symbolic.cpp:278-285
        # (polar cases: every order divides by sums of singular values of matrices drawn from [-1, 1] -- round-off
        # grows with the order on both sides)
        tol = 1e-7 if name.startswith("polar") else 1e-9
        sk = max(1.0, np.abs(bo).max())
        assert np.abs(b - bo).max() <= tol * sk, (k, np.abs(b - bo).max(), sk)
        yk = flat(prop.push_xi(xs[k]))
        yko = flat(oprop.push_xi([xs[k]]))
        sk = max(1.0, np.abs(yko).max())
        assert np.abs(yk - yko).max() <= tol * sk, (k, np.abs(yk - yko).max(), sk)
        # coefficient = bias + Jacobian . x_k: what the ANM loop relies on (check_taylor_prop, tests/symbolic.cpp:96-103)
        # (SVD-W divides by s_i + s_j through clip_div's x y / (y^2 + 1e-12) in the Jacobian and in the recurrence --
        # tensor_svd.cpp:28-31 --, which agree to 1e-12 / (s_i + s_j)^2 only: the reference's own check_taylor_prop
        # compares the two at 1e-4)
        lin = b + np.einsum("bij,bj->bi", J, xs[k].reshape(batch, n))
        assert np.abs(lin - yk).max() <= (1e-5 if name.startswith("polar") else 1e-8) * sk
        ys.append(yk)
    # the series against a direct evaluation at a small a (check_taylor_prop's eps_eval leg, :131-137)
    a = 0.02
    direct, _ = _mk_device(api, build, shape, batch, 1)
    yd = flat(direct.push_xi(sum(x * a ** k for k, x in enumerate(xs))))
    series = sum(v * a ** k for k, v in enumerate(ys))
    assert np.abs(yd - series).max() <= 1e-6 * max(1.0, np.abs(yd).max())


def _spread(batch, n, seed):
    """order-0 matrices Q1 diag(n, ..., 2, 1) Q2: well separated singular values, so that dU/dM (which divides by
    s_j^2 - s_i^2, tensor_svd.cpp:236-262) is well conditioned and the comparison is not a test of clip_div"""
    rng = np.random.default_rng(seed)
    q1 = np.linalg.qr(rng.standard_normal((batch, n, n)))[0]
    q2 = np.linalg.qr(rng.standard_normal((batch, n, n)))[0]
    return q1 @ (np.diag(np.arange(n, 0, -1.0))[None] * rng.uniform(0.9, 1.1, (batch, 1, 1))) @ q2


@pytest.mark.parametrize("n", [2, 4, 5])
@pytest.mark.parametrize("case", ["U", "S", "USW", "S_vector_output"])
def test_svdw_full_recurrences_outside_3x3(api, case, n):
    """batched_svd_w of n x n matrices with U or S read by other operators: the full U, S, W recurrences
    (oprs/linalg.cpp:561-600, tensor_svd.cpp:275-387) and dU/dM, dS/dM (tensor_svd.cpp:147-273) on the vector
    interpreter, the graphs of tests/test_device_ops.py::test_svdw_full_mode at other sizes.  (The columns of U are
    defined up to a sign -- Eigen's JacobiSVD in the reference, LAPACK in the oracle, a one-sided Jacobi on the
    device --; U enters through its elementwise square, which is not.)"""
    batch, N = 6, 5

    def build(x, M):
        u, sv, w = x.batched_svd_w(False)
        uu = u * u
        if case == "U":
            return uu.batched_matmul(w) + uu.batched_transpose()
        if case == "S":
            return (sv * sv).reduce_sum(-1).batched_mul_eye(n).batched_matmul(w) + x
        if case == "USW":
            return uu.batched_matmul(w) * (sv.log() * sv).reduce_sum(-1) + uu
        return M.linear_combine([(2.0, sv), (0.5, sv.pow(2))], 0.25)
    rng = np.random.default_rng(n)
    xs = [_spread(batch, n, 7 + n)] + [0.1 * rng.standard_normal((batch, n, n)) for _ in range(N)]
    prop, keep = _mk_device(api, build, (n, n), batch, N)
    oprop = _mk_oracle(build)
    flat = lambda a: np.asarray(a).reshape(batch, -1)
    y, yo = flat(prop.push_xi(xs[0])), flat(oprop.push_xi([xs[0]]))
    assert y.shape == yo.shape and np.abs(y - yo).max() <= 1e-10 * max(1.0, np.abs(yo).max())
    J = prop.get_jacobian()
    Jo = np.asarray(oprop.get_jacobian()).reshape(J.shape)
    assert np.abs(J - Jo).max() <= 1e-9 * max(1.0, np.abs(Jo).max())
    for k in range(1, N + 1):
        b, bo = flat(prop.compute_next_order_bias()), flat(oprop.compute_next_order_bias())
        assert np.abs(b - bo).max() <= 1e-8 * max(1.0, np.abs(bo).max()), k
        yk, yko = flat(prop.push_xi(xs[k])), flat(oprop.push_xi([xs[k]]))
        assert np.abs(yk - yko).max() <= 1e-8 * max(1.0, np.abs(yko).max()), k
        lin = b + np.einsum("bij,bj->bi", J, xs[k].reshape(batch, n * n))
        assert np.abs(lin - yk).max() <= 1e-6 * max(1.0, np.abs(yk).max())


@pytest.mark.parametrize("m", [2, 3, 4, 5, 6, 7])
def test_oracle_polymat_det_coeff_against_evaluated_polynomial(m):
    """the reference's own check of compute_polymat_det_coeff (tests/tensor.cpp: det of the evaluated polynomial
    matrix against the polynomial of the coefficients) for both paths -- expansion (dim <= 4) and DFT (dim > 4)"""
    rng = np.random.default_rng(m)
    nc, batch = 4, 5
    cs = [rng.standard_normal((batch, m, m)) for _ in range(nc)]
    deg = (nc - 1) * m
    co = np.stack([T_.compute_polymat_det_coeff(cs, k)[:, 0] for k in range(deg + 2)])  # (deg+2, batch)
    assert not np.any(co[deg + 1])
    for a in (0.3, -0.7, 1.1):
        want = np.linalg.det(sum(c * a ** k for k, c in enumerate(cs)))
        got = sum(co[k] * a ** k for k in range(deg + 1))
        assert np.allclose(got, want, rtol=1e-9, atol=1e-9 * np.abs(co).max())


def test_shape_rules(api):
    g = api.graph()
    x = g.placeholder_matrix(4, 6)
    with pytest.raises(A.SanmAssertionError):
        x.batched_matmul(x)                       # (4,6) x (4,6)
    with pytest.raises(A.SanmAssertionError):
        x.batched_det()
    with pytest.raises(A.SanmAssertionError):
        A.batched_mat_inv_mul(x, None, False)
    assert x.batched_matmul(x.batched_transpose()).id >= 0
    with pytest.raises(A.SanmAssertionError):     # SVD-W: square matrices
        x.batched_svd_w(False)
    with pytest.raises(A.SanmUnsupportedError):   # the sum over the batch as well has no batched output
        x.reduce_sum(-2)
    assert x.reduce_sum(1).id >= 0 and x.reduce_sum(2).id >= 0
    y = g.placeholder_matrix(9, 9)                # larger than the interpreter's 8 x 8 linear algebra
    ident = A.SparseLinearDesc(api, sp.identity(81, format="csr"))
    with pytest.raises((A.SanmAssertionError, A.SanmUnsupportedError)):
        A.TaylorCoeffProp(api, y.batched_det() * y, ident, 2, 1, in_size=81)


@pytest.mark.parametrize("n,rank", [(4, 3), (4, 2), (5, 4), (5, 3), (2, 1)])
def test_cofactor_of_rank_deficient_matrices(api, n, rank):
    """Tensor.DetCofactor's rank-deficient leg (tests/tensor.cpp:416-498) at other sizes: the Jacobian of
    batched_det is the cofactor matrix -- non-zero at rank n - 1, exactly zero at rank <= n - 2 (where the
    reference's SVD formula applies its rank test, tensor_linalg.cpp:18-59, and the device takes determinants of the
    minors, which vanish by themselves)."""
    rng = np.random.default_rng(10 * n + rank)
    batch = 4
    x0 = rng.standard_normal((batch, n, rank)) @ rng.standard_normal((batch, rank, n))
    prop, keep = _mk_device(api, lambda x, M: x.batched_det(), (n, n), batch, 1)
    oprop = _mk_oracle(lambda x, M: x.batched_det())
    d, do = prop.push_xi(x0), oprop.push_xi([x0])
    assert np.abs(d).max() <= 1e-12 and np.abs(np.asarray(do)).max() <= 1e-12
    J = prop.get_jacobian().reshape(batch, n, n)
    Jo = np.asarray(oprop.get_jacobian()).reshape(batch, n, n)
    scale = np.abs(x0).max() ** (n - 1)
    assert np.abs(J - Jo).max() <= 1e-10 * scale
    if rank <= n - 2:
        assert np.abs(J).max() <= 1e-12 * scale and not np.any(Jo)
    else:
        assert np.abs(J).max() > 1e-3 * scale
