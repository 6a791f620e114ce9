"""Slice / Concat and graphs over batched vectors (libsanm/oprs/misc.cpp:104-331) through the C ABI: the vector
interpreter of the device path (sanm_amd/csrc/vecprog.h) against the oracle's restatement of the same operator metas
and against the reference's known answer -- the Rosenbrock gradient {515.4, -285.4, -341.6, 2085.4, -482} of
tests/symbolic.cpp:756-763, built with the very slice / concat calls of tests/symbolic.cpp:730-745."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import symbolic as S
from sanm_amd import api as A


def rosen_der(x, lc, cat):
    """tests/symbolic.cpp:730-745, for either operator API (lc = linear_combine, cat = concat)"""
    xm, xm_m1, xm_p1 = x.slice(1, 1, -1), x.slice(1, None, -2), x.slice(1, 2, None)
    x0, x1, xp1, xp2 = x.slice(1, 0, 1), x.slice(1, 1, 2), x.slice(1, -1, None), x.slice(1, -2, -1)
    der0 = lc([(-400.0, x0 * (x1 - x0.pow(2))), (2.0, x0)], -2)
    der1 = lc([(200.0, xm), (-200.0, xm_m1.pow(2)), (-400.0, (xm_p1 - xm.pow(2)) * xm), (2.0, xm)], -2)
    der2 = lc([(200.0, xp1), (-200.0, xp2.pow(2))])
    return cat([der0, der1, der2], 1)


def _device(api, build, n, batch, order):
    g = api.graph()
    y = build(g.placeholder_vector(n), A.linear_combine, A.concat)
    ident = A.SparseLinearDesc(api, sp.identity(batch * n, format="csr"))
    return A.TaylorCoeffProp(api, y, ident, order, batch, in_size=n), (g, ident)


def _oracle(build):
    return S.TaylorCoeffProp(build(S.placeholder(S.ComputingGraph()), S.linear_combine, S.concat))


def test_rosenbrock_gradient_kat(api):
    """the reference's known answer, whole through the product path"""
    x0 = np.array([[1.3, 0.7, 0.8, 1.9, 1.2]])
    prop, keep = _device(api, rosen_der, 5, 1, 1)
    y = prop.push_xi(x0)
    assert y.shape == (1, 5)
    assert np.allclose(y, [[515.4, -285.4, -341.6, 2085.4, -482.0]], rtol=1e-12)
    assert np.allclose(y, _oracle(rosen_der).push_xi([x0]), rtol=1e-14)


@pytest.mark.parametrize("batch", [1, 3])
def test_taylor_series_and_jacobian_of_vector_graphs(api, batch):
    """orders 0..6 and the Jacobian of the Rosenbrock gradient graph against the oracle (the batch > 1 variant runs
    independent rows: the oracle is fed one row at a time, its Slice / Concat being batch-1 like the reference's)"""
    N, n = 6, 5
    rng = np.random.default_rng(3)
    xs = [np.array([1.3, 0.7, 0.8, 1.9, 1.2]) + 0.1 * rng.standard_normal((batch, n))] + \
         [0.3 * rng.standard_normal((batch, n)) for _ in range(N)]
    prop, keep = _device(api, rosen_der, n, batch, N)
    oprops = [_oracle(rosen_der) for _ in range(batch)]
    y = prop.push_xi(xs[0])
    yo = np.concatenate([o.push_xi([xs[0][b:b + 1]]) for b, o in enumerate(oprops)])
    assert np.allclose(y, yo, rtol=1e-13)
    J = prop.get_jacobian()
    Jo = np.concatenate([o.get_jacobian() for o in oprops])
    assert J.shape == (batch, 5, 5) and np.allclose(J, Jo, rtol=1e-12, atol=1e-12)
    ys = [y]
    for k in range(1, N + 1):
        b = prop.compute_next_order_bias()
        bo = np.concatenate([o.compute_next_order_bias() for o in oprops])
        if k == 1:
            assert not np.any(b)  # symbolic.cpp:278-285
        assert np.allclose(b, bo, rtol=1e-10, atol=1e-9 * max(1.0, np.abs(bo).max()))
        yk = prop.push_xi(xs[k])
        yko = np.concatenate([o.push_xi([xs[k][q:q + 1]]) for q, o in enumerate(oprops)])
        assert np.allclose(yk, yko, rtol=1e-10, atol=1e-9 * max(1.0, np.abs(yko).max()))
        # coefficient = bias + Jacobian . x_k (what the ANM loop relies on)
        assert np.allclose(b + np.einsum("bij,bj->bi", J, xs[k]), yk, rtol=1e-9, atol=1e-8 * max(1.0, np.abs(yk).max()))
        ys.append(yk)
    a = 0.03
    direct, _ = _device(api, rosen_der, n, batch, 1)
    yd = direct.push_xi(sum(x * a ** k for k, x in enumerate(xs)))
    series = sum(v * a ** k for k, v in enumerate(ys))
    assert np.abs(yd - series).max() <= 1e-6 * np.abs(yd).max()


def circle_graph(coord, lc, cat):
    """the second user of slice / concat in the reference's tests (tests/symbolic.cpp:842-848): functions of the two
    entries of a coordinate pair, concatenated; here with log, a fractional power, a scalar broadcast and a
    reduction to cover the remaining elementwise operators on vectors"""
    x, y = coord.slice(1, 0, 1), coord.slice(1, 1, 2)
    r2 = coord.pow(2).reduce_sum(-1)            # (B,1) from (B,2)
    f0 = (x * y + r2.log()) * 0.5
    f1 = r2.pow(1.5) - x
    scaled = coord * r2                         # vector x batched scalar
    return cat([f0, f1, scaled], 1)


def test_other_elementwise_operators_on_vectors(api):
    N, batch = 5, 4
    rng = np.random.default_rng(11)
    xs = [np.array([0.9, -0.6]) + 0.1 * rng.standard_normal((batch, 2))] + [0.2 * rng.standard_normal((batch, 2))
                                                                          for _ in range(N)]
    prop, keep = _device(api, circle_graph, 2, batch, N)
    oprops = [_oracle(circle_graph) for _ in range(batch)]
    y = prop.push_xi(xs[0])
    yo = np.concatenate([o.push_xi([xs[0][b:b + 1]]) for b, o in enumerate(oprops)])
    assert y.shape == (batch, 4) and np.allclose(y, yo, rtol=1e-13)
    J, Jo = prop.get_jacobian(), np.concatenate([o.get_jacobian() for o in oprops])
    assert np.allclose(J, Jo, rtol=1e-12, atol=1e-12)
    for k in range(1, N + 1):
        b = prop.compute_next_order_bias()
        bo = np.concatenate([o.compute_next_order_bias() for o in oprops])
        assert np.allclose(b, bo, rtol=1e-10, atol=1e-10)
        yk = prop.push_xi(xs[k])
        yko = np.concatenate([o.push_xi([xs[k][q:q + 1]]) for q, o in enumerate(oprops)])
        assert np.allclose(yk, yko, rtol=1e-10, atol=1e-10)


def test_slice_concat_argument_rules(api):
    g = api.graph()
    x = g.placeholder_vector(5)
    assert x.slice(1, -2, None).id != x.id
    with pytest.raises(A.SanmUnsupportedError):      # misc.cpp:149-150: axis 1, stride 1 only
        x.slice(1, None, None, 2)
    with pytest.raises(A.SanmUnsupportedError):
        x.slice(2, 0, 1)
    with pytest.raises(A.SanmAssertionError):        # misc.cpp:128-131: begin < end inside the tensor
        x.slice(1, 3, 2)
    with pytest.raises(A.SanmAssertionError):
        x.slice(1, 0, 6)
    with pytest.raises(A.SanmUnsupportedError):
        A.concat([x, x], 0)
    # the linear-algebra operators take (batch, rows, cols) matrices, not vectors
    with pytest.raises(A.SanmAssertionError):
        x.batched_det()
    # the ANM drivers run vector graphs too (tests/test_generic_anm.py): x -> x through slice / concat, solved for 2
    y = A.concat([x.slice(1, 0, 2), x.slice(1, 2, None)], 1)
    ident = A.SparseLinearDesc(api, sp.identity(5, format="csr"))
    sol = A.ANMEqnSolver(api, y, ident, ident, np.ones(5), -2 * np.ones(5), api.default_hyper(order=4))
    for _ in range(5):
        if sol.converged():
            break
        sol.next_iter()
    assert sol.converged() and np.allclose(sol.get_x(), 2.0, rtol=1e-9)
    # ... and a non-integer power of a zero is the reference's numerical error here too
    prop = A.TaylorCoeffProp(api, x.pow(0.5), ident, 2, 1, in_size=5)
    with pytest.raises(A.SanmNumericalError):
        prop.push_xi(np.array([[1.0, 0.0, 2.0, 3.0, 4.0]]))


def dup_graph(x, lc, cat):
    """a variable listed twice in one concat, and once more beside a function of itself: the gradient of x is the sum
    over its occurrences (the device's reverse sweep once let two threads add into one slot: ADVICE r3)"""
    return cat([x, x.pow(2), x], 1)


def test_concat_with_a_repeated_operand(api):
    N, batch, n = 4, 3, 5
    rng = np.random.default_rng(3)
    xs = [rng.uniform(0.5, 1.5, (batch, n))] + [0.2 * rng.standard_normal((batch, n)) for _ in range(N)]
    prop, keep = _device(api, dup_graph, n, batch, N)
    oprops = [_oracle(dup_graph) for _ in range(batch)]
    y = prop.push_xi(xs[0])
    yo = np.concatenate([o.push_xi([xs[0][b:b + 1]]) for b, o in enumerate(oprops)])
    assert y.shape == (batch, 3 * n) and np.allclose(y, yo, rtol=1e-13)
    J, Jo = prop.get_jacobian(), np.concatenate([o.get_jacobian() for o in oprops])
    assert np.allclose(J, Jo, rtol=1e-12, atol=1e-12)
    # rows 0..n-1 and 2n..3n-1 are identity blocks, the middle one diag(2 x)
    for b in range(batch):
        assert np.allclose(J[b][:n], np.eye(n)) and np.allclose(J[b][2 * n:], np.eye(n))
        assert np.allclose(J[b][n:2 * n], np.diag(2 * xs[0][b]))
    for k in range(1, N + 1):
        bias = prop.compute_next_order_bias()
        bo = np.concatenate([o.compute_next_order_bias() for o in oprops])
        assert np.allclose(bias, bo, rtol=1e-10, atol=1e-10)
        yk = prop.push_xi(xs[k])
        yko = np.concatenate([o.push_xi([xs[k][q:q + 1]]) for q, o in enumerate(oprops)])
        assert np.allclose(yk, yko, rtol=1e-10, atol=1e-10)


def test_flat_constant_of_nine_values_follows_a_vector_operand(api):
    """A constant given as nine values without a shape is (3,3) by default -- the (T,3,3) constants of the FEA graphs --
    but in a graph over vectors of LENGTH 9 it must stay a vector: placeholder_vector(9) * constant((B,9)) followed by a
    slice was refused ('(batch, n) operands only') before round 4, and the reference accepts it."""
    B, n = 2, 9
    rng = np.random.default_rng(5)
    cval = rng.uniform(0.5, 1.5, (B, n))
    x0 = rng.uniform(0.5, 1.5, (B, n))
    g = api.graph()
    x = g.placeholder_vector(n)
    c = A.constant(g, cval)
    y = (x * c).slice(1, 2, 7)
    ident = A.SparseLinearDesc(api, sp.identity(B * n, format="csr"))
    prop = A.TaylorCoeffProp(api, y, ident, 2, B, in_size=n)
    out = prop.push_xi(x0)
    assert out.shape == (B, 5) and np.allclose(out, (x0 * cval)[:, 2:7], rtol=1e-14)
    J = prop.get_jacobian()
    for b in range(B):
        expect = np.zeros((5, n))
        expect[np.arange(5), np.arange(2, 7)] = cval[b, 2:7]
        assert np.allclose(J[b], expect)
