"""Batched small-matrix kernels of the reference's tensor layer, in numpy.

ORACLE -- test infrastructure only (see oracle/__init__.py).

All tensors are row-major fp64, batch first: matrices are ``(T, n, n)``,
batched scalars ``(T, 1)``, singular values ``(T, n)``.  The reference stores
the same logical layout (libsanm/tensor.h:120-492); its Eigen ``Map``s view the
row-major data as column-major, so every ``*T`` variable in
libsanm/tensor_svd.cpp / tensor_linalg.cpp is the transpose of the logical
matrix.  The formulas below are written on the logical matrices.
"""
from __future__ import annotations

import itertools

import numpy as np

CLIP_EPS = 1e-12  # libsanm/tensor_svd.cpp:28-31


def clip_div(x, y):
    """x*y/(y*y+eps): libsanm/tensor_svd.cpp:28-31."""
    return x * y / (y * y + CLIP_EPS)


# --------------------------------------------------------------------------
# libsanm/tensor_linalg.cpp
# --------------------------------------------------------------------------
def batched_mm(a, b, trans_a=False, trans_b=False):
    """C[b] = op(A[b]) op(B[b]); libsanm/tensor_linalg.cpp:107-210."""
    if trans_a:
        a = np.swapaxes(a, -1, -2)
    if trans_b:
        b = np.swapaxes(b, -1, -2)
    return np.matmul(a, b)


def batched_transpose(a):
    """libsanm/tensor_linalg.cpp:212-283."""
    return np.ascontiguousarray(np.swapaxes(a, -1, -2))


def batched_matinv(a):
    """libsanm/tensor_linalg.cpp:285-317 (Eigen ``inverse()``)."""
    return np.linalg.inv(a)


def batched_determinant(a):
    """(T,n,n) -> (T,1); libsanm/tensor_linalg.cpp:319-353."""
    if a.shape[-1] == 3:
        # closed form, same expansion order as a cofactor expansion on row 0
        d = (a[:, 0, 0] * (a[:, 1, 1] * a[:, 2, 2] - a[:, 1, 2] * a[:, 2, 1])
             - a[:, 0, 1] * (a[:, 1, 0] * a[:, 2, 2] - a[:, 1, 2] * a[:, 2, 0])
             + a[:, 0, 2] * (a[:, 1, 0] * a[:, 2, 1] - a[:, 1, 1] * a[:, 2, 0]))
        return d[:, None]
    return np.linalg.det(a)[:, None]


def batched_cofactor(a):
    """Cofactor matrix through an SVD with a rank test.

    libsanm/tensor_linalg.cpp:18-59, :355-392.  cof(M) = det(U V') * U
    diag(prod_{j!=i} s_j) V'; if rank(M) <= n-2 the result is the zero matrix.
    Eigen's JacobiSVD::rank() counts singular values above
    ``max(s) * n * eps`` (its default threshold).
    """
    T, n, _ = a.shape
    u, s, vt = np.linalg.svd(a)
    thr = s[:, :1] * (n * np.finfo(np.float64).eps)
    rank = (s > thr).sum(axis=1)
    rank = np.where(s[:, 0] == 0, 0, rank)
    sinvd = np.empty_like(s)
    for i in range(n):
        sinvd[:, i] = np.prod(np.delete(s, i, axis=1), axis=1)
    sign = np.linalg.det(np.matmul(u, vt))
    sinvd = np.where(sign[:, None] < 0, -sinvd, sinvd)
    ret = np.matmul(u * sinvd[:, None, :], vt)
    ret[rank + 2 <= n] = 0
    return ret


def batched_mm_vecitem_left(l, r, trans_r=False):
    """out[b,(m,n),p] = sum_k l[b,(m,k),p] r[b,k,n] (or r[b,n,k] if trans_r).

    libsanm/tensor_linalg.cpp:394-434.  ``l`` is (T, M*K, P), ``r`` is
    (T, K, N); result (T, M*N, P).
    """
    T, mk, p = l.shape
    if trans_r:
        r = np.swapaxes(r, -1, -2)
    k, n = r.shape[1], r.shape[2]
    m = mk // k
    l4 = l.reshape(T, m, k, p)
    return np.einsum("bmkp,bkn->bmnp", l4, r).reshape(T, m * n, p)


# --------------------------------------------------------------------------
# libsanm/tensor_svd.cpp
# --------------------------------------------------------------------------
def _rotation_fix_index(s_row, n, eps=1e-3):
    """Literal restatement of the singular-value selection loop.

    libsanm/tensor_svd.cpp:88-128.  Note the ``i = j`` inside a ``for(...;
    ++i)`` loop: after a group [i, j) the scan resumes at j+1.
    Returns the list of indices whose singular value / U column is negated.
    """
    best_idx, best_idx_nr = -1, n + 1
    i = 0
    while i < n:
        j = i + 1
        while j < n and abs(s_row[i] - s_row[j]) < eps:
            j += 1
        nr = j - i
        if nr <= best_idx_nr or (nr == best_idx_nr + 1 and nr % 2 == 1):
            best_idx, best_idx_nr = i, nr
            if nr == 1:
                break
        i = j
        i += 1
    if best_idx_nr == 1 or best_idx_nr % 2 == 0:
        return [best_idx]
    return list(range(best_idx, best_idx + best_idx_nr))


def batched_svd_w(m, require_rotation=False):
    """M = U S U' W with U orthogonal, W = U V'.

    libsanm/tensor_svd.cpp:48-145.  Singular values are sorted descending
    (Eigen JacobiSVD).  With ``require_rotation`` and det(M) < 0 some singular
    values (and the matching U columns) are negated so that det(W) = +1, using
    the reference's own selection rule.
    Returns (U (T,n,n), S (T,n), W (T,n,n)).
    """
    T, n, _ = m.shape
    u, s, vt = np.linalg.svd(m)
    u = u.copy()
    s = s.copy()
    if require_rotation:
        du = np.linalg.det(u) < 0
        dv = np.linalg.det(vt) < 0
        for b in np.nonzero(du != dv)[0]:
            for i in _rotation_fix_index(s[b], n):
                s[b, i] = -s[b, i]
                u[b, :, i] = -u[b, :, i]
    w = np.matmul(u, vt)
    return u, s, w


def svd_w_jacobians(u, s, w, need_u=True, need_s=True, need_w=True):
    """Analytic dU/dM, dS/dM, dW/dM, each (T, rows, n*n), row-major flatten.

    libsanm/tensor_svd.cpp:147-273 (the per-batch FOR4 loop); same formulas as
    the reference's utils/test_svdw_grad.py:27-46 with clip_div in place of
    the plain divisions.
    """
    T, n, _ = u.shape
    v = np.matmul(np.swapaxes(w, 1, 2), u)  # V = W' U
    ut = np.swapaxes(u, 1, 2)
    du = ds = dw = None
    if need_s:
        # dsdm[i, (j,k)] = U'[i,j] V[k,i]
        ds = np.einsum("bij,bki->bijk", ut, v).reshape(T, n, n * n)
    if need_u or need_w:
        # cij[i,j,k,l] = U'[i,k] V[l,j] ; cji = U'[j,k] V[l,i]
        cij = np.einsum("bik,blj->bijkl", ut, v)
        cji = np.einsum("bjk,bli->bijkl", ut, v)
        si = s[:, :, None, None, None]
        sj = s[:, None, :, None, None]
        offdiag = 1.0 - np.eye(n)[None, :, :, None, None]
    if need_w:
        dydm = clip_div(cij - cji, si + sj) * offdiag
        dwdy = np.einsum("bik,bjl->bijkl", u, v)  # dwdy[(i,j),(k,l)] = U[i,k] V[j,l]
        dw = np.matmul(dwdy.reshape(T, n * n, n * n), dydm.reshape(T, n * n, n * n))
    if need_u:
        dxdm = clip_div(cij * sj + cji * si, sj * sj - si * si) * offdiag
        eye = np.eye(n)
        dudx = np.einsum("bik,lj->bijkl", u, eye)  # (l==j) ? U[i,k] : 0
        du = np.matmul(dudx.reshape(T, n * n, n * n), dxdm.reshape(T, n * n, n * n))
    return du, ds, dw


def svd_w_taylor_fwd_p(mk, u0, s0, w0, bm, bp, bpw):
    """Order-k terms of the polar decomposition M = P W, P = U S U'.

    libsanm/tensor_svd.cpp:389-475.  With V0 = W0' U0:
      Q  = S0 V0' Mk' U0
      E  = U0' (Bm - Bp)' U0 + Q + Q'
      X  = clip_div(E_ij, s_i + s_j)
      Pk = (U0 X U0')'
      Wk = U0 diag(clip_div(1, s_i)) U0' (Mk - Bpw - Pk W0)
    Returns (Pk, Wk).
    """
    T, n, _ = mk.shape
    u0t = np.swapaxes(u0, 1, 2)
    v0 = np.matmul(np.swapaxes(w0, 1, 2), u0)
    e = np.matmul(np.matmul(u0t, np.swapaxes(bm - bp, 1, 2)), u0)
    q = s0[:, :, None] * np.matmul(np.matmul(np.swapaxes(v0, 1, 2), np.swapaxes(mk, 1, 2)), u0)
    e = e + q + np.swapaxes(q, 1, 2)
    x = clip_div(e, s0[:, :, None] + s0[:, None, :])
    pkt = np.matmul(np.matmul(u0, x), u0t)
    pk = np.swapaxes(pkt, 1, 2)
    rhs = mk - bpw - np.matmul(pk, w0)
    s0inv = clip_div(1.0, s0)
    wk = np.matmul(np.matmul(u0 * s0inv[:, None, :], u0t), rhs)
    return np.ascontiguousarray(pk), wk


def svd_w_taylor_fwd(mk, mbiask, u0, s0, w0, bu, bw):
    """Order-k U_k, S_k, W_k of M = U S U' W (full mode).

    libsanm/tensor_svd.cpp:275-387.  ``bu`` may be None, in which case only
    W_k is returned (Uk, Sk = None).  Reached only by the reference's unit
    tests (tests/tensor.cpp:841-868); kept for completeness of the oracle.
    Written on the transposed (Eigen-view) matrices exactly like the source,
    then transposed back.
    """
    T, n, _ = mk.shape
    tr = lambda a: np.swapaxes(a, 1, 2)
    cU0T, cW0T, cMkT, cMbT, cBwT = tr(u0), tr(w0), tr(mk), tr(mbiask), tr(bw)
    cV0 = np.matmul(cW0T, tr(cU0T))
    tmp0 = cMkT - cMbT
    eqbT = np.matmul(np.matmul(tr(cV0), tmp0), tr(cU0T))
    rhs = tr(eqbT) - eqbT
    rhs = rhs - np.matmul(np.matmul(tr(cV0), tr(cBwT)), cV0) * s0[:, None, :]
    x = clip_div(rhs, s0[:, :, None] + s0[:, None, :])
    if bu is not None:
        eqbT = eqbT - tr(x) * s0[:, None, :]
    cWkT = np.matmul(np.matmul(cV0, tr(x)), cU0T)
    wk = np.ascontiguousarray(tr(cWkT))
    if bu is None:
        return None, None, wk
    cBuT = tr(bu)
    eqbT = eqbT + cBuT * s0[:, None, :]
    sk = np.einsum("bii->bi", eqbT).copy()
    ukt_u0 = np.zeros_like(eqbT)
    for j in range(n):
        for i in range(j):
            vv = clip_div(eqbT[:, i, j], s0[:, i] - s0[:, j])
            ukt_u0[:, i, j] = vv
            ukt_u0[:, j, i] = -cBuT[:, j, i] - vv
        ukt_u0[:, j, j] = -cBuT[:, j, j] / 2
    cUkT = np.matmul(ukt_u0, cU0T)
    return np.ascontiguousarray(tr(cUkT)), sk, wk


# --------------------------------------------------------------------------
# libsanm/tensor_polymat.cpp
# --------------------------------------------------------------------------
def _det_terms(m):
    """All (sign, permutation) pairs of the Leibniz expansion of an m x m det.

    libsanm/tensor_polymat.cpp:271-323 builds the same set recursively.
    """
    terms = []
    for perm in itertools.permutations(range(m)):
        inv = sum(1 for i in range(m) for j in range(i + 1, m) if perm[i] > perm[j])
        terms.append((-1.0 if inv % 2 else 1.0, perm))
    return terms


def compute_polymat_det_coeff(coeffs, order):
    """Coefficient of a^order in det(sum_i coeffs[i] a^i).

    libsanm/tensor_polymat.cpp:344-379; dims <= 4 use the expansion
    (:201-264, :325-341): each Leibniz term is a product of m scalar
    polynomials, built by truncated Cauchy products (``conv`` :159-168) and a
    final single-coefficient product (``conv_k`` :170-184).  dim > 4: the
    polynomial matrix is evaluated at P = next_pow2(nr_term) roots of unity
    (``fft`` :30-89, here the DFT sum itself), a complex determinant is taken
    at each, and the inverse transform's ``order``-th output is the
    coefficient (:100-136, with its "IDFT not real" assertion).
    ``coeffs``: list of (T,m,m).  Returns (T,1).
    """
    T, m, _ = coeffs[0].shape
    nr_term = (len(coeffs) - 1) * m + 1
    if order >= nr_term:
        return np.zeros((T, 1))
    if order == 0:
        return batched_determinant(coeffs[0])
    if order == 1:
        if len(coeffs) < 2:
            return np.zeros((T, 1))
        return (batched_cofactor(coeffs[0]) * coeffs[1]).reshape(T, -1).sum(axis=1)[:, None]
    if m > 4:
        P = 1
        while P < nr_term:
            P <<= 1
        c = np.stack(coeffs, axis=0)  # (nc, T, m, m)
        accum = np.zeros(T, dtype=complex)
        for i in range(P):
            w = np.exp(2j * np.pi * i * np.arange(len(coeffs)) / P)      # omega_P^(i q), q = 0..nc-1
            val = np.tensordot(w, c, axes=(0, 0))                        # (T, m, m) complex
            ang = -(2 * np.pi) * float(i * order) / P
            accum += np.linalg.det(val) * complex(np.cos(ang), np.sin(ang))
        accum /= P
        assert np.all(np.abs(accum.imag) < 1e-4 * np.maximum(1.0, np.abs(accum.real))), "IDFT not real"
        return np.ascontiguousarray(accum.real)[:, None]
    nc = len(coeffs)
    c = np.stack(coeffs, axis=0)  # (nc, T, m, m)
    ret = np.zeros(T)
    for sign, perm in _det_terms(m):
        # prod = poly(row0) * poly(row1) truncated to degree <= order
        x = c[:, :, 0, perm[0]]
        y = c[:, :, 1, perm[1]]
        if m == 2:
            ret += sign * _conv_k(order, x, y)
            continue
        prod = _conv(order, x, y)
        for r in range(2, m - 1):
            prod = _conv(order, prod, c[:, :, r, perm[r]])
        ret += sign * _conv_k(order, prod, c[:, :, m - 1, perm[m - 1]])
    return ret[:, None]


def _conv(k, x, y):
    """dst[i+j] += x[i]*y[j] for i+j <= k (tensor_polymat.cpp:159-168)."""
    dst = np.zeros((k + 1,) + x.shape[1:])
    for i in range(x.shape[0]):
        for j in range(y.shape[0]):
            if i + j <= k:
                dst[i + j] += x[i] * y[j]
    return dst


def _conv_k(k, x, y):
    """sum_i x[i]*y[k-i] (tensor_polymat.cpp:170-184)."""
    acc = np.zeros(x.shape[1:])
    lo = max(0, k + 1 - y.shape[0])
    for i in range(lo, min(x.shape[0], k + 1)):
        acc += x[i] * y[k - i]
    return acc
