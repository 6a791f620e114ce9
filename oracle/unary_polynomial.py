"""Scalar polynomial helpers of the ANM driver, restated.

ORACLE -- test infrastructure only (see oracle/__init__.py).

Follows libsanm/unary_polynomial.{h,cpp}.  ``solve_eqn`` uses Brent's zero
finder: the reference links the vendored third_party/BRENT (John Burkardt's
C++ transcription of R. Brent, "Algorithms for Minimization Without
Derivatives", 1973, procedure ``zero``); the algorithm is restated here from
the book's description (bisection / secant / inverse quadratic interpolation
with tolerance ``2*macheps*|b| + t``).

``roots`` in the reference is ACM algorithm 30 (Bairstow + Newton,
unary_polynomial.cpp:154-334).  The oracle uses the companion-matrix
eigenvalues (numpy.roots) instead: only *which real roots exist* feeds the
Pade pole search (pade.cpp:113-126), and both methods agree on that up to
round-off for the well-separated roots that matter.  The reference's own KAT
(tests/pade.cpp:16-62) is checked in tests/test_oracle_host.py.
"""
from __future__ import annotations

import numpy as np

MACHEPS = np.finfo(np.float64).eps


def eval_poly(f, x):
    """Horner; unary_polynomial.cpp:71-77."""
    ret = 0.0
    for c in reversed(list(f)):
        ret = ret * x + c
    return ret


def stable_x_range(order):
    """unary_polynomial.cpp:97-103."""
    return float(np.power(1e15, 1.0 / float(order)))


def brent_zero(a, b, t, f):
    """Brent's ``zero`` on a change-of-sign interval [a, b]."""
    sa, sb = a, b
    fa, fb = f(sa), f(sb)
    assert (fa < 0) != (fb < 0) or fa == 0 or fb == 0
    c, fc = sa, fa
    e = sb - sa
    d = e
    while True:
        if abs(fc) < abs(fb):
            sa, sb, c = sb, c, sb
            fa, fb, fc = fb, fc, fb
        tol = 2.0 * MACHEPS * abs(sb) + t
        m = 0.5 * (c - sb)
        if abs(m) <= tol or fb == 0.0:
            break
        if abs(e) < tol or abs(fa) <= abs(fb):
            e = m
            d = e
        else:
            s = fb / fa
            if sa == c:
                p = 2.0 * m * s
                q = 1.0 - s
            else:
                q = fa / fc
                r = fb / fc
                p = s * (2.0 * m * q * (q - r) - (sb - sa) * (r - 1.0))
                q = (q - 1.0) * (r - 1.0) * (s - 1.0)
            if 0.0 < p:
                q = -q
            else:
                p = -p
            s = e
            e = d
            if 2.0 * p < 3.0 * m * q - abs(tol * q) and p < abs(0.5 * s * q):
                d = p / q
            else:
                e = m
                d = e
        sa, fa = sb, fb
        if tol < abs(d):
            sb = sb + d
        elif 0.0 < m:
            sb = sb + tol
        else:
            sb = sb - tol
        fb = f(sb)
        if (0.0 < fb and 0.0 < fc) or (fb <= 0.0 and fc <= 0.0):
            c, fc = sa, fa
            e = sb - sa
            d = e
    return sb


def solve_eqn(f, xmin, xmax, b=0.0, eps=1e-6):
    """x in [xmin, xmax] with f(x) = b; unary_polynomial.cpp:88-95."""
    assert len(f) and xmin < xmax
    fn = lambda x: eval_poly(f, x) - b
    f0, f1 = fn(xmin), fn(xmax)
    assert f0 * f1 <= 0, f"no zero point: f0={f0} f1={f1}"
    return brent_zero(xmin, xmax, eps, fn)


def real_roots(f, tol=1e-8):
    """Real roots of sum f[i] x^i (coefficients low order first).

    Reference: unary_polynomial::roots(f, only_real=True),
    unary_polynomial.cpp:154-334.  Returns None if the solve fails.
    """
    c = np.asarray(list(f), dtype=np.float64)
    n = len(c) - 1
    while n >= 0 and c[n] == 0.0:
        n -= 1
    if n <= 0:
        return []
    try:
        r = np.roots(c[:n + 1][::-1])
    except np.linalg.LinAlgError:
        return None
    out = []
    for z in r:
        if abs(z.imag) <= tol * max(1.0, abs(z.real)):
            out.append(float(z.real))
    return out


def eval_tensor(f, x):
    """Horner over vectors; unary_polynomial.cpp:115-126."""
    ret = None
    for c in reversed(list(f)):
        ret = c.copy() if ret is None else ret * x + c
    return ret
