"""Scalar polynomial helpers of the ANM driver, restated.

ORACLE -- test infrastructure only (see oracle/__init__.py).

Follows libsanm/unary_polynomial.{h,cpp}.  ``solve_eqn`` uses Brent's zero
finder: the reference links the vendored third_party/BRENT (John Burkardt's
C++ transcription of R. Brent, "Algorithms for Minimization Without
Derivatives", 1973, procedure ``zero``).  ``brent_zero`` below is a statement-for-statement
transcription of that routine (same local names ``sa, sb, fa, fb, fc``) because ``solve_a`` must reproduce the
reference's restart parameter bit for bit (tolerance ``2*macheps*|b| + t``); attribution: R. P. Brent 1973,
C++ version by J. Burkardt (LGPL), vendored by the reference under third_party/BRENT/brent.cpp:1003-1130.

``roots`` is ACM algorithm 30 (Ellenberger 1960; Bairstow + Newton), restated below statement by statement
because its FAILURE (None) makes the caller reject the Pade approximant (pade.cpp:113-116) -- a discrete decision
that changes continuation step counts.  Both functions are pinned bit for bit against the reference's own
translation unit compiled here (oracle/build_ref.py -> oracle/_ref/libref_poly_O2.so) and against fixtures
generated from it (tests/golden/ref_poly.json).
"""
from __future__ import annotations

import math

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
    return math.pow(1e15, 1.0 / float(order))  # (libm pow like the reference; numpy differs in the last bit)


class _Brent:
    """State of Brent's zero finder (R. P. Brent, "Algorithms for Minimization Without Derivatives", 1973, chapter 4,
    procedure `zero`): `best` = the point with the smallest |f| seen, `prev` = the best before it, `brk` = the last
    point whose f has the other sign (the zero lies between best and brk), `step` = the step about to be taken,
    `before` = the one before it.  The reference's solve_eqn calls the vendored third_party/BRENT
    (unary_polynomial.cpp:88-95); the formulas are evaluated in the operation order of Brent's publication, which
    makes the zeros bit-identical to the reference's (tests/test_oracle_ref_poly.py); the organisation is ours."""

    def __init__(self, a, b, f):
        self.prev, self.f_prev = a, f(a)
        self.best, self.f_best = b, f(b)
        assert (self.f_prev < 0) != (self.f_best < 0) or self.f_prev == 0 or self.f_best == 0
        self.rebracket()

    def rebracket(self):
        self.brk, self.f_brk = self.prev, self.f_prev
        self.before = self.best - self.prev
        self.step = self.before

    def order_by_residual(self):
        if abs(self.f_brk) < abs(self.f_best):
            self.prev, self.best, self.brk = self.best, self.brk, self.best
            self.f_prev, self.f_best, self.f_brk = self.f_best, self.f_brk, self.f_best

    def bisect(self, half):
        self.before = half
        self.step = half

    def interpolate(self, half, tol):
        ratio = self.f_best / self.f_prev
        if self.prev == self.brk:  # two distinct points: secant
            num = 2.0 * half * ratio
            den = 1.0 - ratio
        else:  # three: inverse quadratic interpolation
            qa = self.f_prev / self.f_brk
            qb = self.f_best / self.f_brk
            num = ratio * (2.0 * half * qa * (qa - qb) - (self.best - self.prev) * (qb - 1.0))
            den = (qa - 1.0) * (qb - 1.0) * (ratio - 1.0)
        if 0.0 < num:
            den = -den
        else:
            num = -num
        two_back = self.before
        self.before = self.step
        if 2.0 * num < 3.0 * half * den - abs(tol * den) and num < abs(0.5 * two_back * den):
            self.step = num / den
        else:
            self.bisect(half)


def brent_zero(a, b, t, f):
    """Brent's ``zero`` on a change-of-sign interval [a, b], absolute tolerance t."""
    z = _Brent(a, b, f)
    while True:
        z.order_by_residual()
        tol = 2.0 * MACHEPS * abs(z.best) + t
        half = 0.5 * (z.brk - z.best)
        if abs(half) <= tol or z.f_best == 0.0:
            return z.best
        if abs(z.before) < tol or abs(z.f_prev) <= abs(z.f_best):
            z.bisect(half)
        else:
            z.interpolate(half, tol)
        z.prev, z.f_prev = z.best, z.f_best
        if tol < abs(z.step):
            z.best = z.best + z.step
        elif 0.0 < half:
            z.best = z.best + tol
        else:
            z.best = z.best - tol
        z.f_best = f(z.best)
        if (0.0 < z.f_best and 0.0 < z.f_brk) or (z.f_best <= 0.0 and z.f_brk <= 0.0):
            z.rebracket()


def solve_eqn(f, xmin, xmax, b=0.0, eps=1e-6):
    """x in [xmin, xmax] with f(x) = b; unary_polynomial.cpp:88-95."""
    assert len(f) and xmin < xmax
    fn = lambda x: eval_poly(f, x) - b
    f0, f1 = fn(xmin), fn(xmax)
    assert f0 * f1 <= 0, f"no zero point: f0={f0} f1={f1}"
    return brent_zero(xmin, xmax, eps, fn)


def _div(x, y):
    """IEEE-754 division (Python raises on a zero divisor, C does not)."""
    try:
        return x / y
    except ZeroDivisionError:
        if x != x or x == 0.0:
            return math.nan
        neg = (math.copysign(1.0, x) < 0) != (math.copysign(1.0, y) < 0)
        return -math.inf if neg else math.inf


def _exp(x):
    try:
        return math.exp(x)
    except OverflowError:
        return math.inf


def _sqrt(x):
    return math.sqrt(x) if x >= 0 or x != x else math.nan


def roots(a, only_real, max_iter=300, tol=1e-8):
    """All roots of sum a[i] x^i by ACM algorithm 30 (K. W. Ellenberger, "On programming the numerical solution
    of polynomial equations", CACM 3(12), 1960: simultaneous Bairstow / Newton iterated synthetic division on the
    polynomial or its reciprocal, accuracy requirement relaxed by a decimal figure after every second failed
    round).

    Restates unary_polynomial::roots, unary_polynomial.cpp:154-334 (defaults unary_polynomial.h:50-52), statement
    by statement in strict double arithmetic: same operation order, same deflation rule, same give-up rule
    (K < 1e-8 => None).  Checked bit for bit against the reference's translation unit compiled with g++ -O2
    (oracle/build_ref.py, tests/test_oracle_ref_poly.py) and against tests/golden/ref_poly.json.

    Returns a list of complex (complex pairs omitted when ``only_real``) or None when the reference returns None.
    The reference's quirks are kept: a zero constant term is stripped without reporting the root 0
    (unary_polynomial.cpp:186-188), and the work arrays keep stale entries between deflations.
    """
    a = [float(v) for v in a]
    assert len(a) >= 2
    n = len(a) - 1
    OFF = 2  # the arrays are indexed -2 .. n
    h = [0.0] * (n + 3)
    b = [0.0] * (n + 3)
    c = [0.0] * (n + 3)
    d = [0.0] * (n + 3)
    e = [0.0] * (n + 3)
    for m, j in enumerate(range(n, -1, -1)):
        h[OFF + j] = a[m]  # reversal: h[0] is the constant term's mirror, i.e. h[j] = a[n - j]
    t = 1
    K = 1.0 / tol
    out = []
    while h[OFF + n] == 0.0:
        n -= 1
        assert n > -2
    p = q = r = 0.0
    ps = qs = pt = qt = 0.0
    rev = 1.0

    # the reference is a goto program; `state` names the label control is at
    state = "INIT"
    while True:
        if state == "INIT":
            if n == 0:
                return out
            ps = qs = pt = qt = s = 0.0
            rev = 1.0
            if n == 1:
                r = _div(-h[OFF + 1], h[OFF + 0])
                state = "LINEAR"
                continue
            for j in range(n, -1, -1):
                if h[OFF + j] != 0.0:
                    s += math.log(abs(h[OFF + j]))
            s = _exp(s / (n + 1))
            for j in range(n, -1, -1):
                h[OFF + j] = _div(h[OFF + j], s)
            if abs(_div(h[OFF + 1], h[OFF + 0])) < abs(_div(h[OFF + n - 1], h[OFF + n])):
                state = "REVERSE"
            else:
                state = "START"
            continue
        if state == "REVERSE":
            t = -t
            for j in range((n - 1) // 2, -1, -1):
                h[OFF + j], h[OFF + n - j] = h[OFF + n - j], h[OFF + j]
            state = "START"
            continue
        if state == "START":
            if qs != 0.0:
                p, q = ps, qs
            else:
                if h[OFF + n - 2] == 0.0:
                    q = 1.0
                    p = -2.0
                else:
                    q = h[OFF + n] / h[OFF + n - 2]
                    p = (h[OFF + n - 1] - q * h[OFF + n - 3]) / h[OFF + n - 2]
                if n == 2:
                    state = "QUADRATIC"
                    continue
                r = 0.0
            state = "ITERATE"
            continue
        if state == "ITERATE":
            nxt = None
            for _ in range(max_iter):
                for j in range(n + 1):  # Bairstow: two synthetic divisions by x^2 + p x + q
                    b[OFF + j] = h[OFF + j] - p * b[OFF + j - 1] - q * b[OFF + j - 2]
                    c[OFF + j] = b[OFF + j] - p * c[OFF + j - 1] - q * c[OFF + j - 2]
                if h[OFF + n - 1] != 0.0 and b[OFF + n - 1] != 0.0:
                    if abs(h[OFF + n - 1] / b[OFF + n - 1]) >= K:
                        b[OFF + n] = h[OFF + n] - q * b[OFF + n - 2]
                    if b[OFF + n] == 0.0:
                        nxt = "QUADRATIC"
                        break
                    if K < abs(h[OFF + n] / b[OFF + n]):
                        nxt = "QUADRATIC"
                        break
                for j in range(n + 1):  # Newton: value and derivative at r
                    d[OFF + j] = h[OFF + j] + r * d[OFF + j - 1]
                    e[OFF + j] = d[OFF + j] + r * e[OFF + j - 1]
                if d[OFF + n] == 0.0:
                    nxt = "LINEAR"
                    break
                if K < abs(h[OFF + n] / d[OFF + n]):
                    nxt = "LINEAR"
                    break
                c[OFF + n - 1] = -p * c[OFF + n - 2] - q * c[OFF + n - 3]
                s = c[OFF + n - 2] * c[OFF + n - 2] - c[OFF + n - 1] * c[OFF + n - 3]
                if s == 0.0:
                    p -= 2.0
                    q *= (q + 1.0)
                else:
                    p += (b[OFF + n - 1] * c[OFF + n - 2] - b[OFF + n] * c[OFF + n - 3]) / s
                    q += (-b[OFF + n - 1] * c[OFF + n - 1] + b[OFF + n] * c[OFF + n - 2]) / s
                if e[OFF + n - 1] == 0.0:
                    r -= 1.0
                else:
                    r -= d[OFF + n] / e[OFF + n - 1]
            if nxt is not None:
                state = nxt
                continue
            ps, qs = pt, qt
            pt, qt = p, q
            if rev < 0.0:
                K /= 10.0
            if K < 1e-8:
                return None
            rev = -rev
            state = "REVERSE"
            continue
        if state == "LINEAR":
            if t < 0:
                r = _div(1.0, r)
            n -= 1
            out.append(complex(r, 0.0))
            for j in range(n, -1, -1):
                if d[OFF + j] != 0.0 and abs(h[OFF + j] / d[OFF + j]) < K:
                    h[OFF + j] = d[OFF + j]
                else:
                    h[OFF + j] = 0.0
            if n == 0:
                return out
            state = "ITERATE"
            continue
        if state == "QUADRATIC":
            if t < 0:
                p = _div(p, q)
                q = _div(1.0, q)
            n -= 2
            disc = q - (p * p / 4.0)
            if 0.0 < disc:
                s = _sqrt(disc)
                if not only_real:
                    out.append(complex(-p / 2.0, s))
                    out.append(complex(-p / 2.0, -s))
            else:
                s = _sqrt((p * p / 4.0) - q)
                qq = -p / 2.0 + s if p < 0.0 else -p / 2.0 - s
                out.append(complex(qq, 0.0))
                out.append(complex(_div(q, qq), 0.0))
            for j in range(n, -1, -1):
                if b[OFF + j] != 0.0 and abs(h[OFF + j] / b[OFF + j]) < K:
                    h[OFF + j] = b[OFF + j]
                else:
                    h[OFF + j] = 0.0
            state = "INIT"
            continue
        raise AssertionError(state)


def real_roots(f, max_iter=300, tol=1e-8):
    """unary_polynomial::roots(f, only_real=true) as pade.cpp:113 calls it: the real parts of the roots the
    algorithm reports as real, in the order it finds them, or None when it gives up."""
    r = roots(f, True, max_iter, tol)
    if r is None:
        return None
    return [z.real for z in r]


def eval_tensor(f, x):
    """Horner over vectors; unary_polynomial.cpp:115-126."""
    ret = None
    for c in reversed(list(f)):
        ret = c.copy() if ret is None else ret * x + c
    return ret
