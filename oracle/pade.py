"""Vector Pade approximant of the ANM series, restated.

ORACLE -- test infrastructure only (see oracle/__init__.py).
Follows libsanm/pade.{h,cpp} (Cochelin et al., "A critical review of
asymptotic numerical methods", appendix 1).
"""
from __future__ import annotations

import numpy as np

from . import unary_polynomial as up


class PadeApproximation:
    def __init__(self, xs, anm_cond, sanity_check=False):
        """pade.cpp:13-105.  ``xs``: list of N+1 vectors (n+1,), last elem = t."""
        self.sanity_check = sanity_check
        self.xs = xs
        self.d = []
        self.d_lo = []
        self.t_nume = []
        self.t0 = 0.0
        self.t_max = 0.0
        self.t_max_a = 0.0
        self.diag = None
        nx = len(xs)
        assert nx >= 3 and xs[0].ndim == 1
        if xs[0].shape[0] < nx * 2 or nx <= 4:
            return
        n = nx - 1
        a = np.zeros((nx, nx))
        eps = np.finfo(np.float64).eps
        orth = [None] * nx
        for i in range(1, n + 1):
            uii = xs[i].copy()
            for j in range(1, i):
                a[i, j] = float(np.dot(xs[i], orth[j]))
                if anm_cond and j == 1:
                    assert abs(a[i, j]) < 1e-4
                    a[i, j] = 0.0
                else:
                    uii = uii - a[i, j] * orth[j]
            aii = float(np.linalg.norm(uii))
            if aii == 0:
                self.d = []
                return
            a[i, i] = aii
            uii = uii / max(aii, eps)
            if aii < eps:
                uii = uii / np.linalg.norm(uii)
            orth[i] = uii

        def solve_d(nn):
            d = [0.0] * nn
            d[0] = 1.0
            for i in range(1, nn):
                s = 0.0
                for j in range(i):
                    s += a[nn - j, nn - i] * d[j]
                y = a[nn - i, nn - i]
                d[i] = -s * y / (y * y + 1e-20)
            return d

        self.d = solve_d(n)
        self.d_lo = solve_d(n - 1)
        self.t_nume = [0.0] * n
        for i in range(n):
            ti = float(xs[i][-1])
            if i == 0:
                self.t0 = ti
            else:
                for j in range(n - i):
                    self.t_nume[i + j] += self.d[j] * ti

    def eval_nume(self, a, d=None, n=None):
        """pade.cpp:181-189."""
        if d is None:
            d, n = self.d, len(self.xs) - 2
        s = np.zeros_like(self.xs[0])
        for i in range(n, 0, -1):
            s = s * a
            scale = up.eval_poly(d[:n - i + 1], a)
            s = s + self.xs[i] * scale
        return s

    def eval_t(self, a):
        return up.eval_poly(self.t_nume, a) / up.eval_poly(self.d, a) + self.t0

    def estimate_valid_range(self, start, eps, limit=0.0):
        """pade.cpp:107-173.  Every discrete decision is recorded with the quantity it was taken on in
        ``self.diag`` (same fields as the product's sanm_anm_pade_diag): ``margin`` of a probe is
        |pn_lo * D_n / D_lo - pn|^2 / (eps^2 |pn|^2) of check(a), pade.cpp:129-138 -- it passes iff margin <= 1."""
        assert start > 0 and eps > 0
        dg = self.diag = {"attempted": True, "built": bool(self.d), "roots_valid": False, "accepted": False,
                          "start": start, "pole": 0.0, "t_max_a": 0.0, "d": list(self.d), "probes": []}
        if not self.d:
            return False
        roots = up.real_roots(self.d)
        if roots is None:
            return False
        dg["roots_valid"] = True
        pole = 0.0
        for r in roots:
            if r > 0 and (pole == 0 or r < pole):
                pole = r
        if pole == 0:
            pole = start * 4
        dg["pole"] = pole
        if pole <= start:
            return False
        n = len(self.xs) - 2
        eps2 = eps * eps

        def check(a):
            denom_n = up.eval_poly(self.d, a)
            denom_lo = up.eval_poly(self.d_lo, a)
            pn = self.eval_nume(a, self.d, n)
            pn_lo = self.eval_nume(a, self.d_lo, n - 1)
            pn_lo = pn_lo * (denom_n / denom_lo) - pn
            num, den = float(np.dot(pn_lo, pn_lo)), float(np.dot(pn, pn))
            ok = num <= den * eps2
            dg["probes"].append((a, num / (den * eps2) if den > 0 else float("inf"), ok))
            return ok

        left = start * 1.001
        right = start + (pole - start) * 0.99
        if not check(left):
            return False
        if limit and right > limit:
            right = limit
        if right > start * 2:
            if check(start * 2):
                left = start * 2
            else:
                right = start * 2
        it = 0
        while it < 8 and right - left > 1e-3:
            mid = (left + right) / 2
            if check(mid):
                left = mid
            else:
                right = mid
            it += 1
        self.t_max_a = left
        self.t_max = self.eval_t(left)
        dg["accepted"] = True
        dg["t_max_a"] = left
        return True

    def solve_a(self, t):
        """pade.cpp:191-201."""
        assert self.t0 <= t <= self.t_max
        if t == self.t_max:
            return self.t_max_a
        c = [self.t_nume[i] - (t - self.t0) * self.d[i] for i in range(len(self.t_nume))]
        return up.solve_eqn(c, 0.0, self.t_max_a, 0.0)

    def eval_xt(self, a):
        """pade.cpp:214-219."""
        ret = self.eval_nume(a)
        ret = ret * (a / up.eval_poly(self.d, a))
        return ret + self.xs[0]
