"""The vector Pade approximant of libsanm/pade.cpp evaluated in HIGH precision: the third arbiter.

ORACLE -- test infrastructure only (see oracle/__init__.py).  Nothing under sanm_amd/ may import this.

Why: the product (device arithmetic) and the fp64 oracle (oracle/pade.py, numpy) take different Pade decisions on
series that agree to 1e-10 (tests/lockstep.py).  Both run the reference's CLASSICAL
Gram-Schmidt sweep (pade.cpp:30-55) in double precision, where the loss of orthogonality is ~ kappa^2 eps; neither is
"the reference".  This module evaluates the SAME algorithm -- same projections, same anm_cond rule, same
`y / (y^2 + 1e-20)` regulariser in solve_d (pade.cpp:67-79), same probes, same bisection (pade.cpp:107-173) -- with the
rounding taken out:

  * every inner product of two series vectors is formed EXACTLY: a_i * b_i = p_i + e_i by Dekker's error-free
    product, the 2n terms summed by math.fsum (correctly rounded), the remainder summed again, six times over
    (relative error < 1e-90: the residual norms of the Gram-Schmidt sweep are differences of such products that cancel
    to kappa^-2 of their size, and kappa is large on these series -- `min_residual_ratio` in the record);
  * everything the algorithm does with the vectors is linear in them, so it is carried out on coefficient vectors over
    the basis {x_1 .. x_N} with the Gram matrix G = [<x_i, x_j>] in mpmath arithmetic (default 200 digits):
    <x_i, q_j> = c_j^T G e_i, |u|^2 = c^T G c, the accept test |pn_lo D_n / D_lo - pn|^2 <= eps^2 |pn|^2 likewise;
  * the poles come from mpmath.polyroots on the exact denominator.

Two outcomes are reported per series: `exact` (roots in high precision: the decision a perfect implementation of the
published method takes) and `ref_roots` (the high-precision denominator rounded to double and handed to the
reference's own root finder, ACM algorithm 30 with its give-up => reject outcome, unary_polynomial.cpp:154-334 /
pade.cpp:113-116: the decision the reference's CODE would take if only its Gram-Schmidt were exact).
"""
from __future__ import annotations

import math

import mpmath as mp
import numpy as np

from . import unary_polynomial as up

_SPLIT = 134217729.0  # 2^27 + 1 (Veltkamp)


def _two_prod(a, b):
    """a * b = p + e exactly, element by element (Dekker 1971; no FMA in numpy)"""
    p = a * b
    ca = _SPLIT * a
    ah = ca - (ca - a)
    al = a - ah
    cb = _SPLIT * b
    bh = cb - (cb - b)
    bl = b - bh
    e = ((ah * bh - p) + ah * bl + al * bh) + al * bl
    return p, e


def exact_dot(a, b, terms=6):
    """<a, b> as an mpmath number: the exact sum of the exact products, to `terms` doubles"""
    p, e = _two_prod(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64))
    vals = np.concatenate([p, e]).tolist()
    total = mp.mpf(0)
    for _ in range(terms):
        s = math.fsum(vals)
        if s == 0.0:
            break
        total += mp.mpf(s)
        vals.append(-s)
    return total


def gram_matrix(xs):
    """G[i][j] = <xs[i], xs[j]>, i, j = 1 .. N (row / column 0 unused), exact"""
    nx = len(xs)
    G = [[mp.mpf(0)] * nx for _ in range(nx)]
    for i in range(1, nx):
        for j in range(1, i + 1):
            G[i][j] = G[j][i] = exact_dot(xs[i], xs[j])
    return G


class PadeHP:
    """pade.cpp:13-105 on coefficient vectors over {x_1..x_N}; `xs` as PadeApproximation takes them"""

    def __init__(self, xs, anm_cond, dps=200, gram=None):
        mp.mp.dps = dps
        self.nx = nx = len(xs)
        self.n = n = nx - 1
        self.ts = [mp.mpf(float(x[-1])) for x in xs]
        self.G = G = gram if gram is not None else gram_matrix(xs)
        self.d, self.d_lo, self.t_nume = [], [], []
        self.diag = None
        self.degenerate = 0
        if xs[0].shape[0] < nx * 2 or nx <= 4:
            return
        eps = mp.mpf(np.finfo(np.float64).eps)
        a = [[mp.mpf(0)] * nx for _ in range(nx)]
        C = [None] * nx  # orth[j] = sum_k C[j][k] x_k

        def gdot(c, i):  # <sum_k c_k x_k, x_i>
            return mp.fsum(c[k] * G[k][i] for k in range(1, nx) if c[k] != 0)

        def gnorm2(c):
            return mp.fsum(c[k] * gdot(c, k) for k in range(1, nx) if c[k] != 0)

        for i in range(1, n + 1):
            u = [mp.mpf(0)] * nx
            u[i] = mp.mpf(1)
            for j in range(1, i):
                a[i][j] = gdot(C[j], i)
                if anm_cond and j == 1:
                    a[i][j] = mp.mpf(0)  # (pade.cpp:40-44: asserted small, then dropped -- projection and all)
                else:
                    u = [uk - a[i][j] * cj for uk, cj in zip(u, C[j])]
            n2 = gnorm2(u)
            if n2 <= 0:  # x_i lies in the span of its predecessors to working precision (pade.cpp:57-60: aii == 0)
                self.degenerate = i
                return
            aii = mp.sqrt(n2)
            a[i][i] = aii
            u = [uk / max(aii, eps) for uk in u]
            if aii < eps:
                nrm = mp.sqrt(gnorm2(u))
                u = [uk / nrm for uk in u]
            C[i] = u
        self.a = a

        def solve_d(nn):
            d = [mp.mpf(0)] * nn
            d[0] = mp.mpf(1)
            for i in range(1, nn):
                s = mp.fsum(a[nn - j][nn - i] * d[j] for j in range(i))
                y = a[nn - i][nn - i]
                d[i] = -s * y / (y * y + mp.mpf("1e-20"))
            return d

        self.d = solve_d(n)
        self.d_lo = solve_d(n - 1)
        self.t_nume = [mp.mpf(0)] * n
        for i in range(1, n):
            for j in range(n - i):
                self.t_nume[i + j] += self.d[j] * self.ts[i]

    # -- evaluation ---------------------------------------------------------------------------------------------
    @staticmethod
    def _poly(c, x):
        s = mp.mpf(0)
        for v in reversed(c):
            s = s * x + v
        return s

    def _nume_weights(self, a, d, n):
        """eval_nume (pade.cpp:181-189) as weights over x_1..x_N: s = sum_i x_i * poly(d[:n-i+1], a) * a^(i-1)"""
        w = [mp.mpf(0)] * self.nx
        for i in range(1, n + 1):
            w[i] = self._poly(d[:n - i + 1], a) * a ** (i - 1)
        return w

    def _quad(self, w):
        G = self.G
        return mp.fsum(w[i] * w[j] * G[i][j] for i in range(1, self.nx) for j in range(1, self.nx))

    def margin(self, a):
        """|pn_lo D_n / D_lo - pn|^2 / |pn|^2 (pade.cpp:129-138, without the eps^2)"""
        a = mp.mpf(a)
        n = self.nx - 2
        dn, dlo = self._poly(self.d, a), self._poly(self.d_lo, a)
        w = self._nume_weights(a, self.d, n)
        wlo = self._nume_weights(a, self.d_lo, n - 1)
        e = [wl * (dn / dlo) - wn for wl, wn in zip(wlo, w)]
        return self._quad(e) / self._quad(w)

    def positive_poles(self):
        """real positive roots of the denominator, high precision"""
        d = [c for c in self.d]
        while d and d[-1] == 0:
            d.pop()
        if len(d) <= 1:
            return []
        roots = mp.polyroots(list(reversed(d)), maxsteps=500, extraprec=4 * mp.mp.prec)
        out = []
        for r in roots:
            if abs(mp.im(r)) <= mp.mpf(10) ** (-mp.mp.dps // 2) * (abs(mp.re(r)) + 1) and mp.re(r) > 0:
                out.append(mp.re(r))
        return sorted(out)

    def estimate_valid_range(self, start, eps, limit=0.0, roots="exact"):
        """the decision sequence of pade.cpp:107-173 with exact quantities.  roots = "exact": poles by
        mpmath.polyroots; "ref": the denominator rounded to double and given to the reference's algorithm 30
        (None => rejected, as pade.cpp:113-116 does)."""
        dg = {"built": bool(self.d), "roots_valid": False, "accepted": False, "start": float(start), "pole": 0.0,
              "t_max_a": 0.0, "probes": [], "roots": roots}
        if not self.d:
            return dg
        start_m = mp.mpf(start)
        if roots == "ref":
            r = up.real_roots([float(c) for c in self.d])
            if r is None:
                return dg
            poles = sorted(mp.mpf(v) for v in r if v > 0)
        else:
            poles = self.positive_poles()
        dg["roots_valid"] = True
        pole = poles[0] if poles else start_m * 4
        dg["pole"] = float(pole)
        if pole <= start_m:
            return dg
        eps2 = mp.mpf(eps) ** 2

        def check(a):
            m = self.margin(a) / eps2
            ok = m <= 1
            dg["probes"].append((float(a), float(m), bool(ok)))
            return ok

        # (the probe points themselves are computed in double, as the reference does: they are inputs, not results)
        left = float(start) * 1.001
        right = float(start) + (float(pole) - float(start)) * 0.99
        if not check(left):
            return dg
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
        dg["accepted"] = True
        dg["t_max_a"] = left
        return dg


def arbitrate(xs, anm_cond, start, eps, limit=0.0, dps=200):
    """both high-precision outcomes for one series: {"exact": diag, "ref_roots": diag, "d": [...]}.  An outcome is
    (accepted, range) like tests/lockstep.py::_outcome."""
    p = PadeHP(xs, anm_cond, dps=dps)
    out = {"d": [float(c) for c in p.d], "degenerate_at": p.degenerate,
           # conditioning of the sweep: the smallest ratio |u_i| / |x_i| (the fp64 sweep loses kappa^2 eps)
           "min_residual_ratio": (min(float(p.a[i][i] / mp.sqrt(p.G[i][i])) for i in range(1, p.n + 1)) if p.d else None)}
    for key, roots in (("exact", "exact"), ("ref_roots", "ref")):
        out[key] = p.estimate_valid_range(start, eps, limit, roots=roots)
    return out
