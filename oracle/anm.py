"""ANM continuation drivers, restated.

ORACLE -- test infrastructure only (see oracle/__init__.py).

Follows libsanm/anm.{h,cpp}: ``ANMDriverHelper`` (solve_expansion_coeffs
anm.cpp:193-312, estimate_valid_range :117-154, eval/solve_a :156-191),
``ANMSolverVecScale`` (:320-445), ``ANMEqnSolver`` (:446-491) and
``ANMImplicitSolver`` (:494-615).  The sparse system follows
libsanm/sparse_solver.cpp: contributions with |c| < 1e-9 are dropped *before*
duplicates are merged (:286-305), the factorisation is an unsymmetric sparse
LU (reference: MKL PARDISO mtype 11, :107-127; here: the same MKL PARDISO with the
same settings through ctypes when the image has MKL -- oracle/pardiso.py --, SuperLU
through scipy otherwise).

Remaps (``SparseLinearDesc``, anm.h:24-73) are held as scipy CSR matrices of
shape (out_size, in_size): ``apply`` (anm.cpp:55-75) is a mat-vec.
"""
from __future__ import annotations

import math
import os
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import pardiso
from . import unary_polynomial as up
from .pade import PadeApproximation
from .symbolic import SANMNumericalError, TaylorCoeffProp


class HyperParam:
    """anm.h:100-114, :237-241."""

    def __init__(self, **kw):
        self.use_pade = False
        self.sanity_check = True
        self.order = 8
        self.maxr = 1e-6
        self.solution_check_tol = 1e-4
        self.xcoeff_l2_penalty = 0.0
        self.converge_rms = 1e-5
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)


def _g(v, prec=6):
    """C's %g / %.3g"""
    return "%.*g" % (prec, float(v))


def assert_allclose(msg, a, b, eps=1e-4):
    """TensorND::assert_allclose (libsanm/tensor.cpp:670-684): |a-b| <=
    eps*max(1, min(|a|,|b|))."""
    a = np.asarray(a).ravel()
    b = np.asarray(b).ravel()
    tol = eps * np.maximum(1.0, np.minimum(np.abs(a), np.abs(b)))
    bad = np.abs(a - b) > tol
    if bad.any():
        i = int(np.nonzero(bad)[0][0])
        raise AssertionError(f"{msg}: mismatch at {i}: {a[i]} vs {b[i]}")


def build_jacobian_csr(remap_out, jac, remap_in, nr_unknown, drop=1e-9, chunk_rows=4096):
    """CSR of remap_out . blockdiag(J_e) . remap_in, with the reference's
    per-contribution drop rule.

    anm.cpp:362-438 / :520-608 enumerate, for every output row i, every
    remap_out entry (tet b, out comp o, c_out), every input comp m and every
    remap_in entry (col j, c_in) and call ``add_constraint(i, j,
    J[b,o,m]*c_out*c_in)``; sparse_solver.cpp:286-305 drops |c| < 1e-9 and
    :250-278 sums duplicates.  A column index == nr_unknown (the ``t`` column
    of the implicit solver) is accumulated into grad_t instead (anm.cpp:575-579).
    Returns (A csr (n,n), grad_t (n,) or None).
    """
    T, odim, idim = jac.shape
    n = nr_unknown
    ro = remap_out.tocsr()
    ri = remap_in.tocsr()
    has_t = ri.shape[1] == n + 1
    rows_l, cols_l, vals_l = [], [], []
    grad_t = np.zeros(n) if has_t else None
    ri_deg = np.diff(ri.indptr)
    for r0 in range(0, n, chunk_rows):
        r1 = min(n, r0 + chunk_rows)
        sub = ro[r0:r1].tocoo()
        # one record per (row, remap_out entry)
        i_out = sub.row + r0
        b = sub.col // odim
        o = sub.col % odim
        c_out = sub.data
        # expand over the input comps m
        m = np.arange(idim)
        i_out = np.repeat(i_out, idim)
        coeff = (jac[b, o, :] * c_out[:, None]).ravel()
        in_row = (b[:, None] * idim + m[None, :]).ravel()
        # expand over remap_in entries of in_row
        deg = ri_deg[in_row]
        rep_i = np.repeat(i_out, deg)
        rep_c = np.repeat(coeff, deg)
        starts = ri.indptr[in_row]
        off = np.arange(deg.sum()) - np.repeat(np.cumsum(deg) - deg, deg)
        pos = np.repeat(starts, deg) + off
        col = ri.indices[pos]
        val = rep_c * ri.data[pos]
        if has_t:
            tm = col == n
            np.add.at(grad_t, rep_i[tm], val[tm])
            keep = ~tm
            rep_i, col, val = rep_i[keep], col[keep], val[keep]
        keep = np.abs(val) >= drop
        rows_l.append(rep_i[keep])
        cols_l.append(col[keep])
        vals_l.append(val[keep])
    A = sp.coo_matrix((np.concatenate(vals_l), (np.concatenate(rows_l), np.concatenate(cols_l))),
                      shape=(n, n)).tocsr()
    A.sum_duplicates()
    return A, grad_t


class SparseSolver:
    """libsanm/sparse_solver.{h,cpp}: factor once, solve many, SpMV.  The factorisation is MKL PARDISO with
    the reference's settings when the image provides MKL (oracle/pardiso.py), SuperLU otherwise."""

    def __init__(self, A):
        self.A = A.tocsr()
        self.lu = None

    def prepare(self, l2=0.0):
        self.l2 = float(l2)
        if self.l2:
            # Tikhonov path (sparse_solver.cpp:366-395): min |Ax-b|^2 + l2 |x|^2 through the normal equations
            # (A'A + l2 I) x = A'b; the reference factors the SPD matrix with PARDISO mtype 2
            M = (self.A.T @ self.A + self.l2 * sp.identity(self.A.shape[0])).tocsc()
            self.lu = spla.splu(M, permc_spec="MMD_AT_PLUS_A")
            return
        if pardiso.available():
            self.lu = pardiso.Pardiso(self.A)
        else:
            self.lu = spla.splu(self.A.tocsc(), permc_spec="MMD_AT_PLUS_A")

    def solve(self, b):
        assert np.all(np.isfinite(b))
        if getattr(self, "l2", 0.0):
            return self.lu.solve(self.A.T @ b)  # sparse_solver.cpp:162-176
        return self.lu.solve(b)

    def apply(self, x):
        return self.A @ x

    def coeff_l2(self):
        return float(np.sqrt((self.A.data ** 2).sum()))


class ANMDriverHelper:
    """anm.h:96-207."""

    def __init__(self, f, remap_inp, remap_out, nr_unknown, hyper):
        assert hyper.order >= 2
        self.hp = hyper
        self.func = f
        self.remap_inp = remap_inp.tocsr()
        self.remap_out = remap_out.tocsr()
        self.max_a_bound = up.stable_x_range(hyper.order)
        self.n = int(nr_unknown)
        self.xt0 = None
        self.iter = 0
        self.t_max = 0.0
        self.t_max_a = 0.0
        self.xt_coeffs = []
        self.t_coeffs = []
        self.pade = None
        self.pade_diags = []  # decision record of every range estimate (PadeApproximation.diag)
        self.verbose = os.environ.get("SANM_VERBOSE") is not None
        self.profile = {}
        self.trace = []  # per-step records (norms of b_k, x_k, t_k) for fixtures

    # ---- hooks -----------------------------------------------------------
    def prepare_inp(self, xt):
        raise NotImplementedError

    def get_grad_t(self):
        raise NotImplementedError

    def on_fx0_computed(self, fx):
        raise NotImplementedError

    # ---- helpers ---------------------------------------------------------
    def _tic(self, tag, t0):
        self.profile[tag] = self.profile.get(tag, 0.0) + (time.perf_counter() - t0)

    def init_xt0(self, x, t):
        x = np.asarray(x, dtype=np.float64).ravel()
        assert x.shape[0] == self.n
        self.xt0 = np.concatenate([x, [float(t)]])

    def _apply_in(self, xt):
        v = self.remap_inp @ self.prepare_inp(xt)
        return v.reshape(self.in_shape)

    def _apply_out(self, y):
        return self.remap_out @ y.ravel()

    def get_t0(self):
        return self.t_coeffs[0]

    # ---- the hot path ----------------------------------------------------
    def solve_expansion_coeffs(self):
        """anm.cpp:193-312."""
        N = self.hp.order
        self.xt_coeffs = [None] * (N + 1)
        self.xt_coeffs[0] = self.xt0
        self.t_coeffs = [float(self.xt0[self.n])]
        rec = {"iter": self.iter, "b_norm": [], "x_norm": [], "t": []}

        t0 = time.perf_counter()
        prop = TaylorCoeffProp(self.func)
        fx0 = self._apply_out(prop.push_xi([self._apply_in(self.xt0)]))
        self._tic("taylor_order0", t0)
        if not self.on_fx0_computed(fx0):
            self.xt_coeffs = self.xt_coeffs[:1]
            return

        solver = None
        xgt = x1 = grad_t = None
        xgt_dot_x1 = t1 = 0.0
        for i in range(1, N + 1):
            t0 = time.perf_counter()
            bi = self._apply_out(prop.compute_next_order_bias())
            self._tic("taylor_next_order", t0)
            if i == 1:
                assert not np.any(bi)
                t0 = time.perf_counter()
                A, gt = build_jacobian_csr(self.remap_out, prop.get_jacobian(), self.remap_inp, self.n)
                self._grad_t_from_build = gt
                self._tic("build_sparse_coeff", t0)
                grad_t = np.asarray(self.get_grad_t(), dtype=np.float64).ravel()
                t0 = time.perf_counter()
                solver = SparseSolver(A)
                solver.prepare(self.hp.xcoeff_l2_penalty)
                self._tic("sparse_prep", t0)
                t0 = time.perf_counter()
                xgt = solver.solve(grad_t)
                self._tic("sparse_solve", t0)
                xbi = bi
                t1 = ti = 1.0 / np.sqrt(float(np.dot(xgt, xgt)) + 1.0)
                # anm.cpp:247-250 (SANM_VERBOSE): |gt|, |xgt|, SparseSolver::coeff_l2
                vtext = "gt=%s xgt=%s jacob=%s" % (_g(np.linalg.norm(grad_t)), _g(np.linalg.norm(xgt)),
                                                   _g(solver.coeff_l2()))
            else:
                t0 = time.perf_counter()
                xbi = solver.solve(bi)
                self._tic("sparse_solve", t0)
                ti = float(np.dot(xbi, x1)) / (t1 - xgt_dot_x1)
            vtext += " %d:(bi=%s xbi=%s)" % (i, _g(np.linalg.norm(bi)), _g(np.linalg.norm(xbi)))  # anm.cpp:257-259
            xi = np.empty(self.n + 1)
            xi[:self.n] = xgt * (-ti) - xbi
            xi[self.n] = ti
            self.xt_coeffs[i] = xi
            if i == 1:
                x1 = xi[:self.n]
                xgt_dot_x1 = float(np.dot(x1, xgt))
            if not self.hp.xcoeff_l2_penalty and self.hp.sanity_check:
                t0 = time.perf_counter()
                Ax = solver.apply(xi[:self.n])
                Ax_r = -(grad_t * ti + bi)
                assert_allclose("ANM check coeff eqn", Ax, Ax_r)
                xdot = float(np.dot(self.xt_coeffs[1], xi))
                if i == 1:
                    assert abs(xdot - 1) < 1e-4, f"xdot={xdot}"
                else:
                    assert abs(xdot) < 1e-4, f"i={i}: xdot={xdot}"
                self._tic("anm_sanity_check", t0)
            rec["b_norm"].append(float(np.linalg.norm(bi)))
            rec["x_norm"].append(float(np.linalg.norm(xi)))
            rec["t"].append(float(ti))
            if i < N:
                t0 = time.perf_counter()
                prop.push_xi([self._apply_in(xi)])
                self._tic("taylor_push", t0)
        t0 = time.perf_counter()
        self.estimate_valid_range()
        self._tic("estimate_valid_range", t0)
        rec["a_bound"] = self.t_max_a
        rec["t_max"] = self.t_max
        rec["pade"] = self.pade is not None
        self.trace.append(rec)
        # the reference's SANM_VERBOSE printout (anm.cpp:200-203, :295-309)
        vtext = "=== ANM iter %d:\n" % self.iter + vtext + "\nbound=%s t=%s\n" % (_g(self.t_max_a), _g(self.t_max))
        vtext += "x(a):" + "".join(" %s" % _g(np.linalg.norm(c), 3) for c in self.xt_coeffs)
        vtext += "\nt(a):" + "".join(" %s," % _g(t, 3) for t in self.t_coeffs) + "\n"
        if self.hp.xcoeff_l2_penalty:
            vtext += "xcoeff_l2_penalty=%s\n" % _g(self.hp.xcoeff_l2_penalty)
        self.verbose_text = vtext
        if self.verbose:
            print(vtext, end="", flush=True)
        self.iter += 1

    def estimate_valid_range(self):
        """anm.cpp:117-154."""
        hp = self.hp
        x1 = float(np.linalg.norm(self.xt_coeffs[1]))
        xback = max(float(np.linalg.norm(self.xt_coeffs[-1])), 1e-15)
        a_bound = math.pow(hp.maxr / xback * x1, 1.0 / float(hp.order - 1))  # libm pow (anm.cpp:126)
        a_bound = min(a_bound, self.max_a_bound)
        self.t_coeffs = [float(c[self.n]) for c in self.xt_coeffs]
        assert self.t_coeffs[1] > 0
        self.t_max_a = a_bound
        self.t_max = up.eval_poly(self.t_coeffs, a_bound)
        assert self.t_max > self.t_coeffs[0], "t does not incr"
        self.pade = None
        use_pade_env = os.environ.get("SANM_PADE") is not None
        diag = {"attempted": False, "accepted": False, "start": a_bound}
        self.pade_diags.append(diag)
        self.a_bound = a_bound
        if (hp.use_pade or use_pade_env) and a_bound < self.max_a_bound:
            pade = PadeApproximation(self.xt_coeffs, not hp.xcoeff_l2_penalty, False)
            self.pade_candidate = pade  # (kept for the lock-step tests even when rejected)
            ok = pade.estimate_valid_range(a_bound, hp.maxr, self.max_a_bound)
            diag.update(pade.diag)
            if ok:
                self.pade = pade
                self.t_max_a = pade.t_max_a
                self.t_max = pade.t_max

    def update_approx(self):
        """anm.cpp:156-159."""
        self.xt0 = self.eval_xt(self.t_max_a)
        self.solve_expansion_coeffs()

    def get_t_upper(self):
        return self.t_max

    def eval_xt(self, a):
        """anm.cpp:166-172."""
        if self.pade is not None:
            return self.pade.eval_xt(a)
        return up.eval_tensor(self.xt_coeffs, a)

    def eval(self, a):
        xt = self.eval_xt(a)
        return xt[:self.n].copy(), float(xt[self.n])

    def solve_a(self, t):
        """anm.cpp:174-191."""
        if t == self.t_max:
            return self.t_max_a
        if self.pade is not None:
            return self.pade.solve_a(t)
        assert self.t_coeffs[0] <= t < self.t_max
        if self.t_max_a > 0:
            l, r = 0.0, self.t_max_a
        else:
            l, r = -self.t_max_a, 0.0
        return up.solve_eqn(self.t_coeffs, l, r, t)

    def get_nr_iter(self):
        return self.iter


class ANMSolverVecScale(ANMDriverHelper):
    """f(x) + t*v = 0; anm.h:209-243, anm.cpp:320-445."""

    def __init__(self, f, remap_inp, remap_out, in_shape, x0, t0, v, hyper=None, _defer=False):
        hyper = hyper or HyperParam()
        super().__init__(f, remap_inp, remap_out, np.asarray(x0).size, hyper)
        self.in_shape = tuple(in_shape)
        self.v = None if v is None else np.asarray(v, dtype=np.float64).ravel()
        assert self.remap_inp.shape[1] == self.n
        if not _defer:
            assert self.v.size == self.n, "currently we assume the system is a full-rank mapping"
            self.init_xt0(x0, t0)
            self.solve_expansion_coeffs()

    def prepare_inp(self, xt):
        return xt[:self.n]

    def get_grad_t(self):
        return self.v

    def check_t0v_match(self, fx):
        """anm.cpp:343-360."""
        a = fx
        b = self.v * self.get_t0()
        maxerr = np.maximum(np.minimum(np.abs(a), np.abs(b)), 1.0) * self.hp.solution_check_tol
        bad = np.abs(a + b) > maxerr
        if bad.any():
            i = int(np.nonzero(bad)[0][0])
            raise SANMNumericalError(f"f(x0)+t0*v is not zero: lhs={a[i]} rhs={b[i]} idx={i} iter={self.iter}")

    def on_fx0_computed(self, fx):
        self.check_t0v_match(fx)
        return True


class ANMEqnSolver(ANMSolverVecScale):
    """Solve f(x) + y = 0; anm.h:245-283, anm.cpp:446-491."""

    def __init__(self, f, remap_inp, remap_out, in_shape, x0, y, hyper=None):
        hyper = hyper or HyperParam()
        super().__init__(f, remap_inp, remap_out, in_shape, x0, 0.0, None, hyper, _defer=True)
        self.converge_rms = hyper.converge_rms
        self.converged = False
        self.residual_rms = 0.0
        self.init_xt0(x0, 0.0)
        self.eqn_y = np.asarray(y, dtype=np.float64).ravel()
        assert self.eqn_y.size == self.n
        self.solve_expansion_coeffs()

    def next_iter(self):
        """anm.cpp:464-478."""
        if self.converged:
            return self
        if self.get_t_upper() >= 1:
            a = self.solve_a(1.0)
        else:
            a = self.t_max_a
        self.xt0 = self.eval_xt(a)
        self.xt0[self.n] = 0.0
        self.solve_expansion_coeffs()
        return self

    def on_fx0_computed(self, fx):
        """anm.cpp:480-491."""
        if self.converged:
            return False
        self.v = fx + self.eqn_y
        self.residual_rms = float(np.sqrt(np.mean(self.v ** 2)))
        if self.residual_rms < self.converge_rms:
            self.converged = True
            return False
        return True

    def get_x(self):
        return self.xt0[:self.n].copy()


class ANMImplicitSolver(ANMDriverHelper):
    """F(x,t) = F(x0,t0), F: R^(n+1) -> R^n; anm.h:285-305, anm.cpp:494-615."""

    def __init__(self, f, remap_inp, remap_out, in_shape, x0, t0, hyper=None):
        hyper = hyper or HyperParam()
        super().__init__(f, remap_inp, remap_out, np.asarray(x0).size, hyper)
        self.in_shape = tuple(in_shape)
        assert self.remap_inp.shape[1] == self.n + 1
        self.fx0 = None
        self._grad_t_from_build = None
        self.init_xt0(x0, t0)
        self.solve_expansion_coeffs()

    def prepare_inp(self, xt):
        return xt

    def get_grad_t(self):
        assert self._grad_t_from_build is not None
        return self._grad_t_from_build

    def on_fx0_computed(self, fx):
        if self.fx0 is None:
            self.fx0 = fx
        else:
            assert_allclose("check f(x0, t0)=f(x, t)", self.fx0, fx, self.hp.solution_check_tol)
        return True
