"""Step-by-step ("lock-step") parity of a device continuation with the oracle's.

Why not simply compare two free-running continuations?  The step sequence is decided by discrete tests (SURVEY
appendix C.7): the outcome of the root finder on the Pade denominator (None => approximant rejected,
pade.cpp:113-116), the smallest positive pole, check(left) and the bisection probes (pade.cpp:129-165).  They are
taken on the denominator of the vector Pade approximant, which comes out of a CLASSICAL Gram-Schmidt sweep over the
nearly parallel series vectors (pade.cpp:30-55) and a triangular solve with its tiny diagonal (pade.cpp:67-79).
Measured here: series that agree to 1e-12 give denominators that agree to 1e-4 only; probe margins of 5.9 against
0.14 at the same point; poles of 0.9 against 1e16; and the root finder (ACM algorithm 30 on an ill-scaled degree-19
polynomial) turns that into different valid / None outcomes -- the reference itself changes its answers with
-march=native (tests/golden/ref_poly.json: native_valid).  Where that happens the decisions are ill-conditioned
functions of the series and no two implementations with different rounding (summation order of a dot product is
enough) take the same ones; the continuations then pass through different intermediate states -- each a valid
approximation within the range criterion -- to the same equilibrium.

What is checked instead, at EVERY step, from a COMMON state:
  1. both sides expand at the same point (the oracle is re-started at the device's restart point, which was first
     compared with the oracle's own evaluation of the approximant at the device's parameter);
  2. residual RMS (1e-7), the series as functions on [0, a_bound] and their first coefficient (1e-6) and the
     plain-series range a_bound (1e-7 + the relative error it inherits from |x_1| and |x_N|) agree;
  3. the outcome of the range estimate -- Pade accepted or not, accepted range -- is either identical, or CERTIFIED
     ill-conditioned: the oracle's own series of that step, perturbed by relative noise of the size of the measured
     device-oracle difference of each series coefficient (>= 1e-13), makes the ORACLE produce the device's flag, and ranges on
     both sides of the device's, within `trials` draws.  A decision the oracle keeps under all perturbations but the
     device takes differently fails the test;
  4. the root finder is outside that uncertainty (bit-exact on both sides, tests/golden/ref_poly.json): on the
     DEVICE's denominator the oracle's implementation must reproduce the device's valid flag exactly;
  5. convergence is declared at the same step, and the equilibrium agrees to 1e-6 relative (north_star).
The free-running oracle's step count is reported next to it; callers assert equality where no event occurred.
"""
import numpy as np

from oracle import unary_polynomial as up
from oracle.pade import PadeApproximation


def series_gaps(dev_coeffs, orc_coeffs, a):
    """per coefficient: relative difference of the two sides' vectors (max norm); and the difference of the two
    SERIES on [0, a]: max_k |dx_k| a^k / max_k |x_k| a^k, k >= 1.  (Near convergence the high-order coefficients
    are tiny and carry the round-off of the bias evaluation at a relative size that their own norm does not
    excuse but their weight in the series does.)"""
    rel, wd, wx = [], 0.0, 0.0
    for k, (d, o) in enumerate(zip(dev_coeffs, orc_coeffs)):
        sc = float(np.abs(o).max())
        df = float(np.abs(np.asarray(d) - o).max())
        rel.append(df / sc if sc > 0 else 0.0)
        if k >= 1:
            wd = max(wd, df * a ** k)
            wx = max(wx, sc * a ** k)
    return rel, (wd / wx if wx > 0 else 0.0)


def _outcome(diag, a_bound):
    return (bool(diag.get("accepted")), float(diag["t_max_a"]) if diag.get("accepted") else float(a_bound))


def certify(osolver, dev_outcome, rel_gaps, rng, trials):
    """the oracle's range estimate on perturbed copies of its own series (coefficient k by relative noise of the
    size of the measured device-oracle difference of that coefficient): returns (certified, draws, noise level)"""
    hp = osolver.hp
    delta = [min(max(g, 1e-13), 1e-4) for g in rel_gaps]
    own = _outcome(osolver.pade_diags[-1], osolver.a_bound)
    draws = [own]
    for _ in range(trials):
        xs = [x * (1.0 + dk * rng.standard_normal(x.shape)) for x, dk in zip(osolver.xt_coeffs, delta)]
        p = PadeApproximation(xs, not hp.xcoeff_l2_penalty, False)
        p.estimate_valid_range(osolver.a_bound, hp.maxr, osolver.max_a_bound)
        draws.append(_outcome(p.diag, osolver.a_bound))
        flags = {d[0] for d in draws}
        same = [d[1] for d in draws if d[0] == dev_outcome[0]]
        if dev_outcome[0] in flags and (not dev_outcome[0] or
                                        min(same) <= dev_outcome[1] * (1 + 1e-6) and max(same) >= dev_outcome[1] * (1 - 1e-6)):
            return True, draws, max(delta)
    return False, draws, max(delta)


class LockStep:
    """run: a sanm_amd.fea.GravityRun after construct(); osolver: the oracle's ANMEqnSolver for the same task."""

    def __init__(self, run, osolver, trials=256, seed=0, series_rtol=1e-6, restart_rtol=1e-4):
        self.run, self.o, self.trials = run, osolver, trials
        self.series_rtol, self.restart_rtol = series_rtol, restart_rtol
        self.rng = np.random.default_rng(seed)
        self.steps, self.events = [], []
        self._compare_expansion()

    # -- one expansion from a common state --------------------------------------------------------------------
    def _compare_expansion(self):
        s, o = self.run.solver, self.o
        k = len(self.steps)
        rec = {"step": k, "rms": (float(s.residual_rms()), float(o.residual_rms))}
        assert abs(rec["rms"][0] - rec["rms"][1]) <= 1e-7 * rec["rms"][1] + 1e-13, f"step {k}: rms {rec['rms']}"
        assert bool(s.converged()) == bool(o.converged), f"step {k}: converged on one side only, rms {rec['rms']}"
        if o.converged:
            rec["converged"] = True
            self.steps.append(rec)
            return
        dd, od = s.pade_diag(), o.pade_diags[-1]
        rel, gap = series_gaps(s.xt_coeffs(), o.xt_coeffs, o.a_bound)
        rec["series_gap"] = gap
        rec["coeff_gap_1_2_N"] = (rel[1], rel[2], rel[-1])
        assert gap <= self.series_rtol and rel[1] <= self.series_rtol, \
            f"step {k}: the series differ by {gap:.2e} on [0, a_bound] (x_1: {rel[1]:.2e})"
        assert bool(dd["attempted"]) == bool(od["attempted"]), f"step {k}: Pade attempted on one side only"
        a_bound_dev = dd["start"] if dd["attempted"] else s.get_t_max_a()
        # a_bound = (maxr |x_1| / |x_N|)^(1/(N-1)), anm.cpp:126: it inherits the relative error of the two norms
        N = len(rel) - 1
        a_tol = 1e-7 + (rel[1] + rel[-1]) / (N - 1)
        assert abs(a_bound_dev - o.a_bound) <= a_tol * o.a_bound, f"step {k}: a_bound {a_bound_dev} vs {o.a_bound}"
        dev = (bool(s.has_pade()), float(s.get_t_max_a()))
        own = _outcome(od, o.a_bound)
        rec.update(device=dev, oracle=own, margin_left=(dd["probes"][0][1] if dd["probes"] else None,
                                                        od["probes"][0][1] if od.get("probes") else None))
        if dd["attempted"] and dd["built"]:
            # the root finder is a deterministic function of the coefficients, bit-exact on both sides
            assert (up.real_roots(list(dd["d"])) is not None) == bool(dd["roots_valid"]), \
                f"step {k}: root finder outcome on the device's own denominator"
        if dev[0] != own[0] or abs(dev[1] - own[1]) > (1e-6 + a_tol) * own[1]:
            ok, draws, delta = certify(o, dev, rel, self.rng, self.trials)
            ev = {"step": k, "device": dev, "oracle": own, "series_gap": gap, "perturbation": delta,
                  "device_roots_valid": bool(dd.get("roots_valid")), "oracle_roots_valid": bool(od.get("roots_valid")),
                  "draws": sorted(set(draws)), "certified": ok}
            self.events.append(ev)
            assert ok, f"step {k}: outcome {dev} vs the oracle's {own} is not an ill-conditioned decision: {ev}"
            # the oracle continues with the device's outcome
            if dev[0]:
                o.pade = o.pade_candidate
                o.pade.t_max_a = dev[1]
                o.pade.t_max = o.pade.eval_t(dev[1])
                o.t_max_a, o.t_max = o.pade.t_max_a, o.pade.t_max
            else:
                o.pade = None
                o.t_max_a = o.a_bound
                o.t_max = up.eval_poly(o.t_coeffs, o.a_bound)
        self.steps.append(rec)

    def step(self):
        s, o = self.run.solver, self.o
        k = len(self.steps)
        # the restart point (ANMEqnSolver::next_iter, anm.cpp:464-478): parameter, then x(a)
        a_o = o.solve_a(1.0) if o.get_t_upper() >= 1 else o.t_max_a
        x_o = o.eval_xt(a_o)[:o.n]
        self.run.step()
        x_d = s.get_x()
        scale = float(np.abs(x_o - o.xt0[:o.n]).max())  # the step's displacement increment
        err = float(np.abs(x_d - x_o).max())
        # (after an adopted outcome the oracle evaluates its approximant at a parameter its own range test did not
        # accept: it is then only as good as the approximant is there)
        forced = bool(self.events) and self.events[-1]["step"] == k - 1
        rtol = 50 * self.restart_rtol if forced else self.restart_rtol
        assert err <= rtol * scale + 1e-12 * float(np.abs(x_o).max()), \
            f"step {k}: restart points differ by {err:.2e} (increment {scale:.2e})"
        self.steps[-1]["restart_rel_err"] = err / scale if scale > 0 else 0.0
        # common state: the oracle expands at the device's restart point
        o.init_xt0(x_d, 0.0)
        o.solve_expansion_coeffs()
        self._compare_expansion()

    def run_to_convergence(self, max_steps=200):
        while not self.run.solver.converged():
            self.step()
            assert len(self.steps) < max_steps
        return self

    @property
    def nr_steps(self):
        return len(self.steps) - 1

    def summary(self):
        def clean(v):
            if isinstance(v, (tuple, list)):
                return [clean(x) for x in v]
            if isinstance(v, (np.floating, np.bool_, np.integer)):
                return v.item()
            return v
        return {"steps": self.nr_steps, "events": [{k: clean(v) for k, v in e.items()} for e in self.events],
                "per_step": [{k: clean(v) for k, v in r.items()} for r in self.steps]}


def counts_compatible(device_steps, oracle_steps):
    """free-running step counts of two continuations with Pade on: equal unless an ill-conditioned Pade decision
    was met on the way (see the module docstring) -- then still the same handful.  Tests that need the decision
    by decision statement use LockStep."""
    return device_steps >= 1 and oracle_steps >= 1 and abs(device_steps - oracle_steps) <= max(2, oracle_steps // 3)
