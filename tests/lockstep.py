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
  1. both sides expand at the same point: the oracle is re-started at the device's restart point, which was first
     compared (a) always, at 1e-8 of the step's increment, with the oracle's evaluation code applied to what the device
     exports -- its series and, with Pade, its own denominator --, and (b) where both sides took the same decision, at
     1e-4 with the oracle's own approximant (_check_restart; round 3 widened (b) by x50 after an adopted outcome);
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

LockStep drives an ANMEqnSolver (next_iter until converged), LockStepPath the path-following drivers ANMSolverVecScale /
ANMImplicitSolver (update_approx), lockstep_vtx_delta_stage one stage of run_with_vtx_delta (implicit solver, then the
order-6 refinement) -- BASELINE config 1 and the CLI's mesh_twist.  With arbiter=True every range estimate is also
taken in high precision (oracle/pade_hp.py) on both sides' series: scripts/pade_arbiter.py.
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
    delta = [min(max(g, 1e-13), 1e-6) for g in rel_gaps]  # (1e-6: what the series comparison admits at most)
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


def _eval_exported(dev_coeffs, dd, has_pade, a):
    """the device's approximant evaluated by the ORACLE's code from what the device exports: its series, and -- with
    Pade -- its own denominator (sanm_anm_pade_diag); pade.cpp:214-219 / unary_polynomial.cpp:115-126"""
    xs = [np.asarray(c, dtype=np.float64) for c in dev_coeffs]
    if has_pade:
        p = PadeApproximation.__new__(PadeApproximation)
        p.xs, p.d = xs, [float(v) for v in dd["d"]]
        return p.eval_xt(a)
    return up.eval_tensor(xs, a)


class _LockStepBase:
    """what the two forms of lock-step share: the comparison of one expansion's range estimate and of the restart
    point.  self.s: device solver (sanm_amd.api), self.o: the oracle's solver of the same class."""

    def _init(self, s, o, trials, seed, series_rtol, restart_rtol, arbiter):
        self.s, self.o, self.trials = s, o, trials
        # arbiter: every range estimate is also taken in high precision (oracle/pade_hp.py) on BOTH sides' series
        self.arbiter = arbiter
        self.series_rtol, self.restart_rtol = series_rtol, restart_rtol
        self.rng = np.random.default_rng(seed)
        self.steps, self.events = [], []

    def _compare_range(self, rec):
        """series, a_bound, and the outcome of the range estimate: identical or certified ill-conditioned (then the
        oracle adopts the device's outcome)"""
        s, o, k = self.s, self.o, rec["step"]
        dd, od = s.pade_diag(), o.pade_diags[-1]
        self._dev_coeffs, self._dev_diag = s.xt_coeffs(), dd
        rel, gap = series_gaps(self._dev_coeffs, o.xt_coeffs, o.a_bound)
        rec["series_gap"] = gap
        rec["coeff_gap_1_2_N"] = (rel[1], rel[2], rel[-1])
        rec["coeff_gaps"] = list(rel)
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
        if self.arbiter and dd["attempted"]:
            rec["arbiter"] = self._arbitrate(self._dev_coeffs, dev, own)
        if dd["attempted"] and dd["built"]:
            # the root finder is a deterministic function of the coefficients, bit-exact on both sides
            assert (up.real_roots(list(dd["d"])) is not None) == bool(dd["roots_valid"]), \
                f"step {k}: root finder outcome on the device's own denominator"
        rec["forced"] = False
        if dev[0] != own[0] or abs(dev[1] - own[1]) > (1e-6 + a_tol) * own[1]:
            ok, draws, delta = certify(o, dev, rel, self.rng, self.trials)
            ev = {"step": k, "device": dev, "oracle": own, "series_gap": gap, "perturbation": delta,
                  "device_roots_valid": bool(dd.get("roots_valid")), "oracle_roots_valid": bool(od.get("roots_valid")),
                  "draws": sorted(set(draws)), "certified": ok}
            self.events.append(ev)
            # (strict=False: a device that runs another algorithm on purpose -- SANM_PADE_ORTH=cgs2 under
            # scripts/pade_arbiter.py -- is recorded, not judged, and the oracle follows it)
            assert ok or not getattr(self, "strict", True), \
                f"step {k}: outcome {dev} vs the oracle's {own} is not an ill-conditioned decision: {ev}"
            rec["forced"] = True
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

    def _check_restart(self, rec, a_d, xt_d, xt_o, nx):
        """the device's restart point xt_d = its approximant at its parameter a_d, twice:
        (i) ALWAYS, against the oracle's evaluation code applied to what the device exports (its series and, with
            Pade, its own denominator): 1e-8 of the step's increment -- the evaluation arithmetic, free of any
            decision;
        (ii) where the two sides took the same decision, against the oracle's OWN approximant at the oracle's
            parameter (restart_rtol = 1e-4 of the increment: two valid approximants of one path).  After an adopted
            outcome the oracle holds an approximant its own range test did not accept at that parameter, so there is
            nothing of the oracle's to compare with (round 3 widened (ii) by x50 there; now the figure is recorded
            only and the oracle restarts at the device's point, from where the next expansion is compared)."""
        k = rec["step"]
        x0 = np.asarray(self._dev_coeffs[0], dtype=np.float64)[:nx]
        scale = float(np.abs(xt_o[:nx] - self.o.xt0[:nx]).max())  # the step's increment
        own = _eval_exported(self._dev_coeffs, self._dev_diag, bool(rec["device"][0]), a_d)
        err_self = float(np.abs(xt_d[:nx] - own[:nx]).max())
        assert err_self <= 1e-8 * max(scale, float(np.abs(own[:nx] - x0).max())) + 1e-12 * float(np.abs(own[:nx]).max()), \
            f"step {k}: the device's restart point is not its own approximant at a = {a_d}: {err_self:.2e}"
        err = float(np.abs(xt_d[:nx] - xt_o[:nx]).max())
        rec["restart_rel_err"] = err / scale if scale > 0 else 0.0
        rec["restart_self_err"] = err_self / scale if scale > 0 else 0.0
        if not rec["forced"]:
            assert err <= self.restart_rtol * scale + 1e-12 * float(np.abs(xt_o[:nx]).max()), \
                f"step {k}: restart points differ by {err:.2e} (increment {scale:.2e})"

    def _arbitrate(self, dev_coeffs, dev, own):
        """the same range estimate in high precision (exact inner products, 200-digit algebra: oracle/pade_hp.py) on
        the device's series and on the oracle's: which fp64 side took the decision the rounding-free algorithm takes"""
        from oracle import pade_hp
        o, hp = self.o, self.o.hp

        def same(a, b):
            return a[0] == b[0] and abs(a[1] - b[1]) <= 2e-3 * max(abs(b[1]), 1e-300)

        out = {}
        for side, xs, actual in (("device", [np.asarray(c, dtype=np.float64) for c in dev_coeffs], dev),
                                 ("oracle", o.xt_coeffs, own)):
            r = pade_hp.arbitrate(xs, not hp.xcoeff_l2_penalty, o.a_bound, hp.maxr, o.max_a_bound)
            e = {"fp64": list(actual)}
            for key in ("exact", "ref_roots"):
                dg = r[key]
                oc = _outcome(dg, o.a_bound)
                e[key] = {"outcome": list(oc), "roots_valid": dg["roots_valid"], "pole": dg["pole"],
                          "margin_left": dg["probes"][0][1] if dg["probes"] else None,
                          "agrees_with_fp64": bool(same(actual, oc))}
            out[side] = e
        out["hp_outcomes_of_both_series_agree"] = {
            key: bool(same(tuple(out["device"][key]["outcome"]), tuple(out["oracle"][key]["outcome"])))
            for key in ("exact", "ref_roots")}
        return out

    @property
    def nr_steps(self):
        return len(self.steps) - 1

    def summary(self):
        def clean(v):
            if isinstance(v, (tuple, list)):
                return [clean(x) for x in v]
            if isinstance(v, dict):
                return {k: clean(x) for k, x in v.items()}
            if isinstance(v, (np.floating, np.bool_, np.integer)):
                return v.item()
            return v
        return {"steps": self.nr_steps, "events": [{k: clean(v) for k, v in e.items()} for e in self.events],
                "per_step": [{k: clean(v) for k, v in r.items()} for r in self.steps]}


class LockStep(_LockStepBase):
    """ANMEqnSolver.  run: a sanm_amd.fea.GravityRun after construct() (or any object with .solver and .step());
    osolver: the oracle's ANMEqnSolver for the same task."""

    def __init__(self, run, osolver, trials=256, seed=0, series_rtol=1e-6, restart_rtol=1e-4, arbiter=False):
        self.run = run
        self._init(run.solver, osolver, trials, seed, series_rtol, restart_rtol, arbiter)
        self._compare_expansion()

    # -- one expansion from a common state --------------------------------------------------------------------
    def _compare_expansion(self):
        s, o = self.s, self.o
        k = len(self.steps)
        rec = {"step": k, "rms": (float(s.residual_rms()), float(o.residual_rms))}
        # (absolute floor: a thousandth of the convergence threshold the residual is compared with -- 1e-13 for the
        # reference's 1e-10 of fea/main.cpp:28; the order-6 refinement of run_with_vtx_delta converges at 1e-5 and
        # ends in residuals that are pure round-off of the nodal sums, 3e-13 against 5e-13)
        floor = max(1e-13, 1e-3 * float(o.converge_rms))
        assert abs(rec["rms"][0] - rec["rms"][1]) <= 1e-7 * rec["rms"][1] + floor, f"step {k}: rms {rec['rms']}"
        assert bool(s.converged()) == bool(o.converged), f"step {k}: converged on one side only, rms {rec['rms']}"
        if o.converged:
            rec["converged"] = True
            self.steps.append(rec)
            return
        self._compare_range(rec)
        self.steps.append(rec)

    def step(self):
        s, o = self.s, self.o
        # the restart point (ANMEqnSolver::next_iter, anm.cpp:464-478): parameter, then x(a)
        a_o = o.solve_a(1.0) if o.get_t_upper() >= 1 else o.t_max_a
        x_o = o.eval_xt(a_o)
        a_d = s.solve_a(1.0) if s.get_t_upper() >= 1 else s.get_t_max_a()
        self.run.step()
        x_d = s.get_x()
        self._check_restart(self.steps[-1], a_d, x_d, x_o, o.n)
        # common state: the oracle expands at the device's restart point
        o.init_xt0(x_d, 0.0)
        o.solve_expansion_coeffs()
        self._compare_expansion()

    def run_to_convergence(self, max_steps=200):
        while not self.s.converged():
            self.step()
            assert len(self.steps) < max_steps
        return self


class LockStepPath(_LockStepBase):
    """The path-following drivers -- ANMSolverVecScale and ANMImplicitSolver (anm.h:209-243, :285-305) -- followed
    with update_approx (anm.cpp:156-159): every expansion compared from a common (x, t), Pade outcomes identical or
    certified, restart points as in LockStep.  dsol: the device solver, osol: the oracle's, same class and inputs."""

    def __init__(self, dsol, osol, trials=256, seed=0, series_rtol=1e-6, restart_rtol=1e-4, arbiter=False):
        self._init(dsol, osol, trials, seed, series_rtol, restart_rtol, arbiter)
        self._compare_expansion()

    def _compare_expansion(self):
        rec = {"step": len(self.steps)}
        self._compare_range(rec)
        rec["t_upper"] = (float(self.s.get_t_upper()), float(self.o.get_t_upper()))
        if not rec["forced"]:
            assert abs(rec["t_upper"][0] - rec["t_upper"][1]) <= 1e-5 * abs(rec["t_upper"][1]) + 1e-12, \
                f"step {rec['step']}: t_upper {rec['t_upper']}"
        self.steps.append(rec)

    def update_approx(self):
        s, o = self.s, self.o
        xt_o = o.eval_xt(o.t_max_a)
        a_d = s.get_t_max_a()
        s.update_approx()
        xt_d = np.asarray(s.xt_coeffs()[0], dtype=np.float64)
        self._check_restart(self.steps[-1], a_d, xt_d, xt_o, o.n + 1)
        o.xt0 = xt_d.copy()
        o.solve_expansion_coeffs()
        self._compare_expansion()

    def follow_to(self, t_dst, inclusive=True, max_steps=200):
        """update_approx until t_upper passes t_dst: `t_upper > t_dst` ends the loop of tests/symbolic.cpp:28-54
        (inclusive), `t_upper >= t_dst` that of fea/main.cpp:193-215 (inclusive=False)"""
        while self.s.get_t_upper() < t_dst or (inclusive and self.s.get_t_upper() == t_dst):
            self.update_approx()
            assert len(self.steps) < max_steps
        return self


class _EqnAdapter:
    """a bare device ANMEqnSolver in the shape LockStep expects of a GravityRun"""

    def __init__(self, solver):
        self.solver = solver

    def step(self):
        self.solver.next_iter()


def lockstep_vtx_delta_stage(api, dmesh, omesh, omat, fixed, config, vtx_delta, vtx_coord, require_refine,
                             refine_f_load=None, arbiter=False):
    """One stage of run_with_vtx_delta (fea/main.cpp:436-580; sanm_amd/fea.py and oracle/fea.py restate it) with
    the device and the oracle in lock step: the displacement-driven ANMImplicitSolver followed to t = 1
    (LockStepPath), then the order-6 ANMEqnSolver refinement from the DEVICE's vertices on both sides (LockStep).
    Returns (device vertices after the stage, {"iter_deform", "iter_refine", "events"})."""
    from oracle import fea as ofea
    from oracle.anm import ANMEqnSolver as OEqn, ANMImplicitSolver as OImplicit
    from sanm_amd import fea as dfea
    from sanm_amd.api import ANMEqnSolver, ANMImplicitSolver
    energy, mc = config["energy_model"], config["material"]
    model = api.fea_model(dmesh.V, dmesh.tets, fixed, energy, float(mc["young"]), float(mc["poisson"]),
                          init_vtx_coord=vtx_coord, vtx_delta=vtx_delta)
    hp = dfea.hyper_from_config(api, config, solution_check_tol=10.0, converge_rms=1e-5)
    dsol = ANMImplicitSolver(api, model.y, model.lt_inp, model.lt_out, model.x0(), 0.0, hp)
    om = ofea.make_forward(omesh, omat, fixed, energy, vtx_coord, vtx_delta)
    osol = OImplicit(om.y, om.lt_inp.mat, om.lt_out, om.lt_inp.out_shape, om.lt_inp.x0, 0.0,
                     ofea.default_hyper(config, solution_check_tol=10.0))
    lp = LockStepPath(dsol, osol, arbiter=arbiter).follow_to(1.0, inclusive=False)
    xt = dsol.eval(dsol.solve_a(1.0))[0]
    xo = osol.eval(osol.solve_a(1.0))[0]
    # (both approximants meet the range criterion at t = 1; same tolerance as two restart points)
    scale = float(np.abs(xo - osol.xt_coeffs[0][:osol.n]).max())
    if not any(r["forced"] for r in lp.steps[-1:]):
        assert np.abs(xt - xo).max() <= 1e-4 * scale + 1e-12 * np.abs(xo).max()
    out = {"iter_deform": int(dsol.get_nr_iter()), "iter_refine": 0, "events": list(lp.events), "path": lp}
    vtx = model.full_vertices(xt, vtx_coord) + vtx_delta
    frms = dfea._force_rms(api, dmesh, fixed, energy, mc, vtx)
    if require_refine or frms >= 1e-10:
        m2 = api.fea_model(dmesh.V, dmesh.tets, fixed, energy, float(mc["young"]), float(mc["poisson"]),
                           init_vtx_coord=vtx)
        hp2 = dfea.hyper_from_config(api, config, converge_rms=1e-5, solution_check_tol=1e-4, order=6)
        f_sub = np.zeros(m2.n) if refine_f_load is None else m2.copy_vtx_values(refine_f_load)
        s2 = ANMEqnSolver(api, m2.y, m2.lt_inp, m2.lt_out, m2.x0(), f_sub, hp2)
        om2 = ofea.make_forward(omesh, omat, fixed, energy, vtx)
        oh2 = ofea.default_hyper(config, converge_rms=1e-5)
        oh2.order = 6
        o2 = OEqn(om2.y, om2.lt_inp.mat, om2.lt_out, om2.lt_inp.out_shape, om2.lt_inp.x0,
                  np.zeros(om2.lt_inp.n) if refine_f_load is None else om2.lt_inp.copy_vtx_values(refine_f_load), oh2)
        ls = LockStep(_EqnAdapter(s2), o2, arbiter=arbiter).run_to_convergence()
        vtx = m2.full_vertices(s2.get_x(), vtx)
        out["iter_refine"] = int(s2.get_nr_iter())
        out["events"] += ls.events
    vtx = np.where(fixed, dmesh.V + vtx_delta, vtx)
    return vtx, out
