#!/usr/bin/env python3
"""Third arbiter for the Pade decisions: every range estimate of the lock-step continuations is
taken a third time in HIGH precision (oracle/pade_hp.py: exact inner products of the series vectors, the reference's
Gram-Schmidt / solve_d / probes / bisection in 200-digit arithmetic) on the device's series and on the fp64 oracle's,
and the record says, per decision, which fp64 side agrees with the rounding-free outcome.

  python scripts/pade_arbiter.py [--api hip|hostsim] [--cases cuboid_nc,...,human_arap16] [--out file.json]

Test infrastructure: drives the product through the C ABI beside the oracle, exactly as tests/lockstep.py does."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import fea as ofea  # noqa: E402
from sanm_amd import fea as dfea  # noqa: E402
from tests.lockstep import LockStep  # noqa: E402

CUBOIDS = ["cuboid_nc", "cuboid_ni", "cuboid_arap", "cuboid_nc_l2", "smoke_6x3x3"]
FULL = ["bob", "armadillo_small", "human_arap16"]


def make_case(api, name):
    if name == "smoke_6x3x3":
        cfg = {"material": {"young": 3e3, "poisson": 0.45, "density": 1000.0}, "g": [0, -9.81, 0],
               "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_c", "order": 12}
        run = dfea.GravityRun(api, dfea.make_cuboid(6, 3, 3, 0.025), dict(cfg), solver_rtol=1e-15).construct()
        _, osolver, _ = ofea.make_gravity_solver(ofea.make_cuboid(6, 3, 3, 0.025), cfg)
        return run, osolver
    if name.startswith("cuboid_"):
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", f"anm_{name}.json")))
        run = dfea.GravityRun(api, dfea.make_cuboid(*gold["dims"], gold["spacing"]), dict(gold["config"]),
                              solver_rtol=1e-15).construct()
        _, osolver, _ = ofea.make_gravity_solver(ofea.make_cuboid(*gold["dims"], gold["spacing"]), gold["config"])
        return run, osolver
    cfg, mesh = dfea.load_named_config(name)
    run = dfea.GravityRun(api, mesh, dict(cfg)).construct()
    cfg2, mesh2 = dfea.load_named_config(name)
    _, osolver, _ = ofea.make_gravity_solver(ofea.TetMesh(mesh2.V, mesh2.tets, mesh2.surface_vtx), cfg2)
    return run, osolver


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--api", default="hip", choices=["hip", "hostsim"])
    ap.add_argument("--cases", default=",".join(CUBOIDS + FULL))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_pade_arbiter.json"))
    ap.add_argument("--lenient", action="store_true",
                    help="record decisions the oracle would not take under any perturbation instead of failing on them "
                         "(a device that runs another orthogonalisation on purpose: SANM_PADE_ORTH=cgs2)")
    args = ap.parse_args()
    if args.api == "hip":
        import sanm_amd
        api = sanm_amd.get_api(0)
    else:
        from tests.hostsim import get_hostsim_api
        api = get_hostsim_api()
    res = {"device_backend": api.backend_name(), "pade_orth": os.environ.get("SANM_PADE_ORTH", "cgs (the reference's)"),
           "precision": "exact inner products + 200-digit algebra "
           "(oracle/pade_hp.py)", "cases": {}}
    tally = {"decisions": 0, "fp64_sides_agree": 0, "events": 0,
             "event_device_matches_hp": 0, "event_oracle_matches_hp": 0, "event_both": 0, "event_neither": 0,
             "hp_of_both_series_agree": 0, "hp_exact_of_both_series_agree": 0,
             "device_matches_hp_overall": 0, "oracle_matches_hp_overall": 0}
    for name in args.cases.split(","):
        t0 = time.time()
        run, osolver = make_case(api, name)
        ls = LockStep(run, osolver, arbiter=True)
        ls.strict = not args.lenient
        ls = ls.run_to_convergence()
        rows = []
        for rec in ls.steps:
            arb = rec.get("arbiter")
            if not arb:
                continue
            d, o = arb["device"], arb["oracle"]
            # the reference's CODE with a rounding-free Gram-Schmidt: the high-precision denominator through the
            # reference's own root finder ("ref_roots")
            dm, om = d["ref_roots"]["agrees_with_fp64"], o["ref_roots"]["agrees_with_fp64"]
            is_event = any(e["step"] == rec["step"] for e in ls.events)
            tally["decisions"] += 1
            tally["fp64_sides_agree"] += int(not is_event)
            tally["device_matches_hp_overall"] += int(dm)
            tally["oracle_matches_hp_overall"] += int(om)
            tally["hp_of_both_series_agree"] += int(arb["hp_outcomes_of_both_series_agree"]["ref_roots"])
            tally["hp_exact_of_both_series_agree"] += int(arb["hp_outcomes_of_both_series_agree"]["exact"])
            if is_event:
                tally["events"] += 1
                key = "event_both" if dm and om else "event_device_matches_hp" if dm else \
                    "event_oracle_matches_hp" if om else "event_neither"
                tally[key] += 1
            rows.append({"step": rec["step"], "event": is_event, "series_gap": rec["series_gap"],
                         "coeff_gaps": rec["coeff_gaps"], "device_fp64": d["fp64"], "oracle_fp64": o["fp64"],
                         "hp_on_device_series": {k: d[k] for k in ("exact", "ref_roots")},
                         "hp_on_oracle_series": {k: o[k] for k in ("exact", "ref_roots")},
                         "hp_outcomes_of_both_series_agree": arb["hp_outcomes_of_both_series_agree"]})
        res["cases"][name] = {"steps": ls.nr_steps, "seconds": round(time.time() - t0, 1), "decisions": rows}
        print(name, "steps", ls.nr_steps, "decisions", len(rows), "events", len(ls.events),
              f"{time.time() - t0:.0f}s", flush=True)
        for r in rows:
            print("  step", r["step"], "EVENT" if r["event"] else "     ", "dev", r["device_fp64"], "orc", r["oracle_fp64"],
                  "| HP(dev)", r["hp_on_device_series"]["ref_roots"]["outcome"],
                  "HP(orc)", r["hp_on_oracle_series"]["ref_roots"]["outcome"],
                  "| exact:", r["hp_on_device_series"]["exact"]["outcome"], flush=True)
    res["tally"] = tally
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps(tally))


if __name__ == "__main__":
    main()
