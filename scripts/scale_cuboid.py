"""Scaling probe on a synthetic cuboid (stand-in for the missing full Armadillo mesh): the
armadillo material / gravity / boundary rule on an nx x ny x nz vertex grid (5 tets per cell).
   python scripts/scale_cuboid.py NX [steps]"""
import json
import sys
import time

import torch  # noqa: F401  (before the HIP library)

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sanm_amd
from sanm_amd import fea

nx = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg, _ = fea.load_named_config("armadillo_small")
cfg = dict(cfg)
cfg.pop("scale", None)
cfg["material"] = dict(cfg["material"], young=float(sys.argv[3]) if len(sys.argv) > 3 else 2.0e4)  # soft: several steps
mesh = fea.make_cuboid(nx, nx, nx, 0.2 / nx)  # a 0.2 m block like the scaled armadillo
api = sanm_amd.get_api(0)
t0 = time.perf_counter()
run = fea.GravityRun(api, mesh, cfg, profile=1)
t1 = time.perf_counter()
run.construct()  # first continuation step (includes the analysis)
t2 = time.perf_counter()
ts = []
for _ in range(steps):
    if run.solver.converged():
        break
    a = time.perf_counter()
    run.step()
    ts.append(time.perf_counter() - a)
st = run.solver.stats()
print(json.dumps({"nx": nx, "T": mesh.nr_tet, "n": st["nr_unknown"], "nnz": st["jacobian_nnz"],
                  "prep_s": round(t1 - t0, 2), "first_step_incl_analysis_s": round(t2 - t1, 2),
                  "step_ms": [round(t * 1e3, 1) for t in ts], "rms": run.rms[-1],
                  "factor_nnz": st["factor_nnz"], "factor_gflop": round(st["factor_flops"] / 1e9, 1),
                  "levels": st["nr_level"], "max_front": st["max_front"],
                  "profile_s_per_step": {k: round(v / max(run.solver.get_nr_iter(), 1), 4)
                                         for k, v in run.solver.profile().items()}}))
