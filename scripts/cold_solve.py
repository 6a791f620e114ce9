"""One cold whole solve of a workload (constructor to convergence), nothing else: for API traces of the first solve in a process.
   python scripts/cold_solve.py [workload]"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench  # noqa: E402
import sanm_amd  # noqa: E402
from sanm_amd import fea  # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "armadillo_small"
api = sanm_amd.get_api(0)
cfg, mesh = bench.load_workload(w)
t0 = time.perf_counter()
run = fea.GravityRun(api, mesh, dict(cfg)).run()
print("time_solve", time.perf_counter() - t0, "steps", run.solver.get_nr_iter(), run.solver.setup_profile())
