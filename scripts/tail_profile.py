"""Host-clock profile (mode 1: synchronised phases) of steady-state ANM steps: where the time between the order
loop and the next step goes.   python scripts/tail_profile.py [workload] [steps]"""
import sys
import time

import os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench  # noqa: E402
import sanm_amd
from sanm_amd import fea

w = sys.argv[1] if len(sys.argv) > 1 else "armadillo_small"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
api = sanm_amd.get_api()
cfg, mesh = bench.load_workload(w)
run = fea.GravityRun(api, mesh, dict(cfg)).construct()
s = run.solver
x0 = s.get_x()
s.run_steps(3, x0)
for mode in (1, 0):
    s.set_profile(mode)
    t0 = time.perf_counter()
    s.run_steps(steps, x0)
    dt = (time.perf_counter() - t0) / steps
    print(f"mode {mode}: {dt * 1e3:.3f} ms per step")
    if mode:
        p, c = s.profile(), s.profile_counts()
        for k in sorted(p, key=lambda k: -p[k]):
            print(f"  {k:32s} {p[k] / steps * 1e6:9.1f} us/step  ({c[k] / steps:.1f} calls)")
