#!/usr/bin/env python3
"""Where a constructor's time goes when the device is waited for right after it: constructor call / wait, for a first and a
second solver of one process (GPU box):   python scripts/ctor_sync_probe.py [workload]"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sanm_amd import api as A, fea as dfea  # noqa: E402

import torch  # noqa: E402  (before the product library, as bench.py does)

torch.cuda.set_device(0)
wl = sys.argv[1] if len(sys.argv) > 1 else "armadillo_small"
api = bench.make_api(0)
cfg, mesh = bench.load_workload(wl)
run = dfea.GravityRun(api, mesh, dict(cfg))
keep = []
for i in range(3):
    bench.device_sync()
    t0 = time.perf_counter()
    s = A.ANMEqnSolver(api, run.model.y, run.model.lt_inp, run.model.lt_out, run.model.x0(), run.f_sub, run.hyper)
    t1 = time.perf_counter()
    bench.device_sync()
    t2 = time.perf_counter()
    prof = s.setup_profile()
    print(f"solver {i}: constructor {t1 - t0:.4f} s, wait for the device {t2 - t1:.4f} s, laps "
          f"{sum(v for k, v in prof.items() if isinstance(v, float) and k != 'analysis_thread'):.4f}", flush=True)
    s._keep = s._keep + (run.model,)
    keep.append(s)
