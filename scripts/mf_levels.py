#!/usr/bin/env python3
"""The level table of the multifrontal analysis of a workload (SANM_MF_DEBUG), CPU only through the host harness:
   python scripts/mf_levels.py refine:armadillo_small:1"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
os.environ["SANM_MF_DEBUG"] = "1"
import bench  # noqa: E402
from dist_plan import jacobian_pattern  # noqa: E402
from sanm_amd import api as A, fea as dfea  # noqa: E402
from tests.hostsim import get_hostsim_api  # noqa: E402

api = get_hostsim_api()
cfg, mesh = bench.load_workload(sys.argv[1])
fixed, _ = dfea.setup_gravity(api, mesh, cfg)
P, coords = jacobian_pattern(mesh, fixed)
t = time.time()
s = A.DirectSolver(api, P, coords)
print("analysis seconds", time.time() - t, s.stats())
