"""Per-(mode, order) steady-state time of the Taylor pass kernel: python scripts/time_passes.py [workload]"""
import sys
import torch  # noqa: F401  (before the HIP library)
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sanm_amd
from sanm_amd import fea

api = sanm_amd.get_api(0)
cfg, mesh = fea.load_named_config(sys.argv[1] if len(sys.argv) > 1 else "armadillo_small")
run = fea.GravityRun(api, mesh, cfg).construct()
run.step()
s = run.solver
names = {0: "EVAL0", 1: "GRAD", 2: "BIAS", 3: "COEFF"}
print("EVAL0 %.1f us  GRAD %.1f us" % (s.time_kernel(0, 50, 0, 0) * 1e3, s.time_kernel(0, 50, 1, 0) * 1e3))
for mode in (2, 3):
    print(names[mode], " ".join("%.0f" % (s.time_kernel(0, 50, mode, k) * 1e3) for k in range(1, 21 if mode == 2 else 20)))
try:
    print("COEFF(k)+BIAS(k+1) fused", " ".join("%.0f" % (s.time_kernel(0, 50, 4, k) * 1e3) for k in range(1, 20)))
except Exception as e:  # interpreter kernels only: no fused pass
    print("fused pass:", e)
