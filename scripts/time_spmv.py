"""steady-state time of the plain CSR SpMV kernel on the armadillo Jacobian: python scripts/time_spmv.py"""
import sys
import torch  # noqa: F401
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sanm_amd
from sanm_amd import fea
api = sanm_amd.get_api(0)
cfg, mesh = fea.load_named_config("armadillo_small")
run = fea.GravityRun(api, mesh, cfg).construct()
run.step()
print("spmv %.1f us" % (run.solver.time_kernel(1, 200, 0, 0) * 1e3))
