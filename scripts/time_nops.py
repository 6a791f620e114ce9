import sys, os
os.environ["SANM_NO_JIT"] = "1"  # the truncation hook lives in the interpreter kernels
import torch
sys.path.insert(0, "/root/repo")
import sanm_amd
from sanm_amd import fea
api = sanm_amd.get_api(0)
cfg, mesh = fea.load_named_config("armadillo_small")
run = fea.GravityRun(api, mesh, cfg).construct()
run.step()
s = run.solver
for nops in range(0, 13):
    os.environ["SANM_DBG_NOPS"] = str(nops)
    print(nops, "COEFF %.1f" % (s.time_kernel(0, 50, 3, 5) * 1e3), "BIAS5 %.1f" % (s.time_kernel(0, 50, 2, 5) * 1e3), "BIAS20 %.1f" % (s.time_kernel(0, 50, 2, 20) * 1e3), "GRAD %.1f" % (s.time_kernel(0, 50, 1, 0) * 1e3))
