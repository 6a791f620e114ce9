"""With a library built with -DSANM_MF_PHASES (SANM_EXTRA_CXXFLAGS): phase times of the panel launches' critical
workgroup on the Jacobian of a workload.   python scripts/mf_phases.py [workload]"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench  # noqa: E402
import sanm_amd  # noqa: E402
from sanm_amd import fea  # noqa: E402
from sanm_amd.api import DirectSolver  # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "armadillo_small"
api = sanm_amd.get_api()
cfg, mesh = bench.load_workload(w)
run = fea.GravityRun(api, mesh, dict(cfg)).construct()
run.step()
A = run.solver.jacobian_csr().tocsr()
A.sort_indices()
coords = mesh.V[run.model.lt_inp.vertex_loc[:, 0]] if hasattr(run.model.lt_inp, "vertex_loc") else None
ds = DirectSolver(api, A, coords)
for _ in range(3):
    print("factor", ds.factor(A))
