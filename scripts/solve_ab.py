"""Factor the Jacobian of a named config once and run `reps` solves back to back: a target for
`rocprofv3 --kernel-trace --stats` when a variant of the solve kernels is to be compared in isolation
(scripts/build_variants.py + SANM_HIP_LIBRARY).   python scripts/solve_ab.py [config] [reps]"""
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, '.')
import sanm_amd  # noqa: E402
from sanm_amd import api as A  # noqa: E402
from sanm_amd import fea as dfea  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "armadillo_small"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
api = sanm_amd.get_api()
cfg, mesh = dfea.load_named_config(name)
run = dfea.GravityRun(api, mesh, dict(cfg)).construct()
J = run.solver.jacobian_csr()
coords = None
s = A.DirectSolver(api, J, coords)
s.factor(J)
b = np.random.default_rng(0).standard_normal(J.shape[0])
for _ in range(reps):
    x = s.solve(b)
print("n", J.shape[0], "nnz", J.nnz, "|x|", float(np.abs(x).max()))
