"""debug: merged top block vs level-by-level solve on the Jacobian of tests/test_direct_solver.py::test_fem_jacobian"""
import os
import sys

import numpy as np
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from oracle import fea as ofea  # noqa: E402
from oracle import symbolic as S  # noqa: E402
from oracle.anm import build_jacobian_csr  # noqa: E402
import sanm_amd  # noqa: E402
from sanm_amd.api import DirectSolver  # noqa: E402

api = sanm_amd.get_api()
mesh = ofea.make_cuboid(9, 5, 4, 0.02)
fixed = np.zeros((mesh.nr_vertices, 3), bool)
fixed[mesh.V[:, 0] < 0.01] = True
om = ofea.make_forward(mesh, ofea.Material(1e4, 0.45), fixed, "neohookean_c")
prop = S.TaylorCoeffProp(om.y)
prop.push_xi([(om.lt_inp.mat @ om.lt_inp.x0).reshape(-1, 3, 3)])
A, _ = build_jacobian_csr(om.lt_out, prop.get_jacobian(), om.lt_inp.mat, om.lt_inp.n)
A = A.tocsr()
A.sort_indices()
coords = mesh.V[om.lt_inp.vertex_loc[:, 0]]
lu = spla.splu(A.tocsc())
rng = np.random.default_rng(0)
b = rng.standard_normal(A.shape[0])
xr = lu.solve(b)
for top in sys.argv[1:] or ["2048", "0"]:
    os.environ["SANM_MF_TOP"] = top
    os.environ["SANM_MF_DEBUG"] = "1"
    ds = DirectSolver(api, A, coords)
    print("factor", ds.factor(A))
    x = ds.solve(b)
    err = np.abs(x - xr)
    print("SANM_MF_TOP", top, "err", err.max() / np.abs(xr).max(), "worst entries", np.argsort(-err)[:8], ds.stats())
