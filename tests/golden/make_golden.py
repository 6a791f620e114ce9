"""Generate the golden fixtures under tests/golden/ (authoring container only).

Two kinds of vectors:

1. ``ref_utils.npz`` -- inputs and outputs of the REFERENCE's own Python
   utilities, imported from /root/reference/utils (they cannot travel to the
   GPU box, their outputs can):
     * utils/test_cofactor.py:8-13   compute_cofactor (SVD route) on 3x3 inputs,
       including a rank-deficient one
     * utils/test_svdw_grad.py:9-46  svdw + svdw_jacobian (dU/dM, dS/dM, dW/dM) at n=3
     * utils/check_single_tet.py     deformation gradient, Cauchy stress and vertex
       normals of the single-tet KAT (rest apex z = 0.022755286528750494)
2. ``anm_*.json`` -- results of the oracle (pinned by 1. and by the invariant
   tests) on small end-to-end cases: per-step residual RMS, step counts,
   a_bound / t_max per step and final vertices.  The GPU tests compare the HIP
   path against these and against the live oracle.  The oracle's linear solves
   were MKL PARDISO (oracle/pardiso.py: the reference's solver and settings) when
   these were generated.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, "/root/reference/utils")

import check_single_tet as cst  # noqa: E402  (reference util)
import test_cofactor as tcof  # noqa: E402  (reference util)
import test_svdw_grad as tsg  # noqa: E402  (reference util)

from oracle import fea  # noqa: E402


def ref_utils():
    rng = np.random.RandomState(20211)
    mats = rng.uniform(-2, 2, (16, 3, 3))
    mats[3, 2] = mats[3, 0] * 0.5 - mats[3, 1]  # rank 2
    mats[5] = np.outer(mats[5, 0], mats[5, 1])  # rank 1
    cof = np.stack([tcof.compute_cofactor(m) for m in mats])
    sv_in = rng.normal(size=(8, 3, 3))
    U, S, W, dU, dS, dW = [], [], [], [], [], []
    for m in sv_in:
        u, s, w = tsg.svdw(m.copy())
        du, ds, dw = tsg.svdw_jacobian(m, u, s, w)
        U.append(u); S.append(s); W.append(w); dU.append(du); dS.append(ds); dW.append(dw)
    # single tet KAT (check_single_tet.main)
    sd = np.zeros((4, 3))
    ang, sp = np.pi * 2 / 3, 0.025
    sd[:3, 0] = sp * np.cos(np.arange(3) * ang)
    sd[:3, 1] = sp * np.sin(np.arange(3) * ang)
    sd[3, 2] = sp
    sr = sd.copy()
    sr[3] = [0, 0, 0.022755286528750494]
    F = cst.comptute_deformation_gradient(sr, sd)
    sig = cst.compute_cauchy_stress(F)
    nrm_rest = cst.compute_vtx_norm(cst.make_shape_matrix(sr))
    nrm_def = cst.compute_vtx_norm(cst.make_shape_matrix(sd))
    np.savez(os.path.join(HERE, "ref_utils.npz"), cof_in=mats, cof_out=cof, svdw_in=sv_in,
             svdw_U=np.stack(U), svdw_S=np.stack(S), svdw_W=np.stack(W), svdw_dU=np.stack(dU),
             svdw_dS=np.stack(dS), svdw_dW=np.stack(dW), tet_rest=sr, tet_deform=sd, tet_F=F,
             tet_cauchy=sig, tet_norm_rest=nrm_rest, tet_norm_deform=nrm_def, tet_mu=cst.mu, tet_k=cst.k)


CASES = {
    # name: (cuboid dims, spacing, config)
    "cuboid_nc": ((8, 3, 3), 0.025, {"material": {"young": 2e3, "poisson": 0.45, "density": 1000.0},
                                     "g": [0, -9.81, 0], "boundary_thresh": 0.05,
                                     "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_c", "order": 20}),
    "cuboid_ni": ((8, 3, 3), 0.025, {"material": {"young": 2e3, "poisson": 0.45, "density": 1000.0},
                                     "g": [0, -9.81, 0], "boundary_thresh": 0.05,
                                     "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_i", "order": 20}),
    "cuboid_arap": ((8, 3, 3), 0.025, {"material": {"young": 2e4, "poisson": 0.45, "density": 1000.0},
                                       "g": [0, -9.81, 0], "boundary_thresh": 0.05,
                                       "boundary_proj_dir": [-1, 0, 0], "energy_model": "arap", "order": 16}),
    "cuboid_nc_nopade_o8": ((6, 4, 3), 0.02, {"material": {"young": 5e3, "poisson": 0.4, "density": 1200.0},
                                              "g": [0, 0, -9.81], "boundary_thresh": 0.1,
                                              "boundary_proj_dir": [1, 0, 0], "energy_model": "neohookean_c",
                                              "order": 8, "disable_pade": True}),
    # Tikhonov path (config/override_l2_penalty.json: xcoeff_l2_penalty), sparse_solver.cpp:366-395
    "cuboid_nc_l2": ((6, 3, 3), 0.025, {"material": {"young": 3e3, "poisson": 0.45, "density": 1000.0},
                                        "g": [0, -9.81, 0], "boundary_thresh": 0.05,
                                        "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_c", "order": 10,
                                        "xcoeff_l2_penalty": 1e-6}),
}


def anm_cases():
    for name, (dims, spacing, cfg) in CASES.items():
        mesh = fea.make_cuboid(*dims, spacing)
        model, solver, f = fea.make_gravity_solver(mesh, cfg)
        x, rms = fea.run_anm(solver)
        out = {"dims": dims, "spacing": spacing, "config": cfg, "iter": solver.get_nr_iter(),
               "residual_rms": rms, "a_bound": [r["a_bound"] for r in solver.trace],
               "t_max": [r["t_max"] for r in solver.trace], "pade": [r["pade"] for r in solver.trace],
               "first_step_t": solver.trace[0]["t"], "first_step_x_norm": solver.trace[0]["x_norm"],
               "nr_unknown": int(model.lt_inp.n),
               "vertices": model.lt_inp.full_vertices(x).tolist()}
        json.dump(out, open(os.path.join(HERE, f"anm_{name}.json"), "w"))
        print(name, "iter", out["iter"], "rms", ["%.2g" % r for r in rms])
    # BASELINE config 1 (test_simple_cuboid_twist.json, ARAP, implicit solver + refine)
    cfg = json.load(open("/root/reference/config/test_simple_cuboid_twist.json"))
    V, stats = fea.test_cuboid_twist(cfg)
    json.dump({"config": cfg, "stats": stats, "vertices": V.tolist()},
              open(os.path.join(HERE, "anm_cuboid_twist.json"), "w"))
    print("cuboid_twist", stats)
    # single tet inverse (config/test_single_tet_inverse.json)
    cfg = json.load(open("/root/reference/config/test_single_tet_inverse.json"))
    V, solver = fea.test_single_tet_inverse(cfg)
    json.dump({"config": cfg, "iter": solver.get_nr_iter(), "vertices": V.tolist()},
              open(os.path.join(HERE, "anm_single_tet_inverse.json"), "w"))
    print("single_tet_inverse apex z", V[3, 2])


if __name__ == "__main__":
    ref_utils()
    anm_cases()
