"""Command-line front end with the reference's calling convention (fea/main.cpp:1064-1102):

    python -m sanm_amd.cli <system config json> <task config json> [<task override json> ...]

The task config and its overrides are merged key by key (nlohmann `json::update`), the TetGen mesh named by
`"mesh"` is read relative to the task config's directory (fea/main.cpp:921-934), and a `"func": "gravity"`
task (fea/main.cpp:984-1046; forward or, with `"inverse": true`, inverse mode) is solved on the device.  The
outputs are the reference's (run_and_save, fea/main.cpp:247-433): `<out_filename>-orig.obj`,
`<out_filename>-i<inverse>-<energy_model>.obj` and the statistics json beside it with the same keys
(`time_prep`, `time_solve`, `order`, `name`, `threads`, `pade`, `iter`, `force_rms_recomp`, `mesh_V`, `mesh_F`,
`displacement`, `nr_inverted`).  The system config's thread count has no meaning on the device and is only
recorded.  Other `func` values of the reference (baselines, rendering helpers, mesh_twist) are outside the hot
path and rejected.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

from . import fea
from .api import TaylorCoeffProp


def read_tetgen(filebase):
    """TetrahedralMesh::from_tetgen_files, fea/tetrahedral_mesh.cpp:206-260: .node / .ele / .face, zero-based,
    no attributes.  Returns (Mesh, surface triangles (nf, 3))."""
    def toks(path):
        with open(path) as f:
            return [t for line in f for t in line.split("#")[0].split()]
    tn = toks(filebase + ".node")
    nv, dim, nattr, bm = (int(t) for t in tn[:4])
    if dim != 3 or nattr != 0:
        raise ValueError(f"{filebase}.node: need 3-D vertices without attributes")
    a = np.array(tn[4:4 + nv * (4 + bm)], dtype=np.float64).reshape(nv, 4 + bm)
    if not np.array_equal(a[:, 0], np.arange(nv)):
        raise ValueError(f"{filebase}.node: vertices must be numbered from zero")
    te = toks(filebase + ".ele")
    nt, npt, nattr = (int(t) for t in te[:3])
    if npt != 4:
        raise ValueError(f"{filebase}.ele: need 4-node tetrahedra")
    e = np.array(te[3:3 + nt * (5 + nattr)], dtype=np.int64).reshape(nt, 5 + nattr)
    tf = toks(filebase + ".face")
    nf, bmark = int(tf[0]), int(tf[1])
    fa = np.array(tf[2:2 + nf * (4 + bmark)], dtype=np.int64).reshape(nf, 4 + bmark)
    tri = fa[:, 1:4]
    return fea.Mesh(a[:, 1:4], e[:, 1:5], np.unique(tri)), tri


def save_obj(path, vertices, triangles):
    """TetrahedralMesh::write_to_file with explicit surfaces, fea/tetrahedral_mesh.cpp:262-267, :295-330"""
    with open(path, "w") as f:
        for v in vertices:
            f.write("v %g %g %g\n" % tuple(v))
        for t in triangles:
            f.write("f %d %d %d\n" % tuple(int(i) + 1 for i in t))


def relative_displacement(v0, v1):
    """fea/main.cpp:219-223"""
    return float(np.sqrt(((v1 - v0) ** 2).sum() / v0.size) / np.linalg.norm(v0.max(0) - v0.min(0)))


def nr_inverted(tets, v0, v1):
    """fea/main.cpp:226-244"""
    def sign(v):
        a, b, c, d = (v[tets[:, i]] for i in range(4))
        return np.einsum("ij,ij->i", np.cross(b - a, c - a), d - a) >= 0
    return int(np.count_nonzero(sign(v0) != sign(v1)))


def run_gravity(api, task_dir, config, sys_config, out=sys.stdout):
    mesh, tri = read_tetgen(os.path.join(task_dir, config["mesh"]))
    inverse = bool(config.get("inverse", False))
    run = fea.GravityRun(api, mesh, config, inverse=inverse)  # scales the mesh, builds the model
    V0 = mesh.V.copy()
    out.write("solving mesh %s%s order=%d:" % (os.path.basename(config["mesh"]), " (inv)" if inverse else "",
                                                int(run.hyper.order)))
    out.flush()
    run.construct()
    while not run.solver.converged():
        run.step()
        out.write(" %.3g" % run.rms[-1])
        out.flush()
    V1 = run.vertices()
    out.write("\ntiming(sec): prep=%.3f solve=%.3f\n" % (run.time_prep, run.time_solve))
    st = run.stats()
    # force balance recomputed from scratch at the solution (compute_force_rms, fea/mesh_template.h:221-262)
    prop = TaylorCoeffProp(api, run.model.y, run.model.lt_inp, 1, mesh.nr_tet)
    y = prop.push_xi(run.solver.get_x())
    resid = run.model.lt_out.to_scipy() @ y.ravel() + run.f_sub
    jstat = {"time_prep": st["time_prep"], "time_solve": st["time_solve"], "order": st["order"],
             "name": "mesh %s" % os.path.basename(config["mesh"]), "threads": int(sys_config.get("threads", 1)),
             "solver_threads": int(sys_config.get("sparse_solver_threads", sys_config.get("threads", 1))),
             "pade": st["pade"], "iter": st["iter"],
             "force_rms_recomp": float(np.sqrt(np.mean(resid ** 2))), "mesh_V": mesh.nr_vertices,
             "mesh_F": mesh.nr_tet, "displacement": relative_displacement(V0, V1),
             "nr_inverted": nr_inverted(mesh.tets, V0, V1), "residual_rms": st["residual_rms"],
             "device": api.backend_name()}
    base = config["out_filename"]
    os.makedirs(os.path.dirname(os.path.abspath(base)), exist_ok=True)
    save_obj(base + "-orig.obj", V0, tri)
    base += "-i%d-%s" % (int(inverse), config["energy_model"])
    save_obj(base + ".obj", V1, tri)
    with open(base + ".json", "w") as f:
        json.dump(jstat, f, indent=1)
    return jstat


def main(argv=None, api=None, out=sys.stdout):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) < 2:
        sys.stderr.write("usage: python -m sanm_amd.cli <system config file> <task config file> "
                         "[<task override json files ...>]\n")
        return -1
    sys_config = json.load(open(argv[0]))
    config = json.load(open(argv[1]))
    for extra in argv[2:]:
        config.update(json.load(open(extra)))
    if api is None:
        import torch  # noqa: F401  (must be loaded before the HIP library)
        import sanm_amd
        api = sanm_amd.get_api(int(os.environ.get("LOCAL_RANK", 0)))
    func = config["func"]
    if func == "gravity":
        t0 = time.perf_counter()
        st = run_gravity(api, os.path.dirname(os.path.abspath(argv[1])), config, sys_config, out)
        out.write("iter=%d force_rms=%.3g displacement=%.4g total=%.2fs\n" %
                  (st["iter"], st["force_rms_recomp"], st["displacement"], time.perf_counter() - t0))
        return 0
    raise ValueError("func %r is outside the device hot path (supported: gravity)" % func)


if __name__ == "__main__":
    try:
        sys.exit(main())
    except Exception as e:  # fea/main.cpp:1104-1112
        sys.stderr.write("caught exception: %s\n" % e)
        sys.exit(2)
