"""Command-line front end with the reference's calling convention (fea/main.cpp:1064-1102):

    python -m sanm_amd.cli <system config json> <task config json> [<task override json> ...]

The task config and its overrides are merged key by key (nlohmann `json::update`), the TetGen mesh named by
`"mesh"` is read relative to the task config's directory (fea/main.cpp:921-934), and a `"func": "gravity"`
task (fea/main.cpp:984-1046; forward or, with `"inverse": true`, inverse mode) is solved on the device.  The
outputs are the reference's (run_and_save, fea/main.cpp:247-433): `<out_filename>-orig.obj`,
`<out_filename>-i<inverse>-<energy_model>.obj` and the statistics json beside it with the same keys
(`time_prep`, `time_solve`, `order`, `name`, `threads`, `pade`, `iter`, `force_rms_recomp`, `mesh_V`, `mesh_F`,
`displacement`, `nr_inverted`).  The system config's thread count has no meaning on the device and is only
recorded.  All five `func` values of the reference's dispatcher (fea/main.cpp:1081-1101) are served: `gravity`
(`<mesh>.bou` fixed-vertex files included, fea/main.cpp:1000-1013), `mesh_twist` (fea/main.cpp:774-919),
`test_single_tet_inverse`, `test_cuboid`, `test_cuboid_twist` (fea/main.cpp:583-772).  Config keys that select
paths outside the device hot path (`baseline`, `save_interm`) raise instead of being ignored.  The reference
prints vertices with `%g` everywhere (6 digits: too coarse for a 1e-6 comparison); next to every `.obj` this
front end also writes `<same name>.vertices.txt` with 17 significant digits.
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


def save_vertices_precise(path, vertices):
    """not in the reference: one vertex per line with 17 significant digits (the .obj files keep %g)"""
    with open(path, "w") as f:
        for v in vertices:
            f.write("%.17g %.17g %.17g\n" % tuple(v))


def save_mesh(base, vertices, triangles, only_vtx=None):
    """save_mesh (fea/main.cpp:61-78): `<base>` ends in .obj; with `only_vtx` just the faces whose vertices all
    lie in the set (TetrahedralMesh::write_to_file with a filter set, fea/tetrahedral_mesh.cpp:295-330)"""
    tri = np.asarray(triangles)
    if only_vtx is not None:
        keep = np.isin(tri, np.fromiter(only_vtx, dtype=np.int64)).all(axis=1) if len(tri) else np.zeros(0, bool)
        tri = tri[keep]
    save_obj(base, vertices, tri)
    save_vertices_precise(base[:-4] + ".vertices.txt", vertices)


def save_out_surface_vtx(config, vertices, surface_vtx):
    """save_out_surface_vtx (fea/main.cpp:80-88; tetrahedral_mesh.cpp:277-293)"""
    if "out_surface_vtx" not in config:
        return
    sv = np.sort(np.asarray(surface_vtx))
    if sv[0] != 0 or sv[-1] != len(sv) - 1:
        raise ValueError("surface vertices are not numbered 0..n-1")
    with open(config["out_surface_vtx"], "w") as f:
        for v in vertices[: len(sv)]:
            f.write("%g %g %g\n" % tuple(v))


def reject_unsupported(config):
    """keys that select code outside the device hot path must not be silently ignored"""
    if config.get("baseline"):
        raise ValueError("config key 'baseline' selects the Newton / Levenberg-Marquardt baselines of the reference "
                         "(fea/baseline.cpp): not part of the device path")
    if config.get("save_interm"):
        raise ValueError("config key 'save_interm' (intermediate meshes through ANMSolverVecScale) is not served "
                         "by this front end")


def read_bou(filebase, nv):
    """<mesh>.bou: 1-based ids of the vertices with all three coordinates fixed (fea/main.cpp:1000-1013);
    None if the file does not exist"""
    path = filebase + ".bou"
    if not os.path.exists(path):
        return None
    ids = np.array([int(t) for t in open(path).read().split()], dtype=np.int64)
    if ids.size and (ids.min() < 1 or ids.max() > nv):
        raise ValueError(f"{path}: vertex ids must be in 1..{nv}")
    fixed = np.zeros((nv, 3), dtype=bool)
    fixed[ids - 1] = True
    return fixed


def _solve_static(api, mesh, tri, fixed, f_load, config, sys_config, name, inverse, out, save=True):
    """run_and_save (fea/main.cpp:247-433), ANM branch"""
    reject_unsupported(config)
    run = fea.GravityRun.from_parts(api, mesh, config, fixed, f_load, inverse=inverse)
    V0 = mesh.V.copy()
    out.write("solving %s%s order=%d:" % (name, " (inv)" if inverse else "", int(run.hyper.order)))
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
    jstat = {"time_prep": st["time_prep"], "time_solve": st["time_solve"], "order": st["order"], "name": name,
             "threads": int(sys_config.get("threads", 1)),
             "solver_threads": int(sys_config.get("sparse_solver_threads", sys_config.get("threads", 1))),
             "pade": st["pade"], "iter": st["iter"],
             "force_rms_recomp": float(np.sqrt(np.mean(resid ** 2))), "mesh_V": mesh.nr_vertices,
             "mesh_F": mesh.nr_tet, "displacement": relative_displacement(V0, V1),
             "nr_inverted": nr_inverted(mesh.tets, V0, V1), "residual_rms": st["residual_rms"],
             "device": api.backend_name()}
    if save:
        base = config["out_filename"]
        os.makedirs(os.path.dirname(os.path.abspath(base)), exist_ok=True)
        save_mesh(base + "-orig.obj", V0, tri)
        base += "-i%d-%s" % (int(inverse), config["energy_model"])
        save_mesh(base + ".obj", V1, tri)
        with open(base + ".json", "w") as f:
            json.dump(jstat, f, indent=1)
        save_out_surface_vtx(config, V1, mesh.surface_vtx)
    return jstat, V1


def run_gravity(api, task_dir, config, sys_config, out=sys.stdout):
    """gravity (fea/main.cpp:984-1046)"""
    filebase = os.path.join(task_dir, config["mesh"])
    mesh, tri = read_tetgen(filebase)
    inverse = bool(config.get("inverse", False))
    fixed_cfg, f_load = fea.setup_gravity(api, mesh, config, boundary=False)  # scales the mesh
    fixed = read_bou(filebase, mesh.nr_vertices)
    if fixed is None:
        out.write("bou file does not exist; fix lowest points ...\n")
        fixed = fea.boundary_by_config(api, mesh, config)
    base = config["out_filename"]
    os.makedirs(os.path.dirname(os.path.abspath(base)), exist_ok=True)
    save_mesh(base + "-boundary.obj", mesh.V, tri, only_vtx=np.nonzero(fixed[:, 0])[0])
    out.write("mesh loading finished %s:\n nr_vtx=%d nr_tet=%d boundary_vtx=%d\n" %
              (filebase, mesh.nr_vertices, mesh.nr_tet, int(fixed[:, 0].sum())))
    st, _ = _solve_static(api, mesh, tri, fixed, f_load, config, sys_config,
                          "mesh %s" % os.path.basename(config["mesh"]), inverse, out)
    return st


def _surface_tris_of_tets(mesh):
    """boundary faces of a tetrahedral mesh (those that belong to exactly one tet), for the .obj of the built-in
    test meshes, which the reference writes through TetrahedralMesh::write_to_file without explicit surfaces"""
    faces = {}
    for t in mesh.tets:
        for a, b, c in ((0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3)):
            key = tuple(sorted((int(t[a]), int(t[b]), int(t[c]))))
            faces[key] = faces.get(key, 0) + 1
    return np.array([k for k, v in faces.items() if v == 1], dtype=np.int64).reshape(-1, 3)


def run_test_single_tet_inverse(api, config, sys_config, out=sys.stdout):
    """test_single_tet_inverse (fea/main.cpp:583-621)"""
    spacing = float(config["spacing"])
    ang = 2 * np.pi / 3
    V = np.zeros((4, 3))
    for i in range(3):
        V[i, 0], V[i, 1] = np.cos(ang * i) * spacing, np.sin(ang * i) * spacing
    V[3, 2] = spacing
    mesh = fea.Mesh(V, np.array([[0, 1, 2, 3]]), np.arange(4))
    fixed = np.zeros((4, 3), dtype=bool)
    fixed[:3] = True
    f_load = np.zeros((4, 3))
    f_load[3, 2] = -1000
    st, V1 = _solve_static(api, mesh, _surface_tris_of_tets(mesh), fixed, f_load, config, sys_config, "single tet inv",
                           True, out)
    for i in range(4):
        out.write("vertex %d: (%.3f, %.3f, %.3f) -> (%.3f, %.3f, %.3f)\n" % (i, *V[i], *V1[i]))
    return st


def run_test_cuboid(api, config, sys_config, out=sys.stdout):
    """test_cuboid (fea/main.cpp:623-663)"""
    nx, ny, nz, sp = int(config["x"]), int(config["y"]), int(config["z"]), float(config["spacing"])
    mesh = fea.make_cuboid(nx, ny, nz, sp)
    fixed = np.zeros((mesh.nr_vertices, 3), dtype=bool)
    fixed[mesh.V[:, 0] <= sp / 2.0] = True
    f_load = np.zeros((mesh.nr_vertices, 3))
    sel = (mesh.V[:, 0] > (nx // 2 - 1) * sp - sp / 2.0) & (mesh.V[:, 2] < sp / 2.0)
    f_load[sel, 2] = -50.0
    inverse = bool(config.get("inverse", False))
    st, _ = _solve_static(api, mesh, _surface_tris_of_tets(mesh), fixed, f_load, config, sys_config,
                          "cuboid inverse" if inverse else "cuboid", inverse, out)
    return st


def run_test_cuboid_twist(api, config, sys_config, out=sys.stdout):
    """test_cuboid_twist (fea/main.cpp:665-772)"""
    reject_unsupported(config)
    V, stats = fea.test_cuboid_twist(api, config)
    nx, ny, nz, sp = int(config["x"]), int(config["y"]), int(config["z"]), float(config["spacing"])
    mesh = fea.make_cuboid(nx, ny, nz, sp)
    if "out_filename" in config:
        base = config["out_filename"]
        os.makedirs(os.path.dirname(os.path.abspath(base)), exist_ok=True)
        save_mesh(base + ".obj", V, _surface_tris_of_tets(mesh))
        with open(base + ".json", "w") as f:
            json.dump(stats, f, indent=1)
    out.write("cuboid twist: %d deformation stages, iterations %s\n" % (len(stats), [s["iter_tot"] for s in stats]))
    return stats


def run_mesh_twist(api, task_dir, config, sys_config, out=sys.stdout):
    """mesh_twist (fea/main.cpp:774-919): the vertices beyond `ratio_hi` along `axis` are rotated / shifted as a
    rigid block, those below `ratio_lo` stay, and the body follows by ANMImplicitSolver continuation."""
    reject_unsupported(config)
    filebase = os.path.join(task_dir, config["mesh"])
    mesh, tri = read_tetgen(filebase)
    if float(config.get("scale", 0)) > 0:
        mesh.V = mesh.V * float(config["scale"])
    out.write("mesh twist: V=%d F=%d\n" % (mesh.nr_vertices, mesh.nr_tet))
    axis = np.asarray(config["axis"], dtype=np.float64)
    base = config["out_filename"]
    os.makedirs(os.path.dirname(os.path.abspath(base)), exist_ok=True)
    proj = mesh.V @ axis
    pmin, pmax = proj.min(), proj.max()
    proj_dist = pmax - pmin
    th0 = pmin + proj_dist * float(config["ratio_lo"])
    th1 = pmin + proj_dist * (1 - float(config["ratio_hi"]))
    on_surface = np.zeros(mesh.nr_vertices, dtype=bool)
    on_surface[mesh.surface_vtx] = True
    cand = on_surface | bool(config.get("include_int_points", False))
    sel = ((proj <= th0) | (proj >= th1)) & cand
    fixed = np.zeros((mesh.nr_vertices, 3), dtype=bool)
    fixed[sel] = True
    bnd_idx = np.nonzero(sel & (proj >= th1))[0]
    out.write("proj range: %g %g thr=%g,%g\n" % (pmin, pmax, th0, th1))
    save_mesh(base + "-orig.obj", mesh.V, tri)
    save_mesh(base + "-boundary.obj", mesh.V, tri, only_vtx=np.nonzero(sel)[0])

    f_load = None
    vtx_cur = mesh.V.copy()
    if config.get("add_gravity", False):
        mc = config["material"]
        f_load = api.gravity_load(mesh.V, mesh.tets, float(mc["density"]), np.asarray(config["g"], dtype=np.float64))
        _, vtx_cur = _solve_static(api, mesh, tri, fixed, f_load, config, sys_config, "gravity_init", False, out,
                                   save=False)
        save_mesh(base + "-gravity.obj", vtx_cur, tri)

    bnd_next = vtx_cur[bnd_idx].copy()

    def apply_trans(c):
        nonlocal bnd_next
        ang = float(c["angle"]) * np.pi / 180
        shift = np.asarray(c["shift"], dtype=np.float64)
        ra = int(c.get("rot_axis", 2))
        small = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        rmat = np.eye(3)
        for i in range(3):
            for j in range(3):
                if i != ra and j != ra:
                    rmat[i, j] = small[i - (i > ra), j - (j > ra)]
        bnd_next = bnd_next @ rmat.T + shift * proj_dist

    for c in config.get("transforms", [config]):
        apply_trans(c)
    delta = np.zeros_like(vtx_cur)
    delta[bnd_idx] = bnd_next - vtx_cur[bnd_idx]
    save_mesh(base + "-boundary-dst.obj", vtx_cur + delta, tri, only_vtx=np.nonzero(sel)[0])
    vtx_new, stat = fea.run_with_vtx_delta(api, mesh, fixed, config, delta, vtx_cur, False, refine_f_load=f_load)
    save_mesh(base + ".obj", vtx_new, tri)
    with open(base + ".json", "w") as f:
        json.dump(stat, f, indent=1)
    save_out_surface_vtx(config, vtx_new, mesh.surface_vtx)
    out.write("mesh twist: iter_deform=%d iter_refine=%d force_rms=%.3g\n" %
              (stat["iter_deform"], stat["iter_refine"], stat["force_rms_recomp"]))
    return stat


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
    task_dir = os.path.dirname(os.path.abspath(argv[1]))
    t0 = time.perf_counter()
    if func == "gravity":
        st = run_gravity(api, task_dir, config, sys_config, out)
        out.write("iter=%d force_rms=%.3g displacement=%.4g total=%.2fs\n" %
                  (st["iter"], st["force_rms_recomp"], st["displacement"], time.perf_counter() - t0))
    elif func == "mesh_twist":
        run_mesh_twist(api, task_dir, config, sys_config, out)
    elif func == "test_single_tet_inverse":
        run_test_single_tet_inverse(api, config, sys_config, out)
    elif func == "test_cuboid":
        run_test_cuboid(api, config, sys_config, out)
    elif func == "test_cuboid_twist":
        run_test_cuboid_twist(api, config, sys_config, out)
    else:
        raise ValueError("unknown func: %s" % func)  # fea/main.cpp:1101
    return 0


if __name__ == "__main__":
    try:
        sys.exit(main())
    except Exception as e:  # fea/main.cpp:1104-1112
        sys.stderr.write("caught exception: %s\n" % e)
        sys.exit(2)
