"""The command-line front end (fea `gravity` task with the reference's argument convention, TetGen input,
obj + stats-json output) against the oracle on the same files."""
import io
import json
import os

import numpy as np

from tests.lockstep import LockStep, lockstep_vtx_delta_stage

from oracle import fea as ofea
from sanm_amd import cli


def _write_tetgen(base, V, tets, surf_tris):
    with open(base + ".node", "w") as f:
        f.write("%d 3 0 0\n" % len(V))
        for i, v in enumerate(V):
            f.write("%d %.17g %.17g %.17g\n" % (i, *v))
    with open(base + ".ele", "w") as f:
        f.write("%d 4 0\n" % len(tets))
        for i, t in enumerate(tets):
            f.write("%d %d %d %d %d\n" % (i, *t))
    with open(base + ".face", "w") as f:
        f.write("%d 1\n" % len(surf_tris))
        for i, t in enumerate(surf_tris):
            f.write("%d %d %d %d 1\n" % (i, *t))


def test_gravity_task_from_files(api, tmp_path):
    om = ofea.make_cuboid(5, 3, 3, 0.03)
    # surface triangles: faces of tets with all three vertices on the surface
    surf = set(om.surface_vtx.tolist())
    tris = []
    for t in om.tets:
        for a, b, c in ((0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3)):
            if t[a] in surf and t[b] in surf and t[c] in surf:
                tris.append((t[a], t[b], t[c]))
    os.makedirs(tmp_path / "model")
    _write_tetgen(str(tmp_path / "model" / "block.1"), om.V, om.tets, tris)
    task = {"func": "gravity", "mesh": "model/block.1", "energy_model": "neohookean_i", "g": [0, -9.81, 0],
            "material": {"type": "young_poisson", "young": 3e3, "poisson": 0.45, "density": 900.0},
            "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0], "out_filename": str(tmp_path / "out" / "block")}
    json.dump(task, open(tmp_path / "task.json", "w"))
    json.dump({"order": 12}, open(tmp_path / "override_order12.json", "w"))
    json.dump({"verbosity": 0, "threads": 4}, open(tmp_path / "sys.json", "w"))
    log = io.StringIO()
    rc = cli.main([str(tmp_path / "sys.json"), str(tmp_path / "task.json"), str(tmp_path / "override_order12.json")],
                  api=api, out=log)
    assert rc == 0
    base = str(tmp_path / "out" / "block")
    st = json.load(open(base + "-i0-neohookean_i.json"))
    for key in ("time_prep", "time_solve", "order", "name", "threads", "pade", "iter", "force_rms_recomp", "mesh_V",
                "mesh_F", "displacement", "nr_inverted"):
        assert key in st
    assert st["order"] == 12 and st["threads"] == 4 and st["mesh_V"] == om.nr_vertices and st["mesh_F"] == om.nr_tet
    assert st["force_rms_recomp"] < 1e-8 and st["nr_inverted"] == 0 and st["displacement"] > 1e-3
    # the same task through the oracle
    cfg = dict(task, order=12)
    omodel, osolver, _ = ofea.make_gravity_solver(ofea.read_tetgen(str(tmp_path / "model" / "block.1")), cfg)
    xo, _ = ofea.run_anm(osolver)
    Vo = omodel.lt_inp.full_vertices(xo)
    assert st["pade"]
    # step count: the same task again beside a second oracle in lock step (tests/lockstep.py); the CLI's count must
    # be the lock-step device's, and EQUAL to the free-running oracle's unless a certified ill-conditioned Pade
    # decision was logged on the way
    from sanm_amd import fea as dfea
    dmesh, _ = cli.read_tetgen(str(tmp_path / "model" / "block.1"))
    run = dfea.GravityRun(api, dmesh, dict(cfg)).construct()
    _, osolver2, _ = ofea.make_gravity_solver(ofea.read_tetgen(str(tmp_path / "model" / "block.1")), cfg)
    ls = LockStep(run, osolver2).run_to_convergence()
    assert ls.nr_steps == st["iter"]
    print("gravity from files: steps", st["iter"], "oracle", osolver.get_nr_iter(), "events",
          [(e["step"], e["device"], e["oracle"]) for e in ls.events])
    if not ls.events:
        assert st["iter"] == osolver.get_nr_iter()
    Vd = np.array([[float(x) for x in line.split()[1:]] for line in open(base + "-i0-neohookean_i.obj")
                   if line.startswith("v ")])
    assert Vd.shape == Vo.shape and np.abs(Vd - Vo).max() <= 1e-5 * np.abs(Vo).max()  # %g keeps 6 digits
    assert sum(1 for line in open(base + "-orig.obj") if line.startswith("f ")) == len(tris)


def _block_files(tmp_path, dims=(5, 3, 3), sp=0.03):
    om = ofea.make_cuboid(*dims, sp)
    surf = set(om.surface_vtx.tolist())
    tris = [(t[a], t[b], t[c]) for t in om.tets for a, b, c in ((0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3))
            if t[a] in surf and t[b] in surf and t[c] in surf]
    os.makedirs(tmp_path / "model", exist_ok=True)
    _write_tetgen(str(tmp_path / "model" / "block.1"), om.V, om.tets, tris)
    json.dump({"verbosity": 0, "threads": 1}, open(tmp_path / "sys.json", "w"))
    return om, tris


def test_gravity_task_takes_the_fixed_set_from_the_bou_file(api, tmp_path):
    """fea/main.cpp:1000-1013: `<mesh>.bou` (1-based vertex ids, all three coordinates fixed) wins over the
    threshold rule of the config.  The same task with and without the file solves two different problems."""
    om, tris = _block_files(tmp_path)
    # fix the face x = max instead of what boundary_thresh / boundary_proj_dir would pick (x = min)
    ids = np.nonzero(om.V[:, 0] >= om.V[:, 0].max() - 1e-12)[0]
    open(tmp_path / "model" / "block.1.bou", "w").write("\n".join(str(i + 1) for i in ids) + "\n")
    task = {"func": "gravity", "mesh": "model/block.1", "energy_model": "neohookean_c", "g": [0, -9.81, 0], "order": 10,
            "material": {"type": "young_poisson", "young": 3e3, "poisson": 0.45, "density": 900.0},
            "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0], "out_filename": str(tmp_path / "out" / "b")}
    json.dump(task, open(tmp_path / "task.json", "w"))
    assert cli.main([str(tmp_path / "sys.json"), str(tmp_path / "task.json")], api=api, out=io.StringIO()) == 0
    Vd = np.loadtxt(str(tmp_path / "out" / "b-i0-neohookean_c.vertices.txt"))
    assert np.array_equal(Vd[ids], om.V[ids]), "the vertices named by the .bou file did not stay fixed"
    assert np.abs(Vd - om.V)[om.V[:, 0] <= 1e-12].max() > 1e-4, "the x = min face must be free to move now"
    # the oracle with the same fixed set
    mat = ofea.Material(3e3, 0.45, 900.0)
    fixed = np.zeros((om.nr_vertices, 3), dtype=bool)
    fixed[ids] = True
    f = ofea.gravity_load(om, mat, np.array([0, -9.81, 0]))
    model, solver, x = ofea.solve_static(om, mat, fixed, "neohookean_c", f, dict(task))
    Vo = model.lt_inp.full_vertices(x)
    assert np.abs(Vd - Vo).max() <= 1e-8 * np.abs(Vo).max()  # the 17-digit vertex file, not the %g .obj
    nb = sum(1 for line in open(str(tmp_path / "out" / "b-boundary.obj")) if line.startswith("f "))
    assert 0 < nb < len(tris)


def test_unsupported_config_keys_raise(api, tmp_path):
    _block_files(tmp_path)
    task = {"func": "gravity", "mesh": "model/block.1", "energy_model": "neohookean_c", "g": [0, -9.81, 0],
            "material": {"young": 3e3, "poisson": 0.45, "density": 900.0}, "boundary_thresh": 0.05,
            "out_filename": str(tmp_path / "out" / "b"), "baseline": {"use_levmar": True}}
    json.dump(task, open(tmp_path / "task.json", "w"))
    import pytest
    with pytest.raises(ValueError, match="baseline"):
        cli.main([str(tmp_path / "sys.json"), str(tmp_path / "task.json")], api=api, out=io.StringIO())


def test_single_tet_inverse_task(api, tmp_path):
    """config/test_single_tet_inverse.json through the CLI; the rest height of the apex against the oracle"""
    json.dump({"verbosity": 0, "threads": 1}, open(tmp_path / "sys.json", "w"))
    task = {"func": "test_single_tet_inverse", "spacing": 0.025, "energy_model": "neohookean_i", "order": 12,
            "material": {"young": 1e7, "poisson": 0.45}, "out_filename": str(tmp_path / "out" / "tet")}
    json.dump(task, open(tmp_path / "task.json", "w"))
    log = io.StringIO()
    assert cli.main([str(tmp_path / "sys.json"), str(tmp_path / "task.json")], api=api, out=log) == 0
    Vd = np.loadtxt(str(tmp_path / "out" / "tet-i1-neohookean_i.vertices.txt"))
    Vo, _ = ofea.test_single_tet_inverse(task)
    assert np.abs(Vd - Vo).max() <= 1e-9 * np.abs(Vo).max()
    assert "vertex 3:" in log.getvalue()


def test_cuboid_task(api, tmp_path):
    json.dump({"verbosity": 0, "threads": 1}, open(tmp_path / "sys.json", "w"))
    task = {"func": "test_cuboid", "x": 6, "y": 3, "z": 3, "spacing": 0.025, "energy_model": "neohookean_c", "order": 10,
            "material": {"young": 1e5, "poisson": 0.4}, "out_filename": str(tmp_path / "out" / "c")}
    json.dump(task, open(tmp_path / "task.json", "w"))
    assert cli.main([str(tmp_path / "sys.json"), str(tmp_path / "task.json")], api=api, out=io.StringIO()) == 0
    st = json.load(open(str(tmp_path / "out" / "c-i0-neohookean_c.json")))
    assert st["force_rms_recomp"] < 1e-8 and st["name"] == "cuboid"
    om = ofea.make_cuboid(6, 3, 3, 0.025)
    fixed = np.zeros((om.nr_vertices, 3), dtype=bool)
    fixed[om.V[:, 0] <= 0.0125] = True
    f = np.zeros((om.nr_vertices, 3))
    f[(om.V[:, 0] > (6 // 2 - 1) * 0.025 - 0.0125) & (om.V[:, 2] < 0.0125), 2] = -50.0
    model, solver, x = ofea.solve_static(om, ofea.Material(1e5, 0.4, 0.0), fixed, "neohookean_c", f, dict(task))
    Vd = np.loadtxt(str(tmp_path / "out" / "c-i0-neohookean_c.vertices.txt"))
    Vo = model.lt_inp.full_vertices(x)
    assert np.abs(Vd - Vo).max() <= 1e-8 * np.abs(Vo).max()
    # step count: the same solve beside a second oracle in lock step; equal to the free-running oracle's unless a
    # certified ill-conditioned Pade decision was logged
    from sanm_amd import fea as dfea
    dmesh = dfea.make_cuboid(6, 3, 3, 0.025)
    run = dfea.GravityRun.from_parts(api, dmesh, dict(task), fixed, f).construct()
    from oracle.anm import ANMEqnSolver as OEqn
    m2 = ofea.make_forward(om, ofea.Material(1e5, 0.4, 0.0), fixed, "neohookean_c")
    o2 = OEqn(m2.y, m2.lt_inp.mat, m2.lt_out, m2.lt_inp.out_shape, m2.lt_inp.x0, m2.lt_inp.copy_vtx_values(f),
              ofea.default_hyper(dict(task), converge_rms=1e-10, solution_check_tol=1e-3))
    ls = LockStep(run, o2).run_to_convergence()
    assert ls.nr_steps == st["iter"]
    print("test_cuboid: steps", st["iter"], "oracle", solver.get_nr_iter(), "events",
          [(e["step"], e["device"], e["oracle"]) for e in ls.events])
    if not ls.events:
        assert st["iter"] == solver.get_nr_iter()


def test_mesh_twist_task(api, tmp_path):
    """mesh_twist (fea/main.cpp:774-919) from TetGen files: one end of a bar held, the other rotated by 20
    degrees about the bar's axis; the oracle's run_with_vtx_delta on the same displacement."""
    om, tris = _block_files(tmp_path, dims=(6, 3, 3), sp=0.025)
    task = {"func": "mesh_twist", "mesh": "model/block.1", "energy_model": "arap", "order": 10,
            "material": {"young": 1e6, "poisson": 0.4}, "axis": [1, 0, 0], "ratio_lo": 0.1, "ratio_hi": 0.1,
            "angle": 20, "shift": [0, 0, 0], "rot_axis": 0, "out_filename": str(tmp_path / "out" / "tw")}
    json.dump(task, open(tmp_path / "task.json", "w"))
    assert cli.main([str(tmp_path / "sys.json"), str(tmp_path / "task.json")], api=api, out=io.StringIO()) == 0
    st = json.load(open(str(tmp_path / "out" / "tw.json")))
    assert st["force_rms_recomp"] < 1e-5 and st["iter_deform"] >= 1
    Vd = np.loadtxt(str(tmp_path / "out" / "tw.vertices.txt"))
    # the same displacement through the oracle
    proj = om.V[:, 0]
    pmin, pmax = proj.min(), proj.max()
    on_surf = np.zeros(om.nr_vertices, dtype=bool)
    on_surf[om.surface_vtx] = True
    lo, hi = pmin + (pmax - pmin) * 0.1, pmin + (pmax - pmin) * 0.9
    sel = ((proj <= lo) | (proj >= hi)) & on_surf
    fixed = np.zeros((om.nr_vertices, 3), dtype=bool)
    fixed[sel] = True
    bnd = np.nonzero(sel & (proj >= hi))[0]
    a = 20 * np.pi / 180
    rmat = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    delta = np.zeros_like(om.V)
    delta[bnd] = om.V[bnd] @ rmat.T - om.V[bnd]
    Vo, ost = ofea.run_with_vtx_delta(om, ofea.Material(1e6, 0.4, 0.0), fixed, "arap", dict(task), delta, om.V.copy(),
                                      False)
    assert np.abs(Vd - Vo).max() <= 1e-6 * np.abs(Vo).max()
    # step counts: the stage again with the oracle in lock step (LockStepPath for the implicit solver, LockStep for
    # the refinement); equal to the free-running oracle's unless a certified ill-conditioned decision was logged
    dmesh, _ = cli.read_tetgen(str(tmp_path / "model" / "block.1"))
    _, rec = lockstep_vtx_delta_stage(api, dmesh, om, ofea.Material(1e6, 0.4, 0.0), fixed, dict(task), delta,
                                      dmesh.V.copy(), False)
    assert (rec["iter_deform"], rec["iter_refine"]) == (st["iter_deform"], st["iter_refine"])
    print("mesh_twist: steps", (st["iter_deform"], st["iter_refine"]), "oracle", (ost["iter_deform"], ost["iter_refine"]),
          "events", [(e["step"], e["device"], e["oracle"]) for e in rec["events"]])
    if not rec["events"]:
        assert (st["iter_deform"], st["iter_refine"]) == (ost["iter_deform"], ost["iter_refine"])
