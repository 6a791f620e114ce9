"""Host-side front-end pieces needed to drive the hot path from a config:
mesh data files, the reference's boundary rule, the gravity task and the
``run_anm`` loop (fea/main.cpp:172-190, :921-1046).  All numerics go through
the C ABI (``sanm_amd.api``); nothing here computes on the CPU beyond input
preparation, and nothing here touches ``oracle/``.
"""
from __future__ import annotations

import json
import os
import time

import numpy as np

from .api import ANMEqnSolver, Api

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "data", "meshes")


class Mesh:
    def __init__(self, vertices, tets, surface_vtx):
        self.V = np.ascontiguousarray(vertices, dtype=np.float64)
        self.tets = np.ascontiguousarray(tets, dtype=np.int32)
        self.surface_vtx = np.ascontiguousarray(surface_vtx, dtype=np.int64)

    @property
    def nr_vertices(self):
        return self.V.shape[0]

    @property
    def nr_tet(self):
        return self.tets.shape[0]


def load_mesh_npz(path):
    d = np.load(path)
    return Mesh(d["vertices"], d["tets"], d["surface_vtx"])


def load_named_config(name):
    """A BASELINE config (task json merged with its overrides) + its mesh."""
    cfg = json.load(open(os.path.join(DATA_DIR, name + ".json")))
    mesh = load_mesh_npz(os.path.join(DATA_DIR, cfg["mesh_npz"]))
    return cfg, mesh


def refine_mesh(mesh, levels=1):
    """Regular 1 -> 8 subdivision of every tet (edge midpoints; the inner octahedron cut along the 1-3 / 0-2 midpoint
    diagonal), `levels` times: an organic mesh at scale from a BASELINE one (bench.py, workload `refine:<name>:<levels>`;
    not a reference function -- the reference loads its big meshes from files that are not in the image).  Orientation
    of the children follows the parent's.  The surface is carried along as FACES (the parent's faces that belong to one
    tet only and join three of its surface vertices; each splits into four): a midpoint is a surface vertex iff its edge
    lies in a surface face -- interior edges that join two surface vertices (thin parts) do not make one."""
    V, T = mesh.V, mesh.tets.astype(np.int64)
    nv = V.shape[0]
    surf = np.zeros(nv, dtype=bool)
    surf[mesh.surface_vtx] = True
    faces = np.sort(np.concatenate([T[:, [1, 2, 3]], T[:, [0, 2, 3]], T[:, [0, 1, 3]], T[:, [0, 1, 2]]], axis=0), axis=1)
    fkey = (faces[:, 0] * nv + faces[:, 1]) * nv + faces[:, 2]
    _, first, cnt = np.unique(fkey, return_index=True, return_counts=True)
    sf = faces[first[cnt == 1]]
    sf = sf[surf[sf].all(axis=1)]
    for _ in range(levels):
        nv = V.shape[0]
        pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
        e = np.concatenate([np.sort(T[:, list(pq)], axis=1) for pq in pairs], axis=0)
        key = e[:, 0] * nv + e[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        a, b = uniq // nv, uniq % nv
        V = np.concatenate([V, 0.5 * (V[a] + V[b])], axis=0)

        def mid(p, q, nv=nv, uniq=uniq):  # midpoint vertex of the edges (p, q)
            lo, hi = np.minimum(p, q), np.maximum(p, q)
            return nv + np.searchsorted(uniq, lo * nv + hi)
        fa, fb, fc = sf[:, 0], sf[:, 1], sf[:, 2]
        mab, mbc, mac = mid(fa, fb), mid(fb, fc), mid(fa, fc)
        sf = np.concatenate([np.stack([fa, mab, mac], 1), np.stack([fb, mab, mbc], 1), np.stack([fc, mac, mbc], 1),
                             np.stack([mab, mbc, mac], 1)], axis=0)
        m = (nv + inv).reshape(6, -1).T  # per tet: midpoints of 01 02 03 12 13 23
        m01, m02, m03, m12, m13, m23 = (m[:, i] for i in range(6))
        v0, v1, v2, v3 = (T[:, i] for i in range(4))
        T = np.concatenate([
            np.stack([v0, m01, m02, m03], 1), np.stack([m01, v1, m12, m13], 1),
            np.stack([m02, m12, v2, m23], 1), np.stack([m03, m13, m23, v3], 1),
            np.stack([m01, m02, m03, m13], 1), np.stack([m01, m12, m02, m13], 1),
            np.stack([m02, m03, m13, m23], 1), np.stack([m02, m12, m23, m13], 1)], axis=0)
        # keep every child's orientation the parent's (positive signed volume stays positive)
        d = V[T[:, 1:]] - V[T[:, :1]]
        vol = np.einsum("ij,ij->i", np.cross(d[:, 0], d[:, 1]), d[:, 2])
        parent_sign = np.sign(vol[: len(v0)])  # child 0 is similar to its parent
        flip = np.sign(vol) != np.tile(parent_sign, 8)
        T[flip] = T[flip][:, [0, 2, 1, 3]]
    out = np.zeros(V.shape[0], dtype=bool)
    out[mesh.surface_vtx] = True
    out[sf.ravel()] = True
    return Mesh(V, T, np.nonzero(out)[0])


def make_cuboid(nx, ny, nz, size):
    """TetrahedralMesh::make_cuboid, fea/tetrahedral_mesh.cpp:93-204 (5 tets / cell)."""
    ii, jj, kk = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    V = np.stack([ii.ravel(), jj.ravel(), kk.ravel()], axis=1).astype(np.float64) * size
    surf = np.nonzero(((ii == 0) | (ii == nx - 1) | (jj == 0) | (jj == ny - 1) |
                       (kk == 0) | (kk == nz - 1)).ravel())[0]
    ci, cj, ck = np.meshgrid(np.arange(nx - 1), np.arange(ny - 1), np.arange(nz - 1), indexing="ij")
    ci, cj, ck = ci.ravel(), cj.ravel(), ck.ravel()
    gid = lambda x, y, z: (x * ny + y) * nz + z
    h = np.stack([gid(ci, cj, ck), gid(ci + 1, cj, ck), gid(ci + 1, cj + 1, ck), gid(ci, cj + 1, ck),
                  gid(ci, cj, ck + 1), gid(ci + 1, cj, ck + 1), gid(ci + 1, cj + 1, ck + 1),
                  gid(ci, cj + 1, ck + 1)], axis=1)
    pat = np.array([(0, 2, 1, 5), (0, 4, 7, 5), (0, 2, 5, 7), (2, 6, 5, 7), (0, 7, 3, 2)])
    tets = h[:, pat].reshape(-1, 4)
    return Mesh(V, tets, surf)


def hyper_from_config(api: Api, config, **over):
    """setup_solver_param, fea/main.cpp:105-121 (+ run_and_save :382-385)."""
    kw = dict(order=int(config.get("order", 20)),
              use_pade=0 if config.get("disable_pade", False) else 1,
              sanity_check=0 if config.get("disable_anm_sanity_check", False) else 1,
              xcoeff_l2_penalty=float(config.get("xcoeff_l2_penalty", 0)),
              converge_rms=1e-10, solution_check_tol=1e-3)
    kw.update(over)
    return api.default_hyper(**kw)


def boundary_by_config(api: Api, mesh: Mesh, config):
    """setup_boundary_by_config (fea/main.cpp:921-982) with gravity()'s default projection direction -g"""
    g = np.asarray(config["g"], dtype=np.float64)
    proj = np.asarray(config.get("boundary_proj_dir", -g), dtype=np.float64)
    flt = config.get("boundary_filter")
    return api.boundary_by_threshold(
        mesh.V, mesh.surface_vtx, proj, float(config["boundary_thresh"]),
        None if flt is None else flt["dir"], 0.0 if flt is None else float(flt["min"]),
        0.0 if flt is None else float(flt["max"]))


def setup_gravity(api: Api, mesh: Mesh, config, boundary=True):
    """gravity(), fea/main.cpp:984-1046, up to run_and_save.  Scales the mesh
    in place (once).  Returns (fixed_mask (nv,3) bool or None, f_load (nv,3)); boundary=False leaves the
    fixed set to the caller (the `.bou` file of the mesh, sanm_amd/cli.py)."""
    mc = config["material"]
    g = np.asarray(config["g"], dtype=np.float64)
    if "scale" in config and not getattr(mesh, "_scaled", False):
        # mesh->resize_inplace (fea/main.cpp:995-997); marked so that a second run on the same Mesh object does
        # not scale it again
        mesh.V = mesh.V * float(config["scale"])
        mesh._scaled = True
    fixed = boundary_by_config(api, mesh, config) if boundary else None
    f_load = api.gravity_load(mesh.V, mesh.tets, float(mc["density"]), g)
    return fixed, f_load


class GravityRun:
    """run_and_save (fea/main.cpp:247-433), ANM branch, on the device path."""

    @classmethod
    def from_parts(cls, api: Api, mesh: Mesh, config, fixed, f_load, inverse=False, **hyper_over):
        """a static solve with the fixed set and the nodal load given by the caller (the test_* tasks, `.bou`
        files)"""
        return cls(api, mesh, config, inverse=inverse, _parts=(np.asarray(fixed, dtype=bool), np.asarray(f_load)),
                   **hyper_over)

    def __init__(self, api: Api, mesh: Mesh, config, inverse=False, shard=None, _parts=None, **hyper_over):
        self.api = api
        self.mesh = mesh
        self.config = config
        t0 = time.perf_counter()
        self.fixed, self.f_load = _parts if _parts is not None else setup_gravity(api, mesh, config)
        mc = config["material"]
        self.model = api.fea_model(mesh.V, mesh.tets, self.fixed, config["energy_model"],
                                   float(mc["young"]), float(mc["poisson"]), inverse=inverse)
        self.f_sub = self.model.copy_vtx_values(self.f_load)
        self.hyper = hyper_from_config(api, config, **hyper_over)
        self.time_prep = time.perf_counter() - t0
        self.solver = None
        self.shard = shard
        self.rms = []
        self.time_solve = 0.0

    def construct(self):
        t0 = time.perf_counter()
        self.solver = ANMEqnSolver(self.api, self.model.y, self.model.lt_inp, self.model.lt_out,
                                   self.model.x0(), self.f_sub, self.hyper, shard=self.shard)
        self.solver._keep = self.solver._keep + (self.model,)  # the graph / remap views belong to the model
        self.rms = [self.solver.residual_rms()]
        self.time_solve += time.perf_counter() - t0
        return self

    def step(self):
        """one ANMEqnSolver::next_iter (one ANM continuation step unless converged)"""
        t0 = time.perf_counter()
        self.solver.next_iter()
        self.rms.append(self.solver.residual_rms())
        self.time_solve += time.perf_counter() - t0

    def run(self, max_iter=100000):
        """run_anm, fea/main.cpp:172-190"""
        if self.solver is None:
            self.construct()
        it = 0
        while not self.solver.converged():
            self.step()
            it += 1
            if it >= max_iter:
                break
        return self

    def vertices(self):
        return self.model.full_vertices(self.solver.get_x(), self.mesh.V)

    def stats(self):
        """the keys the reference writes to its stats json (fea/main.cpp:425-431)"""
        return {"time_prep": self.time_prep, "time_solve": self.time_solve,
                "order": int(self.hyper.order), "pade": bool(self.hyper.use_pade),
                "iter": int(self.solver.get_nr_iter()), "mesh_V": self.mesh.nr_vertices,
                "mesh_F": self.mesh.nr_tet, "residual_rms": self.rms}


# ---------------------------------------------------------------------------
# displacement-driven task (BASELINE config 1): fea/main.cpp:436-580, :665-772
# ---------------------------------------------------------------------------
def run_anm_implicit(solver, t_dest=1.0, max_iter=100000):
    """run_anm(ANMImplicitSolver&), fea/main.cpp:193-215"""
    t_up = [solver.get_t_upper()]
    it = 0
    while solver.get_t_upper() < t_dest:
        solver.update_approx()
        t_up.append(solver.get_t_upper())
        it += 1
        if it >= max_iter:
            raise RuntimeError("implicit ANM did not reach t_dest")
    return solver.eval(solver.solve_a(t_dest))[0], t_up


def _force_rms(api: Api, mesh: Mesh, fixed, energy, mat_cfg, vtx_coord):
    """eval_force_rms, fea/main.cpp:468-474: |f(x)|_rms at the current vertices
    (order-0 pass of the device program + remap_out through a throw-away
    ANMSolverVecScale-free path: TaylorCoeffProp push_xi and the remap matrix)."""
    from .api import TaylorCoeffProp
    m = api.fea_model(mesh.V, mesh.tets, fixed, energy, float(mat_cfg["young"]), float(mat_cfg["poisson"]),
                      init_vtx_coord=vtx_coord)
    prop = TaylorCoeffProp(api, m.y, m.lt_inp, 1, mesh.nr_tet)
    y = prop.push_xi(m.x0())
    f = m.lt_out.to_scipy() @ y.ravel()
    return float(np.sqrt(np.mean(f ** 2)))


def run_with_vtx_delta(api: Api, mesh: Mesh, fixed, config, vtx_delta, vtx_coord, require_refine, refine_f_load=None):
    """run_with_vtx_delta, fea/main.cpp:436-580 (ANM branch).  refine_f_load: (nv,3) nodal load of the refinement
    pass (mesh_twist with add_gravity), zero otherwise."""
    from .api import ANMImplicitSolver
    energy, mc = config["energy_model"], config["material"]
    stat = {}
    model = api.fea_model(mesh.V, mesh.tets, fixed, energy, float(mc["young"]), float(mc["poisson"]),
                          init_vtx_coord=vtx_coord, vtx_delta=vtx_delta)
    hp = hyper_from_config(api, config, solution_check_tol=10.0, converge_rms=1e-5)
    solver = ANMImplicitSolver(api, model.y, model.lt_inp, model.lt_out, model.x0(), 0.0, hp)
    xt, t_up = run_anm_implicit(solver, 1.0)
    vtx_coord = model.full_vertices(xt, vtx_coord) + vtx_delta
    stat["iter_deform"] = solver.get_nr_iter()
    stat["t_upper"] = t_up
    frms = _force_rms(api, mesh, fixed, energy, mc, vtx_coord)
    stat["force_rms_deform"] = frms
    require_refine = require_refine or frms >= 1e-10
    stat["iter_refine"] = 0
    if require_refine:
        m2 = api.fea_model(mesh.V, mesh.tets, fixed, energy, float(mc["young"]), float(mc["poisson"]),
                           init_vtx_coord=vtx_coord)
        hp2 = hyper_from_config(api, config, converge_rms=1e-5, solution_check_tol=1e-4, order=6)
        f_sub = np.zeros(m2.n) if refine_f_load is None else m2.copy_vtx_values(refine_f_load)
        s2 = ANMEqnSolver(api, m2.y, m2.lt_inp, m2.lt_out, m2.x0(), f_sub, hp2)
        rms = [s2.residual_rms()]
        while not s2.converged():
            s2.next_iter()
            rms.append(s2.residual_rms())
        vtx_coord = m2.full_vertices(s2.get_x(), vtx_coord)
        stat["iter_refine"] = s2.get_nr_iter()
        stat["refine_rms"] = rms
    dst = mesh.V + vtx_delta
    vtx_coord = np.where(fixed, dst, vtx_coord)
    stat["force_rms_recomp"] = _force_rms(api, mesh, fixed, energy, mc, vtx_coord)
    stat["iter_tot"] = stat["iter_deform"] + stat["iter_refine"]
    return vtx_coord, stat


def test_cuboid_twist(api: Api, config, stage=None):
    """test_cuboid_twist, fea/main.cpp:665-772 (config/test_simple_cuboid_twist.json).  `stage`: the function that
    runs one prescribed-displacement stage (default run_with_vtx_delta; the parity tests pass one that runs the
    oracle beside it)."""
    stage = stage or run_with_vtx_delta
    nx, ny, nz = int(config["x"]), int(config["y"]), int(config["z"])
    spacing = float(config["spacing"])
    mesh = make_cuboid(nx, ny, nz, spacing)
    x_thresh = spacing * (nx - 1.5)
    vtx_cur = mesh.V.copy()
    fixed = np.zeros((mesh.nr_vertices, 3), dtype=bool)
    left = vtx_cur[:, 0] <= spacing / 2.0
    right = vtx_cur[:, 0] >= x_thresh
    fixed[left | right] = True
    bnd_idx = np.nonzero(right)[0]
    stats = []

    def update_to_next(bnd_next, require_refine):
        nonlocal vtx_cur
        delta = np.zeros_like(vtx_cur)
        delta[bnd_idx] = bnd_next - vtx_cur[bnd_idx]
        vtx_cur, st = stage(api, mesh, fixed, config, delta, vtx_cur, require_refine)
        stats.append(st)

    bnd_init = vtx_cur[bnd_idx].copy()
    remain, finished = float(config["rotate"]), 0.0
    split = float(config.get("rotate_split", 90))
    while remain > 1e-5:
        rot = min(remain, split)
        remain -= rot
        finished += rot
        r = finished * np.pi / 180
        rmat = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
        nxt = bnd_init @ rmat.T
        nxt = nxt + (bnd_init.mean(axis=0) - nxt.mean(axis=0))
        update_to_next(nxt, False)
    bnd_init = vtx_cur[bnd_idx].copy()
    for bd in config["bend"]:
        ang = float(bd["angle"]) * np.pi / 180
        shift = np.asarray(bd["shift"], dtype=np.float64)
        rmat = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
        update_to_next(bnd_init @ rmat.T + shift * spacing, True)
    return vtx_cur, stats
