"""The parts of the reference's ``fea/`` layer that define the hot-path inputs.

ORACLE -- test infrastructure only (see oracle/__init__.py).

* TetGen reader / cuboid generator      fea/tetrahedral_mesh.cpp:93-260
* vertex normals, volumes, shape matrix fea/tetrahedral_mesh.cpp:31-69
* remap_in  (``MeshShapeMatTrans``)      fea/mesh_template.h:20-111
* remap_out (``MeshForceOutputTrans``)   fea/mesh_template.h:132-161
* constitutive graphs (pk1 / cauchy)    fea/material.cpp:10-115
* forward / inverse models              fea/mesh_template.h:174-219
* boundary rule + gravity load          fea/main.cpp:921-1046
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from . import symbolic as S
from .anm import ANMEqnSolver, HyperParam


# ------------------------------------------------------------------ mesh --
class TetMesh:
    def __init__(self, vertices, tets, surface_vtx=None):
        self.V = np.ascontiguousarray(vertices, dtype=np.float64)  # (nv,3)
        self.tets = np.ascontiguousarray(tets, dtype=np.int64)  # (T,4)
        self.surface_vtx = None if surface_vtx is None else np.asarray(sorted(set(int(i) for i in surface_vtx)))

    @property
    def nr_vertices(self):
        return self.V.shape[0]

    @property
    def nr_tet(self):
        return self.tets.shape[0]

    def resize_inplace(self, scale):
        self.V = self.V * scale

    def shape_matrix(self, V=None):
        """Ds/Dm: columns are v_i - v_0 (tetrahedral_mesh.cpp:20-29, :41-45)."""
        V = self.V if V is None else V
        x = V[self.tets]  # (T,4,3)
        d = x[:, 1:, :] - x[:, :1, :]  # (T,3(dm),3(r))
        return np.ascontiguousarray(np.swapaxes(d, 1, 2))  # [e, r, dm]

    def vertex_norms_volumes(self):
        """tetrahedral_mesh.cpp:31-69: per tet 4 normals (T,4,3) and volumes."""
        x = self.V[self.tets]
        v1, v2, v3 = x[:, 1] - x[:, 0], x[:, 2] - x[:, 0], x[:, 3] - x[:, 0]
        det = np.einsum("ij,ij->i", v1, np.cross(v2, v3))
        vol = np.abs(det) / 6
        t1, t2, t3 = np.cross(v2, v3), np.cross(v3, v1), np.cross(v1, v2)
        sgn = np.where(det > 0, -1.0, 1.0)[:, None]
        t1, t2, t3 = t1 * sgn, t2 * sgn, t3 * sgn
        norms = np.stack([-(t1 + t2 + t3), t1, t2, t3], axis=1) * (1.0 / 6)
        return norms, vol


def read_tetgen(filebase):
    """fea/tetrahedral_mesh.cpp:206-260."""
    def toks(path):
        with open(path) as f:
            return f.read().split()
    tn = toks(filebase + ".node")
    nv, dim, nattr, bm = (int(t) for t in tn[:4])
    assert dim == 3 and nattr == 0 and bm == 0
    a = np.array(tn[4:4 + nv * 4], dtype=np.float64).reshape(nv, 4)
    assert np.all(a[:, 0] == np.arange(nv))
    V = a[:, 1:4]
    te = toks(filebase + ".ele")
    nt, npt, nattr = (int(t) for t in te[:3])
    assert npt == 4 and nattr == 0
    e = np.array(te[3:3 + nt * 5], dtype=np.int64).reshape(nt, 5)
    assert np.all(e[:, 0] == np.arange(nt))
    tf = toks(filebase + ".face")
    nf, bmark = int(tf[0]), int(tf[1])
    w = 5 if bmark else 4
    fa = np.array(tf[2:2 + nf * w], dtype=np.int64).reshape(nf, w)
    surf = np.unique(fa[:, 1:4])
    return TetMesh(V, e[:, 1:5], surf)


def make_cuboid(nx, ny, nz, size):
    """fea/tetrahedral_mesh.cpp:93-204 (5 tets per cell)."""
    assert nx >= 2 and ny >= 2 and nz >= 2 and size > 0
    ii, jj, kk = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    V = np.stack([ii.ravel() * size, jj.ravel() * size, kk.ravel() * size], axis=1).astype(np.float64)
    surf = np.nonzero(((ii == 0) | (ii == nx - 1) | (jj == 0) | (jj == ny - 1) |
                       (kk == 0) | (kk == nz - 1)).ravel())[0]
    gid = lambda x, y, z: (x * ny + y) * nz + z
    tets = []
    pat = [(0, 2, 1, 5), (0, 4, 7, 5), (0, 2, 5, 7), (2, 6, 5, 7), (0, 7, 3, 2)]
    for i in range(nx - 1):
        for j in range(ny - 1):
            for k in range(nz - 1):
                h = [gid(i, j, k), gid(i + 1, j, k), gid(i + 1, j + 1, k), gid(i, j + 1, k),
                     gid(i, j, k + 1), gid(i + 1, j, k + 1), gid(i + 1, j + 1, k + 1), gid(i, j + 1, k + 1)]
                for p in pat:
                    tets.append([h[q] for q in p])
    return TetMesh(V, np.array(tets, dtype=np.int64), surf)


# ---------------------------------------------------------------- remaps --
class MeshShapeMatTrans:
    """remap_in: unknown vector -> (T,3,3) shape matrices minus the fixed part.

    fea/mesh_template.h:20-111.  ``fixed_mask`` is (nv,3) bool.  Output index
    ``e*9 + r*3 + (dm-1)`` gets ``+x[v_dm, r] - x[v_0, r]`` (+ delta * t);
    fixed coordinates are folded into ``bias``.
    """

    def __init__(self, mesh, fixed_mask, init_vtx_coord=None, vtx_delta=None):
        self.mesh = mesh
        nv = mesh.nr_vertices
        fixed_mask = np.asarray(fixed_mask, dtype=bool)
        assert fixed_mask.shape == (nv, 3)
        V0 = mesh.V if init_vtx_coord is None else np.asarray(init_vtx_coord, dtype=np.float64)
        self.has_delta = vtx_delta is not None
        free = ~fixed_mask
        self.vtx2uidx = np.full((nv, 3), -1, dtype=np.int64)
        self.vtx2uidx[free] = np.arange(int(free.sum()))  # vertex-major, coord-minor
        self.n = int(free.sum())
        self.x0 = V0[free].copy()
        self.vertex_loc = np.argwhere(free)  # (n,2): vtx, coord
        T = mesh.nr_tet
        rows, cols, vals = [], [], []
        bias = np.zeros((T, 3, 3))
        tets = mesh.tets
        for dm in range(1, 4):
            for r in range(3):
                oidx = np.arange(T) * 9 + r * 3 + (dm - 1)
                for vcol, sgn in ((0, -1.0), (dm, 1.0)):
                    v = tets[:, vcol]
                    u = self.vtx2uidx[v, r]
                    fx = u < 0
                    bias[fx, r, dm - 1] += sgn * V0[v[fx], r]
                    rows.append(oidx[~fx])
                    cols.append(u[~fx])
                    vals.append(np.full(int((~fx).sum()), sgn))
                if vtx_delta is not None:
                    d = vtx_delta[tets[:, dm], r] - vtx_delta[tets[:, 0], r]
                    nz = d != 0
                    rows.append(oidx[nz])
                    cols.append(np.full(int(nz.sum()), self.n))
                    vals.append(d[nz])
        ncol = self.n + (1 if self.has_delta else 0)
        self.mat = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                                 shape=(T * 9, ncol)).tocsr()
        self.bias = bias
        self.out_shape = (T, 3, 3)

    def copy_vtx_values(self, vtx_values):
        """mesh_template.h:113-128."""
        return np.asarray(vtx_values)[self.vertex_loc[:, 0], self.vertex_loc[:, 1]].copy()

    def full_vertices(self, x, base=None):
        """Scatter unknowns back into a (nv,3) vertex array."""
        out = (self.mesh.V if base is None else base).copy()
        out[self.vertex_loc[:, 0], self.vertex_loc[:, 1]] = x[:self.n]
        return out


def make_force_output_trans(inp: MeshShapeMatTrans):
    """remap_out: (T,3,3) stress -> nodal force on the unknowns.

    fea/mesh_template.h:132-161: row (v,c) = sum over tets e adjacent to v
    (in tet order, fea/mesh.cpp:27-52) and j of n_{e,v}[j] * P_e[c, j].
    """
    mesh = inp.mesh
    norms, _ = mesh.vertex_norms_volumes()
    T = mesh.nr_tet
    e = np.repeat(np.arange(T), 4)
    vid = np.tile(np.arange(4), T)
    v = mesh.tets.ravel()
    rows, cols, vals = [], [], []
    for c in range(3):
        u = inp.vtx2uidx[v, c]
        ok = u >= 0
        for j in range(3):
            rows.append(u[ok])
            cols.append(e[ok] * 9 + c * 3 + j)
            vals.append(norms[e[ok], vid[ok], j])
    return sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                         shape=(inp.n, T * 9)).tocsr()


# ------------------------------------------------------------- materials --
class Material:
    """fea/material.cpp:10-18."""

    def __init__(self, young, poisson, density=0.0):
        E, nu = float(young), float(poisson)
        self.young, self.poisson, self.density = E, nu, float(density)
        self.bulk = E / (3 * (1 - nu * 2))
        self.shear = E / (2 * (1 + nu))
        self.lame_first = E * nu / ((1 + nu) * (1 - nu * 2))


def pk1(energy, mat, F, dim=3):
    """First Piola-Kirchhoff stress graphs; fea/material.cpp:55-99."""
    if energy == "neohookean_i":
        k, mu = mat.bulk, mat.shear
        FTinv = S.batched_mat_inv_mul(F, None, True).batched_transpose()
        J = F.batched_det()
        Ic = F.pow(2).reduce_sum(-1)
        J23 = J.pow(-2.0 / 3.0)
        t2 = S.linear_combine([(mu / -3.0, J23 * Ic), (k, J * J), (-k, J)], 0) * FTinv
        return S.linear_combine([(mu, J23 * F), (1.0, t2)])
    if energy == "neohookean_c":
        mu, lam = mat.shear, mat.lame_first
        FTinv = S.batched_mat_inv_mul(F, None, True).batched_transpose()
        J = F.batched_det()
        return S.linear_combine([(mu, F), (-mu, FTinv), (lam, J.log() * FTinv)])
    if energy == "arap":
        return (F - F.batched_svd_w(True)[2]) * mat.shear
    if energy == "stvk_stretch":
        mu = mat.shear
        return S.linear_combine([(mu, F.batched_matmul(F.batched_transpose()).batched_matmul(F)), (-mu, F)])
    raise ValueError(energy)


def cauchy_stress(energy, mat, F, dim=3):
    """fea/material.cpp:20-53."""
    if energy == "neohookean_i":
        k, mu = mat.bulk, mat.shear
        b = F.batched_matmul(F.batched_transpose())
        J = F.batched_det()
        Ic = F.pow(2).reduce_sum(-1)
        J53 = J.pow(-5.0 / 3.0)
        t2 = S.linear_combine([(mu / -3.0, J53 * Ic), (k, J)], -k).batched_mul_eye(dim)
        return S.linear_combine([(mu, J53 * b), (1.0, t2)])
    if energy == "neohookean_c":
        lam, mu = mat.lame_first, mat.shear
        b = F.batched_matmul(F.batched_transpose())
        Jinv = F.batched_det().pow(-1)
        xI = S.linear_combine([(mu, Jinv), (lam, Jinv * Jinv.log())])
        return S.linear_combine([(mu, Jinv * b), (-1.0, xI.batched_mul_eye(dim))])
    raise ValueError(energy)


class ElasticForceModel:
    pass


def make_forward(mesh, mat, fixed_mask, energy, init_vtx_coord=None, vtx_delta=None):
    """fea/mesh_template.h:191-219."""
    m = ElasticForceModel()
    m.cg = S.ComputingGraph()
    m.lt_inp = MeshShapeMatTrans(mesh, fixed_mask, init_vtx_coord, vtx_delta)
    m.lt_out = make_force_output_trans(m.lt_inp)
    Ds = S.placeholder(m.cg) + S.constant(m.cg, m.lt_inp.bias)
    DmInv = S.constant(m.cg, np.linalg.inv(mesh.shape_matrix()))
    F = Ds.batched_matmul(DmInv)
    m.F = F
    m.y = pk1(energy, mat, F)
    return m


def make_inverse(mesh, mat, fixed_mask, energy):
    """fea/mesh_template.h:174-189 (unknown = rest shape, Cauchy stress)."""
    m = ElasticForceModel()
    m.cg = S.ComputingGraph()
    m.lt_inp = MeshShapeMatTrans(mesh, fixed_mask, None, None)
    m.lt_out = make_force_output_trans(m.lt_inp)
    Dm = S.placeholder(m.cg) + S.constant(m.cg, m.lt_inp.bias)
    Ds = S.constant(m.cg, mesh.shape_matrix())
    F = S.batched_mat_inv_mul(Dm, Ds, True)
    m.F = F
    m.y = cauchy_stress(energy, mat, F)
    return m


# ------------------------------------------------------------ gravity task --
def boundary_by_config(mesh, default_proj_dir, config):
    """fea/main.cpp:921-982: fix surface vertices in the lowest slab."""
    V = mesh.V
    d = np.asarray(config.get("boundary_proj_dir", default_proj_dir), dtype=np.float64)
    d = d / np.linalg.norm(d)
    p = V @ d
    thresh = p.min() + (p.max() - p.min()) * float(config["boundary_thresh"])
    surf = np.zeros(mesh.nr_vertices, dtype=bool)
    surf[mesh.surface_vtx] = True
    sel = (p <= thresh) & surf
    if "boundary_filter" in config:
        fc = config["boundary_filter"]
        fd = np.asarray(fc["dir"], dtype=np.float64)
        q = V @ fd
        dd = q.max() - q.min()
        th0, th1 = q.min() + dd * float(fc["min"]), q.min() + dd * float(fc["max"])
        sel &= (q >= th0) & (q <= th1)
    return np.repeat(sel[:, None], 3, axis=1)


def gravity_load(mesh, mat, g):
    """fea/main.cpp:1025-1036: vol*density*g/4 on each tet vertex."""
    _, vol = mesh.vertex_norms_volumes()
    g = np.asarray(g, dtype=np.float64)
    f = np.zeros((mesh.nr_vertices, 3))
    node = (vol * mat.density)[:, None] * g[None, :] / 4
    for j in range(4):
        np.add.at(f, mesh.tets[:, j], node)
    return f


def setup_gravity_task(mesh, config):
    """fea/main.cpp:984-1046 up to the solver construction.

    Returns (material, fixed_mask, f_load_full).  ``mesh`` must already be
    loaded; it is scaled in place by config['scale'].
    """
    mc = config["material"]
    mat = Material(mc["young"], mc["poisson"], mc["density"])
    g = np.asarray(config["g"], dtype=np.float64)
    if "scale" in config:
        mesh.resize_inplace(float(config["scale"]))
    fixed = boundary_by_config(mesh, -g, config)
    return mat, fixed, gravity_load(mesh, mat, g)


def make_gravity_solver(mesh, config, hyper=None, inverse=False):
    """fea/main.cpp:247-433 ``run_and_save`` up to the ANMEqnSolver ctor."""
    mat, fixed, f_load = setup_gravity_task(mesh, config)
    energy = config["energy_model"]
    model = make_inverse(mesh, mat, fixed, energy) if inverse else make_forward(mesh, mat, fixed, energy)
    f_sub = model.lt_inp.copy_vtx_values(f_load)
    if hyper is None:
        hyper = default_hyper(config, converge_rms=1e-10, solution_check_tol=1e-3)
    solver = ANMEqnSolver(model.y, model.lt_inp.mat, model.lt_out, model.lt_inp.out_shape,
                          model.lt_inp.x0, f_sub, hyper)
    return model, solver, f_sub


def run_anm(solver, max_iter=10000):
    """fea/main.cpp:172-190."""
    it = 0
    rms = [solver.residual_rms]
    while not solver.converged:
        solver.next_iter()
        rms.append(solver.residual_rms)
        it += 1
        assert it < max_iter
    return solver.get_x(), rms


# ----------------------------------------------------- reference test tasks --
def default_hyper(config, **over):
    """fea/main.cpp:105-121 (setup_solver_param)."""
    kw = dict(order=int(config.get("order", 20)),
              use_pade=not config.get("disable_pade", False),
              sanity_check=not config.get("disable_anm_sanity_check", False),
              xcoeff_l2_penalty=float(config.get("xcoeff_l2_penalty", 0)))
    kw.update(over)
    return HyperParam(**kw)


def solve_static(mesh, mat, fixed_mask, energy, f_load_full, config, inverse=False):
    """fea/main.cpp:247-433 (run_and_save, ANM branch).  Returns (model, solver, x)."""
    model = make_inverse(mesh, mat, fixed_mask, energy) if inverse else \
        make_forward(mesh, mat, fixed_mask, energy)
    f_sub = model.lt_inp.copy_vtx_values(f_load_full)
    hyper = default_hyper(config, converge_rms=1e-10, solution_check_tol=1e-3)
    solver = ANMEqnSolver(model.y, model.lt_inp.mat, model.lt_out, model.lt_inp.out_shape,
                          model.lt_inp.x0, f_sub, hyper)
    x, _ = run_anm(solver)
    return model, solver, x


def test_single_tet_inverse(config):
    """fea/main.cpp:582-625: one tet, base fixed, -1000 N on the apex, inverse
    (rest-shape) solve.  Known answer: apex rest height 0.022755286528750494
    (reference utils/check_single_tet.py:61)."""
    spacing = float(config["spacing"])
    mc = config["material"]
    mat = Material(mc["young"], mc["poisson"], mc.get("density", 0.0))
    ang = np.pi * 2 / 3
    V = np.zeros((4, 3))
    for i in range(3):
        V[i, 0] = np.cos(ang * i) * spacing
        V[i, 1] = np.sin(ang * i) * spacing
    V[3, 2] = spacing
    mesh = TetMesh(V, np.array([[0, 1, 2, 3]]))
    fixed = np.zeros((4, 3), dtype=bool)
    fixed[:3] = True
    f = np.zeros((4, 3))
    f[3, 2] = -1000
    model, solver, x = solve_static(mesh, mat, fixed, config["energy_model"], f, config, inverse=True)
    return model.lt_inp.full_vertices(x), solver


def run_anm_implicit(solver, t_dest=1.0, max_iter=10000):
    """fea/main.cpp:193-215."""
    it = 0
    t_up = [solver.get_t_upper()]
    while solver.get_t_upper() < t_dest:
        solver.update_approx()
        t_up.append(solver.get_t_upper())
        it += 1
        assert it < max_iter
    return solver.eval(solver.solve_a(t_dest))[0], t_up


def run_with_vtx_delta(mesh, mat, fixed_mask, energy, config, vtx_delta, vtx_coord, require_refine):
    """fea/main.cpp:436-580: displacement-driven solve with ANMImplicitSolver,
    then an optional order-6 ANMEqnSolver refinement.  Updates and returns
    vtx_coord; also returns a stats dict."""
    from .anm import ANMImplicitSolver
    stat = {}
    model = make_forward(mesh, mat, fixed_mask, energy, vtx_coord, vtx_delta)
    hyper = default_hyper(config, solution_check_tol=10.0)
    solver = ANMImplicitSolver(model.y, model.lt_inp.mat, model.lt_out, model.lt_inp.out_shape,
                               model.lt_inp.x0, 0.0, hyper)
    xt, t_up = run_anm_implicit(solver, 1.0)
    vtx_coord = model.lt_inp.full_vertices(xt, vtx_coord) + vtx_delta
    stat["iter_deform"] = solver.get_nr_iter()
    stat["t_upper"] = t_up

    def force_rms(vc):
        m2 = make_forward(mesh, mat, fixed_mask, energy, vc)
        y = S.eval_unary_func(m2.y, (m2.lt_inp.mat @ m2.lt_inp.x0).reshape(m2.lt_inp.out_shape))
        fr = m2.lt_out @ y.ravel()
        return float(np.sqrt(np.mean(fr ** 2)))

    frms = force_rms(vtx_coord)
    stat["force_rms_deform"] = frms
    require_refine = require_refine or frms >= 1e-10
    stat["iter_refine"] = 0
    if require_refine:
        m2 = make_forward(mesh, mat, fixed_mask, energy, vtx_coord)
        hyper2 = default_hyper(config, converge_rms=1e-5)
        hyper2.order = 6
        s2 = ANMEqnSolver(m2.y, m2.lt_inp.mat, m2.lt_out, m2.lt_inp.out_shape, m2.lt_inp.x0,
                          np.zeros(m2.lt_inp.n), hyper2)
        x, rms = run_anm(s2)
        vtx_coord = m2.lt_inp.full_vertices(x, vtx_coord)
        stat["iter_refine"] = s2.get_nr_iter()
        stat["refine_rms"] = rms
    # enforce_dst_boundary (main.cpp:452-461)
    dst = mesh.V + vtx_delta
    vtx_coord = np.where(fixed_mask, dst, vtx_coord)
    stat["force_rms_recomp"] = force_rms(vtx_coord)
    stat["iter_tot"] = stat["iter_deform"] + stat["iter_refine"]
    return vtx_coord, stat


def test_cuboid_twist(config):
    """fea/main.cpp:665-772 (BASELINE config 1 with config/test_simple_cuboid_twist.json)."""
    nx, ny, nz = int(config["x"]), int(config["y"]), int(config["z"])
    spacing = float(config["spacing"])
    mc = config["material"]
    mat = Material(mc["young"], mc["poisson"], mc.get("density", 0.0))
    mesh = make_cuboid(nx, ny, nz, spacing)
    x_thresh = spacing * (nx - 1.5)
    vtx_cur = mesh.V.copy()
    fixed = np.zeros((mesh.nr_vertices, 3), dtype=bool)
    left = vtx_cur[:, 0] <= spacing / 2.0
    right = vtx_cur[:, 0] >= x_thresh
    fixed[left | right] = True
    bnd_idx = np.nonzero(right)[0]
    assert bnd_idx.size
    energy = config["energy_model"]
    stats = []

    def update_to_next(bnd_next, require_refine):
        nonlocal vtx_cur
        delta = np.zeros_like(vtx_cur)
        delta[bnd_idx] = bnd_next - vtx_cur[bnd_idx]
        vtx_cur, st = run_with_vtx_delta(mesh, mat, fixed, energy, config, delta, vtx_cur, require_refine)
        stats.append(st)

    bnd_init = vtx_cur[bnd_idx].copy()
    remain = float(config["rotate"])
    finished = 0.0
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
        nxt = bnd_init @ rmat.T + shift * spacing
        update_to_next(nxt, True)
    return vtx_cur, stats
