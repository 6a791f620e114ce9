"""ctypes binding of the C ABI in include/sanm_hip.h, plus thin Python classes
that mirror the reference's C++ interface for this path (same names and
argument meaning): ``SymbolVar`` / ``ComputingGraph`` (libsanm/oprs.h),
``TaylorCoeffProp`` (libsanm/symbolic.h:337-383), ``ANMEqnSolver`` /
``ANMSolverVecScale`` / ``ANMImplicitSolver`` (libsanm/anm.h:209-305) and the
fea model builders (fea/mesh_template.h:174-219).

This module only talks to the shared library it is given; it contains no
numerical fallback.  ``sanm_amd.get_api()`` loads the HIP product library.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_u64p = C.POINTER(C.c_uint64)
c_u32p = C.POINTER(C.c_uint32)
c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)


class SanmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[sanm_hip error {code}] {msg}")
        self.code = code
        self.msg = msg


class SanmAssertionError(SanmError):
    """SANMAssertionError (libsanm/utils.h:34-50)"""


class SanmNumericalError(SanmError):
    """SANMNumericalError (libsanm/utils.h:34-50)"""


class SanmUnsupportedError(SanmError):
    pass


_ERR = {1: SanmAssertionError, 2: SanmNumericalError, 4: SanmUnsupportedError}


class HyperParamC(C.Structure):
    _fields_ = [("use_pade", C.c_int), ("sanity_check", C.c_int), ("order", C.c_int),
                ("maxr", C.c_double), ("solution_check_tol", C.c_double),
                ("xcoeff_l2_penalty", C.c_double), ("converge_rms", C.c_double),
                ("solver_rtol", C.c_double), ("solver_maxit", C.c_int), ("solver_kind", C.c_int),
                ("profile", C.c_int), ("solver_refine", C.c_int)]


class StatsC(C.Structure):
    _fields_ = [("nr_unknown", C.c_int64), ("nr_tet", C.c_int64), ("jacobian_nnz", C.c_int64),
                ("assembly_contribs", C.c_int64), ("nr_linear_solve", C.c_int64),
                ("linear_iters_total", C.c_int64), ("linear_iters_last", C.c_int64),
                ("linear_relres_last", C.c_double), ("arena_bytes", C.c_double),
                ("factor_nnz", C.c_int64), ("nr_front", C.c_int64), ("nr_level", C.c_int64),
                ("max_front", C.c_int64), ("factor_flops", C.c_double),
                ("factor_flops_own", C.c_double), ("factor_flops_top", C.c_double),
                ("nr_subtree", C.c_int64), ("nr_subtree_own", C.c_int64),
                ("dist_schur_doubles", C.c_int64), ("dist_inbox_doubles", C.c_int64),
                ("factor_flops_top_own", C.c_double), ("factor_flops_critical", C.c_double),
                ("nr_dist_stage", C.c_int64), ("front_store_doubles", C.c_int64)]


ENERGY = {"neohookean_i": 0, "neohookean_c": 1, "arap": 2, "stvk_stretch": 3}

# every symbol include/sanm_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "sanm_hip_init", "sanm_hip_last_error", "sanm_hip_backend_name", "sanm_hip_abi_version",
    "sanm_hip_comm_available", "sanm_hip_comm_unique_id", "sanm_hip_comm_init", "sanm_hip_comm_destroy", "sanm_hip_comm_query",
    "sanm_graph_create", "sanm_graph_destroy", "sanm_graph_placeholder", "sanm_graph_constant", "sanm_graph_placeholder_vector", "sanm_graph_placeholder_matrix", "sanm_graph_constant_matrix", "sanm_graph_slice", "sanm_graph_concat",
    "sanm_graph_linear_combine", "sanm_graph_multiply", "sanm_graph_pow", "sanm_graph_log",
    "sanm_graph_reduce_sum", "sanm_graph_batched_matmul", "sanm_graph_batched_mat_inv_mul",
    "sanm_graph_batched_det", "sanm_graph_batched_transpose", "sanm_graph_batched_mul_eye",
    "sanm_graph_batched_svd_w",
    "sanm_sparse_desc_create", "sanm_sparse_desc_destroy", "sanm_sparse_desc_get",
    "sanm_sparse_desc_set_out_coords", "sanm_direct_solver_create", "sanm_direct_solver_destroy",
    "sanm_direct_solver_factor", "sanm_direct_solver_solve", "sanm_direct_solver_stats", "sanm_direct_solver_apply", "sanm_direct_solver_coeff_l2",
    "sanm_taylor_create", "sanm_taylor_destroy", "sanm_taylor_push_xi",
    "sanm_taylor_compute_next_order_bias", "sanm_taylor_output_size", "sanm_taylor_get_jacobian", "sanm_taylor_get_var",
    "sanm_taylor_reset",
    "sanm_hyper_param_default", "sanm_anm_eqn_solver_create", "sanm_anm_eqn_solver_create_sharded",
    "sanm_anm_vecscale_solver_create",
    "sanm_anm_implicit_solver_create", "sanm_anm_solver_destroy", "sanm_anm_next_iter",
    "sanm_anm_update_approx", "sanm_anm_restart", "sanm_anm_run_steps", "sanm_anm_spec_source", "sanm_rtc_compile_check", "sanm_rtc_cache_stats", "sanm_rtc_cache_probe", "sanm_direct_solver_dist_plan", "sanm_test_set_p2p", "sanm_anm_time_kernel", "sanm_anm_pass_timing", "sanm_anm_converged", "sanm_anm_residual_rms", "sanm_anm_get_x",
    "sanm_anm_get_t_upper", "sanm_anm_get_t_max_a", "sanm_anm_solve_a", "sanm_anm_eval",
    "sanm_anm_nr_iter", "sanm_anm_nr_xt_coeffs", "sanm_anm_xt_coeff", "sanm_anm_has_pade",
    "sanm_anm_get_stats", "sanm_anm_get_stats_sized", "sanm_anm_setup_profile", "sanm_rtc_cache_drop_memory", "sanm_fea_spec_source", "sanm_rtc_source_key", "sanm_rtc_compile_to_file", "sanm_rtc_embedded_hits", "sanm_anm_profile", "sanm_anm_profile_counts", "sanm_anm_profile_launches", "sanm_anm_set_profile", "sanm_anm_debug_inject", "sanm_anm_trace", "sanm_anm_pade_diag", "sanm_anm_verbose_text", "sanm_anm_jacobian_csr",
    "sanm_fea_model_create", "sanm_fea_model_destroy", "sanm_fea_model_nr_unknown",
    "sanm_fea_model_graph", "sanm_fea_model_output_var", "sanm_fea_model_F_var",
    "sanm_fea_model_remap_inp", "sanm_fea_model_remap_out", "sanm_fea_model_x0",
    "sanm_fea_model_copy_vtx_values", "sanm_fea_model_scatter", "sanm_fea_gravity_load",
    "sanm_fea_boundary_by_threshold", "sanm_poly_solve_eqn", "sanm_poly_real_roots", "sanm_poly_roots",
    "sanm_pade_create", "sanm_pade_destroy", "sanm_pade_estimate_valid_range", "sanm_pade_get_t_max", "sanm_pade_solve_a",
    "sanm_pade_eval_xt",
]


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Api:
    """One loaded shared library."""

    def __init__(self, lib: C.CDLL):
        self.lib = lib
        lib.sanm_hip_last_error.restype = C.c_char_p
        lib.sanm_hip_backend_name.restype = C.c_char_p
        for name in ("sanm_fea_model_graph", "sanm_fea_model_remap_inp", "sanm_fea_model_remap_out"):
            getattr(lib, name).restype = C.c_void_p
        for name in SYMBOLS:
            fn = getattr(lib, name)
            if fn.restype is C.c_int and name.endswith("_destroy") and name != "sanm_hip_comm_destroy":
                fn.restype = None
        self._initialised = False

    # -- plumbing ---------------------------------------------------------
    def check(self, rc):
        if rc != 0:
            msg = self.lib.sanm_hip_last_error().decode()
            raise _ERR.get(rc, SanmError)(rc, msg)

    def init(self, device=0):
        self.check(self.lib.sanm_hip_init(C.c_int(device)))
        self._initialised = True
        return self

    def comm_available(self):
        """True when this process can load RCCL from C++ (no collective, never raises)"""
        return bool(self.lib.sanm_hip_comm_available())

    def comm_unique_id(self):
        """ncclUniqueId (128 bytes) of a new communicator; rank 0 calls this and hands the bytes to all ranks"""
        buf = C.create_string_buffer(128)
        self.check(self.lib.sanm_hip_comm_unique_id(buf, C.c_size_t(128)))
        return buf.raw

    def comm_init(self, rank, world, uid: bytes):
        """join the library's RCCL communicator (collective over all ranks)"""
        self.check(self.lib.sanm_hip_comm_init(C.c_int(rank), C.c_int(world), C.c_char_p(uid), C.c_size_t(len(uid))))

    def comm_query(self):
        """(size, rank) of the live communicator as RCCL reports them; (0, 0) without one"""
        w, r = C.c_int(0), C.c_int(0)
        self.check(self.lib.sanm_hip_comm_query(C.byref(w), C.byref(r)))
        return w.value, r.value

    def comm_destroy(self):
        self.check(self.lib.sanm_hip_comm_destroy())

    def backend_name(self):
        return self.lib.sanm_hip_backend_name().decode()

    # -- builders ---------------------------------------------------------
    def graph(self):
        return ComputingGraph(self)

    def sparse_desc(self, mat):
        return SparseLinearDesc(self, mat=mat)

    def default_hyper(self, **kw):
        hp = HyperParamC()
        self.lib.sanm_hyper_param_default(C.byref(hp), 1)
        for k, v in kw.items():
            if not hasattr(hp, k):
                raise AttributeError(k)
            setattr(hp, k, v)
        return hp

    # -- host scalar helpers ----------------------------------------------
    def poly_solve_eqn(self, f, xmin, xmax, b=0.0, eps=1e-6):
        f = _f64(f)
        out = C.c_double()
        self.check(self.lib.sanm_poly_solve_eqn(_dp(f), C.c_int(f.size), C.c_double(xmin),
                                                C.c_double(xmax), C.c_double(b), C.c_double(eps),
                                                C.byref(out)))
        return out.value

    def poly_real_roots(self, f):
        f = _f64(f)
        roots = np.zeros(max(f.size, 1))
        nr = C.c_int()
        self.check(self.lib.sanm_poly_real_roots(_dp(f), C.c_int(f.size), _dp(roots), C.byref(nr)))
        return None if nr.value < 0 else roots[:nr.value].copy()

    def poly_roots(self, f, only_real=False, max_iter=300, tol=1e-8):
        """unary_polynomial::roots: complex array, or None where the reference returns None."""
        f = _f64(f)
        re, im = np.zeros(max(f.size, 1)), np.zeros(max(f.size, 1))
        nr = C.c_int()
        self.check(self.lib.sanm_poly_roots(_dp(f), C.c_int(f.size), C.c_int(int(only_real)), C.c_int(max_iter),
                                            C.c_double(tol), _dp(re), _dp(im), C.byref(nr)))
        return None if nr.value < 0 else re[:nr.value] + 1j * im[:nr.value]

    # -- fea helpers --------------------------------------------------------
    def gravity_load(self, vertices, tets, density, g):
        V = _f64(vertices)
        t = np.ascontiguousarray(tets, dtype=np.int32)
        g = _f64(g)
        out = np.zeros_like(V)
        self.check(self.lib.sanm_fea_gravity_load(C.c_int64(V.shape[0]), _dp(V), C.c_int64(t.shape[0]),
                                                  t.ctypes.data_as(c_i32p), C.c_double(density), _dp(g),
                                                  _dp(out)))
        return out

    def boundary_by_threshold(self, vertices, surface_vtx, proj_dir, thresh, filter_dir=None,
                              filter_min=0.0, filter_max=0.0):
        V = _f64(vertices)
        surf = np.zeros(V.shape[0], dtype=np.uint8)
        surf[np.asarray(surface_vtx, dtype=np.int64)] = 1
        pd = _f64(proj_dir)
        fd = None if filter_dir is None else _f64(filter_dir)
        out = np.zeros((V.shape[0], 3), dtype=np.uint8)
        self.check(self.lib.sanm_fea_boundary_by_threshold(
            C.c_int64(V.shape[0]), _dp(V), surf.ctypes.data_as(c_u8p), _dp(pd), C.c_double(thresh),
            None if fd is None else _dp(fd), C.c_double(filter_min), C.c_double(filter_max),
            out.ctypes.data_as(c_u8p)))
        return out.astype(bool)

    def fea_model(self, vertices, tets, fixed_mask, energy, young, poisson, inverse=False,
                  init_vtx_coord=None, vtx_delta=None):
        return FeaModel(self, vertices, tets, fixed_mask, energy, young, poisson, inverse,
                        init_vtx_coord, vtx_delta)


class SymbolVar:
    """libsanm/oprs.h:14-63"""

    def __init__(self, graph, vid):
        self.graph = graph
        self.id = int(vid)

    def _mk(self, fn, *args):
        out = C.c_int()
        self.graph.api.check(fn(self.graph.h, *args, C.byref(out)))
        return SymbolVar(self.graph, out.value)

    def __add__(self, rhs):
        if isinstance(rhs, SymbolVar):
            return linear_combine([(1.0, self), (1.0, rhs)])
        return linear_combine([(1.0, self)], float(rhs))

    def __sub__(self, rhs):
        if isinstance(rhs, SymbolVar):
            return linear_combine([(1.0, self), (-1.0, rhs)])
        return self + (-float(rhs))

    def __rsub__(self, lhs):
        return linear_combine([(-1.0, self)], float(lhs))

    def slice(self, axis, begin=None, end=None, stride=1):
        """SymbolVar::slice (oprs.h:60): x[:, begin:end]; None like the reference's Maybe<int>"""
        lib = self.graph.api.lib
        return self._mk(lib.sanm_graph_slice, C.c_int(self.id), C.c_int(int(axis)), C.c_int(begin is not None),
                        C.c_int(0 if begin is None else int(begin)), C.c_int(end is not None),
                        C.c_int(0 if end is None else int(end)), C.c_int(int(stride)))

    def __mul__(self, rhs):
        lib = self.graph.api.lib
        if isinstance(rhs, SymbolVar):
            return self._mk(lib.sanm_graph_multiply, C.c_int(self.id), C.c_int(rhs.id))
        return linear_combine([(float(rhs), self)], 0.0)

    def reduce_sum(self, axis, keepdim=True):
        return self._mk(self.graph.api.lib.sanm_graph_reduce_sum, C.c_int(self.id), C.c_int(axis))

    def batched_transpose(self):
        return self._mk(self.graph.api.lib.sanm_graph_batched_transpose, C.c_int(self.id))

    def batched_matinv(self):
        return batched_mat_inv_mul(self, None, True)

    def batched_matmul(self, rhs):
        return self._mk(self.graph.api.lib.sanm_graph_batched_matmul, C.c_int(self.id), C.c_int(rhs.id))

    def batched_det(self):
        return self._mk(self.graph.api.lib.sanm_graph_batched_det, C.c_int(self.id))

    def batched_mul_eye(self, dim):
        return self._mk(self.graph.api.lib.sanm_graph_batched_mul_eye, C.c_int(self.id), C.c_int(dim))

    def pow(self, exp):
        return self._mk(self.graph.api.lib.sanm_graph_pow, C.c_int(self.id), C.c_double(exp))

    def log(self):
        return self._mk(self.graph.api.lib.sanm_graph_log, C.c_int(self.id))

    def batched_svd_w(self, require_rotation=False):
        usw = (C.c_int * 3)()
        self.graph.api.check(self.graph.api.lib.sanm_graph_batched_svd_w(
            self.graph.h, C.c_int(self.id), C.c_int(1 if require_rotation else 0), usw))
        return [SymbolVar(self.graph, usw[i]) for i in range(3)]


def batched_mat_inv_mul(x, a, is_left):
    """libsanm/oprs.h:66-71"""
    return x._mk(x.graph.api.lib.sanm_graph_batched_mat_inv_mul, C.c_int(x.id),
                 C.c_int(-1 if a is None else a.id), C.c_int(1 if is_left else 0))


def linear_combine(vars_, bias=0.0):
    """libsanm/oprs.h:76-77"""
    g = vars_[0][1].graph
    n = len(vars_)
    cs = (C.c_double * n)(*[float(c) for c, _ in vars_])
    vs = (C.c_int * n)(*[v.id for _, v in vars_])
    out = C.c_int()
    g.api.check(g.api.lib.sanm_graph_linear_combine(g.h, C.c_int(n), cs, vs, C.c_double(bias), C.byref(out)))
    return SymbolVar(g, out.value)


class ComputingGraph:
    def __init__(self, api, handle=None, owned=True):
        self.api = api
        self.owned = owned
        if handle is None:
            h = C.c_void_p()
            api.check(api.lib.sanm_graph_create(C.byref(h)))
            handle = h
        self.h = handle

    def __del__(self):
        if getattr(self, "owned", False) and self.h:
            self.api.lib.sanm_graph_destroy(self.h)
            self.h = None

    def placeholder(self):
        out = C.c_int()
        self.api.check(self.api.lib.sanm_graph_placeholder(self.h, C.byref(out)))
        return SymbolVar(self, out.value)

    def placeholder_vector(self, size):
        """a (batch, size) vector input (graphs over it may use slice / concat: sanm_graph_placeholder_vector)"""
        out = C.c_int()
        self.api.check(self.api.lib.sanm_graph_placeholder_vector(self.h, C.c_int(int(size)), C.byref(out)))
        v = SymbolVar(self, out.value)
        v.vec_size = int(size)
        return v

    def placeholder_matrix(self, rows, cols):
        """a (batch, rows, cols) matrix input (sanm_graph_placeholder_matrix): sizes other than 3 x 3 run on the
        vector interpreter"""
        out = C.c_int()
        self.api.check(self.api.lib.sanm_graph_placeholder_matrix(self.h, C.c_int(int(rows)), C.c_int(int(cols)),
                                                                  C.byref(out)))
        v = SymbolVar(self, out.value)
        if (int(rows), int(cols)) != (3, 3):
            v.vec_size = int(rows) * int(cols)
        return v

    def constant(self, val):
        val = _f64(val)
        batch = val.shape[0]
        out = C.c_int()
        if val.ndim == 3 and val.shape[1:] != (3, 3):
            self.api.check(self.api.lib.sanm_graph_constant_matrix(
                self.h, _dp(val), C.c_int64(batch), C.c_int(val.shape[1]), C.c_int(val.shape[2]), C.byref(out)))
            return SymbolVar(self, out.value)
        size = int(np.prod(val.shape[1:]))
        self.api.check(self.api.lib.sanm_graph_constant(self.h, _dp(val), C.c_int64(batch),
                                                        C.c_int(size), C.byref(out)))
        return SymbolVar(self, out.value)


def concat(vars_, axis):
    """concat (oprs.h; misc.cpp:321-331)"""
    vars_ = list(vars_)
    g = vars_[0].graph
    ids = (C.c_int * len(vars_))(*[v.id for v in vars_])
    out = C.c_int()
    g.api.check(g.api.lib.sanm_graph_concat(g.h, C.c_int(len(vars_)), ids, C.c_int(int(axis)), C.byref(out)))
    return SymbolVar(g, out.value)


def placeholder(cg):
    return cg.placeholder()


def constant(cg, val):
    return cg.constant(val)


class SparseLinearDesc:
    """libsanm/anm.h:24-85; built from a scipy CSR matrix (out_size, in_size)."""

    def __init__(self, api, mat=None, handle=None, owned=True):
        self.api = api
        self.owned = owned
        if handle is not None:
            self.h = handle
            return
        m = mat.tocsr()
        m.sort_indices()
        self.shape = m.shape
        rp = np.ascontiguousarray(m.indptr, dtype=np.uint64)
        ix = np.ascontiguousarray(m.indices, dtype=np.uint64)
        cf = _f64(m.data)
        h = C.c_void_p()
        api.check(api.lib.sanm_sparse_desc_create(C.c_int64(m.shape[0]), C.c_int64(m.shape[1]),
                                                  rp.ctypes.data_as(c_u64p), ix.ctypes.data_as(c_u64p),
                                                  _dp(cf), C.byref(h)))
        self.h = h

    def __del__(self):
        if getattr(self, "owned", False) and self.h:
            self.api.lib.sanm_sparse_desc_destroy(self.h)
            self.h = None

    def to_scipy(self):
        import scipy.sparse as sp
        o, i, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        self.api.check(self.api.lib.sanm_sparse_desc_get(self.h, C.byref(o), C.byref(i), C.byref(nnz),
                                                         None, None, None))
        rp = np.zeros(o.value + 1, dtype=np.uint64)
        ix = np.zeros(nnz.value, dtype=np.uint64)
        cf = np.zeros(nnz.value)
        self.api.check(self.api.lib.sanm_sparse_desc_get(self.h, None, None, None,
                                                         rp.ctypes.data_as(c_u64p),
                                                         ix.ctypes.data_as(c_u64p), _dp(cf)))
        return sp.csr_matrix((cf, ix.astype(np.int64), rp.astype(np.int64)), shape=(o.value, i.value))


class DirectSolver:
    """SparseSolver (libsanm/sparse_solver.h:17-87): analyse a pattern once,
    factor / solve many times."""

    def __init__(self, api, A, coords=None):
        self.api = api
        A = A.tocsr()
        A.sort_indices()
        self.n = A.shape[0]
        self.rp = np.ascontiguousarray(A.indptr, dtype=np.uint32)
        self.col = np.ascontiguousarray(A.indices, dtype=np.uint32)
        cd = None if coords is None else _f64(coords)
        h = C.c_void_p()
        api.check(api.lib.sanm_direct_solver_create(C.c_int64(self.n), self.rp.ctypes.data_as(c_u32p),
                                                    self.col.ctypes.data_as(c_u32p),
                                                    None if cd is None else _dp(cd), C.byref(h)))
        self.h = h

    def __del__(self):
        if getattr(self, "h", None):
            self.api.lib.sanm_direct_solver_destroy(self.h)
            self.h = None

    def dist_plan(self):
        """the tree-to-ranks plan of a solver created under SANM_MF_PLAN_WORLD (test hook, sanm_hip_test.h)"""
        out, n = np.zeros(1 << 14), C.c_int64()
        self.api.check(self.api.lib.sanm_direct_solver_dist_plan(self.h, C.c_int64(out.size), _dp(out), C.byref(n)))
        assert n.value <= out.size
        G, S = int(out[0]), int(out[1])
        p = 12
        sf = out[p:p + S * G].reshape(S, G)
        sn = out[p + S * G:p + 2 * S * G].reshape(S, G)
        ex = out[p + 2 * S * G:p + 2 * S * G + 2 * S].reshape(S, 2)
        rfl = sf[0].tolist()
        q = p + 2 * S * G + 2 * S
        nx = int(out[q])
        xf = out[q + 1:q + 1 + 5 * nx].reshape(nx, 5)
        return {"schur_transfers": [{"stage": int(r[0]), "src": int(r[1]), "dst": int(r[2]), "doubles": r[3],
                                     "src_stage": int(r[4])} for r in xf],
                "world": G, "nr_stage": S, "total_flops": out[2], "top_flops": out[3], "nr_subtree": int(out[4]),
                "schur_exchange_doubles": out[5], "inbox_exchange_doubles": out[6], "top_nnz": out[7],
                "factor_nnz": out[8], "critical_flops": out[9], "imbalance": out[10],
                "rank_flops": rfl, "stage_flops": sf.tolist(), "stage_nnz": sn.tolist(),
                "stage_schur_doubles": ex[:, 0].tolist(), "stage_schur_max_recv_doubles": ex[:, 1].tolist()}

    def factor(self, A):
        A = A.tocsr()
        A.sort_indices()
        val = _f64(A.data)
        assert val.size == self.col.size
        bad = C.c_int()
        self.api.check(self.api.lib.sanm_direct_solver_factor(self.h, _dp(val), C.byref(bad)))
        return bad.value

    def solve(self, b):
        b = _f64(b)
        x = np.zeros(self.n)
        self.api.check(self.api.lib.sanm_direct_solver_solve(self.h, _dp(b), _dp(x)))
        return x

    def apply(self, x):
        """SparseSolver::apply: A @ x with the values of the last factor()"""
        x = _f64(x)
        y = np.zeros(self.n)
        self.api.check(self.api.lib.sanm_direct_solver_apply(self.h, _dp(x), _dp(y)))
        return y

    def coeff_l2(self):
        """SparseSolver::coeff_l2: Frobenius norm of the values of the last factor()"""
        out = C.c_double()
        self.api.check(self.api.lib.sanm_direct_solver_coeff_l2(self.h, C.byref(out)))
        return out.value

    def stats(self):
        nnz, fl = C.c_int64(), C.c_double()
        nf, nl, mf, rp, nsv = (C.c_int32() for _ in range(5))
        self.api.check(self.api.lib.sanm_direct_solver_stats(self.h, C.byref(nnz), C.byref(fl), C.byref(nf),
                                                             C.byref(nl), C.byref(mf), C.byref(rp), C.byref(nsv)))
        return {"nnz_factors": nnz.value, "flops": fl.value, "nr_front": nf.value, "nr_level": nl.value,
                "max_front": mf.value, "root_pivots": rp.value, "nr_supervar": nsv.value}


class TaylorCoeffProp:
    """libsanm/symbolic.h:337-383 on the device."""

    def __init__(self, api, y: SymbolVar, remap_inp: SparseLinearDesc, max_order, nr_tet, in_size=9):
        """nr_tet: the batch; in_size: elements of the placeholder per batch item (9 for (T,3,3) graphs, the vector
        length for graphs over sanm_graph_placeholder_vector)"""
        self.api = api
        self.T = int(nr_tet)
        self.in_size = int(in_size)
        h = C.c_void_p()
        api.check(api.lib.sanm_taylor_create(y.graph.h, C.c_int(y.id), remap_inp.h, C.c_int(max_order),
                                             C.byref(h)))
        self.h = h
        self._keep = (y.graph, remap_inp)
        sz = C.c_int()
        api.check(api.lib.sanm_taylor_output_size(h, C.byref(sz)))
        self.out_size = sz.value
        self.out_shape = (3, 3) if sz.value == 9 else (sz.value,)

    def __del__(self):
        if getattr(self, "h", None):
            self.api.lib.sanm_taylor_destroy(self.h)
            self.h = None

    def push_xi(self, x):
        x = _f64(x).ravel()
        y = np.zeros((self.T,) + self.out_shape)
        self.api.check(self.api.lib.sanm_taylor_push_xi(self.h, _dp(x), _dp(y)))
        return y

    def compute_next_order_bias(self):
        b = np.zeros((self.T,) + self.out_shape)
        self.api.check(self.api.lib.sanm_taylor_compute_next_order_bias(self.h, _dp(b)))
        return b

    def get_jacobian(self):
        j = np.zeros((self.T, self.out_size, self.in_size))
        self.api.check(self.api.lib.sanm_taylor_get_jacobian(self.h, _dp(j)))
        return j

    def get_var(self, var: SymbolVar, order, size):
        out = np.zeros((self.T, size))
        self.api.check(self.api.lib.sanm_taylor_get_var(self.h, C.c_int(var.id), C.c_int(order), _dp(out)))
        return out

    def reset(self):
        self.api.check(self.api.lib.sanm_taylor_reset(self.h))


class PadeApproximation:
    """libsanm/pade.h:21-62 on its own (the ANM solvers hold one internally): xs is (nr_coeff, len) with t as the last
    entry of every coefficient"""

    def __init__(self, api, xs, anm_cond):
        self.api = api
        xs = _f64(np.stack([np.asarray(x, dtype=np.float64).ravel() for x in xs]))
        self.len = xs.shape[1]
        self.h = C.c_void_p()
        api.check(api.lib.sanm_pade_create(C.c_int(xs.shape[0]), C.c_int64(self.len), _dp(xs),
                                           C.c_int(1 if anm_cond else 0), C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None):
            self.api.lib.sanm_pade_destroy(self.h)
            self.h = None

    def estimate_valid_range(self, start, eps, limit=0.0):
        ok = C.c_int()
        self.api.check(self.api.lib.sanm_pade_estimate_valid_range(self.h, C.c_double(start), C.c_double(eps),
                                                                   C.c_double(limit), C.byref(ok)))
        return bool(ok.value)

    def _tmax(self):
        t, a = C.c_double(), C.c_double()
        self.api.check(self.api.lib.sanm_pade_get_t_max(self.h, C.byref(t), C.byref(a)))
        return t.value, a.value

    def get_t_max(self):
        return self._tmax()[0]

    def get_t_max_a(self):
        return self._tmax()[1]

    def solve_a(self, t):
        a = C.c_double()
        self.api.check(self.api.lib.sanm_pade_solve_a(self.h, C.c_double(t), C.byref(a)))
        return a.value

    def eval_xt(self, a):
        out = np.zeros(self.len)
        self.api.check(self.api.lib.sanm_pade_eval_xt(self.h, C.c_double(a), _dp(out)))
        return out


class _ANMSolver:
    def __init__(self, api):
        self.api = api
        self.h = C.c_void_p()
        self._keep = None
        self.n = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.api.lib.sanm_anm_solver_destroy(self.h)
            self.h = None

    def _get(self, fn, ctype):
        out = ctype()
        self.api.check(fn(self.h, C.byref(out)))
        return out.value

    # ANMDriverHelper (libsanm/anm.h:116-139)
    def update_approx(self):
        self.api.check(self.api.lib.sanm_anm_update_approx(self.h))

    def get_t_upper(self):
        return self._get(self.api.lib.sanm_anm_get_t_upper, C.c_double)

    def get_t_max_a(self):
        return self._get(self.api.lib.sanm_anm_get_t_max_a, C.c_double)

    def solve_a(self, t):
        out = C.c_double()
        self.api.check(self.api.lib.sanm_anm_solve_a(self.h, C.c_double(t), C.byref(out)))
        return out.value

    def eval(self, a):
        x = np.zeros(self.n)
        t = C.c_double()
        self.api.check(self.api.lib.sanm_anm_eval(self.h, C.c_double(a), _dp(x), C.byref(t)))
        return x, t.value

    def get_nr_iter(self):
        return self._get(self.api.lib.sanm_anm_nr_iter, C.c_int64)

    def xt_coeffs(self):
        nr = self._get(self.api.lib.sanm_anm_nr_xt_coeffs, C.c_int)
        out = []
        for i in range(nr):
            v = np.zeros(self.n + 1)
            self.api.check(self.api.lib.sanm_anm_xt_coeff(self.h, C.c_int(i), _dp(v)))
            out.append(v)
        return out

    def has_pade(self):
        return bool(self._get(self.api.lib.sanm_anm_has_pade, C.c_int))

    def stats(self):
        st = StatsC()
        self.api.check(self.api.lib.sanm_anm_get_stats_sized(self.h, C.byref(st), C.c_size_t(C.sizeof(st))))
        return {k: getattr(st, k) for k, _ in StatsC._fields_}

    def setup_profile(self):
        """host seconds of the constructor's phases (sanm_anm_setup_profile): {"tet_order", "program", "jit",
        "remap_tables", "pattern", "analysis", "analysis_thread", "analysis_device", "solver_vectors"} plus "jit_source" in {"embedded", "compiled", "disk_hit", "memory_hit", "none"}"""
        lib = self.api.lib
        lib.sanm_anm_setup_profile.restype = C.c_int
        n = lib.sanm_anm_setup_profile(self.h, C.c_int(0), None, None)
        names = (C.c_char_p * max(n, 1))()
        secs = (C.c_double * max(n, 1))()
        n = lib.sanm_anm_setup_profile(self.h, C.c_int(n), names, secs)
        out = {}
        for i in range(n):
            k = names[i].decode()
            if k.startswith("jit_"):
                out["jit_source"] = k[4:]
            else:
                out[k] = secs[i]
        return out

    def profile(self):
        n = self.api.lib.sanm_anm_profile(self.h, C.c_int(0), None, None)
        if n < 0:
            self.api.check(-n)
        names = (C.c_char_p * max(n, 1))()
        secs = (C.c_double * max(n, 1))()
        n = self.api.lib.sanm_anm_profile(self.h, C.c_int(n), names, secs)
        return {names[i].decode(): secs[i] for i in range(n)}

    def profile_counts(self):
        """how often each profile tag was entered (same keys as profile())"""
        keys = list(self.profile().keys())
        cnt = (C.c_double * max(len(keys), 1))()
        n = self.api.lib.sanm_anm_profile_counts(self.h, C.c_int(len(keys)), cnt)
        return {keys[i]: cnt[i] for i in range(min(n, len(keys)))}

    def profile_launches(self):
        """kernel launches queued inside each profile tag (same keys as profile())"""
        keys = list(self.profile().keys())
        cnt = (C.c_double * max(len(keys), 1))()
        n = self.api.lib.sanm_anm_profile_launches(self.h, C.c_int(len(keys)), cnt)
        return {keys[i]: cnt[i] for i in range(min(n, len(keys)))}

    def debug_inject(self, kind, order, index, value, scale=False):
        """test hook, see sanm_anm_debug_inject"""
        self.api.check(self.api.lib.sanm_anm_debug_inject(self.h, C.c_int(kind), C.c_int(order), C.c_int64(index),
                                                          C.c_double(value), C.c_int(1 if scale else 0)))

    def set_profile(self, mode, clear=True):
        """0: off, 1: host clock around synchronised phases, 2: device events (no synchronisation)"""
        self.api.check(self.api.lib.sanm_anm_set_profile(self.h, C.c_int(mode), C.c_int(1 if clear else 0)))

    def trace(self):
        n = self.api.lib.sanm_anm_trace(self.h, C.c_int(0), None, None, None)
        b, x, t = np.zeros(max(n, 1)), np.zeros(max(n, 1)), np.zeros(max(n, 1))
        n = self.api.lib.sanm_anm_trace(self.h, C.c_int(n), _dp(b), _dp(x), _dp(t))
        return {"b_norm": b[:n].tolist(), "x_norm": x[:n].tolist(), "t": t[:n].tolist()}

    def verbose_text(self):
        """the reference's SANM_VERBOSE printout of the last expansion (sanm_anm_verbose_text)"""
        self.api.lib.sanm_anm_verbose_text.restype = C.c_int64
        n = self.api.lib.sanm_anm_verbose_text(self.h, None, C.c_int64(0))
        buf = C.create_string_buffer(n + 1)
        self.api.lib.sanm_anm_verbose_text(self.h, buf, C.c_int64(n + 1))
        return buf.value.decode()

    def pade_diag(self):
        """decisions of the last Pade range estimate (sanm_anm_pade_diag)"""
        head = np.zeros(8)
        d = np.zeros(64)
        probes = np.zeros(3 * 32)
        nd = C.c_int()
        self.api.check(self.api.lib.sanm_anm_pade_diag(self.h, _dp(head), _dp(d), C.c_int(64), C.byref(nd),
                                                       _dp(probes), C.c_int(32)))
        npr = int(head[7])
        return {"attempted": bool(head[0]), "built": bool(head[1]), "roots_valid": bool(head[2]),
                "accepted": bool(head[3]), "start": head[4], "pole": head[5], "t_max_a": head[6],
                "d": d[:nd.value].copy(),
                "probes": [(probes[3 * i], probes[3 * i + 1], bool(probes[3 * i + 2])) for i in range(npr)]}

    def spec_source(self):
        """HIP source of the pass kernels specialised for this solver's graph (sanm_anm_spec_source)."""
        self.api.lib.sanm_anm_spec_source.restype = C.c_int64
        n = self.api.lib.sanm_anm_spec_source(self.h, None, C.c_int64(0))
        buf = C.create_string_buffer(n + 1)
        self.api.lib.sanm_anm_spec_source(self.h, buf, C.c_int64(n + 1))
        return buf.value.decode()

    def time_kernel(self, kernel, reps, mode=2, order=1):
        out = C.c_double()
        self.api.check(self.api.lib.sanm_anm_time_kernel(self.h, C.c_int(kernel), C.c_int(reps), C.c_int(mode),
                                                         C.c_int(order), C.byref(out)))
        return out.value

    def pass_timing(self, enable, fetch=True):
        tot, cnt = C.c_double(), C.c_int64()
        self.api.check(self.api.lib.sanm_anm_pass_timing(self.h, C.c_int(1 if enable else 0),
                                                         C.byref(tot) if fetch else None,
                                                         C.byref(cnt) if fetch else None))
        return tot.value, cnt.value

    def jacobian_csr(self):
        import scipy.sparse as sp
        n, nnz = C.c_int64(), C.c_int64()
        self.api.check(self.api.lib.sanm_anm_jacobian_csr(self.h, C.byref(n), C.byref(nnz), None, None, None))
        rp = np.zeros(n.value + 1, dtype=np.uint32)
        col = np.zeros(nnz.value, dtype=np.uint32)
        val = np.zeros(nnz.value)
        self.api.check(self.api.lib.sanm_anm_jacobian_csr(self.h, None, None, rp.ctypes.data_as(c_u32p),
                                                          col.ctypes.data_as(c_u32p), _dp(val)))
        return sp.csr_matrix((val, col.astype(np.int64), rp.astype(np.int64)), shape=(n.value, n.value))


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)


class ANMEqnSolver(_ANMSolver):
    """libsanm/anm.h:245-283.  ``shard=(rank, world, allreduce)`` runs the Taylor
    passes and the assembly on this rank's tet range; ``allreduce(ptr, count)`` must
    sum ``count`` doubles in place at device pointer ``ptr`` over all ranks
    (see sanm_amd/dist.py)."""

    def __init__(self, api, y: SymbolVar, remap_inp, remap_out, x0, f_y, hyper, shard=None):
        super().__init__(api)
        x0, f_y = _f64(x0).ravel(), _f64(f_y).ravel()
        self.n = x0.size
        self._keep = (y.graph, remap_inp, remap_out)
        if shard is None:
            api.check(api.lib.sanm_anm_eqn_solver_create(y.graph.h, C.c_int(y.id), remap_inp.h, remap_out.h,
                                                         _dp(x0), _dp(f_y), C.c_int64(self.n),
                                                         C.byref(hyper), C.byref(self.h)))
            return
        rank, world, fn = shard
        if fn is None:  # the library's own RCCL communicator (Api.comm_init), all-reduce on the solver's stream
            self._cb = None
            api.check(api.lib.sanm_anm_eqn_solver_create_sharded(
                y.graph.h, C.c_int(y.id), remap_inp.h, remap_out.h, _dp(x0), _dp(f_y), C.c_int64(self.n),
                C.byref(hyper), C.c_int(rank), C.c_int(world), C.cast(None, ALLREDUCE_FN), None, C.byref(self.h)))
            return

        def _cb(user, ptr, count):
            try:
                fn(ptr, count)
                return 0
            except Exception as e:  # noqa: BLE001 - reported through the C error path
                import traceback
                traceback.print_exc()
                return 1

        self._cb = ALLREDUCE_FN(_cb)  # keep the trampoline alive
        api.check(api.lib.sanm_anm_eqn_solver_create_sharded(
            y.graph.h, C.c_int(y.id), remap_inp.h, remap_out.h, _dp(x0), _dp(f_y), C.c_int64(self.n),
            C.byref(hyper), C.c_int(rank), C.c_int(world), self._cb, None, C.byref(self.h)))

    def next_iter(self):
        self.api.check(self.api.lib.sanm_anm_next_iter(self.h))
        return self

    def converged(self):
        return bool(self._get(self.api.lib.sanm_anm_converged, C.c_int))

    def residual_rms(self):
        return self._get(self.api.lib.sanm_anm_residual_rms, C.c_double)

    def run_steps(self, count, x0):
        """exactly `count` more completed steps (restarting from x0 when the solve converges); returns restarts"""
        x0 = _f64(x0).ravel()
        r = C.c_int()
        self.api.check(self.api.lib.sanm_anm_run_steps(self.h, C.c_int(count), _dp(x0), C.byref(r)))
        return r.value

    def restart(self, x0):
        x0 = _f64(x0).ravel()
        self.api.check(self.api.lib.sanm_anm_restart(self.h, _dp(x0)))
        return self

    def get_x(self):
        x = np.zeros(self.n)
        self.api.check(self.api.lib.sanm_anm_get_x(self.h, _dp(x)))
        return x


class ANMSolverVecScale(_ANMSolver):
    """libsanm/anm.h:209-243"""

    def __init__(self, api, y, remap_inp, remap_out, x0, t0, v, hyper):
        super().__init__(api)
        x0, v = _f64(x0).ravel(), _f64(v).ravel()
        self.n = x0.size
        self._keep = (y.graph, remap_inp, remap_out)
        api.check(api.lib.sanm_anm_vecscale_solver_create(
            y.graph.h, C.c_int(y.id), remap_inp.h, remap_out.h, _dp(x0), C.c_double(t0), _dp(v),
            C.c_int64(self.n), C.byref(hyper), C.byref(self.h)))


class ANMImplicitSolver(_ANMSolver):
    """libsanm/anm.h:285-305"""

    def __init__(self, api, y, remap_inp, remap_out, x0, t0, hyper):
        super().__init__(api)
        x0 = _f64(x0).ravel()
        self.n = x0.size
        self._keep = (y.graph, remap_inp, remap_out)
        api.check(api.lib.sanm_anm_implicit_solver_create(
            y.graph.h, C.c_int(y.id), remap_inp.h, remap_out.h, _dp(x0), C.c_double(t0),
            C.c_int64(self.n), C.byref(hyper), C.byref(self.h)))


class FeaModel:
    """ElasticForceModel (fea/mesh.h:149-226) built by make_forward / make_inverse."""

    def __init__(self, api, vertices, tets, fixed_mask, energy, young, poisson, inverse,
                 init_vtx_coord, vtx_delta):
        self.api = api
        V = _f64(vertices)
        t = np.ascontiguousarray(tets, dtype=np.int32)
        fm = np.ascontiguousarray(fixed_mask, dtype=np.uint8)
        assert V.shape[1] == 3 and t.shape[1] == 4 and fm.shape == V.shape
        iv = None if init_vtx_coord is None else _f64(init_vtx_coord)
        vd = None if vtx_delta is None else _f64(vtx_delta)
        h = C.c_void_p()
        api.check(api.lib.sanm_fea_model_create(
            C.c_int64(V.shape[0]), _dp(V), C.c_int64(t.shape[0]), t.ctypes.data_as(c_i32p),
            fm.ctypes.data_as(c_u8p), C.c_int(ENERGY[energy]), C.c_double(young), C.c_double(poisson),
            C.c_int(1 if inverse else 0), None if iv is None else _dp(iv),
            None if vd is None else _dp(vd), C.byref(h)))
        self.h = h
        self.nv, self.T = V.shape[0], t.shape[0]
        n = C.c_int64()
        api.check(api.lib.sanm_fea_model_nr_unknown(h, C.byref(n)))
        self.n = n.value
        self.cg = ComputingGraph(api, C.c_void_p(api.lib.sanm_fea_model_graph(h)), owned=False)
        self.y = SymbolVar(self.cg, api.lib.sanm_fea_model_output_var(h))
        self.F = SymbolVar(self.cg, api.lib.sanm_fea_model_F_var(h))
        self.lt_inp = SparseLinearDesc(api, handle=C.c_void_p(api.lib.sanm_fea_model_remap_inp(h)), owned=False)
        self.lt_out = SparseLinearDesc(api, handle=C.c_void_p(api.lib.sanm_fea_model_remap_out(h)), owned=False)

    def __del__(self):
        if getattr(self, "h", None):
            self.api.lib.sanm_fea_model_destroy(self.h)
            self.h = None

    def x0(self):
        x = np.zeros(self.n)
        self.api.check(self.api.lib.sanm_fea_model_x0(self.h, _dp(x)))
        return x

    def copy_vtx_values(self, vtx_values):
        v = _f64(vtx_values)
        out = np.zeros(self.n)
        self.api.check(self.api.lib.sanm_fea_model_copy_vtx_values(self.h, _dp(v), _dp(out)))
        return out

    def full_vertices(self, x, base):
        out = _f64(base).copy()
        x = _f64(x)
        self.api.check(self.api.lib.sanm_fea_model_scatter(self.h, _dp(x), _dp(out)))
        return out
