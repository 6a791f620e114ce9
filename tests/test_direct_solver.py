"""Multifrontal LU (SparseSolver role, libsanm/sparse_solver.cpp) against
scipy's sparse LU: the reference's own check is Ax = b on random systems
(tests/tensor.cpp:44-85)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import fea as ofea
from oracle import symbolic as S
from oracle.anm import build_jacobian_csr
from sanm_amd.api import DirectSolver


def _check(api, A, coords=None, tol=1e-9, nrhs=3, seed=0):
    rng = np.random.default_rng(seed)
    A = A.tocsr()
    A.sort_indices()
    ds = DirectSolver(api, A, coords)
    assert ds.factor(A) == 0
    lu = spla.splu(A.tocsc())
    for _ in range(nrhs):
        b = rng.standard_normal(A.shape[0])
        x = ds.solve(b)
        xr = lu.solve(b)
        assert np.abs(x - xr).max() <= tol * np.abs(xr).max()
    # SparseSolver::apply / coeff_l2 (sparse_solver.cpp:202-223)
    v = rng.standard_normal(A.shape[0])
    assert np.abs(ds.apply(v) - A @ v).max() <= 1e-12 * max(1.0, np.abs(A @ v).max())
    assert ds.coeff_l2() == pytest.approx(np.sqrt((A.data ** 2).sum()), rel=1e-12)
    # refactor with new values on the same pattern (one analysis, many steps)
    A2 = A.copy()
    A2.data = A.data * (1 + 0.1 * rng.standard_normal(A.nnz))
    A2 = A2 + sp.diags(np.abs(A2).sum(axis=1).A1) * np.sign(A.diagonal()[0])
    A2 = sp.csr_matrix(A2)
    A2.sort_indices()
    if A2.nnz == A.nnz:
        assert ds.factor(A2) == 0
        b = rng.standard_normal(A.shape[0])
        assert np.abs(ds.solve(b) - spla.spsolve(A2.tocsc(), b)).max() <= tol * 10
    return ds


def test_random_block_unsymmetric(api):
    rng = np.random.default_rng(1)
    nb = 70
    B = sp.random(nb, nb, density=0.07, random_state=2)
    B = ((B + B.T) != 0).astype(float) + sp.identity(nb)
    A = sp.kron(B, np.ones((3, 3))).tocsr()
    A.data = rng.standard_normal(A.nnz)
    A = A + sp.diags(np.full(3 * nb, 45.0))
    ds = _check(api, A)
    st = ds.stats()
    assert st["nr_supervar"] <= nb and st["nr_front"] >= 1


def test_scalar_pattern_no_blocks(api):
    # 2-D 5-point Laplacian: supervariables are single unknowns, several levels
    k = 23
    T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(k, k))
    A = sp.kron(sp.identity(k), T) + sp.kron(T, sp.identity(k)) + 0.1 * sp.identity(k * k)
    ds = _check(api, sp.csr_matrix(A))
    assert ds.stats()["nr_level"] >= 3


def test_tiny_and_diagonal(api):
    _check(api, sp.csr_matrix(np.array([[4.0, 1.0], [2.0, 3.0]])))
    _check(api, sp.csr_matrix(sp.diags(np.arange(1.0, 40.0))))
    # disconnected components -> forest
    A = sp.block_diag([sp.csr_matrix(np.array([[3.0, 1], [1, 2]])), sp.diags([5.0, 6.0]),
                       sp.csr_matrix(np.array([[2.0, -1, 0], [-1, 2, -1], [0, -1, 2]]))])
    _check(api, sp.csr_matrix(A))


@pytest.mark.parametrize("with_coords", [True, False])
def test_fem_jacobian(api, with_coords):
    """the actual ANM Jacobian of a cuboid (negative definite), with and
    without the coordinate hint (principal-axis vs graph-distance dissection)"""
    mesh = ofea.make_cuboid(9, 5, 4, 0.02)
    fixed = np.zeros((mesh.nr_vertices, 3), bool)
    fixed[mesh.V[:, 0] < 0.01] = True
    om = ofea.make_forward(mesh, ofea.Material(1e4, 0.45), fixed, "neohookean_c")
    prop = S.TaylorCoeffProp(om.y)
    prop.push_xi([(om.lt_inp.mat @ om.lt_inp.x0).reshape(-1, 3, 3)])
    A, _ = build_jacobian_csr(om.lt_out, prop.get_jacobian(), om.lt_inp.mat, om.lt_inp.n)
    coords = mesh.V[om.lt_inp.vertex_loc[:, 0]] if with_coords else None
    ds = _check(api, A, coords, tol=1e-8)
    st = ds.stats()
    assert st["nr_supervar"] <= A.shape[0]
    assert st["nr_level"] >= 3 and st["max_front"] < A.shape[0]


@pytest.mark.parametrize("case", ["fem", "grid3d"])
def test_merged_top_block(api, monkeypatch, case):
    """SANM_MF_TOP: the root and the level below it multiplied out into one dense operator after the factorisation
    (mf_kernels.h, top_gemm_kernel / top_solve_kernel; device back end only -- the host harness ignores the
    switch): same solutions, several right-hand sides in a row (the block's solution must not land on the
    right-hand side other workgroups are still reading), with and without index maps in its products."""
    monkeypatch.setenv("SANM_MF_TOP", "2048")
    if case == "fem":
        test_fem_jacobian(api, True)
        test_fem_jacobian(api, False)
    else:
        test_grid3d_wide_separators(api)
        test_scalar_pattern_no_blocks(api)


def test_singular_matrix_reports_bad_pivot(api):
    A = sp.csr_matrix(np.array([[1.0, 2.0, 0], [2.0, 4.0, 0], [0, 0, 1.0]]))
    ds = DirectSolver(api, A)
    assert ds.factor(A) >= 1


def _grid3d_wide_separators(api):
    """3-D 7-point stencil with unsymmetric values and the coordinate hint: the top
    separators have 200+ pivots, i.e. many 32-wide panels per front, partial last
    panels and workgroups that read panel tiles other workgroups of the same launch
    are busy with (a data race there only shows on fronts this wide)."""
    k = 15
    rng = np.random.default_rng(7)
    I, T = sp.identity(k), sp.diags([-1.0, -1.0], [-1, 1], shape=(k, k))
    P = sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T)
    P = sp.csr_matrix(P)
    A = P.copy()
    A.data = P.data * (1 + 0.3 * rng.standard_normal(P.nnz))
    A = sp.csr_matrix(A + sp.diags(np.full(k ** 3, 7.0)))
    g = np.arange(k, dtype=float)
    coords = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    ds = _check(api, A, coords=coords, nrhs=2)
    st = ds.stats()
    assert st["max_front"] >= 200 and st["nr_level"] >= 4
    return st


def test_grid3d_wide_separators(api):
    _grid3d_wide_separators(api)


@pytest.mark.parametrize("min_k", ["32", "96"])
def test_two_level_blocking_forced(api, monkeypatch, min_k):
    """fronts of 1000+ pivots are factored with two blocking levels (outer blocks of 4 panels, the rest of
    the trailing matrix updated by one MFMA GEMM per outer block); SANM_MF_OUTER_MIN_K forces that path on
    the fronts of the small test systems, partial panels and fronts narrower than an outer block included.
    (The host test harness has its own serial factorisation and ignores the switch.)"""
    monkeypatch.setenv("SANM_MF_OUTER_MIN_K", min_k)
    test_grid3d_wide_separators(api)
    test_random_block_unsymmetric(api)
    test_scalar_pattern_no_blocks(api)
    test_tiny_and_diagonal(api)


def test_solve_kernels_for_vectors_beyond_lds(api, monkeypatch):
    """fronts whose solve vectors do not fit the LDS (20,000+ rows, million-tet meshes) take plain mat-vec level
    kernels; SANM_MF_LDS_MAX forces them on the small test systems.  (Ignored by the host test harness.)"""
    monkeypatch.setenv("SANM_MF_LDS_MAX", "64")
    test_grid3d_wide_separators(api)
    test_random_block_unsymmetric(api)
    test_tiny_and_diagonal(api)


@pytest.mark.parametrize("width", ["24", "48", "96"])
def test_big_fronts_cut_into_chains_forced(api, monkeypatch, width):
    """fronts of more than 1536 pivots are cut into chains of ~1000-pivot fronts (multifrontal.cpp, split_big_fronts:
    the explicit inverses of a k-pivot front cost 2 k^3 + 2 k^2 b on top of its LU); SANM_MF_SPLIT_K forces the cut
    on the fronts of the small test systems.  Same elimination order, no more factor entries (fewer where a chunk does not touch all of the pivots after it), fewer
    flops, more levels;
    together with the backward kernel for long rows (bwd_wide_kernel, from 4096 rows on; SANM_MF_WIDE_MIN_M) at each
    of its rows-per-workgroup settings.  (The host harness takes the chains, and ignores the kernel switches.)"""
    k = 14
    T = sp.diags([-1, 2.2, -1], [-1, 0, 1], shape=(k, k))
    I = sp.identity(k)
    A = sp.csr_matrix(sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T))
    A = sp.csr_matrix(sp.kron(A, np.array([[1.0, 0.2, 0], [0.1, 1.0, 0.3], [0, 0.2, 1.0]])))
    monkeypatch.setenv("SANM_MF_SPLIT_K", "0")
    plain = _check(api, A).stats()
    monkeypatch.setenv("SANM_MF_SPLIT_K", width)
    for rows in ["1", "2", "4"]:
        monkeypatch.setenv("SANM_MF_WIDE_MIN_M", "0")
        monkeypatch.setenv("SANM_MF_WIDE_R", rows)
        cut = _check(api, A).stats()
    assert cut["nnz_factors"] <= plain["nnz_factors"] and cut["nr_supervar"] == plain["nr_supervar"]
    assert cut["nr_level"] > plain["nr_level"] and cut["nr_front"] > plain["nr_front"]
    assert cut["flops"] < plain["flops"]
    assert cut["max_front"] <= plain["max_front"]
    test_random_block_unsymmetric(api)
    test_tiny_and_diagonal(api)
    test_fem_jacobian(api, True)


@pytest.mark.parametrize("variant", ["staged", "wide", "beyond_lds", "chains"])
def test_two_phase_levels_forced(api, monkeypatch, variant):
    """heights of the tree whose boundary-operator products (2 k^2 b flops per front) outweigh two more launches per
    sweep keep -L21 / -U12 in the F[B,A] / F[A,B] slots and are solved in two dependent launches per direction
    (mf_types.h, Level::two_phase; the top of a 32^3 block and beyond).  SANM_MF_TWO_PHASE=1 forces every height
    with a boundary, here through each family of level kernels: LDS-staged, the wide backward kernel, the kernels for
    vectors beyond the LDS, and over chains of cut fronts.  (The host harness has its own solve: it only sees the
    flop count change.)"""
    monkeypatch.setenv("SANM_MF_TWO_PHASE", "0")
    plain = _grid3d_wide_separators(api)
    monkeypatch.setenv("SANM_MF_TWO_PHASE", "1")
    if variant == "wide":
        monkeypatch.setenv("SANM_MF_WIDE_MIN_M", "0")
    elif variant == "beyond_lds":
        monkeypatch.setenv("SANM_MF_LDS_MAX", "64")
    elif variant == "chains":
        monkeypatch.setenv("SANM_MF_SPLIT_K", "48")
    forced = _grid3d_wide_separators(api)
    if variant != "chains":
        assert forced["flops"] < plain["flops"] and forced["nnz_factors"] == plain["nnz_factors"]
    test_random_block_unsymmetric(api)
    test_scalar_pattern_no_blocks(api)
    test_tiny_and_diagonal(api)
    test_fem_jacobian(api, True)


@pytest.mark.parametrize("seed", range(4))
def test_reference_sparse_solver_case(api, seed):
    """Tensor.SparseSolver (tests/tensor.cpp:44-70): a random 8 x 8 system in [-1, 1] with one structural zero per
    row, A x = b to Catch2's Approx.  (Its second half -- prepare(alpha), the regularised normal equations -- is
    the solver of the Tikhonov path: tests/test_tikhonov.py.)"""
    rng = np.random.default_rng(seed)
    N = 8
    A = rng.uniform(-1, 1, (N, N))
    b = rng.uniform(-1, 1, N)
    for i in range(N):
        A[i, (i + 2) % N] = 0
    S = sp.csr_matrix(A)
    S.sort_indices()
    ds = DirectSolver(api, S)
    ds.factor(S)
    x = ds.solve(b)
    assert np.allclose(A @ x, b, rtol=1.2e-5, atol=1e-12)
    assert np.abs(x - np.linalg.solve(A, b)).max() <= 1e-9 * np.abs(x).max()


@pytest.mark.parametrize("leaf", ["8", "20", "32"])
def test_small_fronts_in_one_workgroup_forced(api, monkeypatch, leaf):
    """levels of a thousand or more fronts of at most 96 pivots are factored by small_front_kernel (mf_kernels.h, round
    5: pivot block in LDS, unblocked LU, L and U inverted in place, the boundary products as a tile loop in the same
    workgroup) instead of the panel chain and the GEMM passes.  SANM_MF_SMALL_MIN_FRONTS=1 forces it on every level of
    the small test systems whose fronts fit; the leaf size varies the pivot counts (1 ... 96: partial 32-panels, an odd /
    even LDS stride, fronts without a boundary, single-pivot fronts) and SANM_MF_MERGE=none keeps the lower levels
    small.  Perturbed pivots must still be counted.  (The host harness ignores the switch.)"""
    monkeypatch.setenv("SANM_MF_SMALL_MIN_FRONTS", "1")
    monkeypatch.setenv("SANM_MF_LEAF", leaf)
    monkeypatch.setenv("SANM_MF_MERGE", "none")
    test_random_block_unsymmetric(api)
    test_scalar_pattern_no_blocks(api)
    test_tiny_and_diagonal(api)
    test_fem_jacobian(api, True)
    test_fem_jacobian(api, False)
    test_grid3d_wide_separators(api)
    for seed in range(4):
        test_reference_sparse_solver_case(api, seed)
    # a singular pivot block in a leaf: the perturbation is counted and the refined solve still works
    A = sp.block_diag([sp.csr_matrix(np.array([[1e-30, 1.0], [1.0, 1.0]])), sp.diags([2.0, 3.0, 4.0])]).tocsr()
    A.sort_indices()
    ds = DirectSolver(api, A)
    assert ds.factor(A) >= 1


def test_small_front_size_classes_beside_each_other(api, monkeypatch):
    """the size classes of small_front_kernel on a level go to queues of their own (backend_hip.hip, mf_factor_levels):
    the fronts are independent, so the solution is the one of the classes queued one after the other
    (SANM_MF_SMALL_SERIAL=1), bit for bit.  (The host harness ignores both switches.)"""
    monkeypatch.setenv("SANM_MF_SMALL_MIN_FRONTS", "1")
    monkeypatch.setenv("SANM_MF_MERGE", "none")
    if True:
        n = 14
        idx = np.arange(n ** 3).reshape(n, n, n)
        rows, cols = [idx.ravel()], [idx.ravel()]
        for ax in range(3):
            a = np.take(idx, np.arange(n - 1), axis=ax).ravel()
            b = np.take(idx, np.arange(1, n), axis=ax).ravel()
            rows += [a, b]
            cols += [b, a]
        rows, cols = np.concatenate(rows), np.concatenate(cols)
        rng = np.random.default_rng(11)
        A = sp.csr_matrix((rng.standard_normal(rows.size) * 0.3, (rows, cols)), shape=(n ** 3, n ** 3))
        A = sp.csr_matrix(A + sp.diags(np.full(n ** 3, 8.0)))
        A.sort_indices()
    g = np.stack(np.meshgrid(*[np.arange(14.0)] * 3, indexing="ij"), -1).reshape(-1, 3)
    b = np.random.default_rng(12).standard_normal(A.shape[0])
    out = []
    for serial in (None, "1"):
        if serial:
            monkeypatch.setenv("SANM_MF_SMALL_SERIAL", serial)
        ds = DirectSolver(api, A, g)
        assert ds.factor(A) == 0
        out.append(ds.solve(b))
    assert np.array_equal(out[0], out[1])
    assert np.abs(A @ out[0] - b).max() < 1e-9


def test_parallel_analysis_gives_the_sequential_ordering(api, monkeypatch):
    """the nested dissection hands the subtrees at the top of its tree to host threads and numbers the nodes as the
    sequential loop does (multifrontal.cpp, NestedDissection::dissect; the row loops of the supervariable graph and of
    the scatter map run on threads too): same tree, same elimination order, hence the same factorisation -- the
    statistics of the analysis and the bits of a solve with 1 and with 8 threads."""
    mesh = ofea.make_cuboid(32, 20, 16, 0.02)
    nv = mesh.nr_vertices
    T = mesh.tets
    rows = np.repeat(T, 4, axis=1).ravel()
    cols = np.tile(T, (1, 4)).ravel()
    G = sp.csr_matrix((np.ones(rows.size), (rows, cols)), shape=(nv, nv))
    G.sum_duplicates()
    G.data[:] = 1.0
    rng = np.random.default_rng(5)
    A = sp.kron(G, np.ones((3, 3)), format="csr")
    A.data = rng.standard_normal(A.nnz) * 0.1
    A = sp.csr_matrix(A + sp.diags(np.full(3 * nv, 40.0)))
    A.sort_indices()
    coords = np.repeat(mesh.V, 3, axis=0)
    b = rng.standard_normal(A.shape[0])
    out = []
    for threads in ("1", "8"):
        monkeypatch.setenv("SANM_MF_ND_THREADS", threads)
        ds = DirectSolver(api, A, coords)
        assert ds.factor(A) == 0
        out.append((ds.stats(), ds.solve(b)))
    assert out[0][0] == out[1][0] and out[0][0]["nr_supervar"] == nv >= 8192
    assert np.array_equal(out[0][1], out[1][1])
    assert np.abs(A @ out[0][1] - b).max() < 1e-9


def test_supervariables_from_runs_of_equal_rows(api, monkeypatch):
    """the supervariables come from runs of consecutive rows with one column list (the 3 x 3 blocks of a mesh Jacobian),
    the pattern of the runs then goes through the hashing of closed neighbourhoods (multifrontal.cpp, build_sv_graph;
    round 6).  On a symmetric pattern that is the partition, the numbering and the ordering the hashing of the whole
    pattern gives (SANM_MF_SV_RUNS=0) -- same statistics, same bits of a solve; SANM_MF_DEBUG makes the analysis check
    its parent positions entry by entry against a search of their own and look up the place of every entry of A.  In an unsymmetric pattern
    rows of a run may differ in their COLUMNS: the run is then a supervariable with explicit zeros, and the system is
    still solved."""
    monkeypatch.setenv("SANM_MF_DEBUG", "1")
    mesh = ofea.make_cuboid(14, 9, 7, 0.05)
    nv = mesh.nr_vertices
    T = mesh.tets
    rows = np.repeat(T, 4, axis=1).ravel()
    cols = np.tile(T, (1, 4)).ravel()
    G = sp.csr_matrix((np.ones(rows.size), (rows, cols)), shape=(nv, nv))
    G.sum_duplicates()
    rng = np.random.default_rng(11)
    A = sp.kron(G, np.ones((3, 3)), format="csr")
    A.data = rng.standard_normal(A.nnz) * 0.1
    A = sp.csr_matrix(A + sp.diags(np.full(3 * nv, 30.0)))
    A.sort_indices()
    coords = np.repeat(mesh.V, 3, axis=0)
    b = rng.standard_normal(A.shape[0])
    out = []
    for runs in ("0", "1"):
        monkeypatch.setenv("SANM_MF_SV_RUNS", runs)
        for c in (coords, None):
            ds = DirectSolver(api, A, c)
            assert ds.factor(A) == 0
            out.append((ds.stats(), ds.solve(b)))
    for i in (0, 1):
        assert out[i][0] == out[2 + i][0] and out[i][0]["nr_supervar"] == nv
        assert np.array_equal(out[i][1], out[2 + i][1])
        assert np.abs(A @ out[i][1] - b).max() < 1e-9
    # unsymmetric: pairs of rows with one column list, columns that differ inside a pair
    n = 600
    P = sp.random(n // 2, n, density=0.02, random_state=5, format="csr")
    P = sp.vstack([P[i // 2] for i in range(n)]).tocsr()  # row 2i+1 lists the columns of row 2i
    P = (P + sp.kron(sp.eye(n // 2), np.ones((2, 2)))).tocsr()  # both diagonals in both rows
    P.data = rng.standard_normal(P.nnz) * 0.05
    P = sp.csr_matrix(P + sp.diags(np.full(n, 4.0)))
    P.sort_indices()
    assert (P != P.T).nnz > 0 and np.array_equal(P[0].indices, P[1].indices)
    bu = rng.standard_normal(n)
    xs = []
    for runs in ("0", "1"):
        monkeypatch.setenv("SANM_MF_SV_RUNS", runs)
        ds = DirectSolver(api, P, None)
        assert ds.factor(P) == 0
        xs.append((ds.stats()["nr_supervar"], ds.solve(bu)))
        assert np.abs(P @ xs[-1][1] - bu).max() < 1e-10
    assert xs[1][0] <= n // 2 < xs[0][0]  # (the hashing of A + A' tells the rows of a pair apart)


@pytest.mark.parametrize("leaf", ["8", "32"])
def test_forward_operator_not_transposed(api, monkeypatch, leaf):
    """levels of fronts of at most 128 pivots keep the boundary block of their forward operator TRANSPOSED in the dead
    F[P,B] slot and sweep it with a thread per boundary row (mf_types.h, Level::fwd_t; round 5) -- the default, so every
    other test of this file runs it; SANM_MF_FWD_T=0 is the row-wise form of rounds 1-4 on the same systems, with the
    small-front kernel forced on and off (it writes the same slot).  (The host harness has its own solve.)"""
    monkeypatch.setenv("SANM_MF_FWD_T", "0")
    monkeypatch.setenv("SANM_MF_LEAF", leaf)
    for small in ("1", "0"):
        monkeypatch.setenv("SANM_MF_SMALL_MIN_FRONTS", small)
        test_random_block_unsymmetric(api)
        test_scalar_pattern_no_blocks(api)
        test_fem_jacobian(api, True)
        test_grid3d_wide_separators(api)
    monkeypatch.setenv("SANM_MF_FWD_T", "1")
    monkeypatch.setenv("SANM_MF_SMALL_MIN_FRONTS", "1")
    test_fem_jacobian(api, True)
    test_grid3d_wide_separators(api)


@pytest.mark.parametrize("leaf", ["8", "32"])
def test_selective_zero_fill_and_assigned_schur_blocks(api, monkeypatch, leaf):
    """Round 6: the factorisation's prologue zeroes only what is accumulated into (zero_kernel: rows P, the P / A columns
    of rows A, the P columns of rows B -- not F[A,B], F[B,A], F[B,B]); the F[B,B] block of a front with children is
    ASSIGNED by round 0 of the extend-add (schur_gather_kernel), a front without children writes its Schur complement
    without reading the block.  Same bits as the memset of the whole storage (SANM_MF_FULL_ZERO=1), also when the
    storage is filled with NaNs first (SANM_MF_POISON=1: nothing unwritten is read), with the small-front kernel, the
    panel chain, two-phase levels and chains of cut fronts, over two factorisations on the same storage.  (The host
    harness zeroes everything and adds: its own results are the reference the other tests hold the device to.)"""
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
    rng = np.random.default_rng(3)
    b = rng.standard_normal(A.shape[0])
    A2 = A.copy()
    A2.data = A.data * (1 + 0.05 * rng.standard_normal(A.nnz))
    monkeypatch.setenv("SANM_MF_LEAF", leaf)
    for extra in ({}, {"SANM_MF_SMALL_MIN_FRONTS": "0"}, {"SANM_MF_TWO_PHASE": "1"}, {"SANM_MF_SPLIT_K": "48"},
                  {"SANM_MF_OUTER_MIN_K": "32"}):
        for k, v in extra.items():
            monkeypatch.setenv(k, v)
        sols = []
        # (the selective path is the default from 16 GB of front storage on; forced here)
        for mode in ({"SANM_MF_FULL_ZERO": "1"}, {"SANM_MF_SELECTIVE_ZERO": "1"}, {"SANM_MF_POISON": "1"},
                     {"SANM_MF_SELECTIVE_ZERO": "1", "SANM_MF_EA_ROWS": "16"}):
            for k, v in mode.items():
                monkeypatch.setenv(k, v)
            ds = DirectSolver(api, A, coords)
            assert ds.factor(A) == 0
            x1 = ds.solve(b)
            assert ds.factor(A2) == 0  # (a second factorisation on the storage the first one left behind)
            x2 = ds.solve(b)
            sols.append((x1, x2))
            for k in mode:
                monkeypatch.delenv(k)
        assert all(np.all(np.isfinite(x)) for s in sols for x in s)
        for s in sols[1:]:
            assert np.array_equal(s[0], sols[0][0]) and np.array_equal(s[1], sols[0][1])
        assert np.abs(A @ sols[0][0] - b).max() <= 1e-8 * np.abs(b).max()
        for k in extra:
            monkeypatch.delenv(k)


def test_solve_launch_forms_give_the_same_bits(api, monkeypatch):
    """Round 6: the sweeps' launches go out over flat lists of the (front, row block) pairs that exist (levels of 64+
    fronts; MfSolveBlock) and preload the row chunks of the level's TYPICAL row, the rest of a longer row in the tail
    loop.  Neither changes a sum: the box grids (SANM_MF_SOLVE_LISTS=0), lists forced onto every level (=1) and chunks
    for the longest row (SANM_MF_LS_WIDTH=0) or far too few (=0.3) give the same solutions bit for bit, with the
    transposed forward operator and without, one-phase and two-phase levels, several right-hand sides in a row."""
    mesh = ofea.make_cuboid(12, 6, 5, 0.02)
    fixed = np.zeros((mesh.nr_vertices, 3), bool)
    fixed[mesh.V[:, 0] < 0.01] = True
    om = ofea.make_forward(mesh, ofea.Material(1e4, 0.45), fixed, "neohookean_c")
    prop = S.TaylorCoeffProp(om.y)
    prop.push_xi([(om.lt_inp.mat @ om.lt_inp.x0).reshape(-1, 3, 3)])
    A, _ = build_jacobian_csr(om.lt_out, prop.get_jacobian(), om.lt_inp.mat, om.lt_inp.n)
    A = A.tocsr()
    A.sort_indices()
    coords = mesh.V[om.lt_inp.vertex_loc[:, 0]]
    rng = np.random.default_rng(5)
    bs = [rng.standard_normal(A.shape[0]) for _ in range(3)]
    monkeypatch.setenv("SANM_MF_LEAF", "8")  # (many small fronts: levels of 64+ fronts exist on this small mesh)
    for extra in ({}, {"SANM_MF_FWD_T": "0"}, {"SANM_MF_TWO_PHASE": "1"}):
        for k, v in extra.items():
            monkeypatch.setenv(k, v)
        ref = None
        for mode in ({"SANM_MF_SOLVE_LISTS": "0", "SANM_MF_LS_WIDTH": "0"}, {"SANM_MF_SOLVE_LISTS": "0"}, {},
                     {"SANM_MF_SOLVE_LISTS": "1"}, {"SANM_MF_SOLVE_LISTS": "1", "SANM_MF_LS_WIDTH": "0.3"}):
            for k, v in mode.items():
                monkeypatch.setenv(k, v)
            ds = DirectSolver(api, A, coords)
            assert ds.factor(A) == 0
            xs = [ds.solve(b) for b in bs]
            del ds
            for k in mode:
                monkeypatch.delenv(k)
            if ref is None:
                ref = xs
                assert max(np.abs(A @ x - b).max() for x, b in zip(xs, bs)) <= 1e-8
            else:
                assert all(np.array_equal(x, r) for x, r in zip(xs, ref)), mode
        for k in extra:
            monkeypatch.delenv(k)
