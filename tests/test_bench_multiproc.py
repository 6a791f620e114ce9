"""bench.py's N > 1 path (one process per GPU, barrier, max-over-ranks time,
whole-job value) exercised with world_size 2 on CPU: gloo backend, the
test-only host harness in place of the HIP library, a tiny cuboid workload."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import bench
from tests.hostsim import get_hostsim_api
bench.make_api = lambda local_rank: get_hostsim_api()
bench.device_sync = lambda: None
out = bench.main(["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "cuboid:5,3,3",
                  "--no-cpu-baseline", "--dist-backend", "gloo", "--at-scale-workload", "cuboid:6,3,3",
                  "--at-scale-steps", "2", "--at-scale-large-workload", "cuboid:7,3,3", "--at-scale-large-steps", "2"]
                 + {extra!r})
if os.environ["RANK"] == "0":
    assert out is not None
else:
    assert out is None
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_two_ranks(extra):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER.format(root=ROOT, extra=extra)], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1, outs[0][0]
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")], "only rank 0 prints"
    return json.loads(lines[0])


def _check_common(d):
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    fam = d["roofline_families"]
    # (the all-reduces of the tet-sharded mode are a family of their own, priced against nothing)
    assert set(fam) - {"collective"} == {"solve", "factor", "taylor", "io", "asm", "tail"}
    assert d["roofline"]["family"] in fam and d["roofline"]["bound"] == "hbm"
    assert abs(sum(f["share_of_step"] for f in fam.values()) - 1) < 1e-6
    assert d["roofline_whole_step"]["algorithmic_bytes_per_step"] > 0
    # the reference's own metric (one whole solve, construction -> convergence) and the at-scale leg
    # ride in the same line, for every N
    e = d["end_to_end"]
    assert e["iter"] >= 1 and e["time_solve"] > 0 and e["cold"]["converged"]
    assert set(e["setup_seconds"]) == {"analysis", "analysis_thread", "analysis_device", "solver_vectors", "tables", "jit_cold", "jit_cold_source", "jit_cached", "jit_cached_source"}
    assert e["cached"]["iter"] == e["cold"]["iter"]
    a = d["at_scale"]
    assert a["steps"] == 2 and a["value"] > 0 and a["rccl_ranks"] == d["rccl_ranks"]
    assert a["roofline"]["family"] in a["roofline_families"] and a["roofline"]["bound"] == "hbm"
    assert "cuboid:6,3,3" in a["config"]["workload"] and "dist_solver" in a["config"]
    assert a["end_to_end"]["iter"] >= 1
    # a third leg in the regime where the distributed direct solver acts, for every N
    b = d["at_scale_large"]
    assert b["steps"] == 2 and b["value"] > 0 and "cuboid:7,3,3" in b["config"]["workload"]
    assert "dist_solver" in b["config"] and b["end_to_end"]["iter"] >= 1 and "factor" in b["roofline_families"]
    if d["config"]["parallelism"].startswith("tet-shard"):
        assert a["config"]["parallelism"].startswith("tet-shard") and a["collective_ms_per_step"] > 0


def test_two_rank_replicas_gloo():
    d = _run_two_ranks(["--parallelism", "replicas"])
    _check_common(d)
    assert d["scaling"] == "weak" and d["config"]["parallelism"] == "replicas"
    # whole-job value = N * K / max-over-ranks time
    assert abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 3 / 1e3)) < 1e-6 * d["value"]


def test_two_rank_shard_is_the_default_and_terminates():
    """N > 1 defaults to ONE tet-sharded problem (BASELINE config 4); the measurement steps after the timed
    region issue collectives too, so every rank has to run them (they once ran on rank 0 only: a deadlock)."""
    d = _run_two_ranks([])
    _check_common(d)
    assert d["scaling"] == "strong" and d["config"]["parallelism"].startswith("tet-shard")
    assert "collective" in d["roofline_families"] and d["roofline_families"]["collective"]["bound"] == "xgmi"
    assert d["roofline"]["frac"] > 0 and d["roofline"]["launches_per_step"] is not None
    # one problem: value = K / max-over-ranks time
    assert abs(d["value"] - 3 / (d["ms_per_step"] * 3 / 1e3)) < 1e-6 * d["value"]


def test_plain_command_with_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2 ...` with no torchrun around it and no WORLD_SIZE in the environment: the process
    becomes the launcher of two fresh rank processes (bench.spawn_ranks), rank 0's line comes out on its stdout and
    says n_gpus == 2; the collective's own rank count is reported (gloo here: the host harness has no RCCL)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SANM_BENCH_TEST_HOOK"] = "tests.hostsim.bench_hook"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", "cuboid:5,3,3", "--no-cpu-baseline", "--dist-backend", "gloo",
                        "--at-scale-workload", "cuboid:6,3,3", "--at-scale-steps", "2",
                        "--at-scale-large-workload", "cuboid:7,3,3", "--at-scale-large-steps", "2"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    _check_common(d)
    assert d["n_gpus"] == 2 and d["backend"] == "hostsim"
    assert d["rccl_ranks"] == 2 and "gloo" in d["collective_impl"]
    assert d["scaling"] == "strong" and d["collective_ms_per_step"] > 0


def test_launcher_reports_a_failing_rank():
    """a rank that dies takes the launcher's exit status with it (and the other ranks are ended, not left waiting)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SANM_BENCH_TEST_HOOK"] = "tests.hostsim.no_such_hook_module"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--workload", "cuboid:5,3,3", "--no-cpu-baseline", "--dist-backend", "gloo"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "launcher: rank" in r.stderr


def test_plain_command_two_ranks_with_the_distributed_direct_solver():
    """the same launcher with the subtree-distributed direct solver forced on (SANM_DIST_SOLVER=1; host harness, gloo):
    bench.py's timed region, its family measurement steps (all ranks) and the line work with the solver's own
    collectives in the sparse_prep / sparse_solve brackets"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SANM_BENCH_TEST_HOOK"] = "tests.hostsim.bench_hook"
    env["SANM_DIST_SOLVER"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", "cuboid:10,5,5", "--no-cpu-baseline", "--dist-backend", "gloo",
                        "--at-scale-workload", "cuboid:6,3,3", "--at-scale-steps", "2",
                        "--at-scale-large-workload", "cuboid:7,3,3", "--at-scale-large-steps", "2"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    _check_common(d)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "strong"
    # more collectives than the tet-sharded driver's alone: per step 1 + 1 + (order - 1) = 13 at order 12, plus two per
    # factorisation and two per solve
    assert d["roofline_families"]["collective"]["launches_per_step"] >= 0


def test_refined_workload_mesh_is_a_conforming_subdivision():
    """bench.py's `refine:<config>:<levels>` workloads (an organic mesh at scale: every tet of a BASELINE mesh cut into
    8): volume conserved, every child oriented like its parent, the subdivision conforming (an interior face belongs
    to exactly two tets, a boundary face to one, and the boundary faces are 4 x the parent's)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from sanm_amd import fea as dfea

    def faces(T):
        f = np.concatenate([np.sort(T[:, list(c)], axis=1) for c in [(0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3)]], axis=0)
        _, cnt = np.unique(f, axis=0, return_counts=True)
        return cnt

    def volumes(m):
        d = m.V[m.tets[:, 1:]] - m.V[m.tets[:, :1]]
        return np.einsum("ij,ij->i", np.cross(d[:, 0], d[:, 1]), d[:, 2]) / 6

    _, coarse = dfea.load_named_config("bob")
    fine = dfea.refine_mesh(coarse, 1)
    assert fine.nr_tet == 8 * coarse.nr_tet
    v0, v1 = volumes(coarse), volumes(fine)
    assert abs(v1.sum() - v0.sum()) <= 1e-12 * abs(v0.sum())
    assert np.all(np.sign(v1) == np.tile(np.sign(v0), 8)) and np.all(np.abs(v1) > 0)
    c0, c1 = faces(coarse.tets), faces(fine.tets)
    assert set(np.unique(c1)) <= {1, 2}
    assert (c1 == 1).sum() == 4 * (c0 == 1).sum()
    # the refined surface is the set of vertices of the faces that belong to one tet only (not every
    # midpoint of two surface vertices -- interior edges join surface vertices in thin parts)
    f = np.concatenate([np.sort(fine.tets[:, list(c)], axis=1) for c in [(0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3)]], axis=0)
    uf, cnt = np.unique(f, axis=0, return_counts=True)
    assert set(uf[cnt == 1].ravel().tolist()) == set(np.asarray(fine.surface_vtx).tolist())
    twice = dfea.refine_mesh(dfea.make_cuboid(3, 3, 3, 0.1), 2)
    assert twice.nr_tet == 64 * 5 * 8 and set(np.unique(faces(twice.tets))) <= {1, 2}


def test_launcher_ends_the_ranks_when_one_dies_inside_a_collective():
    """a rank that is killed while the others wait for it in a collective (SANM_BENCH_TEST_DIE_AT: rank 1 exits hard
    before its second barrier) must not leave the launcher hanging: it reports the dead rank's status, ends the rest by
    PID and exits non-zero well within the test's time limit"""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SANM_BENCH_TEST_HOOK"] = "tests.hostsim.bench_hook"
    env["SANM_BENCH_TEST_DIE_AT"] = "1:2"  # rank 1, at its 2nd barrier
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", "cuboid:5,3,3", "--no-cpu-baseline", "--dist-backend", "gloo",
                        "--at-scale-workload", "none"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "launcher: rank 1 exited with status" in r.stderr, r.stderr[-2000:]
    assert time.time() - t0 < 120
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], "no line from a job that lost a rank"
