"""The RCCL all-reduce callback of the tet-sharded path on a real GPU: a
single-rank process group (only one GPU is available to the test box) checks the
zero-copy device-pointer wrapping and the collective call itself."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_allreduce_callback_single_rank():
    import torch
    import torch.distributed as dist
    from sanm_amd import dist as sdist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        fn = sdist.make_rccl_allreduce()
        x = torch.arange(1000, dtype=torch.float64, device="cuda") * 0.5
        fn(x.data_ptr(), x.numel())
        assert np.array_equal(x.cpu().numpy(), np.arange(1000) * 0.5)
    finally:
        dist.destroy_process_group()


def _cfg():
    # (Pade off: the step count of two runs whose nodal sums are formed in different orders is only guaranteed equal
    # without its ill-conditioned decisions -- tests/lockstep.py; tests/test_sharded.py runs both settings)
    return {"material": {"young": 3e3, "poisson": 0.45, "density": 1000.0}, "g": [0, -9.81, 0],
            "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_c", "order": 12,
            "disable_pade": True}


def test_sharded_hip_path_with_the_library_communicator_equals_the_unsharded_solve():
    """sanm_anm_eqn_solver_create_sharded on the HIP backend with the library's own RCCL communicator
    (ncclAllReduce queued on the solver's stream): one rank, so every collective is the identity and the result
    must equal the unsharded solve -- same step count, same vertices.  The test box has one GPU; what this proves
    is that the sharded code path (tet range, per-order all-reduce of b_k, all-reduce of the Jacobian values and
    of f(x0), RCCL bound by dlopen, solver stream) runs on the device."""
    import sanm_amd
    from sanm_amd import dist as sdist
    from sanm_amd import fea as dfea
    api = sanm_amd.get_api(0)
    assert api.backend_name() == "hip"
    sdist.init_native_comm(api, 0, 1)
    try:
        mesh = lambda: dfea.make_cuboid(8, 4, 4, 0.025)
        ref = dfea.GravityRun(api, mesh(), _cfg(), solver_rtol=1e-15).run()
        run = dfea.GravityRun(api, mesh(), _cfg(), shard=(0, 1, None), solver_rtol=1e-15).run()
        assert run.solver.get_nr_iter() == ref.solver.get_nr_iter()
        V, Vr = run.vertices(), ref.vertices()
        assert np.abs(V - Vr).max() / np.abs(Vr).max() < 1e-9
        assert run.rms[-1] < 1e-10
    finally:
        api.comm_destroy()


def test_sharded_hip_path_with_the_callback_equals_the_unsharded_solve():
    """the same through the C ABI's all-reduce callback (torch.distributed nccl group of one rank)"""
    import torch
    import torch.distributed as dist
    import sanm_amd
    from sanm_amd import dist as sdist
    from sanm_amd import fea as dfea
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        api = sanm_amd.get_api(0)
        ncall = [0]
        base = sdist.make_rccl_allreduce()

        def counted(ptr, count):
            ncall[0] += 1
            base(ptr, count)

        mesh = lambda: dfea.make_cuboid(8, 4, 4, 0.025)
        ref = dfea.GravityRun(api, mesh(), _cfg(), solver_rtol=1e-15).run()
        run = dfea.GravityRun(api, mesh(), _cfg(), shard=(0, 1, counted), solver_rtol=1e-15).run()
        steps = run.solver.get_nr_iter()
        assert steps == ref.solver.get_nr_iter()
        assert np.abs(run.vertices() - ref.vertices()).max() / np.abs(ref.vertices()).max() < 1e-9
        assert ncall[0] == steps * (1 + 1 + (12 - 1)) + 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["armadillo_small", "human_arap16"])
def test_full_size_configs_through_the_sharded_solver(name):
    """BASELINE configs 4 and 5 take `sanm_anm_eqn_solver_create_sharded` (symbolic.cpp:525-536 is what it replaces)
    with the library's RCCL communicator.  The test box has one GPU, so world = 1 -- every line of the N-rank code
    runs (tet range, gather restricted to own tets, ncclAllReduce of f(x0) / the Jacobian values / b_k per order on
    the solver's stream, solve without the fused ends), the collectives reduce over one rank.  Checked like the
    unsharded full-size runs: convergence, the oracle's equilibrium (tests/golden/full_*.npz) to 1e-6, the
    continuation step by step beside a live oracle (tests/lockstep.py), and the free-running step count identical
    to the oracle's wherever no ill-conditioned Pade decision was met."""
    import json
    import sanm_amd
    from oracle import fea as ofea
    from sanm_amd import dist as sdist
    from sanm_amd import fea as dfea
    from tests.lockstep import LockStep
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    api = sanm_amd.get_api(0)
    assert sdist.init_native_comm(api, 0, 1)
    try:
        assert api.comm_query() == (1, 0)  # ncclCommCount / ncclCommUserRank of the live communicator
        cfg, mesh = dfea.load_named_config(name)
        run = dfea.GravityRun(api, mesh, dict(cfg), shard=(0, 1, None)).run()
        assert run.solver.converged() and run.rms[-1] < 1e-10
        steps = run.solver.get_nr_iter()
        gold = np.load(os.path.join(root, "tests", "golden", f"full_{name}.npz"))
        Vo, osteps = gold["vertices"], int(gold["steps"])
        V = run.vertices()
        err = float(np.abs(V - Vo).max() / np.abs(Vo).max())
        assert err <= 1e-6, err
        # lock-step beside the oracle
        cfg2, mesh2 = dfea.load_named_config(name)
        run2 = dfea.GravityRun(api, mesh2, dict(cfg2), shard=(0, 1, None)).construct()
        cfg3, mesh3 = dfea.load_named_config(name)
        _, osolver, _ = ofea.make_gravity_solver(ofea.TetMesh(mesh3.V, mesh3.tets, mesh3.surface_vtx), cfg3)
        ls = LockStep(run2, osolver).run_to_convergence()
        assert ls.nr_steps == steps
        rec = {"config": name, "path": "sanm_anm_eqn_solver_create_sharded, world 1, library communicator",
               "device_steps": int(steps), "oracle_free_running_steps": osteps, "vertex_rel_err": err,
               "lockstep": ls.summary()}
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        json.dump(rec, open(os.path.join(root, "gpurun_out", f"parity_steps_sharded_{name}.json"), "w"), indent=1)
        print(json.dumps({k: rec[k] for k in ("config", "device_steps", "oracle_free_running_steps", "vertex_rel_err")}),
              "events:", [(e["step"], e["device"], e["oracle"]) for e in ls.events])
        if not ls.events:
            assert steps == osteps
    finally:
        api.comm_destroy()


WORKER_2ON1 = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("gloo")
import sanm_amd
from sanm_amd import fea as dfea, dist as sdist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
api = sanm_amd.get_api(0)
assert api.backend_name() == "hip"
cfg, mesh = dfea.load_named_config({name!r})
p2p_log = []
p2p_cb = sdist.set_staged_p2p(api, p2p_log) if os.environ.get("TEST_P2P") else None
run = dfea.GravityRun(api, mesh, dict(cfg), shard=(rank, world, sdist.make_staged_allreduce())).run()
gold = np.load(os.path.join({root!r}, "tests", "golden", "full_" + {name!r} + ".npz"))
V, Vo = run.vertices(), gold["vertices"]
print("RESULT " + json.dumps(dict(rank=rank, steps=int(run.solver.get_nr_iter()), gold_steps=int(gold["steps"]),
                                  err=float(np.abs(V - Vo).max() / np.abs(Vo).max()), rms=float(run.rms[-1]),
                                  vsum=float(V.sum()), st=run.solver.stats(),
                                  p2p=[len(p2p_log)] + [int(sum(c[i] for c in p2p_log)) for i in range(4)])), flush=True)
dist.barrier()
dist.destroy_process_group()
"""


def _two_ranks_on_one_gpu(name, env_extra, world=2):
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    base_env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    base_env.update(env_extra)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(world):
        env = dict(base_env, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER_2ON1.format(root=root, name=name)],
                                      env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    try:
        for p in procs:
            so, se = p.communicate(timeout=900)
            assert p.returncode == 0, se[-3000:]
            res.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][0][7:]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return sorted(res, key=lambda r: r["rank"])


def _check_tree_distribution(res):
    for r in res:
        assert r["steps"] == r["gold_steps"] == 2 and r["err"] < 1e-9 and r["rms"] < 1e-10
        assert r["st"]["nr_subtree"] >= len(res) and r["st"]["nr_subtree_own"] >= 1  # every rank owns a subtree
    assert len({r["vsum"] for r in res}) == 1
    total, top = res[0]["st"]["factor_flops"], res[0]["st"]["factor_flops_top"]
    own = [r["st"]["factor_flops_own"] for r in res]
    top_own = [r["st"]["factor_flops_top_own"] for r in res]
    # every front has one owner: subtrees and top fronts partition the work
    assert abs(sum(own) + top - total) <= 1e-9 * total and abs(sum(top_own) - top) <= 1e-9 * total
    crit = res[0]["st"]["factor_flops_critical"]
    assert max(o + t for o, t in zip(own, top_own)) <= crit * (1 + 1e-9) and crit < total
    return own, top_own, total, crit


def test_subtree_distributed_solver_on_the_device_two_ranks_on_one_gpu():
    """The distributed direct solver (multifrontal.cpp, MfSchedule::Dist: every front of the elimination tree one owner,
    stages with exchanges between them) on the HIP backend: two ranks share cuda:0 (staged gloo all-reduce),
    SANM_DIST_SOLVER=1 forces the distribution on the BASELINE-size mesh of config 4.  Each rank zeroes, factors and
    solves its own fronts only; Schur complements, inbox rows and solution entries are exchanged (copy2d_kernel,
    mf_factor_piece / mf_solve_piece).  Every entry of every exchange has one writer, so the result must be the oracle's
    equilibrium in the oracle's 2 steps, identical on both ranks."""
    res = _two_ranks_on_one_gpu("armadillo_small", {"SANM_DIST_SOLVER": "1"})
    own, top_own, total, crit = _check_tree_distribution(res)
    print([(r["rank"], r["steps"], r["err"]) for r in res], "own", own, "top own", top_own, "total", total, "critical", crit)
    # (armadillo_small's Jacobian graph has two components, i.e. the elimination forest two roots: at two ranks each
    # gets one -- no top, one stage; the four-rank case below has the stages)
    assert res[0]["st"]["nr_dist_stage"] >= 1
    assert max(own) <= 0.8 * sum(own)


def test_tree_distributed_solver_three_stages_four_ranks_on_one_gpu():
    """the same over FOUR ranks sharing cuda:0: the top of the tree is mapped onto rank sets -- the
    root to rank 0, the two separators below it to ranks 0 and 2, beside each other --, three stages, exchanges before
    the second and the third; every rank owns a subtree; equilibrium, step count and bits as above."""
    res = _two_ranks_on_one_gpu("armadillo_small", {"SANM_DIST_SOLVER": "1"}, world=4)
    own, top_own, total, crit = _check_tree_distribution(res)
    print("4 ranks: own", own, "top own", top_own, "total", total, "critical", crit)
    assert res[0]["st"]["nr_dist_stage"] >= 2
    assert sum(t > 0 for t in top_own) >= 2  # separators beside each other on different owners
    assert crit <= 0.75 * total


def test_tree_distributed_solver_over_point_to_point_transfers_four_ranks_on_one_gpu():
    """the three-stage case again with the exchanges as POINT-TO-POINT transfers (the branch a multi-GPU node takes with
    `ncclSend` / `ncclRecv` / `ncclBroadcast`): the test hook `sanm_test_set_p2p`, every transfer staged through the host and
    a gloo group because the four ranks share cuda:0 (sanm_amd/dist.py, set_staged_p2p).  Pack (copy2d_kernel), transfer,
    unpack on the HIP backend; Schur complements and inbox rows to the one rank that needs them, every stage's pivots to
    everyone.  Equilibrium, step count and bits as with the all-reduce form."""
    res = _two_ranks_on_one_gpu("armadillo_small", {"SANM_DIST_SOLVER": "1", "TEST_P2P": "1"}, world=4)
    _check_tree_distribution(res)
    assert res[0]["st"]["nr_dist_stage"] >= 2
    for r in res:
        ncalls, sends, recvs, bcasts, doubles = r["p2p"]
        assert ncalls > 0 and bcasts > 0 and doubles > 0
    assert sum(r["p2p"][1] for r in res) == sum(r["p2p"][2] for r in res) > 0


WORKER_AT_SCALE = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("gloo")
import bench
import sanm_amd
from sanm_amd import fea as dfea, dist as sdist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
api = sanm_amd.get_api(0)
cfg, mesh = bench.load_workload({name!r})
p2p_log = []
p2p_cb = sdist.set_staged_p2p(api, p2p_log) if os.environ.get("TEST_P2P") else None
run = dfea.GravityRun(api, mesh, dict(cfg), shard=(rank, world, sdist.make_staged_allreduce())).run(max_iter={max_iter})
st = run.solver.stats()
V = run.vertices()
out = dict(rank=rank, steps=int(run.solver.get_nr_iter()), rms=float(run.rms[-1]), vsum=float(V.sum()), st=st,
           converged=bool(run.solver.converged()),
           p2p=[len(p2p_log)] + [int(sum(c[i] for c in p2p_log)) for i in range(4)])
if rank == 0:
    if p2p_cb is not None:
        api.lib.sanm_test_set_p2p(type(p2p_cb)(), None)
    del run
    cfg1, mesh1 = bench.load_workload({name!r})
    ref = dfea.GravityRun(api, mesh1, dict(cfg1)).run(max_iter={max_iter})
    Vr = ref.vertices()
    out.update(ref_steps=int(ref.solver.get_nr_iter()), err=float(np.abs(V - Vr).max() / np.abs(Vr).max()),
               ref_flops=ref.solver.stats()["factor_flops"], ref_store=ref.solver.stats()["front_store_doubles"])
print("RESULT " + json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("name,world,max_iter", [("refine:armadillo_small:1", 2, 60), ("refine:armadillo_small:1", 4, 60), ("refine:armadillo_small:2", 4, 2)])
def test_tree_distributed_solver_at_scale_ranks_on_one_gpu(name, world, max_iter):
    """the distributed direct solver where it is ON BY DEFAULT (from 50 GFLOP per factorisation), tet-sharded over ranks that
    share cuda:0, the exchanges point to point (test hook, staged through the host): the 338 k-tet leg of the bench over two
    and over four ranks to convergence (the top of the tree mapped onto rank sets, three stages), and two continuation steps
    of the 2.7 M-tet leg over four (6.6 TFLOP per factorisation, Schur transfers of hundreds of MB; a rank stores the fronts
    it factors and the Schur blocks it receives -- four whole front stores would not fit the device).  All ranks
    end on the same bits; rank 0 then runs the problem unsharded: same steps, vertices to 1e-9 (the tet-sharded sums of
    b_k differ in their order), the ranks' own flops plus the top's add up to the unsharded count."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    base_env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    base_env["TEST_P2P"] = "1"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(world):
        env = dict(base_env, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER_AT_SCALE.format(root=root, name=name, max_iter=max_iter)],
                                      env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    try:
        outs = [p.communicate(timeout=1500) for p in procs]
        bad = [(i, p.returncode) for i, p in enumerate(procs) if p.returncode != 0]
        if bad:  # (every rank's own story: the first one to fail takes the others' collectives down with it)
            os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
            for i, (_, se) in enumerate(outs):
                open(os.path.join(root, "gpurun_out", f"dist_at_scale_rank{i}.err"), "w").write(se)
        assert not bad, (bad, [o[1][-600:] for o in outs])
        for so, _ in outs:
            res.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][0][7:]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res.sort(key=lambda r: r["rank"])
    r0 = res[0]
    assert len({r["vsum"] for r in res}) == 1 and len({r["steps"] for r in res}) == 1 and r0["steps"] == r0["ref_steps"]
    assert r0["err"] < 1e-9, r0["err"]
    if max_iter >= 60:
        assert r0["converged"] and r0["rms"] < 1e-10
    assert r0["st"]["nr_subtree"] >= world and all(r["st"]["nr_subtree_own"] >= 1 for r in res)
    own = [r["st"]["factor_flops_own"] for r in res]
    top = r0["st"]["factor_flops_top"]
    assert abs(sum(own) + top - r0["ref_flops"]) <= 1e-9 * r0["ref_flops"] and max(own) < 1.4 * r0["ref_flops"] / world
    assert sum(r["p2p"][1] for r in res) == sum(r["p2p"][2] for r in res) > 0
    held = [r["st"]["front_store_doubles"] for r in res]
    assert all(0 < h < r0["ref_store"] for h in held) and r0["ref_store"] <= sum(held) < 1.5 * r0["ref_store"]
    print("front stores GB", [round(h * 8 / 1e9, 2) for h in held], "of", round(r0["ref_store"] * 8 / 1e9, 2))
    print(name, world, "ranks: steps", r0["steps"], "err", r0["err"], "own GF", [round(o / 1e9, 1) for o in own], "top GF", round(top / 1e9, 1),
          "critical GF", round(r0["st"]["factor_flops_critical"] / 1e9, 1), "stages", r0["st"]["nr_dist_stage"],
          "p2p calls / sends / receives / broadcasts / doubles", [r["p2p"] for r in res])


def test_subtree_distributed_solver_over_chains_and_two_phase_levels_on_the_device():
    """the same with the round-4 schedule features forced onto this mesh: fronts cut into chains (SANM_MF_SPLIT_K),
    every height two-phase (SANM_MF_TWO_PHASE: boundary operators not multiplied out, two launches per sweep) and the
    wide backward kernel (SANM_MF_WIDE_MIN_M) -- through mf_factor_piece / mf_solve_piece and the three exchanges.
    The arithmetic of the solves differs from the default schedule's, so the comparison is with the golden equilibrium
    (1e-9) and between the ranks (bit for bit), not with the default run's bits."""
    res = _two_ranks_on_one_gpu("armadillo_small", {"SANM_DIST_SOLVER": "1", "SANM_MF_SPLIT_K": "96",
                                                     "SANM_MF_TWO_PHASE": "1", "SANM_MF_WIDE_MIN_M": "512"})
    for r in res:
        assert r["steps"] == r["gold_steps"] == 2 and r["err"] < 1e-9 and r["rms"] < 1e-10
        assert r["st"]["nr_subtree"] >= 2
    assert max(r["st"]["nr_level"] for r in res) > 8
    assert res[0]["vsum"] == res[1]["vsum"]


def test_two_ranks_on_one_gpu_run_the_world_2_branch_of_the_sharded_hip_path():
    """World = 2 on the device.  RCCL refuses two ranks on one GPU and the test box has one, so the all-reduce goes
    through the C ABI's callback, staged through a gloo group (sanm_amd.dist.make_staged_allreduce) -- not a
    measurement, but it makes the world = 2 branch of the library (tet ranges of rank 0 and 1, gathers restricted
    to own tets, all-reduce of partial nodal sums that are NOT the whole sum) execute on the HIP backend before
    the driver's first multi-GPU run, on the BASELINE-size mesh of config 4: both ranks must reach the oracle's
    equilibrium (tests/golden/full_armadillo_small.npz) in the oracle's 2 steps and agree with each other bit for
    bit.  Then `python bench.py --gpus 2` as a PLAIN COMMAND: the launcher itself starts the two ranks."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    base_env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = _two_ranks_on_one_gpu("armadillo_small", {})
    print(res)
    for r in res:
        assert r["steps"] == r["gold_steps"] == 2 and r["err"] < 1e-9 and r["rms"] < 1e-10
    assert res[0]["vsum"] == res[1]["vsum"]
    # the launcher
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "3",
                        "--workload", "armadillo_small", "--no-cpu-baseline", "--dist-backend", "gloo",
                        "--at-scale-workload", "none"],
                       env=base_env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["backend"] == "hip" and d["rccl_ranks"] == 2
    assert d["config"]["parallelism"].startswith("tet-shard") and d["collective_ms_per_step"] > 0
    # warm-up = construct (step 1), step 2, converged -> restart: the first whole solve took the oracle's 2 steps
    assert d["config"]["steps_per_solve"][0] == 2, d["config"]["steps_per_solve"]
    # ... which is what the line's end_to_end object timed from the constructor on (the reference's time_solve)
    assert d["end_to_end"]["iter"] == 2 and d["end_to_end"]["cold"]["converged"]
    # (armadillo's Neo-Hookean graph at order 20 is in the set compiled ahead of time into the library: no run-time
    # compilation in either constructor)
    assert d["end_to_end"]["setup_seconds"]["jit_cold_source"] == "embedded"
    assert d["end_to_end"]["setup_seconds"]["jit_cached_source"] == "embedded"


_TWO_DEVICE_WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r})
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch
import torch.distributed as dist
rank = int(os.environ["RANK"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", init_method="file://" + os.environ["RDZV"], rank=rank, world_size=2,
                        device_id=torch.device("cuda", rank))
import sanm_amd
from sanm_amd import dist as sdist
from sanm_amd import fea as dfea
api = sanm_amd.get_api(rank)
ok = sdist.init_native_comm(api, rank, 2)
assert ok, "the library communicator did not come up on both ranks"
world, r = api.comm_query()
assert (world, r) == (2, rank), (world, r)
cfg = {{"material": {{"young": 3e3, "poisson": 0.45, "density": 1000.0}}, "g": [0, -9.81, 0], "boundary_thresh": 0.05,
       "boundary_proj_dir": [-1, 0, 0], "energy_model": "neohookean_c", "order": 12, "disable_pade": True}}
mesh = lambda: dfea.make_cuboid(12, 5, 5, 0.025)
ref = dfea.GravityRun(api, mesh(), dict(cfg), solver_rtol=1e-15).run()
run = dfea.GravityRun(api, mesh(), dict(cfg), shard=(rank, 2, None), solver_rtol=1e-15).run()
V, Vr = run.vertices(), ref.vertices()
print(json.dumps({{"rank": rank, "steps": int(run.solver.get_nr_iter()), "ref_steps": int(ref.solver.get_nr_iter()),
                  "err": float(np.abs(V - Vr).max() / np.abs(Vr).max()), "vsum": float(V.sum()), "rms": float(run.rms[-1])}}), flush=True)
api.comm_destroy()
dist.barrier()
dist.destroy_process_group()
"""


def test_native_rccl_allreduce_between_two_devices(tmp_path):
    """ncclAllReduce of the library communicator ACROSS TWO DEVICES: sanm_amd.dist.init_native_comm (identifier
    broadcast, ncclCommInitRank on both ranks), ncclCommCount == 2, a tet-sharded solve whose nodal sums really are
    partial on each rank, compared with the unsharded solve on the same devices.  The pool's test boxes have ONE GPU:
    there the test skips and says so -- it is here so that the first box with two devices runs the collective before
    the scaling bench does."""
    import json
    import subprocess
    import sys
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip(f"{ndev} GPU visible: ncclAllReduce between two devices cannot run on this box (the world = 2 branch "
                    "of the library runs on one GPU through a staged gloo all-reduce in the tests above)")
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["RDZV"] = str(tmp_path / "store")
    procs = [subprocess.Popen([sys.executable, "-c", _TWO_DEVICE_WORKER.format(root=root)], env=dict(env, RANK=str(r)),
                              cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=900))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
        res.append(json.loads([l for l in so.splitlines() if l.startswith("{")][-1]))
    for r in res:
        assert r["steps"] == r["ref_steps"] and r["err"] < 1e-9 and r["rms"] < 1e-10
    assert res[0]["vsum"] == res[1]["vsum"]
