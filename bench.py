#!/usr/bin/env python3
"""Benchmark of the ANM hot path: ANM continuation steps / second.

  python bench.py --gpus N --steps K --warmup W

A "step" is one completed ANMDriverHelper::solve_expansion_coeffs
(libsanm/anm.cpp:193-312): order-0 evaluation, Jacobian, CSR assembly, linear
solver preparation and N=order (bias, solve, coefficient) rounds, plus the
range estimate / Pade.  The workload is the BASELINE metric's configuration:
config/armadillo.json (Neo-Hookean compressible, order 20, Pade and sanity
checks on).  The full Armadillo mesh is missing from the reference snapshot
(.MISSING_LARGE_BLOBS), so the stand-in is the shipped Armadillo-small.1
(config/armadillo_small.json: same material, load and boundary rule).

The continuation runs from the rest state exactly as `fea` does; when a solve
converges before W+K steps are done, the next solve starts again from the rest
state on the same solver (sanm_anm_restart), so exactly K completed steps are
timed.  For N > 1 every rank runs the whole problem on its own GPU (replicas:
see DESIGN.md "Multi-GPU"), value = N*K / max-over-ranks time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=12)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--workload", default="armadillo_small")
    p.add_argument("--solver-rtol", type=float, default=1e-12)
    p.add_argument("--solver-kind", type=int, default=1, help="0: Jacobi-PCG, 1: multifrontal LU")
    p.add_argument("--profile", type=int, default=0)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-steps", type=int, default=2)
    p.add_argument("--parallelism", default="replicas", choices=["replicas", "shard"],
                   help="N>1: independent replicas (weak scaling) or one tet-sharded problem with an "
                        "RCCL all-reduce of b_k per Taylor order (strong scaling)")
    p.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL on ROCm) or gloo (CPU tests)")
    return p.parse_args(argv)


def cpu_baseline(workload, cpu_steps):
    """The oracle (numpy port of the reference algorithm; MKL PARDISO on one thread for the direct
    solve when the image has MKL, SuperLU otherwise) timed on the host for `cpu_steps` ANM steps of
    the same workload."""
    import numpy as np  # noqa: F401
    from oracle import fea as ofea
    from oracle import pardiso
    from sanm_amd import fea as dfea
    cfg, mesh = dfea.load_named_config(workload)
    omesh = ofea.TetMesh(mesh.V, mesh.tets, mesh.surface_vtx)
    t0 = time.perf_counter()
    model, solver, _ = ofea.make_gravity_solver(omesh, cfg)  # ctor = first step
    steps = 1
    while steps < cpu_steps and not solver.converged:
        solver.next_iter()
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "ANM steps/s", "cores": 1, "kind": "port",
            "sample": f"{steps} ANM step(s) of {workload} (order {cfg.get('order', 20)}) from the rest state, "
                      f"numpy oracle + {'MKL PARDISO (1 thread)' if pardiso.available() else 'SuperLU'}, "
                      f"{dt:.1f} s incl. graph/remap setup",
            "profile": {k: round(v, 3) for k, v in solver.profile.items()}}


def make_api(local_rank):
    """The HIP product library on GPU `local_rank` (tests replace this hook)."""
    import sanm_amd
    return sanm_amd.get_api(local_rank)


def device_sync():
    import torch
    torch.cuda.synchronize()


def load_workload(name):
    """A named BASELINE config, 'cuboid:nx,ny,nz' (test-sized synthetic cantilever) or 'block:N' (the
    armadillo material / load / boundary rule on an N^3-vertex block of 5 (N-1)^3 tets: the scaling stand-in
    SURVEY 8d names for the missing full Armadillo mesh; block:60 = 1.03 M tets)."""
    from sanm_amd import fea as dfea
    if name.startswith("block:"):
        nx = int(name.split(":")[1])
        cfg, _ = dfea.load_named_config("armadillo_small")
        cfg = dict(cfg)
        cfg.pop("scale", None)
        cfg["material"] = dict(cfg["material"], young=2.0e4)  # soft enough to need several steps
        return cfg, dfea.make_cuboid(nx, nx, nx, 0.2 / nx)
    if name.startswith("cuboid:"):
        nx, ny, nz = (int(v) for v in name.split(":")[1].split(","))
        cfg = {"material": {"young": 3e3, "poisson": 0.45, "density": 1000.0}, "g": [0, -9.81, 0],
               "boundary_thresh": 0.05, "boundary_proj_dir": [-1, 0, 0],
               "energy_model": "neohookean_c", "order": 12}
        return cfg, dfea.make_cuboid(nx, ny, nz, 0.025)
    return dfea.load_named_config(name)


def main(argv=None):
    args = parse(argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # torch must load its HIP runtime before libsanm_hip.so pulls in the system one
    # (the other order leaves torch without visible devices)
    import torch  # noqa: F401
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)

    from sanm_amd import fea as dfea
    api = make_api(local_rank)
    cfg, mesh = load_workload(args.workload)
    shard = None
    if world > 1 and args.parallelism == "shard":
        from sanm_amd import dist as sdist
        fn = sdist.make_rccl_allreduce() if args.dist_backend == "nccl" else sdist.make_host_allreduce()
        shard = (rank, world, fn)
    run = dfea.GravityRun(api, mesh, cfg, shard=shard, solver_rtol=args.solver_rtol,
                          solver_kind=args.solver_kind, profile=args.profile)
    x0 = run.model.x0()

    def barrier():
        if dist is not None:
            dist.barrier()
        device_sync()

    # ---- continuation with restarts: exactly W + K completed steps ----------
    state = {"started": False, "solves": 0, "steps_per_solve": []}

    def one_step():
        if not state["started"]:
            run.construct()  # first step of the first solve
            state["started"] = True
            state["cur"] = 1
            return
        s = run.solver
        if not s.converged():
            before = s.get_nr_iter()
            s.next_iter()
            if s.get_nr_iter() > before:
                state["cur"] += 1
                return
        # converged (the converged call only evaluates f(x0)): begin a new solve
        state["solves"] += 1
        state["steps_per_solve"].append(state["cur"])
        s.restart(x0)
        state["cur"] = 1

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    it0 = run.solver.get_nr_iter() if state["started"] else 0
    for _ in range(args.steps):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    assert run.solver.get_nr_iter() - it0 == args.steps
    if dist is not None:
        import torch
        dev = "cuda" if args.dist_backend == "nccl" else "cpu"
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    stats = run.solver.stats()
    out = None

    if rank == 0:
        # ---- roofline of the dominant kernel: the Taylor pass (graph interpreter) ---
        # Measured live with HIP events on the solver's stream over two extra ANM
        # steps (same launch mix as the timed region: eval0 + grad + N bias + N-1
        # coefficient passes per step).
        n, nnz, T = stats["nr_unknown"], stats["jacobian_nnz"], stats["nr_tet"]
        N = int(cfg.get("order", 20))
        run.solver.pass_timing(True, fetch=False)
        it_before = run.solver.get_nr_iter()
        while run.solver.get_nr_iter() - it_before < 2:
            one_step()
        pass_ms, pass_cnt = run.solver.pass_timing(False)
        steps_meas = run.solver.get_nr_iter() - it_before
        avg_ms = pass_ms / max(pass_cnt, 1)
        # algorithmic bytes (state-streaming model): per tet and step
        #   8 * [S*N(N-1)/2 + N*(C+S+9)],  S/C = per-order state / per-tet constants
        SC = {"neohookean_c": (20, 45), "neohookean_i": (22, 45), "arap": (27, 39)}
        S_, C_ = SC.get(cfg["energy_model"], (20, 45))
        bytes_step = 8.0 * T * (S_ * N * (N - 1) / 2 + N * (C_ + S_ + 9))
        launches_step = pass_cnt / max(steps_meas, 1)
        if pass_cnt > 0 and avg_ms > 0:
            alg_bytes = bytes_step / launches_step
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        else:  # a backend without event timing (the CPU test harness)
            alg_bytes, achieved = bytes_step / (2 * N + 1.5), 0.0
        # HBM-side bytes per launch of that kernel from the committed PMC profile
        # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, corrected per
        # MI355X_MICROARCH.md; profiles/r01_pmc_traffic.md) -- not collected live
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pmc.get("workload") == args.workload and pmc.get("order") == N:
                traffic = pmc["kernels"]["taylor_pass_kernel"]["traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "ANM continuation steps/sec (armadillo, Neo-Hookean, order 20)",
            "value": (1 if shard else world) * args.steps / dt, "unit": "ANM steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None,
            "dtype": "f64",
            "data": ("real mesh Armadillo-small.1 (stand-in for the missing Armadillo.1), rest state"
                     if args.workload == "armadillo_small" else f"workload {args.workload}, rest state"),
            "config": {"workload": (f"config/{args.workload}.json" if ":" not in args.workload
                                    else f"synthetic {args.workload} (armadillo material, load and boundary rule)")
                                   + f": {cfg['energy_model']}, order "
                                   f"{N}, T={T}, n={n}, nnz={nnz}, pade on, sanity check on",
                       "parallelism": ("tet-shard + all-reduce(b_k)/order" if shard else "replicas") if world > 1 else "single",
                       "linear_solver": "jacobi-pcg" if args.solver_kind == 0 else "multifrontal-lu",
                       "solver_stats": {k: stats[k] for k in ("factor_nnz", "factor_flops", "nr_front",
                                                              "nr_level", "max_front")},
                       "profile": run.solver.profile(),
                       "steps_per_solve": state["steps_per_solve"]},
            "roofline": {"bound": "hbm", "kernel": "taylor_pass_kernel", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "avg_launch_us": avg_ms * 1e3,
                         "launches_per_step": launches_step,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "algorithmic_bytes_per_step": bytes_step},
        }
        if not args.no_cpu_baseline and world == 1 and not args.workload.startswith("block:"):
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
