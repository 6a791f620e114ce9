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
timed.  For N > 1 the default is ONE problem whose tets are sharded over the
ranks (BASELINE config 4: strong scaling, one all-reduce of b_k per Taylor order,
value = K / max-over-ranks time); `--parallelism replicas` runs N independent
copies instead (weak scaling, value = N*K / time).  See DESIGN.md "Multi-GPU".

Beside the headline the line carries, for every N, the same measurement on two
refinements of the same mesh (`at_scale`: every tet cut into 8, 338 k tets -- the
plausible size of the missing Armadillo.1; `at_scale_large`: into 64, 2.7 M tets --
the regime where factorisation and solves are the step and the distributed direct
solver acts) and `end_to_end`: one whole solve from the solver's constructor to
convergence, the reference's own time_solve (fea/main.cpp:382, :418-425).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (the pool's host driver only offers dmabuf IPC: RCCL between processes needs this before the runtime starts;
# the GPU boxes export it already)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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
    p.add_argument("--cpu-threads", type=int, default=None, help="CPU baseline threads (default min(32, cores))")
    p.add_argument("--cpu-baseline-only", action="store_true",
                   help="print the cpu_baseline object alone (no GPU work); used for the 1-thread figure")
    p.add_argument("--cpu-seconds", type=float, default=20.0, help="stepping time of the CPU baseline sample")
    p.add_argument("--parallelism", default="shard", choices=["replicas", "shard"],
                   help="N>1: one tet-sharded problem with an RCCL all-reduce of b_k per Taylor order (strong "
                        "scaling, the default: BASELINE config 4) or independent replicas (weak scaling)")
    p.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL on ROCm) or gloo (CPU tests)")
    p.add_argument("--at-scale-workload", default="auto",
                   help="second leg of the line (`at_scale`): the same measurement on a mesh of the plausible size of the "
                        "missing Armadillo.1; auto = refine:armadillo_small:1 (338 k tets) when the headline workload is "
                        "the default, none otherwise; 'none' skips it")
    p.add_argument("--at-scale-steps", type=int, default=10)
    p.add_argument("--at-scale-warmup", type=int, default=1)
    p.add_argument("--at-scale-large-workload", default="auto",
                   help="third leg of the line (`at_scale_large`): the regime where factorisation and solves are the step "
                        "and where a distributed direct solver can act; auto = refine:armadillo_small:2 (2.7 M tets, "
                        "1.97 M unknowns) when the headline workload is the default, none otherwise; 'none' skips it")
    p.add_argument("--at-scale-large-steps", type=int, default=3)
    p.add_argument("--at-scale-large-warmup", type=int, default=1)
    p.add_argument("--no-end-to-end", action="store_true",
                   help="skip the `end_to_end` object (whole solves from solver construction to convergence: the "
                        "reference's time_solve, fea/main.cpp:382, :418-425)")
    p.add_argument("--callback-allreduce", action="store_true",
                   help="shard mode: all-reduce through the C ABI's callback (torch.distributed) instead of the "
                        "library's own RCCL communicator")
    return p.parse_args(argv)


def cpu_baseline(workload, budget_s=20.0, threads=None):
    """The build's own CPU path, timed on this box's host cores: the same C++ host code as the
    product linked with the CPU backend of tests/hostsim -- tet-sharded worker threads for the Taylor passes and
    the assembly (libsanm/symbolic.cpp:525-536), serial remaps and BLAS-1 on the main thread, and the reference's
    own direct solver, MKL PARDISO, bound with dlopen and set up as libsanm/sparse_solver.cpp:107-127 does
    (mtype 11, iparm[34] = 1, iparm[1] = 3 when threaded, phase 12 at every step, phase 33 per solve).  All host
    cores the process may use.  Sample: whole solves from the rest state until `budget_s` seconds of stepping
    are spent; the per-tag breakdown mirrors the reference's ScopedProfiler tags."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # the reference's multi-threaded configuration is 32 threads (config/sys-mt32.json:3); fewer if the box has fewer
    cores = min(32, avail) if threads is None else threads
    os.environ["SANM_CPU_THREADS"] = str(cores)
    os.environ.setdefault("MKL_THREADING_LAYER", "GNU")  # libgomp is what this process has loaded already
    # (pinned: MKL would otherwise size its team by the box -- 256 hardware threads on the GPU host)
    os.environ["OMP_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = str(cores)
    os.environ.setdefault("MKL_DYNAMIC", "FALSE")
    from tests.hostsim import get_hostsim_native_api
    from sanm_amd import fea as dfea
    setup = {}
    t0 = time.perf_counter()
    capi = get_hostsim_native_api()  # -O3 -march=native build of the harness, made on this box on first use
    setup["harness_build_or_load"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    try:
        import ctypes
        ctypes.CDLL("libmkl_rt.so", mode=ctypes.RTLD_GLOBAL)  # (page-in of MKL: tens of seconds on a cold box)
    except OSError:
        pass
    setup["mkl_load"] = time.perf_counter() - t0
    # MKL's first PARDISO call in a process pays a start-up (thread team, code paths paged in) of 0.8-6 s depending on
    # the box -- none of it work of the solve.  One throw-away solve of a 54-vertex cuboid takes
    # it before the clock of `end_to_end` starts; its own time is reported as the cold surcharge.
    t0 = time.perf_counter()
    try:
        wcfg, wmesh = load_workload("cuboid:6,3,3")
        wrun = dfea.GravityRun(capi, wmesh, wcfg, solver_kind=2)
        wrun.construct()
        del wrun
    except Exception as e:  # noqa: BLE001 -- the warm-up is a courtesy to the baseline, not part of it
        print("cpu baseline: PARDISO warm-up failed:", e, file=sys.stderr)
    setup["pardiso_warmup"] = time.perf_counter() - t0
    cfg, mesh = load_workload(workload)
    t0 = time.perf_counter()
    run = dfea.GravityRun(capi, mesh, cfg, solver_kind=2, profile=1)
    setup["model_tables"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    run.construct()  # the solver's mesh-only tables (untimed) + the first step incl. PARDISO's analysis
    setup["solver_tables_and_first_step"] = time.perf_counter() - t0
    setup_s = sum(setup.values())
    s = run.solver
    x0 = run.model.x0()
    # the reference's own metric for this leg (fea/main.cpp:382, :418-425): construction -> convergence of the FIRST solve
    while not s.converged() and time.perf_counter() - t0 < 20 * budget_s:
        s.next_iter()
    e2e = {"time_solve": time.perf_counter() - t0, "iter": int(s.get_nr_iter()), "converged": bool(s.converged()),
           "setup_seconds": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in s.setup_profile().items()}}
    e2e["steps_per_sec"] = e2e["iter"] / e2e["time_solve"]
    # warm = what the clock above saw (PARDISO started before it); cold = with the start-up of the process's first call
    e2e["warm"] = {"time_solve": e2e["time_solve"], "steps_per_sec": e2e["steps_per_sec"]}
    e2e["cold"] = {"time_solve": e2e["time_solve"] + setup["pardiso_warmup"],
                   "steps_per_sec": e2e["iter"] / (e2e["time_solve"] + setup["pardiso_warmup"]),
                   "first_pardiso_call_seconds": setup["pardiso_warmup"]}
    s.set_profile(1)  # clears what the constructor accumulated
    steps, t_step = 0, 0.0
    while t_step < budget_s:
        t0 = time.perf_counter()
        it0 = s.get_nr_iter()
        s.restart(x0)
        while not s.converged() and time.perf_counter() - t0 < 4 * budget_s:
            s.next_iter()
        steps += s.get_nr_iter() - it0
        t_step += time.perf_counter() - t0
    prof = s.profile()
    tags = ("taylor_order0", "jacobian", "taylor_next_order", "taylor_push", "remap_out", "build_sparse_coeff",
            "sparse_prep", "sparse_solve", "anm_sanity_check", "estimate_valid_range", "solve_expansion_coeffs")
    return {"value": steps / t_step, "unit": "ANM steps/s", "cores": cores, "kind": "port",
            "impl": "NON-REFERENCE CPU port: the C++ host path of this build (tests/hostsim, g++ -O3 -march=native on this "
                    "box: worker threads over tet ranges for the Taylor passes and the assembly, serial remaps as in the "
                    "reference's SparseLinearDesc::apply) + MKL PARDISO (dlopen, reference settings, "
                    f"MKL_NUM_THREADS={cores})",
            "sample": f"{steps} ANM steps ({workload}, order {cfg.get('order', 20)}: whole solves from the rest "
                      f"state) in {t_step:.1f} s on {cores} threads of {avail} available (32 = the reference's "
                      f"sys-mt32 configuration); setup {setup_s:.1f} s not counted",
            "setup_seconds": {k: round(v, 2) for k, v in setup.items()},
            # the keys the reference's stats json carries (fea/main.cpp:425-431; render/gen_table_figs.py:60-66)
            "time_solve": t_step, "iter": steps, "threads": cores, "order": int(cfg.get("order", 20)),
            "pade": not cfg.get("disable_pade", False), "end_to_end": e2e,
            "seconds_per_step": {k: round(prof.get(k, 0.0) / max(steps, 1), 4) for k in tags}}


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
    if name.startswith("refine:"):
        # 'refine:<BASELINE config>:<levels>': its mesh with every tet cut into 8, `levels` times (an organic mesh at
        # scale: refine:armadillo_small:1 = 338 k tets, :2 = 2.7 M)
        _, base, levels = name.split(":")
        cfg, mesh = dfea.load_named_config(base)
        if "scale" in cfg:  # (scale before refining: setup_gravity scales a mesh object once)
            mesh.V = mesh.V * float(cfg["scale"])
            mesh._scaled = True
        fine = dfea.refine_mesh(mesh, int(levels))
        fine._scaled = True
        return cfg, fine
    return dfea.load_named_config(name)


def metric_name(workload, cfg):
    """BASELINE.json's metric, with the workload that was actually run"""
    energy = {"neohookean_c": "Neo-Hookean", "neohookean_i": "Neo-Hookean incompressible",
              "arap": "ARAP"}.get(cfg["energy_model"], cfg["energy_model"])
    mesh = {"armadillo_small": "armadillo-small: stand-in for the missing Armadillo.1"}.get(workload, workload)
    return f"ANM continuation steps/sec ({mesh}, {energy}, order {int(cfg.get('order', 20))})"


# kernels of each family (names as rocprofv3 reports them; profiles/*_kernel_stats.md)
FAMILY_KERNELS = {
    "solve": "mfk::fwd_level_tr_kernel + mfk::fwd_level_kernel + mfk::bwd_level_kernel + permute_out_dot_kernel",
    "factor": "mfk::update_kernel + gemm1/gemm2 + extend_add + panel_finalize + diag + scatter",
    "taylor": "spec_pass* (taylor_pass_kernel: EVAL0, GRAD, COEFF+BIAS per order)",
    "io": "gather_rows3_kernel (remap_out; remap_in is fused into the Taylor passes)",
    "asm": "assemble3_kernel (assemble_kernel without the row triples) + nonfinite_kernel",
    "collective": "ncclAllReduce of f(x0), the Jacobian values and b_k per order (tet-sharded mode)",
    "tail": "sanity_check_multi + Pade (multi_dot, gs_update, scale_rsqrt, lincomb2_diff_norms_multi) + "
            "next_coeff / dot / lincomb + host root finder",
}
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: dense fp64 matrix peak


def measure_families(run, one_step, cfg, stats, args, nsteps=2):
    """Two more ANM steps with device events around the phases of solve_expansion_coeffs (profile mode 2:
    no synchronisation is added).  Returns per-family time, launches, SURVEY 8(d)'s algorithmic bytes and
    the achieved rate."""
    s = run.solver
    n, nnz, T = stats["nr_unknown"], stats["jacobian_nnz"], stats["nr_tet"]
    N = int(cfg.get("order", 20))
    s.set_profile(2)
    s.pass_timing(True, fetch=False)
    it0 = s.get_nr_iter()
    t0 = time.perf_counter()
    while s.get_nr_iter() - it0 < nsteps:
        one_step()
    prof = s.profile()  # waits for the device
    wall_ms = (time.perf_counter() - t0) * 1e3
    cnt = s.profile_counts()
    lch = s.profile_launches()
    pass_ms, pass_cnt = s.pass_timing(False)
    s.set_profile(0)
    k = s.get_nr_iter() - it0
    g = lambda tag: prof.get(tag, 0.0) * 1e3 / k  # ms per step
    c = lambda tag: cnt.get(tag, 0.0) / k
    nl = lambda tag: lch.get(tag, 0.0) / k        # kernel launches per step queued inside the tag's brackets
    whole = g("solve_expansion_coeffs")
    # brackets nest: remap_out and the per-order all-reduce sit inside taylor_next_order, the all-reduces of
    # f(x0) / the Jacobian values inside taylor_order0 / build_sparse_coeff
    taylor_tags = ("taylor_order0", "jacobian", "taylor_next_order", "taylor_push")
    t = {"taylor": sum(g(x) for x in taylor_tags) - g("remap_out") - g("allreduce"),
         "io": g("remap_out"), "asm": g("build_sparse_coeff"), "factor": g("sparse_prep"),
         "solve": g("sparse_solve"), "collective": g("allreduce")}
    t["tail"] = max(whole - sum(t.values()), 0.0)
    # SURVEY.md 8(d), per ANM step
    SC = {"neohookean_c": (20, 45), "neohookean_i": (22, 45), "arap": (27, 39)}
    S_, C_ = SC.get(cfg["energy_model"], (20, 45))
    fnnz = stats["factor_nnz"]
    B = {"taylor": 8.0 * T * (S_ * N * (N - 1) / 2 + N * (C_ + S_ + 9)),
         "io": N * 432.0 * T,
         "asm": T * (81 + 144 * 2) * 8.0 + nnz * 12.0,
         "factor": fnnz * 8.0,          # the "1" of nnz(L+U)*8*(1+N)
         "solve": fnnz * 8.0 * N,       # N solves per step
         "tail": (2 + 3 * N) * 8.0 * (n + 1) + (2.5 * N * N + 44 * N) * 8.0 * (n + 1) + N * (12.0 * nnz + 16.0 * n)}
    # The Pade basis (Gram-Schmidt, 2.5 N^2 of SURVEY's vector passes: 2 units per pair for the projection, 3 for
    # the update) is built by RIDERS: extra workgroups of the remap_out gather (projections), of the solve's last
    # kernel (update + norm) and of next_coeff (scaling) -- DESIGN.md section 8.  Its bytes are booked where its
    # time lands; SANM_NO_RIDERS / SANM_GS_MODE=tail put kernels and bytes back into the tail.
    riders = not (os.environ.get("SANM_NO_RIDERS") or os.environ.get("SANM_GS_MODE") in ("tail", "side")) \
        and not cfg.get("disable_pade", False)
    if riders:
        gs = 2.5 * N * N * 8.0 * (n + 1)
        B["tail"] -= gs
        B["io"] += 0.4 * gs
        B["solve"] += 0.6 * gs
    B["collective"] = 0.0
    tail_launches = nl("solve_expansion_coeffs") - sum(nl(x) for x in taylor_tags + ("build_sparse_coeff", "sparse_prep", "sparse_solve"))
    launches = {"taylor": sum(nl(x) for x in taylor_tags) - nl("remap_out") - nl("allreduce"), "io": nl("remap_out"),
                "asm": nl("build_sparse_coeff"), "solve": nl("sparse_solve"), "factor": nl("sparse_prep"),
                "tail": tail_launches, "collective": nl("allreduce")}
    fam = {}
    for name in ("solve", "factor", "taylor", "io", "asm", "tail") + (("collective",) if t["collective"] > 0 else ()):
        ms = t[name]
        e = {"ms_per_step": ms, "share_of_step": ms / whole if whole > 0 else 0.0, "bound": "hbm",
             "algorithmic_per_step": B[name], "kernels": FAMILY_KERNELS.get(name, name),
             "achieved": B[name] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
        e["frac"] = e["achieved"] / HBM_PEAK_GBS
        L = launches[name]
        e["launches_per_step"] = L
        e["avg_launch_us"] = ms * 1e3 / L if L else None
        e["algorithmic_bytes_per_launch"] = B[name] / L if L else None
        fam[name] = e
    if "collective" in fam:
        fam["collective"]["bound"] = "xgmi"  # latency-bound all-reduces of n doubles: not priced against HBM
    # the factorisation is dense arithmetic on the fp64 matrix cores: priced in flops
    f = fam["factor"]
    f["bound"] = "mfma"
    f["flops_per_step"] = stats["factor_flops"]
    f["achieved_tflops"] = stats["factor_flops"] / (f["ms_per_step"] * 1e-3) / 1e12 if f["ms_per_step"] > 0 else 0.0
    f["peak_tflops"] = FP64_MFMA_PEAK_TFLOPS
    f["frac_mfma"] = f["achieved_tflops"] / FP64_MFMA_PEAK_TFLOPS
    if pass_cnt:
        # the pass launches themselves, one HIP event pair around each (no bracket overhead, no gaps between them)
        t = fam["taylor"]
        t["launches_per_step"] = pass_cnt / max(k, 1)  # the pass launches themselves (the bracket count includes riders' hosts)
        t["avg_launch_us_events"] = pass_ms / pass_cnt * 1e3
        t["achieved_by_launch_events"] = B["taylor"] / 1e9 / (pass_ms / pass_cnt * t["launches_per_step"] * 1e-3)
        t["frac_by_launch_events"] = t["achieved_by_launch_events"] / HBM_PEAK_GBS
    return {"families": fam, "bytes_step": sum(B.values()), "ms_step_measured": whole,
            "ms_step_wall_measured": wall_ms / max(k, 1)}


def cpu_baseline_single_thread(workload, budget_s=8.0):
    """the same baseline on ONE thread (the reference's sys-mt1.json), in a process of its own: the worker pool
    and MKL's thread count are fixed when the host backend is created"""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-threads", "1",
                            "--cpu-seconds", str(budget_s), "--workload", workload],
                           capture_output=True, text=True, timeout=600, cwd=ROOT)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
        return {"value": d["value"], "unit": d["unit"], "cores": 1, "sample": d["sample"],
                "seconds_per_step": d["seconds_per_step"], "setup_seconds": d.get("setup_seconds")}
    except Exception as e:  # noqa: BLE001 - a missing figure must not cost the bench line
        return {"error": str(e)[:200]}


def visible_gpu_count():
    """GPUs this process could hand to its ranks, WITHOUT touching the GPU or importing torch: the entries of
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set, else the kernel driver's topology
    (/sys/class/kfd: the nodes with SIMDs).  None when nothing can be said (no kfd in sight: let the ranks find out)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    base = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(base):
        return None
    n = 0
    for node in os.listdir(base):
        try:
            props = dict(ln.split()[:2] for ln in open(os.path.join(base, node, "properties")) if len(ln.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0 and int(props.get("gfx_target_version", "0")) > 0:
            n += 1
    return n


def spawn_ranks(n, argv, dist_backend="nccl"):
    """`python bench.py --gpus N` as a plain command (no torchrun around it): this process becomes the launcher of N
    fresh rank processes, one per GPU -- the reference's scaling harness is likewise one command per thread count
    (render/run_armadillo_exprs.sh:30-36).  The launcher itself makes no GPU call and does not import torch (it has
    nothing to compute; what the pool forbids is REPLACING a process that has initialised the GPU by another program
    -- fresh children like these ranks are fine either way).  The ranks are ordinary children with RANK / LOCAL_RANK /
    WORLD_SIZE in their environment and rendezvous through a FILE store in a private temporary directory (no port to
    pick, so no race for one: SANM_BENCH_RDZV_FILE); rank 0 prints the JSON line on the stdout they inherit, and the
    launcher exits with the first non-zero child status (the other ranks are then ended by PID, so a rank that died
    before or inside a collective cannot leave the rest waiting for it)."""
    import shutil
    import subprocess
    import tempfile
    if dist_backend == "nccl":
        have = visible_gpu_count()
        if have is not None and have < n:
            print(f"bench.py launcher: --gpus {n} with the nccl (RCCL) backend needs {n} visible GPUs, this machine "
                  f"shows {have}; nothing started", file=sys.stderr, flush=True)
            return 2
    tmp = tempfile.mkdtemp(prefix="sanm_bench_rdzv_")
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   SANM_BENCH_RDZV_FILE=os.path.join(tmp, "store"), SANM_BENCH_SPAWNED="1")
        for k in ("MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        procs.append(subprocess.Popen(cmd, env=env, cwd=os.getcwd()))
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"bench.py launcher: rank {procs.index(p)} exited with status {code}; ending the other "
                          "ranks", file=sys.stderr, flush=True)
                    time.sleep(2.0)  # (let them fail by themselves first: their messages are worth more)
                    for q in live:
                        if q.poll() is None:
                            q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(tmp, ignore_errors=True)
    return rc


def pmc_traffic(workload, order, family):
    """HBM-side bytes per launch of a family's kernels from the committed PMC profile (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, separate passes, corrected per MI355X_MICROARCH.md) -- not collected live"""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        entries = [pmc] + list(pmc.get("workloads", {}).values())
        for e in entries:
            if e.get("workload") == workload and e.get("order") == order:
                return e["families"][family]["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


class Leg:
    """One workload on this rank: the model, the solver (the FIRST solve timed from construction to convergence = the
    reference's time_solve), then exactly W + K completed steps with restarts, then the family measurement."""

    def __init__(self, api, workload, args, shard, dist, rank, world):
        from sanm_amd import fea as dfea
        self.api, self.workload, self.args, self.shard, self.dist = api, workload, args, shard, dist
        self.rank, self.world = rank, world
        self.cfg, mesh = load_workload(workload)
        self.run = dfea.GravityRun(api, mesh, self.cfg, shard=shard, solver_rtol=args.solver_rtol,
                                   solver_kind=args.solver_kind, profile=args.profile)
        self.x0 = self.run.model.x0()
        self.state = {"started": False, "solves": 0, "steps_per_solve": [], "cur": 0}
        self.e2e = None

    _barriers = 0

    def barrier(self):
        Leg._barriers += 1
        die = os.environ.get("SANM_BENCH_TEST_DIE_AT")  # tests only: "<rank>:<n>" -- that rank exits hard at its n-th barrier
        if die and die == f"{self.rank}:{Leg._barriers}":
            os._exit(17)
        if self.dist is not None:
            self.dist.barrier()
        device_sync()

    def max_over_ranks(self, v):
        if self.dist is None:
            return v
        import torch
        dev = "cuda" if self.args.dist_backend == "nccl" else "cpu"
        tt = torch.tensor([v], dtype=torch.float64, device=dev)
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())

    def one_step(self):
        st, run = self.state, self.run
        if not st["started"]:
            run.construct()  # first step of the first solve
            st["started"] = True
            st["cur"] = 1
            return
        s = run.solver
        if not s.converged():
            before = s.get_nr_iter()
            s.next_iter()
            if s.get_nr_iter() > before:
                st["cur"] += 1
                return
        # converged (the converged call only evaluates f(x0)): begin a new solve
        st["solves"] += 1
        st["steps_per_solve"].append(st["cur"])
        s.restart(self.x0)
        st["cur"] = 1

    def first_solve(self):
        """SURVEY 8(d) "Metric": ANM steps / wall time of the solve phase, from the solver's construction to convergence
        (fea/main.cpp:382 starts the clock before the constructor, :418-425 stops it after run_anm).  Cold: the pass
        kernels of the graph are compiled in this call unless the on-disk cache holds them."""
        run = self.run
        self.barrier()
        t0 = time.perf_counter()
        run.construct()
        t_ctor = time.perf_counter() - t0
        s = run.solver
        guard = 0
        while not s.converged() and guard < 10000:
            s.next_iter()
            guard += 1
        device_sync()
        dt = self.max_over_ranks(time.perf_counter() - t0)
        self.state.update(started=True, cur=0)
        it = int(s.get_nr_iter())
        setup = s.setup_profile()
        self.e2e = {"time_solve": dt, "iter": it, "steps_per_sec": it / dt if dt > 0 else 0.0,
                    "converged": bool(s.converged()), "time_prep": run.time_prep, "constructor_seconds": t_ctor,
                    "setup_seconds": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in setup.items()}}
        self.state["steps_per_solve"].append(it)
        self.state["solves"] += 1
        s.restart(self.x0)  # (a completed step: the steady-state part below counts from here)
        self.state["cur"] = 1
        return self.e2e

    def second_solve(self):
        """the same from a second constructor on the same model with the in-process kernel cache dropped: the pass
        kernels come from the on-disk cache, as in every later process"""
        from sanm_amd import fea as dfea
        lib = self.api.lib
        if hasattr(lib, "sanm_rtc_cache_drop_memory"):
            lib.sanm_rtc_cache_drop_memory()
        run2 = dfea.GravityRun.__new__(dfea.GravityRun)
        run2.__dict__.update(self.run.__dict__)  # same model, load and hyper-parameters; a solver of its own
        run2.solver, run2.rms, run2.time_solve = None, [], 0.0
        self.barrier()
        t0 = time.perf_counter()
        run2.construct()
        t_ctor = time.perf_counter() - t0
        s = run2.solver
        guard = 0
        while not s.converged() and guard < 10000:
            s.next_iter()
            guard += 1
        device_sync()
        dt = self.max_over_ranks(time.perf_counter() - t0)
        it = int(s.get_nr_iter())
        setup = s.setup_profile()
        out = {"time_solve": dt, "iter": it, "steps_per_sec": it / dt if dt > 0 else 0.0, "constructor_seconds": t_ctor,
               "setup_seconds": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in setup.items()}}
        del run2, s
        return out

    def timed(self, steps, warmup):
        run, st = self.run, self.state
        for _ in range(warmup):
            self.one_step()
        self.barrier()
        t0 = time.perf_counter()
        it0 = run.solver.get_nr_iter() if st["started"] else 0
        if st["started"]:
            # the K timed steps in one call (sanm_anm_run_steps: the same next_iter / restart sequence as one_step,
            # driven from C++ like the reference's own loop, fea/main.cpp:172-190, without the interpreter in between)
            st["solves"] += run.solver.run_steps(steps, self.x0)
        else:
            for _ in range(steps):
                self.one_step()
        self.barrier()
        dt = time.perf_counter() - t0
        assert run.solver.get_nr_iter() - it0 == steps
        self.dt = self.max_over_ranks(dt)
        self.steps, self.warmup = steps, warmup
        self.stats = run.solver.stats()
        # what every rank factors (the direct solver distributed by subtrees): rank 0 reports the table
        self.rank_stats = None
        if self.dist is not None:
            mine = {k: self.stats[k] for k in ("factor_flops_own", "factor_flops_top_own", "nr_subtree_own")}
            mine["front_store_bytes"] = 8 * self.stats.get("front_store_doubles", 0)  # (what THIS rank holds of the fronts)
            gathered = [None] * self.world
            self.dist.all_gather_object(gathered, mine)
            self.rank_stats = gathered
        # where the step goes: device-event brackets around the phases of two more steps.  Every rank runs them (the
        # sharded solver's collectives need all ranks); rank 0 reports.
        self.meas = measure_families(run, self.one_step, self.cfg, self.stats, self.args)
        return self

    def report(self, coll):
        """the fields of the line that describe this leg (rank 0)"""
        args, cfg, stats, meas, world, shard = self.args, self.cfg, self.stats, self.meas, self.world, self.shard
        n, nnz, T = stats["nr_unknown"], stats["jacobian_nnz"], stats["nr_tet"]
        N = int(cfg.get("order", 20))
        ms_per_step = self.dt / self.steps * 1e3
        fam = meas["families"]
        dom = max((k for k in fam if fam[k]["bound"] == "hbm"), key=lambda k: fam[k]["ms_per_step"])
        d = fam[dom]
        whole = {"algorithmic_bytes_per_step": meas["bytes_step"],
                 "achieved": meas["bytes_step"] / (ms_per_step * 1e-3) / 1e9, "unit": "GB/s",
                 "frac": meas["bytes_step"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
        wl = self.workload
        return {
            "metric": metric_name(wl, cfg),
            "value": (1 if shard else world) * self.steps / self.dt, "unit": "ANM steps/s", "n_gpus": world,
            "steps": self.steps, "warmup": self.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if (shard or world == 1) else "weak", "vs_baseline": None,
            "dtype": "f64", "backend": self.api.backend_name(),
            # the ranks the data path's collective spans as the communication library reports them
            # (ncclCommCount of the library communicator in the default tet-sharded mode)
            "rccl_ranks": coll["ranks"], "collective_impl": coll["impl"],
            "collective_ms_per_step": fam["collective"]["ms_per_step"] if "collective" in fam else 0.0,
            "data": ("real mesh Armadillo-small.1 (stand-in for the missing Armadillo.1), rest state"
                     if wl == "armadillo_small" else f"workload {wl}, rest state"),
            "config": {"workload": (f"config/{wl}.json" if ":" not in wl
                                    else f"synthetic {wl} (armadillo material, load and boundary rule)"
                                    if not wl.startswith("refine:") else
                                    f"{wl} (config/{wl.split(':')[1]}.json with every tet of its mesh cut into "
                                    f"8, {wl.split(':')[2]} time(s))")
                                   + f": {cfg['energy_model']}, order "
                                   f"{N}, T={T}, n={n}, nnz={nnz}, pade on, sanity check on",
                       "parallelism": ("tet-shard + all-reduce(b_k)/order" if shard else "replicas") if world > 1 else "single",
                       "linear_solver": "jacobi-pcg" if args.solver_kind == 0 else "multifrontal-lu",
                       "solver_stats": {k: stats[k] for k in ("factor_nnz", "factor_flops", "nr_front",
                                                              "nr_level", "max_front")},
                       # rank 0's share when the direct solver is distributed over the tree (nr_subtree > 0):
                       # its subtrees (own) and top fronts (top_own) of the whole top, the stages, and the critical path
                       # of the factorisation in flops (factor_flops / factor_flops_critical = speed-up if flops-bound)
                       "dist_solver": dict({k: stats[k] for k in ("nr_subtree", "nr_subtree_own", "factor_flops_own",
                                                                  "factor_flops_top", "factor_flops_top_own",
                                                                  "factor_flops_critical", "nr_dist_stage")},
                                           front_store_bytes=8 * stats.get("front_store_doubles", 0),
                                           exchange_bytes={"schur_per_factorisation": 8 * stats["dist_schur_doubles"],
                                                           "inbox_per_solve": 8 * stats["dist_inbox_doubles"],
                                                           "solution_per_solve": 8 * n if stats["nr_subtree"] else 0},
                                           per_rank=self.rank_stats),
                       "steps_per_solve": self.state["steps_per_solve"]},
            # the family of kernels the step spends most of its time in (HBM-bound families only; the
            # factorisation is priced against the fp64 matrix-core peak in roofline_families)
            "roofline": {"bound": "hbm", "kernel": d["kernels"], "family": dom, "achieved": d["achieved"],
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["achieved"] / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(wl, N, dom), "avg_launch_us": d["avg_launch_us"],
                         "launches_per_step": d["launches_per_step"],
                         "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
                         "algorithmic_bytes_per_step": d["algorithmic_per_step"],
                         "share_of_step": d["ms_per_step"] / max(meas["ms_step_measured"], 1e-9)},
            "roofline_families": fam,
            "roofline_whole_step": whole,
        }


def main(argv=None):
    args = parse(argv)
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.workload, args.cpu_seconds, args.cpu_threads)), flush=True)
        return None
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain `python bench.py --gpus N`: start the N ranks ourselves (before anything touches the GPU)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:] if argv is None else argv, args.dist_backend))
    hook = os.environ.get("SANM_BENCH_TEST_HOOK")
    if hook:
        # tests only (tests/test_bench_multiproc.py): a module that replaces make_api / device_sync with the host
        # harness so that the N > 1 launcher can run in the GPU-less container.  The line says which backend ran.
        import importlib
        importlib.import_module(hook).install(sys.modules[__name__])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher created WORLD_SIZE={world} ranks; reporting "
              f"n_gpus={world}", file=sys.stderr, flush=True)
    # end_to_end measures a COLD construction: the run-time compiled pass kernels go to a cache directory of this
    # process's own (and come back from it for the `cached` figure) unless the caller chose one
    jit_tmp = None
    if "SANM_JIT_CACHE_DIR" not in os.environ and not os.environ.get("SANM_NO_JIT_CACHE"):
        import tempfile
        jit_tmp = tempfile.mkdtemp(prefix="sanm_jit_bench_")
        os.environ["SANM_JIT_CACHE_DIR"] = jit_tmp
    dist = None
    # torch must load its HIP runtime before libsanm_hip.so pulls in the system one
    # (the other order leaves torch without visible devices)
    import torch  # noqa: F401
    if world > 1:
        import torch.distributed as dist
        rdzv = os.environ.get("SANM_BENCH_RDZV_FILE")  # (ranks started by spawn_ranks: a file store, no port)
        kw = dict(init_method=f"file://{rdzv}", rank=rank, world_size=world) if rdzv else {}
        if args.dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **kw)
        else:
            dist.init_process_group(args.dist_backend, **kw)

    if args.dist_backend != "nccl" and torch.cuda.device_count() > 0:
        # host-side process group with the HIP backend: ranks may share a GPU (tests on a single-GPU box)
        local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
    api = make_api(local_rank)
    shard = None
    # which collective implementation the data path uses, and how many ranks IT says it spans
    coll = {"impl": "none (single rank)" if world == 1 else "none on the data path (independent replicas)",
            "ranks": world if world == 1 else (dist.get_world_size() if dist is not None else 1)}
    if world > 1 and args.parallelism == "shard":
        from sanm_amd import dist as sdist
        native = args.dist_backend == "nccl" and not args.callback_allreduce
        if native:
            # the library's own RCCL communicator: ncclAllReduce queued on the solver's stream.  Should it fail to
            # come up on ANY rank (RCCL not loadable from C++, ...), all ranks take the callback path together.
            native = sdist.init_native_comm(api, rank, world)
            if not native:
                print(f"rank {rank}: library communicator unavailable on some rank; all-reduce through "
                      "torch.distributed", file=sys.stderr, flush=True)
        if native:
            fn = None
            # ncclCommCount of the library's communicator: the rank count RCCL itself reports
            coll = {"impl": "library ncclAllReduce queued on the solver stream (RCCL, dlopen)",
                    "ranks": api.comm_query()[0]}
        else:
            if args.dist_backend == "nccl":
                fn = sdist.make_rccl_allreduce()
            elif api.backend_name() == "hip":
                fn = sdist.make_staged_allreduce()  # several ranks on one GPU (single-GPU test boxes): staged through the host
            else:
                fn = sdist.make_host_allreduce()
            coll = {"impl": f"C-ABI callback -> torch.distributed all_reduce ({dist.get_backend()})",
                    "ranks": dist.get_world_size()}
        shard = (rank, world, fn)

    # ---- headline leg: BASELINE's metric configuration -------------------------------------------------------------
    leg = Leg(api, args.workload, args, shard, dist, rank, world)
    e2e = None
    if not args.no_end_to_end:
        e2e = {"workload": args.workload,
               "what": "one whole solve from the solver's constructor to convergence (the reference's time_solve, "
                       "fea/main.cpp:382, :418-425; iter = get_nr_ieter); `cold`: the first constructor of the process, "
                       "pass kernels compiled at run time unless setup_seconds.jit_source says otherwise; `cached`: a "
                       "second constructor on the same model with the in-process kernel cache dropped (kernels from "
                       "the on-disk cache, as in every later process)",
               "cold": leg.first_solve()}
    leg.timed(args.steps, args.warmup)
    if e2e is not None:
        e2e["cached"] = leg.second_solve()
        c, w = e2e["cold"]["setup_seconds"], e2e["cached"]["setup_seconds"]
        e2e["time_solve"], e2e["iter"] = e2e["cold"]["time_solve"], e2e["cold"]["iter"]
        # (analysis: what the constructor WAITS for the direct solver's analysis, which runs on a thread of its own beside
        # the tables -- analysis_thread is that thread's own clock)
        e2e["setup_seconds"] = {"analysis": c.get("analysis"), "analysis_thread": c.get("analysis_thread"),
                                "analysis_device": c.get("analysis_device"), "solver_vectors": c.get("solver_vectors"), "tables": round(sum(
            c.get(k, 0.0) for k in ("tet_order", "program", "remap_tables", "pattern")), 4),
            "jit_cold": c.get("jit"), "jit_cold_source": c.get("jit_source"),
            "jit_cached": w.get("jit"), "jit_cached_source": w.get("jit_source")}
    out = leg.report(coll) if rank == 0 else None
    if out is not None and e2e is not None:
        out["end_to_end"] = e2e

    # ---- at-scale legs: the same measurement on a mesh of the plausible size of the missing Armadillo.1 (338 k tets),
    # and on one where factorisation + solves are 86 % of the step (2.7 M tets: the regime the distributed direct
    # solver is built for).  One cold solve each (no cached duplicate), then W + K steps.
    default_wl = args.workload == "armadillo_small"
    legs = [("at_scale", args.at_scale_workload, "refine:armadillo_small:1", args.at_scale_steps, args.at_scale_warmup),
            ("at_scale_large", args.at_scale_large_workload, "refine:armadillo_small:2", args.at_scale_large_steps,
             args.at_scale_large_warmup)]
    leg = None  # (the headline solver's device memory goes first)
    import gc
    for key, wl2, auto_wl, k2, w2 in legs:
        if wl2 == "auto":
            wl2 = auto_wl if default_wl else "none"
        if not wl2 or wl2 == "none":
            continue
        gc.collect()
        t0 = time.perf_counter()
        leg2 = Leg(api, wl2, args, shard, dist, rank, world)
        t_model = time.perf_counter() - t0
        cold2 = leg2.first_solve() if not args.no_end_to_end else None
        leg2.timed(k2, w2)
        if rank == 0:
            r2 = leg2.report(coll)
            for k in ("higher_is_better", "vs_baseline", "dtype", "backend", "n_gpus", "unit"):
                r2.pop(k, None)
            r2["model_build_seconds"] = t_model
            if cold2 is not None:
                r2["end_to_end"] = cold2
            out[key] = r2
        del leg2

    if out is not None:
        if not args.no_cpu_baseline and world == 1 and not args.workload.startswith("block:"):
            try:
                out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_seconds, args.cpu_threads)
                cb = out["cpu_baseline"]
                cb["gpu_over_cpu"] = out["value"] / cb["value"] if cb["value"] > 0 else None
                cb["single_thread"] = cpu_baseline_single_thread(args.workload)
                if "end_to_end" in out and cb.get("end_to_end"):
                    out["end_to_end"]["cpu"] = cb["end_to_end"]
            except Exception as e:  # noqa: BLE001 -- a failing baseline leg must not cost the measured line
                import traceback
                traceback.print_exc()
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if jit_tmp:
        import shutil
        shutil.rmtree(jit_tmp, ignore_errors=True)
    return out


if __name__ == "__main__":
    main()
