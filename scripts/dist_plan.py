#!/usr/bin/env python3
"""What the distributed direct solver (multifrontal.cpp, MfSchedule::Dist: every front one owner, the top of the tree
mapped proportionally onto rank sets, stages with exchanges between them) gives on a workload at G ranks: the analysis
run as rank 0 of G on the workload's Jacobian pattern (vertex adjacency of the free dofs: no numerics, CPU only through
the host harness), the flops and factor entries of every rank in every stage, the critical path, the size of the
transfers -- and, from the family times of a single-GPU bench line of the same workload (profiles/), the step they predict.

  python scripts/dist_plan.py --workload refine:armadillo_small:2 --world 2,4,8 \
      --bench-json profiles/r05_bench_refine_armadillo_small_2.json --out profiles/r06_dist_plan_armadillo_x64.json"""
import argparse
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402


def jacobian_pattern(mesh, fixed):
    """3x3 blocks for every pair of free vertices that share a tet (the pattern AnmDriver's JacobianPattern finds)"""
    nv = mesh.nr_vertices
    T = mesh.tets
    rows = np.repeat(T, 4, axis=1).ravel()
    cols = np.tile(T, (1, 4)).ravel()
    A = sp.csr_matrix((np.ones(rows.size, dtype=np.int8), (rows, cols)), shape=(nv, nv))
    A.sum_duplicates()
    free = ~fixed[:, 0]
    A = A[free][:, free]
    B = sp.kron(A, np.ones((3, 3), dtype=np.int8), format="csr")
    coords = np.repeat(mesh.V[free], 3, axis=0)
    return B, coords


# xGMI: 7 links x ~153 GB/s bidirectional per GPU (MI355X_MICROARCH.md); one pair of ranks shares one link.  The model
# takes 50 GB/s per direction for a point-to-point transfer (two thirds of the link's one-way peak) and 15 us for a
# small collective on the solver's stream.
P2P_GBS = 50.0
COLL_US = 15.0


def _makespan(work, xfers, S, G, xfer_ms, reverse=False):
    """end of the last (stage, rank) when a rank runs its stages in order and stage s of rank r starts once the stages
    that send to it are done and their data has arrived.  work[s][r] in ms; xfers: the plan's Schur transfers (the edges
    of the tree between owners); reverse: the backward sweep (root stage first, data flows from dst back to src)."""
    end = np.zeros((S, G))
    order = range(S - 1, -1, -1) if reverse else range(S)
    for st in order:
        for r in range(G):
            prev = st + 1 if reverse else st - 1
            t0 = end[prev, r] if 0 <= prev < S else 0.0
            for x in xfers:
                if not reverse and x["stage"] == st and x["dst"] == r:
                    t0 = max(t0, end[x["src_stage"], x["src"]] + xfer_ms(x))
                if reverse and x["src_stage"] == st and x["src"] == r:
                    t0 = max(t0, end[x["stage"], x["dst"]] + xfer_ms(x))
            end[st, r] = t0 + work[st][r]
    return float(end.max())


def predict(plan, fam, n, nnz, order):
    """step time at G ranks from the plan's per-stage tables and the single-GPU family times `fam` (ms per step from a
    bench line of the same workload).  Factorisation: every (stage, rank) takes the measured time in proportion to its
    flops, the stages of a rank run in order, a stage waits for the Schur complements it receives (point-to-point at
    P2P_GBS).  Solves: the same walk forward and backward with the factor entries as the work, a small collective per
    stage boundary and one gather of n doubles at the end.  Taylor / io / assembly by tets + their all-reduces (ring:
    2 (G - 1) / G of the vector over one link); tail replicated."""
    G, S = plan["world"], plan["nr_stage"]
    sf, sn = np.array(plan["stage_flops"]), np.array(plan["stage_nnz"])
    xf = plan["schur_transfers"]
    fwork = sf / plan["total_flops"] * fam["factor"]
    factor = _makespan(fwork, xf, S, G, lambda x: x["doubles"] * 8 / (P2P_GBS * 1e9) * 1e3 + COLL_US * 1e-3)
    no_xfer = _makespan(fwork, xf, S, G, lambda x: 0.0)
    nsolve = order + 1
    swork = sn / plan["factor_nnz"] * fam["solve"] / nsolve / 2  # one sweep of one solve
    small = lambda x: COLL_US * 1e-3
    gather_ms = n * 8 / (P2P_GBS * 1e9) * 1e3
    one_solve = _makespan(swork, xf, S, G, small) + _makespan(swork, xf, S, G, small, reverse=True) + gather_ms
    solve = nsolve * one_solve
    ring = 2.0 * (G - 1) / G
    allred = ((order + 1) * n * 8 + nnz * 8) * ring / (P2P_GBS * 1e9) * 1e3 + (order + 2) * COLL_US * 1e-3
    # the same all-reduces done directly over the G - 1 links of the full mesh (reduce-scatter + all-gather, a chunk of
    # 1 / G per peer and direction at a time) instead of a ring through one link: what RCCL can do on xGMI, not charged
    # in `step_ms`
    allred_mesh = ((order + 1) * n * 8 + nnz * 8) * 2.0 / G / (P2P_GBS * 1e9) * 1e3 + 2 * (order + 2) * COLL_US * 1e-3
    compute_sharded = (fam["taylor"] + fam["io"] + fam["asm"]) / G
    sharded = compute_sharded + allred
    step1 = sum(fam.values())
    step = factor + solve + sharded + fam["tail"]
    step_mesh = factor + solve + compute_sharded + allred_mesh + fam["tail"]
    return {"factor_ms": factor, "factor_ms_without_transfers": no_xfer, "solve_ms": solve,
            "taylor_io_asm_ms": sharded, "allreduce_ms_ring_one_link": allred, "allreduce_ms_full_mesh": allred_mesh,
            "step_ms_with_full_mesh_allreduce": step_mesh, "speedup_with_full_mesh_allreduce": step1 / step_mesh,
            "tail_ms": fam["tail"],
            "step_ms": step, "step_ms_one_gpu": step1, "speedup": step1 / step,
            "factor_speedup": fam["factor"] / factor, "solve_speedup": fam["solve"] / solve,
            "factor_plus_solves_speedup": (fam["factor"] + fam["solve"]) / (factor + solve)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="block:32")
    ap.add_argument("--world", default="2,4,8")
    ap.add_argument("--out", default=None)
    ap.add_argument("--bench-json", default=None,
                    help="a single-GPU bench line of the same workload (profiles/*.json): its family times make the "
                         "prediction; without it the plan's flop / entry shares only")
    ap.add_argument("--bench-key", default=None, help="leg of that line (at_scale, at_scale_large); default: the headline")
    args = ap.parse_args()
    import bench
    from sanm_amd import api as A, fea as dfea
    from tests.hostsim import get_hostsim_api
    api = get_hostsim_api()
    cfg, mesh = bench.load_workload(args.workload)
    fixed, _ = dfea.setup_gravity(api, mesh, cfg)
    P, coords = jacobian_pattern(mesh, fixed)
    print(f"{args.workload}: {mesh.nr_tet} tets, n = {P.shape[0]}, nnz = {P.nnz}", flush=True)
    rec = {"workload": args.workload, "nr_tet": int(mesh.nr_tet), "n": int(P.shape[0]), "plans": [],
           "model": {"p2p_GBs": P2P_GBS, "small_collective_us": COLL_US}}
    fam = None
    if args.bench_json:
        d = json.loads(open(args.bench_json).read().strip().splitlines()[-1])
        if args.bench_key:
            d = d[args.bench_key]
        fam = {k: v["ms_per_step"] for k, v in d["roofline_families"].items() if k != "collective"}
        rec["one_gpu"] = {"source": args.bench_json, "key": args.bench_key, "ms_per_step": d["ms_per_step"],
                          "families_ms": fam}
    os.environ["SANM_DIST_SOLVER"] = "1"
    order = int(cfg.get("order", 20))
    for w in [int(v) for v in args.world.split(",")]:
        os.environ["SANM_MF_PLAN_WORLD"] = str(w)
        s = A.DirectSolver(api, P, coords)
        plan = s.dist_plan()
        del s
        tot, top, crit = plan["total_flops"], plan["top_flops"], plan["critical_flops"]
        plan["factor_speedup_if_flops_bound"] = tot / crit
        if fam:
            plan["prediction"] = predict(plan, fam, P.shape[0], P.nnz, order)
        rec["plans"].append(plan)
        print(f"world {w}: {plan['nr_subtree']} subtrees, {plan['nr_stage']} stages, top {top / tot:.1%} of {tot / 1e12:.2f} "
              f"TFLOP (mapped onto the ranks), subtree imbalance {plan['imbalance']:.2f}, critical path {crit / tot:.1%} -> "
              f"factor x{tot / crit:.2f} if flops-bound; Schur transfers {plan['schur_exchange_doubles'] * 8 / 1e9:.2f} GB in "
              f"{len(plan['schur_transfers'])} point-to-point pieces", flush=True)
        if fam:
            p = plan["prediction"]
            print(f"   predicted: factor {fam['factor']:.1f} -> {p['factor_ms']:.1f} ms ({p['factor_ms_without_transfers']:.1f} "
                  f"without the transfers), solves {fam['solve']:.1f} -> {p['solve_ms']:.1f}, Taylor/io/asm -> "
                  f"{p['taylor_io_asm_ms']:.1f}, step "
                  f"{p['step_ms_one_gpu']:.1f} -> {p['step_ms']:.1f} ms: x{p['speedup']:.2f} (factor + solves x"
                  f"{p['factor_plus_solves_speedup']:.2f}); with the all-reduces over the full mesh instead of a ring "
                  f"through one link ({p['allreduce_ms_ring_one_link']:.1f} -> {p['allreduce_ms_full_mesh']:.1f} ms): "
                  f"{p['step_ms_with_full_mesh_allreduce']:.1f} ms, x{p['speedup_with_full_mesh_allreduce']:.2f}", flush=True)
    os.environ.pop("SANM_MF_PLAN_WORLD", None)
    if args.out:
        json.dump(rec, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
