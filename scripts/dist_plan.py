#!/usr/bin/env python3
"""What stage 1 of the distributed direct solver (multifrontal.cpp, MfSchedule::Dist) would give on a workload at G
ranks: the analysis run as rank 0 of G on the workload's Jacobian pattern (vertex adjacency of the free dofs: no
numerics, CPU only through the host harness), the flops of the replicated top and of every rank's subtrees, the size of
the exchanges -- and, from rates measured on one MI355X (profiles/), the factorisation and solve times they predict.

  python scripts/dist_plan.py --workload block:48 --world 2,4,8 [--out profiles/r04_dist_plan_block48.json]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="block:32")
    ap.add_argument("--world", default="2,4,8")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import bench
    from sanm_amd import api as A, fea as dfea
    from tests.hostsim import get_hostsim_api
    api = get_hostsim_api()
    cfg, mesh = bench.load_workload(args.workload)
    fixed, _ = dfea.setup_gravity(api, mesh, cfg)
    P, coords = jacobian_pattern(mesh, fixed)
    print(f"{args.workload}: {mesh.nr_tet} tets, n = {P.shape[0]}, nnz = {P.nnz}", flush=True)
    rec = {"workload": args.workload, "nr_tet": int(mesh.nr_tet), "n": int(P.shape[0]), "plans": []}
    os.environ["SANM_DIST_SOLVER"] = "1"
    for w in [int(v) for v in args.world.split(",")]:
        os.environ["SANM_MF_PLAN_WORLD"] = str(w)
        s = A.DirectSolver(api, P, coords)
        plan = s.dist_plan()
        del s
        tot, top = plan["total_flops"], plan["top_flops"]
        own = max(plan["rank_flops"])
        plan["factor_flops_of_the_slowest_rank"] = top + own
        plan["factor_speedup_if_flops_bound"] = tot / (top + own)
        plan["solve_bytes_speedup"] = plan["factor_nnz"] / (plan["top_nnz"] + max(plan["rank_nnz"]))
        rec["plans"].append(plan)
        print(f"world {w}: {plan['nr_subtree']} subtrees, top {top / tot:.1%} of {tot / 1e12:.2f} TFLOP, slowest rank's "
              f"subtrees {own / tot:.1%} (imbalance {plan['imbalance']:.2f}) -> factor x{tot / (top + own):.2f}, solve bytes "
              f"x{plan['solve_bytes_speedup']:.2f} (top {plan['top_nnz'] / plan['factor_nnz']:.1%} of the entries); "
              f"Schur exchange {plan['schur_exchange_doubles'] * 8 / 1e9:.2f} GB, inbox {plan['inbox_exchange_doubles'] * 8 / 1e6:.2f} MB "
              f"per solve", flush=True)
    os.environ.pop("SANM_MF_PLAN_WORLD", None)
    if args.out:
        json.dump(rec, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
