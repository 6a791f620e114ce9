#!/usr/bin/env python3
"""The distributed direct solver at a size where it matters, on ONE GPU: two ranks share cuda:0 (staged gloo all-reduce)
and run a block workload with SANM_DIST_SOLVER=1; the result must be the unsharded solve's.  Not a measurement (two
ranks on one device) -- a correctness run of the pieces / exchanges on fronts of thousands of pivots.
  python scripts/dist_on_one_gpu.py [block:32]"""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
WORKER = r"""
import json, os, sys, hashlib
import numpy as np
sys.path.insert(0, {root!r})
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
if world > 1:
    dist.init_process_group("gloo")
import bench, sanm_amd
from sanm_amd import fea as dfea, dist as sdist
api = sanm_amd.get_api(0)
cfg, mesh = bench.load_workload({name!r})
shard = (rank, world, sdist.make_staged_allreduce()) if world > 1 else None
run = dfea.GravityRun(api, mesh, dict(cfg), shard=shard).run(max_iter=60)
V = run.vertices()
st = run.solver.stats()
print("RESULT " + json.dumps(dict(rank=rank, world=world, steps=int(run.solver.get_nr_iter()), rms=float(run.rms[-1]),
                                  md5=hashlib.md5(np.ascontiguousarray(V).tobytes()).hexdigest()[:12],
                                  vsum=float(V.sum()), own=st["factor_flops_own"], top=st["factor_flops_top"],
                                  total=st["factor_flops"], nr_subtree=st["nr_subtree"])), flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
"""


def run(name, world, env_extra):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    base.update(env_extra)
    procs = []
    for r in range(world):
        env = dict(base)
        if world > 1:
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER.format(root=ROOT, name=name)], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    out = []
    for p in procs:
        so, se = p.communicate(timeout=1800)
        assert p.returncode == 0, se[-3000:]
        out.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][0][7:]))
    return out


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "block:32"
    ref = run(name, 1, {})[0]
    two = run(name, 2, {"SANM_DIST_SOLVER": "1"})
    print(json.dumps({"workload": name, "unsharded": ref, "two_ranks_distributed_solver": two}, indent=1))
    for r in two:
        assert r["steps"] == ref["steps"] and r["rms"] < 1e-10 and r["nr_subtree"] >= 2
        assert abs(r["vsum"] - ref["vsum"]) <= 1e-9 * abs(ref["vsum"])
    assert two[0]["md5"] == two[1]["md5"]
    print("distributed solver on", name, ": same continuation as the unsharded solve; ranks identical:", two[0]["md5"],
          "unsharded:", ref["md5"])
