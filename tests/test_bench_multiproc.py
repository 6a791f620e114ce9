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
                  "--no-cpu-baseline", "--dist-backend", "gloo"])
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


def test_two_rank_replicas_gloo():
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER.format(root=ROOT)], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1, outs[0][0]
    d = json.loads(lines[0])
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")], "only rank 0 prints"
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["scaling"] == "weak" and d["config"]["parallelism"] == "replicas"
    # whole-job value = N * K / max-over-ranks time
    assert abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 3 / 1e3)) < 1e-6 * d["value"]
    assert d["roofline"]["kernel"] == "taylor_pass_kernel"
