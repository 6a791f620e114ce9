#!/usr/bin/env python3
"""Which switch moves the free-running step sequence of BASELINE config 5 (human, ARAP, order 16)?  (VERDICT r5,
next-round item 2: round 4's build took 9 steps like the oracle, round 5's takes 8.)  Each round-5 change that
re-associates a sum has an environment switch that restores the old arithmetic; the device continuation is run in a
child process per setting (the switches are read once per process) and the step count, the per-step residual / range /
Pade flag and an md5 of the first expansion's x_1 are recorded.

    python scripts/config5_bisect.py [--config human_arap16] [--out gpurun_out/r06_config5_bisect.json]
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

SETTINGS = [
    ("default", {}),
    ("asm_no_triples", {"SANM_ASM_NO_TRIPLES": "1"}),
    ("rin_no_pack", {"SANM_RIN_NO_PACK": "1"}),
    ("mf_fwd_t_0", {"SANM_MF_FWD_T": "0"}),
    ("mf_small_0", {"SANM_MF_SMALL_MIN_FRONTS": "0"}),
    ("mf_no_tile_lists", {"SANM_MF_NO_TILE_LISTS": "1"}),
    ("mf_small_serial", {"SANM_MF_SMALL_SERIAL": "1"}),
    ("all_old", {"SANM_ASM_NO_TRIPLES": "1", "SANM_RIN_NO_PACK": "1", "SANM_MF_FWD_T": "0",
                 "SANM_MF_SMALL_MIN_FRONTS": "0", "SANM_MF_NO_TILE_LISTS": "1"}),
]

CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, %(root)r)
import numpy as np
import sanm_amd
from sanm_amd import fea as dfea
api = sanm_amd.get_api(0)
cfg, mesh = dfea.load_named_config(%(name)r)
run = dfea.GravityRun(api, mesh, dict(cfg)).construct()
s = run.solver
x1 = np.ascontiguousarray(s.xt_coeffs()[1])
seq = [(float(s.residual_rms()), float(s.get_t_max_a()), bool(s.has_pade()))]
while not s.converged() and len(seq) < 200:
    run.step()
    seq.append((float(s.residual_rms()), float(s.get_t_max_a()), bool(s.has_pade())))
V = run.vertices()
print("RESULT " + json.dumps({"steps": int(s.get_nr_iter()), "converged": bool(s.converged()), "seq": seq,
                               "x1_md5": hashlib.md5(x1.tobytes()).hexdigest(),
                               "vertices_md5": hashlib.md5(np.ascontiguousarray(V).tobytes()).hexdigest()}))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="human_arap16")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_config5_bisect.json"))
    args = ap.parse_args()
    rec = {"config": args.config, "settings": []}
    for tag, env in SETTINGS:
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "name": args.config}], env=e,
                           capture_output=True, text=True, timeout=900)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        if p.returncode or not line:
            rec["settings"].append({"tag": tag, "env": env, "error": (p.stderr or p.stdout)[-800:]})
            print(tag, "FAILED", p.returncode, file=sys.stderr)
            continue
        r = json.loads(line[-1][7:])
        r.update({"tag": tag, "env": env})
        rec["settings"].append(r)
        print(tag, r["steps"], r["x1_md5"][:8], [round(t[1], 4) for t in r["seq"]], flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rec, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
