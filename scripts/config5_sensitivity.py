"""How unique is the answer of BASELINE config 5 (human, ARAP, order 16)?  The free-running device continuation is
repeated on copies of the mesh whose vertex coordinates are perturbed by `eps` relative (default 1e-13: three digits
above the rounding of the input file's decimal coordinates, ten below anything physical), and the equilibria reached
are clustered by their distance to the unperturbed run's.  Prints one json line.
    python scripts/config5_sensitivity.py [n_trials=12] [eps=1e-13] [config=human_arap16]"""
import json
import sys

import numpy as np

sys.path.insert(0, '.')
import sanm_amd  # noqa: E402
from sanm_amd import fea as dfea  # noqa: E402

n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-13
name = sys.argv[3] if len(sys.argv) > 3 else "human_arap16"
api = sanm_amd.get_api()


def solve(seed):
    cfg, mesh = dfea.load_named_config(name)
    if seed is not None:
        rng = np.random.default_rng(seed)
        mesh.V = mesh.V * (1.0 + eps * rng.standard_normal(mesh.V.shape))
    run = dfea.GravityRun(api, mesh, dict(cfg)).run(max_iter=200)
    return run.vertices(), int(run.solver.get_nr_iter()), bool(run.solver.converged()), float(run.rms[-1])


V0, steps0, ok0, rms0 = solve(None)
scale = np.abs(V0).max()
out = {"config": name, "eps": eps, "base": {"steps": steps0, "converged": ok0, "rms": rms0}, "trials": []}
for s in range(n_trials):
    V, steps, ok, rms = solve(s)
    out["trials"].append({"seed": s, "steps": steps, "converged": ok, "rms": rms,
                          "vertex_rel_dist_to_base": float(np.abs(V - V0).max() / scale)})
d = np.array([t["vertex_rel_dist_to_base"] for t in out["trials"]])
out["same_equilibrium_1e-6"] = int((d <= 1e-6).sum())
out["other_equilibrium"] = int((d > 1e-3).sum())
print(json.dumps(out))
