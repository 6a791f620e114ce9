"""Step counts and coefficient agreement device vs oracle on a named config, with and without iterative refinement
in the device's direct solver.  python scripts/parity_refine.py human_arap16"""
import json, os, sys
import numpy as np
sys.path.insert(0, '.')
import sanm_amd
from sanm_amd import fea as dfea
from oracle import fea as ofea
api = sanm_amd.get_api()
name = sys.argv[1] if len(sys.argv) > 1 else "human_arap16"
cfg2, mesh2 = dfea.load_named_config(name)
omesh = ofea.TetMesh(mesh2.V, mesh2.tets, mesh2.surface_vtx)
omodel, o, _ = ofea.make_gravity_solver(omesh, cfg2)
ora = [dict(rms=o.residual_rms, a=o.t_max_a, t=o.t_max, pade=o.pade is not None, coeffs=[c.copy() for c in o.xt_coeffs])]
while not o.converged:
    o.next_iter()
    ora.append(dict(rms=o.residual_rms, a=o.t_max_a, t=o.t_max, pade=o.pade is not None,
                    coeffs=[c.copy() for c in o.xt_coeffs] if not o.converged else None))
Vo = omodel.lt_inp.full_vertices(o.get_x() if hasattr(o, "get_x") else o.xt0[:-1])
print("oracle steps", o.get_nr_iter())
out = {"oracle_steps": int(o.get_nr_iter())}
for refine in (0, 1):
    cfg, mesh = dfea.load_named_config(name)
    run = dfea.GravityRun(api, mesh, dict(cfg), solver_refine=refine).construct()
    s = run.solver
    k = 0
    while True:
        line = "refine %d step %d | dev rms=%.6e a=%.8g pade=%d" % (refine, k, s.residual_rms(), s.get_t_max_a(), s.has_pade())
        if k < len(ora):
            line += " | ora rms=%.6e a=%.8g pade=%d" % (ora[k]["rms"], ora[k]["a"], ora[k]["pade"])
            if not s.converged() and ora[k]["coeffs"] is not None:
                cd = s.xt_coeffs()
                co = ora[k]["coeffs"]
                line += " | coeff rel diff " + " ".join("%.1e" % (np.abs(cd[i] - co[i]).max() / np.abs(co[i]).max()) for i in (1, 2, 8, len(co) - 1))
        print(line, flush=True)
        if s.converged():
            break
        s.next_iter()
        k += 1
    V = run.vertices()
    out["device_steps_refine%d" % refine] = int(s.get_nr_iter())
    out["vertex_rel_err_refine%d" % refine] = float(np.abs(V - Vo).max() / np.abs(Vo).max())
    print("device steps", s.get_nr_iter(), "vertex rel err", out["vertex_rel_err_refine%d" % refine])
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/parity_%s.json" % name, "w"))
print(json.dumps(out))
