import sys, numpy as np
sys.path.insert(0, '.')
import sanm_amd
from sanm_amd import fea as dfea
from oracle import fea as ofea
api = sanm_amd.get_api()
name = sys.argv[1] if len(sys.argv) > 1 else "human_arap16"
cfg, mesh = dfea.load_named_config(name)
run = dfea.GravityRun(api, mesh, dict(cfg)).construct()
s = run.solver
cfg2, mesh2 = dfea.load_named_config(name)
omesh = ofea.TetMesh(mesh2.V, mesh2.tets, mesh2.surface_vtx)
omodel, o, _ = ofea.make_gravity_solver(omesh, cfg2)
for it in range(12):
    line = "step %d | dev rms=%.6e a=%.8g t=%.8g pade=%d | ora rms=%.6e a=%.8g t=%.8g pade=%d" % (
        it, s.residual_rms(), s.get_t_max_a(), s.get_t_upper(), s.has_pade(),
        o.residual_rms, o.t_max_a, o.t_max, o.pade is not None)
    if not s.converged() and not o.converged:
        cd, co = s.xt_coeffs(), o.xt_coeffs
        if len(cd) > 2 and len(co) > 2:
            line += " | coeff rel diff " + " ".join("%.1e" % (np.abs(cd[i] - co[i]).max() / np.abs(co[i]).max()) for i in (1, 2, 8, 16))
    print(line, flush=True)
    if s.converged() and o.converged:
        break
    if not s.converged(): s.next_iter()
    if not o.converged: o.next_iter()
print("dev iters", s.get_nr_iter(), "oracle iters", o.get_nr_iter())
