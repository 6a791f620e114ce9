"""Is the device path deterministic?  The first expansion of a named config, several times in one process:
bit patterns of some series coefficients, the accepted range and the Pade decision."""
import hashlib, sys
import numpy as np
sys.path.insert(0, '.')
import sanm_amd
from sanm_amd import fea as dfea
api = sanm_amd.get_api()
name = sys.argv[1] if len(sys.argv) > 1 else "human_arap16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for r in range(reps):
    cfg, mesh = dfea.load_named_config(name)
    run = dfea.GravityRun(api, mesh, dict(cfg)).construct()
    s = run.solver
    c = s.xt_coeffs()
    J = s.jacobian_csr()
    h = lambda a: hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]
    print(r, "jac", h(J.data), "x1", h(c[1]), "x2", h(c[2]), "x8", h(c[8]), "xN", h(c[-1]), "a=%.17g" % s.get_t_max_a(), "pade", s.has_pade(), flush=True)
