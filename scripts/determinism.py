"""Reproducibility of the device arithmetic across run-time compilers and kernel flavours.

One variant per process (the hiprtc that gets loaded is decided by what the process imports first):
    python scripts/determinism.py <config> [--torch-first] [--tag NAME]
prints ONE json line: md5 of the Jacobian values and of x_1, x_2, x_8, x_N of the first expansion, the accepted
range and the Pade outcome.  scripts/determinism.sh runs the variant matrix and compares:
    rocm    : /opt/rocm's hiprtc (7.2) builds the pass kernels                      } must be bit-identical
    torch   : torch imported first => the wheel's hiprtc (7.0)                      }  (library built with
    nojit   : SANM_NO_JIT=1, the interpreter kernels compiled ahead of time by hipcc } -ffp-contract=off, fma()
    jit_pop : run-time kernels with the per-operator convolution loops               }  explicit)
              (SANM_NO_CONV_FUSION=1: the loop structure of the interpreter kernels)
The default run-time kernels (fused convolution loop: another summation order by design) are reported beside them."""
import hashlib
import json
import sys

args = sys.argv[1:]
if "--torch-first" in args:
    import torch  # noqa: F401  (loads the wheel's libhiprtc before libsanm_hip.so asks for one)
    args.remove("--torch-first")
tag = "run"
if "--tag" in args:
    i = args.index("--tag")
    tag = args[i + 1]
    del args[i:i + 2]
import numpy as np  # noqa: E402

sys.path.insert(0, '.')
import sanm_amd  # noqa: E402
from sanm_amd import fea as dfea  # noqa: E402

api = sanm_amd.get_api()
name = args[0] if args else "human_arap16"
if ":" in name:  # a synthetic workload of bench.py (block:N, cuboid:x,y,z)
    import bench
    cfg, mesh = bench.load_workload(name)
else:
    cfg, mesh = dfea.load_named_config(name)
run = dfea.GravityRun(api, mesh, dict(cfg)).construct()
s = run.solver
c = s.xt_coeffs()
J = s.jacobian_csr()
h = lambda a: hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
d = s.pade_diag()
print(json.dumps({"tag": tag, "config": name, "jac": h(J.data), "x1": h(c[1]), "x2": h(c[2]), "x8": h(c[8]), "xN": h(c[-1]),
                  "a": s.get_t_max_a().hex(), "pade": bool(s.has_pade()),
                  "margin_left": d["probes"][0][1] if d["probes"] else None}), flush=True)
