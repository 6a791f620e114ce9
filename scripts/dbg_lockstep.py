import json, sys, numpy as np
sys.path.insert(0,'.')
import sanm_amd
from tests.lockstep import LockStep
from oracle import fea as ofea
from sanm_amd import fea as dfea
api=sanm_amd.get_api()
gold=json.load(open("tests/golden/anm_cuboid_nc.json"))
run=dfea.GravityRun(api,dfea.make_cuboid(*gold["dims"],gold["spacing"]),dict(gold["config"]),solver_rtol=1e-15).construct()
_,o,_=ofea.make_gravity_solver(ofea.make_cuboid(*gold["dims"],gold["spacing"]),gold["config"])
ls=LockStep(run,o)
try:
    ls.run_to_convergence()
    print("OK steps", ls.nr_steps)
except AssertionError as e:
    print("FAIL", str(e)[:300])
    dd=run.solver.pade_diag(); od=o.pade_diags[-1]
    print("device diag", {k:v for k,v in dd.items() if k!="d"})
    print("oracle diag", {k:v for k,v in od.items() if k!="d"})
    print("d dev", dd["d"][:6]); print("d orc", od["d"][:6])
for r in ls.steps: print(r)
