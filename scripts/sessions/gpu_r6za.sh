#!/bin/bash
# two builds of the library: same bits on a small-front workload?  then alternating timing    usage: gpu_r6za.sh <tag> <libA> <libB>
set -u
TAG=$1; LA=$2; LB=$3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for L in $LA $LB; do
SANM_HIP_LIBRARY=$ROOT/$L SANM_MF_SMALL_MIN_FRONTS=1 python - <<'PY'
import hashlib, os, sys
sys.path.insert(0, '.')
import numpy as np, scipy.sparse as sp
import sanm_amd
from sanm_amd.api import DirectSolver
api = sanm_amd.get_api(0)
rng = np.random.default_rng(1)
k = 40
T = sp.diags([-1, 2.5, -1], [-1, 0, 1], shape=(k, k))
A3 = sp.kron(sp.kron(sp.identity(k), sp.identity(k)), T) + sp.kron(sp.kron(sp.identity(k), T), sp.identity(k)) + sp.kron(sp.kron(T, sp.identity(k)), sp.identity(k))
A = sp.csr_matrix(A3); A.sort_indices()
A.data = A.data * (1 + 0.01 * rng.standard_normal(A.nnz))
ds = DirectSolver(api, A, None)
assert ds.factor(A) == 0
b = rng.standard_normal(A.shape[0])
x = ds.solve(b)
print(os.environ["SANM_HIP_LIBRARY"].split("/")[-1], "md5", hashlib.md5(x.tobytes()).hexdigest(), "resid", float(np.abs(A @ x - b).max()), ds.stats()["nr_front"])
PY
done
run() {  # name, workload, steps, env...
  local name=$1 wl=$2 steps=$3; shift 3
  env "$@" timeout 600 python bench.py --workload $wl --steps $steps --warmup 1 --no-cpu-baseline --no-end-to-end --at-scale-workload none --at-scale-large-workload none > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY
import json
d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1]); f=d["roofline_families"]
print("$name", round(d["value"],3), round(d["ms_per_step"],3), "factor", round(f["factor"]["ms_per_step"],2), "solve", round(f["solve"]["ms_per_step"],3))
PY
}
for rep in 1 2 3; do
  run x8_A_$rep refine:armadillo_small:1 10 SANM_HIP_LIBRARY=$ROOT/$LA
  run x8_B_$rep refine:armadillo_small:1 10 SANM_HIP_LIBRARY=$ROOT/$LB
done
for rep in 1 2; do
  run x64_A_$rep refine:armadillo_small:2 3 SANM_HIP_LIBRARY=$ROOT/$LA
  run x64_B_$rep refine:armadillo_small:2 3 SANM_HIP_LIBRARY=$ROOT/$LB
done
