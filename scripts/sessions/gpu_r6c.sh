#!/bin/bash
# rest of the GPU suite from test_gpu_dist on + setup laps of the two at-scale workloads      usage: gpu_r6c.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests/test_gpu_dist.py tests/test_gpu_fullsize.py tests/test_matrix_dims.py tests/test_oracle_anm.py tests/test_oracle_kernels.py tests/test_oracle_ref_poly.py tests/test_pade_orth.py tests/test_rtc.py tests/test_sharded.py tests/test_tikhonov.py tests/test_vector_graphs.py -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
grep -h "4 ranks\|critical" $OUT/pytest_gpu.log | head
for WL in refine:armadillo_small:1 refine:armadillo_small:2; do
  N=$(echo $WL | tr ':' '_')
  SANM_DEBUG_SETUP=1 SANM_MF_DEBUG=1 timeout 600 python bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --at-scale-workload none --at-scale-large-workload none > $OUT/$N.json 2> $OUT/$N.err
  echo "== $WL"; grep "setup\]\|mf analysis\|svgraph\|laps" $OUT/$N.err | head -80
done
