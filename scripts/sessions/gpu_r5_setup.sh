#!/bin/bash
# setup laps (SANM_DEBUG_SETUP, SANM_MF_DEBUG) of the solver's construction on both bench workloads   usage: gpu_r5_setup.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for WL in armadillo_small refine:armadillo_small:1; do
  N=$(echo $WL | tr ':' '_')
  SANM_DEBUG_SETUP=1 SANM_MF_DEBUG=1 timeout 300 python bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --at-scale-workload none --at-scale-large-workload none > $OUT/$N.json 2> $OUT/$N.err
  echo "== $WL"; grep "setup\]\|mf analysis" $OUT/$N.err | head -60
done
