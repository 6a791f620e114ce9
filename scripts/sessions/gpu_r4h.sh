#!/bin/bash
# round 4, session h: records -- config 5 sensitivity of the final build, block:60, the distributed solver on block:32
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4h
mkdir -p $OUT
cd $ROOT
timeout 900 python scripts/dist_on_one_gpu.py block:32 > $OUT/dist_block32.json 2> $OUT/dist_block32.err; tail -3 $OUT/dist_block32.json; tail -3 $OUT/dist_block32.err
timeout 900 python bench.py --steps 3 --warmup 1 --workload block:60 --no-cpu-baseline > $OUT/bench_block60.json 2> $OUT/bench_block60.err; cut -c1-300 $OUT/bench_block60.json
timeout 1200 python scripts/config5_sensitivity.py human_arap16 12 > $OUT/sens_human.json 2> $OUT/sens_human.err; tail -5 $OUT/sens_human.json
