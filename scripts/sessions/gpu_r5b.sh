#!/bin/bash
# round 4, session 5b: coordinate-axis cut candidates for all sets of 1000+ in big problems: GPU suite, md5s of the BASELINE meshes (must not move),
# bench lines of the blocks
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5b
mkdir -p $OUT
cd $ROOT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|rc=" $OUT/pytest.log | tail -3
for w in armadillo_small bob human_arap16 block:32; do
  timeout 600 python scripts/determinism.py $w --tag axis_cuts >> $OUT/determinism.jsonl 2>> $OUT/determinism.err
done
cut -c1-170 $OUT/determinism.jsonl
for w in block:32 block:48 block:60; do
  timeout 1200 python bench.py --steps 10 --warmup 3 --workload $w --no-cpu-baseline > $OUT/bench_${w/:/}.json 2>> $OUT/bench.err
  cut -c1-200 $OUT/bench_${w/:/}.json
done
bash scripts/prof_block.sh prof_r5b_block48 block:48 > $OUT/prof_block48.txt 2>&1; tail -16 $OUT/prof_block48.txt
