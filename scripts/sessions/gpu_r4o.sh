#!/bin/bash
# round 4, session o: the build with chains in place of big fronts + the wide backward kernel: GPU suite, bench lines
# of every workload, determinism of the BASELINE meshes (their bits must not have moved), kernel stats on block:48
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4o
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|rc=" $OUT/pytest.log | tail -3
for w in armadillo_small block:32; do
  timeout 600 python scripts/determinism.py $w --tag chains_wide >> $OUT/determinism.jsonl 2>> $OUT/determinism.err
done
tail -2 $OUT/determinism.jsonl | cut -c1-300
for w in block:32 block:48 block:60; do
  timeout 1200 python bench.py --steps 10 --warmup 3 --workload $w --no-cpu-baseline > $OUT/bench_${w/:/}.json 2>> $OUT/bench.err
  cut -c1-200 $OUT/bench_${w/:/}.json
done
bash scripts/prof_block.sh prof_r4o_block48 block:48 > $OUT/prof_block48.txt 2>&1; cat $OUT/prof_block48.txt | tail -18
