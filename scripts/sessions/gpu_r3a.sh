#!/bin/bash
# round 3, first GPU session: GPU tests, determinism matrix, default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3a
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|rc=" $OUT/pytest.log | tail -3
timeout 900 bash scripts/determinism.sh human_arap16 armadillo_small > $OUT/determinism.log 2>&1
tail -8 $OUT/determinism.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cut -c1-400 $OUT/bench.json
