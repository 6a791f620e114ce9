#!/bin/bash
# round 6, first session: re-pin config 5 on this tree (bisect over the round-5 switches, default-algorithm arbiter,
# 1e-13 sensitivity) and the default bench line as the round's starting point      usage: gpu_r6a.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python scripts/config5_bisect.py --out $OUT/r06_config5_bisect.json > $OUT/bisect.log 2>&1; echo "bisect rc=$?"
cat $OUT/bisect.log
timeout 1500 python scripts/pade_arbiter.py --cases human_arap16 --out $OUT/r06_pade_arbiter.json > $OUT/arbiter.log 2>&1; echo "arbiter rc=$?"
tail -30 $OUT/arbiter.log
timeout 900 python scripts/config5_sensitivity.py 12 1e-13 human_arap16 > $OUT/r06_sensitivity_human_arap16.json 2> $OUT/sens.err; echo "sens rc=$?"
python -c "
import json; d=json.load(open('$OUT/r06_sensitivity_human_arap16.json')); print('base', d['base'], 'same', d['same_equilibrium_1e-6'], 'other', d['other_equilibrium'], [t['steps'] for t in d['trials']])"
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"])
print("e2e", d["end_to_end"])
a=d["at_scale"]; print("at_scale", a["value"], a["ms_per_step"], a.get("roofline"), a["end_to_end"])
PY
