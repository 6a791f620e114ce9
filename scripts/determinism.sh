#!/bin/bash
# usage (GPU box): bash scripts/determinism.sh [config ...]  -> gpurun_out/determinism.json + verdict lines
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/determinism.jsonl
mkdir -p $ROOT/gpurun_out
: > $OUT
for cfg in "${@:-human_arap16 armadillo_small}"; do
 for c in $cfg; do
  python scripts/determinism.py $c --tag rocm >> $OUT 2>/dev/null
  python scripts/determinism.py $c --torch-first --tag torch >> $OUT 2>/dev/null
  SANM_NO_JIT=1 python scripts/determinism.py $c --tag nojit >> $OUT 2>/dev/null
  SANM_NO_CONV_FUSION=1 python scripts/determinism.py $c --tag jit_pop >> $OUT 2>/dev/null
  SANM_NO_CONV_FUSION=1 python scripts/determinism.py $c --torch-first --tag jit_pop_torch >> $OUT 2>/dev/null
 done
done
python - <<'PY'
import json, os
rows=[json.loads(l) for l in open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/determinism.jsonl")) if l.startswith("{")]
key=lambda r: tuple(r[k] for k in ("jac","x1","x2","x8","xN","a","pade"))
for cfg in sorted({r["config"] for r in rows}):
    t={r["tag"]: r for r in rows if r["config"]==cfg}
    for r in t.values(): print(json.dumps(r))
    same=lambda a,b: a in t and b in t and key(t[a])==key(t[b])
    print(cfg, "rocm == torch (hiprtc 7.2 vs 7.0):", same("rocm","torch"))
    print(cfg, "jit_pop == jit_pop_torch:", same("jit_pop","jit_pop_torch"))
    print(cfg, "jit_pop == nojit (run-time compiled vs ahead-of-time, same loop structure):", same("jit_pop","nojit"))
PY
