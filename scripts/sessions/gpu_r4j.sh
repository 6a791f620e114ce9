#!/bin/bash
# round 4, session j: kernel trace of block:48 with chains (W = 1024): the level launches by grid, with their durations
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4j
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SANM_MF_DEBUG=1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o run -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload block:48 > $OUT/log.txt 2> $OUT/err.txt
cd $ROOT
grep "mf level" $OUT/err.txt > $OUT/levels.txt
python - <<PY
import csv, glob, re, collections
rows = list(csv.DictReader(open(glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0])))
agg = collections.OrderedDict()
for r in rows:
    n = re.sub(r"^void ", "", r["Kernel_Name"]); n = n.replace("sanm_hip::(anonymous namespace)::", "").replace("sanm_hip::", "")
    n = re.sub(r"\(.*", "", n)
    if "level" not in n and "big" not in n and "gemm" not in n and "extend" not in n: continue
    key = (n, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), int(r.get("LDS_Block_Size", 0) or 0))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
with open("$OUT/by_grid.txt", "w") as f:
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"{k[0]:44s} grid {k[1]:6d} x {k[2]:5d} x {k[3]:4d} lds {k[4]:6d}  n {c:5d}  avg {t/c:9.1f} us  total {t/1e3:8.2f} ms\n")
print(open("$OUT/by_grid.txt").read()[:6000])
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
