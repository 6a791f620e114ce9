#!/bin/bash
# round 4, session t: timeline of one factorisation of block:48 with look-ahead (do the two queues overlap?)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4t
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SANM_MF_LA_CUS=${LA_CUS:-0}
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o run -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --workload block:48 > $OUT/log.txt 2> $OUT/err.txt
cd $ROOT
python - <<PY
import csv, glob, re
rows = list(csv.DictReader(open(glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0])))
print(list(rows[0].keys()))
ev = []
for r in rows:
    n = re.sub(r"^void ", "", r["Kernel_Name"]); n = n.replace("sanm_hip::(anonymous namespace)::", "").replace("sanm_hip::", "").replace("mfk::", "")
    n = re.sub(r"\(.*", "", n)
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?"), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])))
ev.sort()
# the last factorisation: from the last scatter_kernel on
idx = [i for i, e in enumerate(ev) if e[2].startswith("scatter_kernel")][-1]
t0 = ev[idx][0]
with open("$OUT/timeline.txt", "w") as f:
    for e in ev[idx:]:
        if e[2].startswith("fwd_level") or e[2].startswith("permute_in"): break
        f.write(f"{(e[0]-t0)/1e3:10.1f} us  +{(e[1]-e[0])/1e3:8.1f}  q{e[3]}  {e[2]:28s} {e[4]}x{e[5]}x{e[6]}\n")
lines = open("$OUT/timeline.txt").read().splitlines()
print(len(lines), "launches")
# show the region around the biggest tall launches
tl = [i for i, l in enumerate(lines) if "gemm2_tall" in l]
for i in tl[3:5]:
    print("\n".join(lines[max(0, i - 4): i + 22]))
    print("-----")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
