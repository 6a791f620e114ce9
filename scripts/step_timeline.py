"""One steady-state ANM step of a rocprofv3 kernel trace as a timeline: per kernel (in order of first appearance) the
number of launches, the summed duration and the summed idle time BEFORE each launch; optionally every launch.
   python scripts/step_timeline.py <dir or kernel_trace.csv> [step index from the end, default 3] [--all]"""
import collections
import csv
import glob
import re
import sys

path = sys.argv[1]
if not path.endswith(".csv"):
    path = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 3
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = re.sub(r"^void ", "", n).replace("sanm_hip::(anonymous namespace)::", "").replace("sanm_hip::", "")
    return re.sub(r"[<(].*", "", n)


# a step begins with the order-0 pass (spec_pass0 / taylor_pass_kernel<0,...>)
def is_eval0(r):
    n = r["Kernel_Name"]
    return n.startswith("spec_pass0") or "taylor_pass_kernel<0" in n


starts = [i for i, r in enumerate(rows) if is_eval0(r)]
b, e = starts[-which - 1], starts[-which]
step = rows[b:e]
t0 = int(step[0]["Start_Timestamp"])
print(f"step of {len(step)} launches, {(int(rows[e]['Start_Timestamp']) - t0) / 1e3:.1f} us start to next start")
agg = collections.OrderedDict()
prev_end = None
for r in step:
    s, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else s - prev_end
    a = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0, 0.0])
    a[0] += 1
    a[1] += (en - s) / 1e3
    a[2] += gap / 1e3
    if "--all" in sys.argv:
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(en - s) / 1e3:7.2f}  gap {gap / 1e3:7.2f}  {short(r['Kernel_Name'])}"
              f"  grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
    prev_end = en
tail_gap = (int(rows[e]["Start_Timestamp"]) - prev_end) / 1e3
print(f"{'kernel':34s} {'n':>5s} {'dur us':>9s} {'gap-before us':>14s} {'avg dur':>8s} {'avg gap':>8s}")
td = tg = 0
for k, (n, d, g) in agg.items():
    print(f"{k:34s} {n:5d} {d:9.1f} {g:14.1f} {d / n:8.2f} {g / n:8.2f}")
    td += d
    tg += g
print(f"{'total':34s} {sum(v[0] for v in agg.values()):5d} {td:9.1f} {tg:14.1f}   (+ {tail_gap:.1f} us idle before the next step)")
