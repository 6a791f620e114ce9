"""GPU idle time between consecutive kernels of a rocprofv3 kernel trace, grouped by (previous, next) kernel:
   python scripts/trace_gaps.py gpurun_out/prof_<tag>/stats/run_kernel_trace.csv [min_gap_us]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 3000
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = re.sub(r"^void ", "", n).replace("sanm_hip::(anonymous namespace)::", "").replace("sanm_hip::", "")
    return re.sub(r"[<(].*", "", n)


busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
cnt, tot = collections.Counter(), collections.Counter()
allgap = 0
for a, b in zip(rows[:-1], rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g > 5e5:  # between steps / setup
        continue
    allgap += max(g, 0)
    if g > thr:
        k = (short(a["Kernel_Name"]), short(b["Kernel_Name"]))
        cnt[k] += 1
        tot[k] += g
print(f"busy {busy / 1e6:.1f} ms, idle in gaps < 0.5 ms: {allgap / 1e6:.1f} ms")
for k, v in tot.most_common(25):
    print(f"{k[0]:28s} -> {k[1]:28s} n={cnt[k]:5d} tot={v / 1e6:7.2f} ms avg={v / cnt[k] / 1e3:6.1f} us")
