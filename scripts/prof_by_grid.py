"""Per-launch-shape durations of one kernel from a rocprofv3 kernel trace:
   python scripts/prof_by_grid.py <dir> <kernel name substring>"""
import collections
import csv
import glob
import sys

trace = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
by = collections.OrderedDict()
for r in csv.DictReader(open(trace)):
    if sys.argv[2] in r["Kernel_Name"]:
        key = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        by.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("grid %-18s n %5d  avg %7.2f us  total %8.2f ms" % (k, len(v), sum(v) / len(v), sum(v) / 1e3))
    tot += sum(v)
print("total %.2f ms" % (tot / 1e3))
