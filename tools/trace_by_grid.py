#!/usr/bin/env python3
"""Average kernel durations of a rocprofv3 --kernel-trace CSV per (kernel name, grid size).  usage: trace_by_grid.py DIR [substr]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_rec"
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if sub not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"].split("(")[0][-70:], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
    acc.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (name, grid), v in acc.items():
    v = sorted(v)[: max(1, len(v) * 3 // 4)] if len(v) > 4 else v          # drop the slowest quarter (first launches)
    print("%-70s blocks=%6d  n=%3d  avg %8.2f us  min %8.2f" % (name, grid, len(v), sum(v) / len(v) / 1e3, v[0] / 1e3))
