#!/usr/bin/env python3
"""Per-kernel time of ONE late step of a rocprofv3 --kernel-trace run (warm-up and library autotuning excluded):
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 <bench script>;  python3 tools/step_kernels.py OUT STEPS out.csv [ANCHOR PER_STEP]
STEPS = number of equal steps the run executed (warm-up included); the last 1/STEPS of the dispatches is summarised -- or, exactly, with ANCHOR (a substring
of a kernel name that runs PER_STEP times in every step, first thing in its forward): the dispatches from the first anchor launch of the second-to-last
step up to the first anchor launch of the last step = one whole step (forward, backward, optimizer) whatever the steps' dispatch counts are."""
import collections
import csv
import glob
import sys

root, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = rows[int(len(rows) * (steps - 1) / steps):]
if len(sys.argv) > 5:
    anchor, per_step = sys.argv[4], int(sys.argv[5])
    idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    if len(idx) < 2 * per_step:
        sys.exit(f"anchor {anchor!r} found {len(idx)} times: need at least {2 * per_step}")
    last = rows[idx[-2 * per_step]:idx[-per_step]]
tot, cnt = collections.Counter(), collections.Counter()
for r in last:
    name = r["Kernel_Name"].replace("void ", "")
    name = name[:name.find("(")] if "(" in name else name
    tot[name[:150]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[name[:150]] += 1
total = sum(tot.values())
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalNs", "AverageNs", "Percentage"])
    for k, v in tot.most_common(60):
        w.writerow([k, cnt[k], v, round(v / cnt[k], 1), round(100.0 * v / total, 2)])
print(f"{len(last)} dispatches, {total / 1e6:.2f} ms of kernels, wall span {(int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])) / 1e6:.2f} ms -> {out}")
