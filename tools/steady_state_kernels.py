#!/usr/bin/env python3
"""Per-step kernel table of the steady state from a rocprofv3 --kernel-trace CSV (development tool).

The --stats table of a bench.py run is dominated by the GEMM-solution tuning of the first forward; this takes only the last `--steps` steps, found
by counting the launches of a kernel that runs a known number of times per step (default: the 7 x 7 RecConv2d block, twice per RecNeXt-M3 forward)."""
import argparse
import collections
import csv

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--marker", default="k_recconv_cpl7b")
ap.add_argument("--per-step", type=int, default=2)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--top", type=int, default=40)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
need = a.steps * a.per_step
if len(marks) < need + a.per_step:
    raise SystemExit(f"only {len(marks)} launches of {a.marker}")
first = marks[-need - 1] + 1            # right after the last marker launch of the step before the window
last = marks[-1]
win = rows[first:last + 1]
span = (int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])) / a.steps / 1e3
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = r["Kernel_Name"]
    agg[k][0] += d
    agg[k][1] += 1
busy = sum(v[0] for v in agg.values()) / a.steps / 1e3
print(f"window: {a.steps} steps, {span:.1f} us per step wall (marker to marker), {busy:.1f} us per step of kernel time, {len(win) / a.steps:.1f} launches per step")
for k, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
    print(f"{d / a.steps / 1e3:9.1f} us/step {c / a.steps:7.1f} x {d / c / 1e3:8.1f} us  {k[:150]}")
