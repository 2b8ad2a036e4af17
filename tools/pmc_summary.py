#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter values per kernel (development tool)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "rcx"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if pat not in name:
            continue
        short = name.split("(")[0][-60:]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for cn, vals in sorted(d.items()):
        print(f"   {cn:28s} mean={sum(vals)/len(vals):16.1f}  n={len(vals)}")
