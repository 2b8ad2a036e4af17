#!/usr/bin/env python3
"""HBM bytes per launch of every rcx:: kernel from two rocprofv3 PMC passes of ANY command (tools/collect_train.sh uses it for one block's forward + backward):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d RAW/pmc_fetch -- <cmd>;  rocprofv3 --pmc WRITE_SIZE ... -d RAW/pmc_write -- <cmd>
    python3 tools/pmc_kernels.py RAW out.json [note]
Same counters, units and gfx950 correction as tools/profile_summary.py (2 * FETCH_SIZE + WRITE_SIZE, KiB); stamped with the kernel sources' sha256."""
import csv
import glob
import importlib.util
import json
import os
import sys
from collections import defaultdict

raw, out = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_rcx_build", os.path.join(root, "recnext_amd", "build.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)


def short(name):
    name = name.replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


pmc = defaultdict(lambda: defaultdict(list))
for sub in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(os.path.join(raw, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "rcx::" in r.get("Kernel_Name", ""):
                pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
kernels = []
for k, d in sorted(pmc.items()):
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
        continue
    fetch, write = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    kernels.append({"kernel": k, "launches_sampled": len(d["FETCH_SIZE"]), "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
                    "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0})
json.dump({"note": sys.argv[3] if len(sys.argv) > 3 else "", "correction": "2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes (gfx950; tools/profile_summary.py)",
           "library_sources_sha256": mod.source_fingerprint(), "kernels": kernels}, open(out, "w"), indent=1)
for k in kernels:
    print(f"{k['kernel'][:110]:110s} {k['hbm_bytes_per_launch'] / 1e6:9.1f} MB")
