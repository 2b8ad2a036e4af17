#!/usr/bin/env python3
"""Condense raw rocprofv3 output (tools/collect_profiles.sh) into the small files kept under profiles/.

  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `python3 bench.py ...` (top kernels)
  <tag>_traffic.json       per kernel: mean FETCH_SIZE / WRITE_SIZE per launch and HBM bytes per launch with the
                           gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE counts half of a 16-B/lane
                           streaming read: doubled; both counters are in KiB)
  <tag>_bench.json         the JSON line bench.py printed in the profiled run
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag, raw, out = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(out, exist_ok=True)


def short(name):
    name = name.replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


stats = glob.glob(os.path.join(raw, "kt", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        keep = rows[:40] + [r for r in rows[40:] if "rcx::" in r["Name"]]      # the top of the list plus every kernel of this library
        for r in keep:
            w.writerow([short(r["Name"])[:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

pmc = defaultdict(lambda: defaultdict(list))
for sub in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(os.path.join(raw, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "rcx::" in r.get("Kernel_Name", ""):
                pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
kernels = []
for k, d in sorted(pmc.items()):
    fetch = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])) if "FETCH_SIZE" in d else None
    write = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"])) if "WRITE_SIZE" in d else None
    rec = {"kernel": k, "launches_sampled": len(d.get("FETCH_SIZE", [])), "FETCH_SIZE_KiB_per_launch": fetch,
           "WRITE_SIZE_KiB_per_launch": write}
    if fetch is not None and write is not None:
        rec["hbm_bytes_per_launch"] = (2.0 * fetch + write) * 1024.0
        rec["correction"] = ("2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes (gfx950: FETCH_SIZE tallies 128-B requests at 64 B; checked on the "
                             "whole-plane kernels, which read x exactly once: 1.02-1.05 x their algorithmic bytes)")
    kernels.append(rec)
# Composite units (what bench.py lists as ONE token mixer but the device runs as several launches: a RecAttn2d unit, the three-launch split schedule of
# 128 x 128 / level 4): HBM bytes per UNIT = sum over the unit's device kernels of bytes per launch x launches, divided by the units of the run.  The unit
# names and their counts come from the bench line of the same profiled run; passes = launches of a kernel that runs once per forward pass (Downsample's
# k_down7m2_cpt) or steps + warm-up + 1.
import re
bench_line = None
p_log = os.path.join(raw, "kt_bench.log")
if os.path.exists(p_log):
    for line in open(p_log):
        if line.startswith('{"metric"'):
            bench_line = json.loads(line)
if bench_line and "token_mixers" in bench_line:
    tm = bench_line["token_mixers"]
    m = re.search(r"last (\d+) warm-up steps", tm.get("measured_over", ""))
    survey = int(m.group(1)) if m else 8
    once = [k for k in kernels if "k_down7m2_cpt" in k["kernel"] and k["launches_sampled"]]
    passes = once[0]["launches_sampled"] if once else 31
    for ent in tm.get("per_kernel", []):
        name = ent["kernel"]
        if name.startswith("RecAttn2d token mixer"):
            pat = r"rcx::qkc::|rcx::lanes::k_down5_lanes|rcx::lanes::k_upadd_lanes|rcx::upcpt::k_upadd_cpt|rcx::upcpt::k_down5_cpt|rcx::cpl14::k_upadd_cpl|rcx::cpl14::k_down5_cpl|rcx::k_linattn|rcx::k_conv_generic"
        elif name.startswith("rcx split schedule"):
            pat = r"rcx::upcpt::k_down5_cpt<|rcx::upcpt::k_upadd_cpt<|rcx::lanes::k_down5_lanes|rcx::lanes::k_upadd_lanes|float, false, 3, 0, 16>|rcx::lanes::k_recconv_lanes_banded<64, 3, 16, \d, \d, \d, float>"
        else:
            continue
        comp = [k for k in kernels if re.search(pat, k["kernel"]) and k.get("hbm_bytes_per_launch") is not None]
        units = ent["launches"] / float(survey) * passes
        if comp and units > 0:
            kernels.append({"kernel": name, "composite": True, "bench_metric": bench_line.get("metric"), "units_in_run": units, "passes": passes,
                            "composed_of": [{"kernel": k["kernel"], "launches": k["launches_sampled"], "hbm_bytes_per_launch": k["hbm_bytes_per_launch"]} for k in comp],
                            "hbm_bytes_per_launch": sum(k["hbm_bytes_per_launch"] * k["launches_sampled"] for k in comp) / units,
                            "correction": "sum over the unit's device kernels of (2*FETCH_SIZE + WRITE_SIZE) x launches, divided by the units of the run"})
import importlib.util
_spec = importlib.util.spec_from_file_location("_rcx_build", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "recnext_amd", "build.py"))
_build = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_build)
json.dump({"tag": tag, "command": "python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline",
           "library_sources_sha256": _build.source_fingerprint(),      # bench.py quotes these bytes only while the kernel sources are the ones measured
           "kernels": kernels},
          open(os.path.join(out, f"{tag}_traffic.json"), "w"), indent=1)

for log in ("kt_bench.log",):
    p = os.path.join(raw, log)
    if os.path.exists(p):
        for line in open(p):
            if line.startswith('{"metric"'):
                open(os.path.join(out, f"{tag}_bench.json"), "w").write(line)
print("wrote", sorted(os.listdir(out)))
