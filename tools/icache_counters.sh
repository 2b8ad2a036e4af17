#!/bin/bash
# Run ON THE GPU BOX: instruction-cache counters of the bench command, per rcx kernel.
TAG=${1:-ic}; shift || true
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/p -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline "$@" > $OUT/p.log 2>&1 || { tail -5 $OUT/p.log; exit 1; }
python3 - "$OUT" <<'EOF'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/p/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.Counter()
for r in csv.DictReader(open(f)):
    if "rcx::" in r["Kernel_Name"]:
        acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": seen[r["Kernel_Name"]] += 1
print(f"{'kernel':66s} launches  icache_req/launch  miss_rate  dup_miss_rate  ifetch/launch  issue-stall")
for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:9]:
    n = seen[k] or 1; req = c["SQC_ICACHE_REQ"] or 1.0
    print(f"{k[:66]:66s} {n:8d}  {req/n:16.0f}  {c['SQC_ICACHE_MISSES']/req:9.3f}  {c['SQC_ICACHE_MISSES_DUPLICATE']/req:13.3f}  {c['SQ_IFETCH']/n:13.0f}  {c['SQ_WAIT_INST_ANY']/(c['SQ_WAVE_CYCLES'] or 1):10.2f}")
EOF
