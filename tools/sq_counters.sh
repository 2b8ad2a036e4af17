#!/bin/bash
# Run ON THE GPU BOX: SQ wave-state counters of the bench command (own pass, counters only), summarised per rcx kernel.
#   tools/sq_counters.sh <tag> [bench.py arguments]
TAG=${1:-sq}; shift || true
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline "$@" > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
python3 - "$OUT" <<'EOF'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/pmc/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "rcx::" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
seen = collections.Counter()
for r in csv.DictReader(open(f)):
    if "rcx::" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_WAVE_CYCLES": seen[r["Kernel_Name"]] += 1
print(f"{'kernel':70s} launches  wait_any  wait_inst  active_inst  (of wave cycles)   lds_conflict/lds_active")
for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    w = c["SQ_WAVE_CYCLES"] or 1.0
    la = c["SQ_LDS_IDX_ACTIVE"] or 1.0
    print(f"{k[:70]:70s} {seen[k]:8d}  {c['SQ_WAIT_ANY']/w:8.2f}  {c['SQ_WAIT_INST_ANY']/w:9.2f}  {c['SQ_ACTIVE_INST_ANY']/w:11.2f}                    {c['SQ_LDS_BANK_CONFLICT']/la:6.2f}")
EOF
