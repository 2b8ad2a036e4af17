#!/bin/bash
OUT=gpurun_out/sq_train; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc -- python3 tools/bench_train.py --which hip --steps 3 "$@" > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
python3 - "$OUT" <<'EOF'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/pmc/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "rcx::" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": seen[k] += 1
print(f"{'kernel':84s} launches  parked  stalled  issuing  lds_conflict/lds_active  wave_cycles")
for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:28]:
    w = c["SQ_WAVE_CYCLES"] or 1.0; la = c["SQ_LDS_IDX_ACTIVE"] or 1.0
    print(f"{k[:84]:84s} {seen[k]:8d}  {c['SQ_WAIT_ANY']/w:6.2f}  {c['SQ_WAIT_INST_ANY']/w:7.2f}  {c['SQ_ACTIVE_INST_ANY']/w:7.2f}  {c['SQ_LDS_BANK_CONFLICT']/la:10.2f}  {w:14.0f}")
EOF
