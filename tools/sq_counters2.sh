#!/bin/bash
# Run ON THE GPU BOX: two more counter passes over the bench command (issue-stall breakdown, memory-instruction backpressure), per rcx kernel.
TAG=${1:-sq2}; shift || true
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
python3 - "$OUT" <<'EOF'
import csv, glob, sys, collections
for p in ("p1", "p2"):
    f = glob.glob(sys.argv[1] + "/" + p + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "rcx::" in r["Kernel_Name"]: acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    names = sorted({c for v in acc.values() for c in v} - {"SQ_WAVE_CYCLES"})
    print("pass", p, "(fractions of SQ_WAVE_CYCLES):", " ".join(n.replace("SQ_", "") for n in names))
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:8]:
        w = c["SQ_WAVE_CYCLES"] or 1.0
        print(f"  {k[:64]:64s}", " ".join(f"{c[n]/w:9.3f}" for n in names))
EOF
