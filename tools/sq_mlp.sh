#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the fused channel-mixer kernels (own passes, counters only) on tools/bench_mlp.py.
OUT=gpurun_out/sq_mlp; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export REPS=4
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1 -- python3 tools/bench_mlp.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p2 -- python3 tools/bench_mlp.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/p3 -- python3 tools/bench_mlp.py > /dev/null 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for p in ("p1", "p2", "p3"):
    fs = glob.glob(sys.argv[1] + f"/{p}/*/*counter_collection.csv")
    if not fs: print("no output for", p); continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "k_channel_mlp" not in k: continue
        acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()): print(f"    {n:28s} {v:16.0f}")
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
