#!/bin/bash
# rocprofv3 average duration of the RecConv2d kernel as a function of the batch (workgroup count), to read off
# how many workgroups a CU really holds.  usage: tools/sweep_batch.sh C H LEVEL "N1 N2 ..."
C=$1; H=$2; L=$3; NS=$4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in $NS; do
  rm -rf gpurun_out/kt_sw
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_sw -- python3 tools/run_block.py --shape $n,$C,$H,$H,$L --iters 20 > /dev/null 2>&1
  python3 - "$n" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/kt_sw/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_recconv" in r["Name"]:
        print("N=%s avg %.2f us min %.2f  %s" % (sys.argv[1], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Name"][12:60]))
PY
done
