#!/bin/bash
# usage (GPU box): tools/run_cpt_variants.sh <outfile> [N=256]  -- every tools/cpt_one_v* binary: fresh and loop timing, checksum, phase stamps
out=$1; N=${2:-256}
mkdir -p "$(dirname $out)"; : > $out
for b in tools/cpt_one_v*; do
  [ -x "$b" ] || continue
  echo "=== $b" >> $out
  timeout -k 10 90 $b $N 40 1 >> $out 2>&1 || { echo "FAILED $b" >> $out; exit 1; }
  timeout -k 10 90 $b $N 40 0 2>&1 | head -2 >> $out
done
