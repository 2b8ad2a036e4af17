mkdir -p gpurun_out/r5a; out=gpurun_out/r5a/nsweep.txt; : > $out
for b in tools/cpt_one_v0_s0p0k1e0 tools/cpt_one_v0_s0p0k1e0DCP2N1; do
  for n in 32 64 128 256 512; do
    echo "=== $b N=$n" >> $out
    timeout -k 10 90 $b $n 40 1 >> $out 2>&1 || exit 1
  done
done
