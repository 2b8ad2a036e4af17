set -u
mkdir -p gpurun_out/h37
cd tools
for b in cpt_bench cpt_bench_st4 cpt_bench_st12; do
  for i in 1 2; do echo "== $b 56"; timeout -k 10 60 ./$b 56 64 256 1 50 || exit 1; done
  echo "== $b 28"; timeout -k 10 60 ./$b 28 128 256 1 50 || exit 1
done > ../gpurun_out/h37/out.txt 2>&1
