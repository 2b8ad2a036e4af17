set -u
mkdir -p gpurun_out/h64
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/h64/kt -- python3 tools/bench_blocks.py --sets m3_b128 --dtypes bf16 --iters 30 > gpurun_out/h64/blocks.log 2>&1
f=$(find gpurun_out/h64/kt -name "*kernel_stats.csv" | head -1); grep "rcx::" $f | cut -c1-100,130-220 | head -8 > gpurun_out/h64/kernels.txt; rm -rf gpurun_out/h64/kt
cat gpurun_out/h64/kernels.txt
