set -u
mkdir -p gpurun_out/h1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T="timeout -k 10 400"
$T python -m pytest tests/test_recconv_gpu.py -q -x --maxfail=5 2>&1 | tail -15 > gpurun_out/h1/t1.log
tail -4 gpurun_out/h1/t1.log
$T python tools/bench_blocks.py --sets m3,m1,m5 --dtypes bf16,fp32 --iters 50 --json gpurun_out/h1/blocks.json > gpurun_out/h1/blocks.log 2>&1; tail -30 gpurun_out/h1/blocks.log | cut -c1-220
RCX_CPL14=0 $T python tools/bench_blocks.py --sets m3 --dtypes bf16 --iters 50 > gpurun_out/h1/blocks_lanes.log 2>&1; tail -6 gpurun_out/h1/blocks_lanes.log | cut -c1-220
$T rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/h1/kt -- python3 tools/bench_blocks.py --sets m3 --dtypes bf16 --iters 30 > gpurun_out/h1/kt.log 2>&1
find gpurun_out/h1/kt -name "*kernel_stats.csv" | head -1 | xargs head -12 | cut -c1-200
timeout -k 10 600 python bench.py --steps 30 --warmup 10 > gpurun_out/h1/bench.json 2> gpurun_out/h1/bench.err; cut -c1-1500 gpurun_out/h1/bench.json
