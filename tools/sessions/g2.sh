set -u
mkdir -p gpurun_out/g2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T="timeout 300"
$T python -m pytest tests/test_recconv_gpu.py -q -k "channel_per_lane_14x14 or golden or full_size or sweep or repeated" --maxfail=8 2>&1 | tail -25 > gpurun_out/g2/t1.log
tail -4 gpurun_out/g2/t1.log
timeout 120 ./tools/ubench/issue1 > gpurun_out/g2/issue1.log 2>&1; cat gpurun_out/g2/issue1.log
$T python tools/bench_blocks.py --sets m3 --dtypes bf16,fp32 --iters 50 --json gpurun_out/g2/blocks.json > gpurun_out/g2/blocks.log 2>&1; tail -12 gpurun_out/g2/blocks.log
RCX_CPL14=0 $T python tools/bench_blocks.py --sets m3 --dtypes bf16 --iters 50 > gpurun_out/g2/blocks_lanes.log 2>&1; tail -6 gpurun_out/g2/blocks_lanes.log
$T rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/g2/kt -- python3 tools/bench_blocks.py --sets m3 --dtypes bf16 --iters 30 > gpurun_out/g2/kt.log 2>&1
find gpurun_out/g2/kt -name "*kernel_stats.csv" | head -1 | xargs head -12
