set -u
mkdir -p gpurun_out/g1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_recconv_gpu.py -q -k "channel_per_lane_14x14 or golden or full_size or sweep" --maxfail=8 2>&1 | tail -25 > gpurun_out/g1/t1.log
tail -5 gpurun_out/g1/t1.log
./tools/ubench/d16_probe > gpurun_out/g1/d16.log 2>&1; cat gpurun_out/g1/d16.log
./tools/ubench/issue1 > gpurun_out/g1/issue1.log 2>&1; cat gpurun_out/g1/issue1.log
python tools/bench_blocks.py --sets m3 --dtypes bf16,fp32 --iters 50 --json gpurun_out/g1/blocks.json > gpurun_out/g1/blocks.log 2>&1; tail -12 gpurun_out/g1/blocks.log
RCX_CPL14=0 python tools/bench_blocks.py --sets m3 --dtypes bf16 --iters 50 > gpurun_out/g1/blocks_lanes.log 2>&1; tail -6 gpurun_out/g1/blocks_lanes.log
