set -u
mkdir -p gpurun_out/h22
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/bench_blocks.py --sets m3 --dtypes fp16,bf16 --iters 50 --json gpurun_out/h22/blocks_fp16.json 2>&1 | grep -v amdgpu | cut -c1-250 | tail -8
