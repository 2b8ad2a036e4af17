set -u
mkdir -p gpurun_out/h12
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_recconv_gpu.py -q -x -k "detection_pyramid" 2>&1 | tail -3
timeout -k 10 300 python tools/bench_blocks.py --sets m3_coco,m3_512 --dtypes bf16 --iters 30 --eager --json gpurun_out/h12/coco_blocks.json 2>&1 | grep -v amdgpu | cut -c1-400 | tail -12
