set -u
mkdir -p gpurun_out/h14
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_recconv_gpu.py tests/test_models.py -q -x 2>&1 | tail -3
timeout -k 10 400 python tools/bench_configs.py > gpurun_out/h14/configs.jsonl 2> gpurun_out/h14/configs.err; cut -c1-300 gpurun_out/h14/configs.jsonl
timeout -k 10 300 python tools/bench_blocks.py --sets m3,m1,m5 --dtypes bf16,fp32 --iters 50 --json gpurun_out/h14/blocks.json > gpurun_out/h14/blocks.log 2>&1; tail -3 gpurun_out/h14/blocks.log | cut -c1-200
timeout -k 10 400 python tools/bench_train.py --model recnext_m3 --batch 128 --steps 8 --which hip > gpurun_out/h14/train.log 2>&1; tail -4 gpurun_out/h14/train.log | cut -c1-300
