set -u
mkdir -p gpurun_out/h59
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/bench_configs.py cfg2,cfg4,cfg5 2>&1 | grep config > gpurun_out/h59/other_configs.jsonl
cat gpurun_out/h59/other_configs.jsonl
timeout -k 10 300 python tools/bench_train.py --model recnext_m3 --batch 128 --steps 8 --which hip 2>&1 | tail -1 | cut -c1-200
timeout -k 10 300 python tools/bench_train.py --model recnext_a3 --batch 128 --steps 8 --which hip 2>&1 | tail -1 | cut -c1-200
timeout -k 10 300 python tools/bench_backward.py 128 2>&1 | grep -v amdgpu.ids > gpurun_out/h59/blocks_fwd_bwd.jsonl; head -4 gpurun_out/h59/blocks_fwd_bwd.jsonl
