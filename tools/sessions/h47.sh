set -u
mkdir -p gpurun_out/h47
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_backward_gpu.py tests/test_train_gpu.py -q -x 2>&1 | tail -3
timeout -k 10 300 python tools/bench_train.py --model recnext_m3 --batch 128 --steps 8 --which hip 2>&1 | tail -1 | cut -c1-200
RCX_BWD_NESTED=0 timeout -k 10 300 python tools/bench_train.py --model recnext_m3 --batch 128 --steps 8 --which hip 2>&1 | tail -1 | cut -c1-200
timeout -k 10 300 python tools/bench_backward.py 128 2>&1 | grep -v amdgpu.ids > gpurun_out/h47/blocks_fwd_bwd.jsonl; head -4 gpurun_out/h47/blocks_fwd_bwd.jsonl
