set -u
mkdir -p gpurun_out/h53
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_backward_gpu.py tests/test_train_gpu.py tests/test_models.py -q -x 2>&1 | tail -3
timeout -k 10 300 python tools/bench_train.py --model recnext_a3 --batch 128 --steps 6 --which hip 2>&1 | tail -1 | cut -c1-200
timeout -k 10 300 python tools/bench_train.py --model recnext_m3 --batch 128 --steps 8 --which hip 2>&1 | tail -1 | cut -c1-200
