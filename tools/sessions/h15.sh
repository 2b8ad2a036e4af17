set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_backward_gpu.py -q -x -k "linear_attention_core_backward or recattn2d_training" 2>&1 | grep -v "^$" | tail -25 | cut -c1-220
timeout -k 10 600 python -m pytest tests/test_models.py tests/test_train_gpu.py -q -x 2>&1 | tail -3
timeout -k 10 300 python tools/bench_train.py --model recnext_a3 --batch 128 --steps 6 --which hip 2>&1 | tail -1 | cut -c1-300
