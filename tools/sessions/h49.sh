set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_backward_gpu.py tests/test_train_gpu.py tests/test_models.py -q -x 2>&1 | tail -3
