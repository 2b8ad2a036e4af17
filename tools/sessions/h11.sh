set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_backward_gpu.py -q -x 2>&1 | grep -v "^$" | tail -30 | cut -c1-250
timeout -k 10 300 python -m recnext_amd.speed --help 2>&1 | tail -15
