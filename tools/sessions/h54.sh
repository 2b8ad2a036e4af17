set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_backward_gpu.py -q -x -k "dwconv_backward" 2>&1 | tail -3
