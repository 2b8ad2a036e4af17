set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_recconv_gpu.py tests/test_fuzz_gpu.py -q -x 2>&1 | tail -3
timeout -k 10 300 python tools/fuzz_lanes.py 120 4242 2>&1 | tail -2
