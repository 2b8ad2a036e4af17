set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/bench_attn.py 2>&1 | tail -4 | cut -c1-200
timeout -k 10 600 python -m pytest tests/test_recconv_gpu.py -q -x -k "linear_attention or recattn" 2>&1 | tail -2
