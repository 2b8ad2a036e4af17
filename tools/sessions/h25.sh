set -u
mkdir -p gpurun_out/h25
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_recconv_gpu.py -q -x -k "linear_attention or recattn" 2>&1 | tail -4
timeout -k 10 600 python -m pytest tests/test_models.py tests/test_backward_gpu.py -q -x 2>&1 | tail -3
timeout -k 10 300 python tools/bench_attn.py 2>&1 | tail -12 | cut -c1-200
RCX_ATTN_MFMA=0 timeout -k 10 300 python tools/bench_attn.py 2>&1 | tail -6 | cut -c1-200
