set -u
mkdir -p gpurun_out/h38
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_backward_gpu.py -q -x -k "fused_backward or fp32_gradients or bf16_input or deterministic" 2>&1 | tail -15
