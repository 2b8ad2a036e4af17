set -u
mkdir -p gpurun_out/h3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_recconv_gpu.py -q -x -k "tiled_channel" 2>&1 | tail -25 > gpurun_out/h3/t1.log
tail -25 gpurun_out/h3/t1.log
