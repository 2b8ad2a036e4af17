set -u
mkdir -p gpurun_out/h5
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_recconv_gpu.py -q -k "tiled_channel" 2>&1 | grep -v "^\s*$" | grep "FAILED\|passed\|failed\|assert np\|^E  " | cut -c1-300 | tail -40
