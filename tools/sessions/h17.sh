set -u
mkdir -p gpurun_out/h17
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_recconv_gpu.py -q -x -k "tiled_channel or fp16 or golden or full_size or sweep" 2>&1 | tail -4
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/h17/kt -- python3 tools/run_shapes.py --shapes "256,96,28,28,3;256,160,28,28,3;256,80,28,28,3;255,96,28,28,3" --iters 20 > gpurun_out/h17/kt.log 2>&1
python3 tools/trace_by_grid.py gpurun_out/h17/kt | tee gpurun_out/h17/img2.txt
RCX_CPT=w timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/h17/kt0 -- python3 tools/run_shapes.py --shapes "256,96,28,28,3;256,160,28,28,3;256,80,28,28,3" --iters 20 > gpurun_out/h17/kt0.log 2>&1
python3 tools/trace_by_grid.py gpurun_out/h17/kt0 | tee gpurun_out/h17/banded.txt
rm -rf gpurun_out/h17/kt gpurun_out/h17/kt0
