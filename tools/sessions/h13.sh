set -u
mkdir -p gpurun_out/h13
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_recconv_gpu.py -q -x -k "channel_per_lane_14x14 or golden or fp16 or repeated or full_size" 2>&1 | tail -4
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/h13/kt -- python3 tools/run_shapes.py --shapes "128,256,14,14,2;256,256,14,14,2;512,256,14,14,2;256,192,14,14,2;256,320,14,14,2" --iters 20 > gpurun_out/h13/kt.log 2>&1
python3 tools/trace_by_grid.py gpurun_out/h13/kt | tee gpurun_out/h13/xl.txt
RCX_CPL14_LDS=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/h13/kt0 -- python3 tools/run_shapes.py --shapes "256,256,14,14,2;256,192,14,14,2" --iters 20 > gpurun_out/h13/kt0.log 2>&1
python3 tools/trace_by_grid.py gpurun_out/h13/kt0 | tee gpurun_out/h13/reg.txt
rm -rf gpurun_out/h13/kt gpurun_out/h13/kt0
