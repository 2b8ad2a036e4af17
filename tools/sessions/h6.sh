set -u
mkdir -p gpurun_out/h6
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T="timeout -k 10 400"
$T python -m pytest tests/test_recconv_gpu.py -q -x 2>&1 | tail -4 > gpurun_out/h6/t1.log; tail -4 gpurun_out/h6/t1.log
$T rocprofv3 --kernel-trace --output-format csv -d gpurun_out/h6/kt -- python3 tools/run_shapes.py --shapes "256,64,56,56,4;256,128,28,28,3;256,48,56,56,4;256,96,28,28,3;256,80,56,56,4;256,160,28,28,3;128,64,56,56,4;512,128,28,28,3" --iters 20 > gpurun_out/h6/kt.log 2>&1
python3 tools/trace_by_grid.py gpurun_out/h6/kt | tee gpurun_out/h6/cpt.txt
RCX_CPT=0 $T rocprofv3 --kernel-trace --output-format csv -d gpurun_out/h6/kt0 -- python3 tools/run_shapes.py --shapes "256,64,56,56,4;256,128,28,28,3;256,48,56,56,4;256,96,28,28,3;256,80,56,56,4;256,160,28,28,3" --iters 20 > gpurun_out/h6/kt0.log 2>&1
python3 tools/trace_by_grid.py gpurun_out/h6/kt0 | tee gpurun_out/h6/lanes.txt
rm -rf gpurun_out/h6/kt gpurun_out/h6/kt0
