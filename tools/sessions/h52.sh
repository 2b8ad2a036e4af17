set -u
mkdir -p gpurun_out/h52
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_backward_gpu.py -q -x -k "linear_attention or recattn" 2>&1 | tail -3
timeout -k 10 300 python tools/bench_train.py --model recnext_a3 --batch 128 --steps 6 --which hip 2>&1 | tail -1 | cut -c1-200
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/h52/kt -- python3 tools/bench_train.py --model recnext_a3 --batch 128 --steps 6 --which hip > gpurun_out/h52/kt.log 2>&1
f=$(find gpurun_out/h52/kt -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/h52/a3_train_kernel_stats.csv; rm -rf gpurun_out/h52/kt
grep -E "rcx::" gpurun_out/h52/a3_train_kernel_stats.csv | head -8 | cut -c1-90,150-260
