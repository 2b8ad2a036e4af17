set -u
mkdir -p gpurun_out/h51
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/h51/kt -- python3 tools/bench_configs.py cfg4 > gpurun_out/h51/kt.log 2>&1
f=$(find gpurun_out/h51/kt -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/h51/a3_fwd_kernel_stats.csv; rm -rf gpurun_out/h51/kt
tail -1 gpurun_out/h51/kt.log
