set -u
mkdir -p gpurun_out/h58
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in recnext_a3 recnext_m3; do
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/h58/kt_$m -- python3 tools/bench_train.py --model $m --batch 128 --steps 6 --which hip > gpurun_out/h58/kt_$m.log 2>&1
f=$(find gpurun_out/h58/kt_$m -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/h58/${m}_train_kernel_stats.csv; rm -rf gpurun_out/h58/kt_$m
tail -1 gpurun_out/h58/kt_$m.log | cut -c1-160
done
