set -u
mkdir -p gpurun_out/g3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 bash tools/sweep_batch.sh 256 14 2 "64 128 256 384 512 1024" > gpurun_out/g3/sweep.log 2>&1; cat gpurun_out/g3/sweep.log
timeout 600 bash tools/pmc_block.sh 256,256,14,14,2 bf16 > gpurun_out/g3/pmc.log 2>&1; tail -40 gpurun_out/g3/pmc.log
