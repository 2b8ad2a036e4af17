set -u
mkdir -p gpurun_out/h2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
S14="32,256,14,14,2;64,256,14,14,2;128,256,14,14,2;192,256,14,14,2;256,256,14,14,2;320,256,14,14,2;512,256,14,14,2;768,256,14,14,2;1024,256,14,14,2;256,128,14,14,2;256,64,14,14,2"
S7="64,512,7,7,1;128,512,7,7,1;256,512,7,7,1;384,512,7,7,1;512,512,7,7,1;1024,512,7,7,1"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/h2/kt -- python3 tools/run_shapes.py --shapes "$S14;$S7" --iters 20 > gpurun_out/h2/kt.log 2>&1
python3 tools/trace_by_grid.py gpurun_out/h2/kt | tee gpurun_out/h2/sweep.txt
rm -rf gpurun_out/h2/kt
timeout -k 10 600 bash tools/pmc_block.sh 256,256,14,14,2 bf16 > gpurun_out/h2/pmc14.log 2>&1; tail -40 gpurun_out/h2/pmc14.log
