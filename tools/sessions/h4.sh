set -u
mkdir -p gpurun_out/h4
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/debug_cpt.py 56 4 > gpurun_out/h4/d56.log 2>&1; grep -v amdgpu.ids gpurun_out/h4/d56.log | cut -c1-250
timeout -k 10 200 python tools/debug_cpt.py 28 3 > gpurun_out/h4/d28.log 2>&1; grep -v amdgpu.ids gpurun_out/h4/d28.log | cut -c1-250
