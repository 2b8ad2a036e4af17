cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/h45
timeout -k 10 200 python tools/host_profile.py > gpurun_out/h45/out.txt 2>&1
