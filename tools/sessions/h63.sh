cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/h63
timeout -k 10 300 python tools/host_profile_step.py recnext_a3 32 > gpurun_out/h63/a3.txt 2>&1
