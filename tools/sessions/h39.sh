cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/h39
timeout -k 10 200 python tools/debug_cplbwd.py 2 8 7 1 > gpurun_out/h39/out.txt 2>&1
