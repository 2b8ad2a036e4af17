set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout -k 10 300 python tools/bench_configs.py cfg4 2>&1 | tail -1
RCX_UPADD_CPL=0 timeout -k 10 300 python tools/bench_configs.py cfg4 2>&1 | tail -1
done
