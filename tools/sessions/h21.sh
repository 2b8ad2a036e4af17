set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 500 python tools/fuzz_lanes.py 250 12345 2>&1 | tail -5
RCX_CPT=all RCX_CPT_GRID=40 timeout -k 10 400 python tools/fuzz_lanes.py 150 777 2>&1 | tail -4
