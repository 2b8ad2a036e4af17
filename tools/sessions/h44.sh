set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for b in 64 128; do timeout -k 10 300 python tools/bench_train.py --model recnext_m3 --batch $b --steps 8 --which hip --graph 2>&1 | tail -4 | cut -c1-300; done
