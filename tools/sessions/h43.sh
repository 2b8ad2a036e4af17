set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for b in 32 64 128 256; do timeout -k 10 300 python tools/bench_train.py --model recnext_m3 --batch $b --steps 8 --which hip 2>&1 | tail -1 | cut -c1-200; done
