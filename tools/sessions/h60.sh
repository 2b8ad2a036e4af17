set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for s in 6 8 6; do timeout -k 10 300 python tools/bench_train.py --model recnext_a3 --batch 128 --steps $s --which hip 2>&1 | tail -1 | cut -c1-200; done
RCX_ATTN_MFMA=0 timeout -k 10 300 python tools/bench_train.py --model recnext_a3 --batch 128 --steps 6 --which hip 2>&1 | tail -1 | cut -c1-200
