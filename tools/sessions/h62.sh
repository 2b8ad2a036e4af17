set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout -k 10 300 python tools/bench_train.py --model recnext_a3 --batch 256 --steps 8 --which hip 2>&1 | tail -1 | cut -c1-160
RCX_ATTN_MFMA=0 timeout -k 10 300 python tools/bench_train.py --model recnext_a3 --batch 256 --steps 8 --which hip 2>&1 | tail -1 | cut -c1-160
done
RCX_ATTN_MFMA=0 RCX_WGRAD_CPL=0 timeout -k 10 300 python tools/bench_train.py --model recnext_a3 --batch 256 --steps 8 --which hip 2>&1 | tail -1 | cut -c1-160
