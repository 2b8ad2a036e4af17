set -u
mkdir -p gpurun_out/h28
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cat > /tmp/bw14.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import recnext_amd
dev = torch.device("cuda:0")
n, c, h, level = 128, 256, 14, 2
mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level).to(dev).train()
x = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
g = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
for _ in range(8):
    y = mod(x); y.backward(g); x.grad = None
    for p in mod.parameters(): p.grad = None
torch.cuda.synchronize()
PY
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/h28/kt -- python3 /tmp/bw14.py > gpurun_out/h28/kt.log 2>&1
f=$(find gpurun_out/h28/kt -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | head -30 | cut -c1-170
