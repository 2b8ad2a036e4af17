cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5r
cat > gpurun_out/r5r/bwd_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, recnext_amd
dev = torch.device("cuda:0")
n, c, h, level = 128, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level).to(dev).to(torch.bfloat16).train()
x = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
g = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
for i in range(12):
    y = mod(x); y.backward(g); x.grad = None
    for p in mod.parameters(): p.grad = None
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5r/bwd56 -- python3 gpurun_out/r5r/bwd_one.py 64 56 4 > /dev/null 2>&1
f=$(ls gpurun_out/r5r/bwd56/*/*kernel_stats.csv | head -1); cut -c1-160 $f | head -40 > gpurun_out/r5r/bwd56_stats.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5r/bwd28 -- python3 gpurun_out/r5r/bwd_one.py 128 28 3 > /dev/null 2>&1
f=$(ls gpurun_out/r5r/bwd28/*/*kernel_stats.csv | head -1); cut -c1-160 $f | head -40 > gpurun_out/r5r/bwd28_stats.txt
rm -rf gpurun_out/r5r/bwd56 gpurun_out/r5r/bwd28
cat gpurun_out/r5r/bwd56_stats.txt
