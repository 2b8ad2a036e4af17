set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cat > /tmp/bw7.py <<'PY'
import os, sys, torch, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import recnext_amd
dev = torch.device("cuda:0")
for (n, c, h, level) in [(128, 512, 7, 1), (128, 256, 14, 2)]:
    mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level).to(dev).train()
    x = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    def fwd():
        return mod(x)
    def both():
        y = mod(x); y.backward(g); x.grad = None
        for p in mod.parameters(): p.grad = None
    for name, fn in (("fwd", fwd), ("fwd+bwd", both)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): fn()
        e.record(); torch.cuda.synchronize()
        print(os.environ.get("RCX_TRAIN_FUSED", "1"), (n, c, h, level), name, round(s.elapsed_time(e) / 50 * 1e3, 1), "us")
PY
python3 /tmp/bw7.py 2>&1 | grep -v amdgpu
RCX_TRAIN_FUSED=0 python3 /tmp/bw7.py 2>&1 | grep -v amdgpu
python3 /tmp/bw7.py 2>&1 | grep -v amdgpu
