"""Fused (rcx_cplbwd.hip) against per-step backward: per-output error report.  python tools/debug_cplbwd.py N C HW LEVEL [mode] [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recnext_amd
from recnext_amd import ops

n, c, hw, level = (int(v) for v in sys.argv[1:5])
mode = sys.argv[5] if len(sys.argv) > 5 else "bilinear"
dtype = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[6] if len(sys.argv) > 6 else "f32"]
dev = torch.device("cuda:0")
torch.manual_seed(11)
mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level, bias=True, mode=mode).to(dev)
wpack, bpack = mod.packed_params()
x = torch.randn(n, c, hw, hw, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
gy = torch.randn(n, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
_, saved = ops.recconv2d_forward_train(x, wpack, bpack, level, 5, mode)
os.environ.pop("RCX_BWD_FUSED", None)
gx1, gw1, gb1 = ops.recconv2d_backward(x, gy, wpack, saved, level, 5, mode, need_bias=True)
os.environ["RCX_BWD_FUSED"] = "0"
gx0, gw0, gb0 = ops.recconv2d_backward(x, gy, wpack, saved, level, 5, mode, need_bias=True)
print("gx", float((gx1.float() - gx0.float()).abs().max()), float(gx0.float().abs().max()))
d = (gx1.float() - gx0.float()).abs().amax(dim=(0, 1))
print("gx err by pixel:\n", d)
for j in range(level + 2):
    a1, a0 = gw1[j].view(5, 5, c), gw0[j].view(5, 5, c)
    print("gw job", j, float((a1 - a0).abs().max()), float(a0.abs().max()), "gb", float((gb1[j] - gb0[j]).abs().max()), float(gb0[j].abs().max()))
    print((a1 - a0).abs().amax(dim=2))
