#!/usr/bin/env python3
"""Check that RCX_PLANE_ABLATE (diagnostic build) changes results, and time the kernel per setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recnext_amd
from tools.bench_blocks import time_fn
dev = torch.device("cuda:0")
n, c, h, w, level = 256, 64, 56, 56, 4
mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level).to(dev).eval()
x = torch.randn(n, c, h, w, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    ref = None
    for ab in [0, 1, 2, 4, 7, 8, 16, 32, 63]:
        os.environ["RCX_PLANE_ABLATE"] = str(ab)
        y = mod(x); torch.cuda.synchronize()
        if ref is None: ref = y
        med, mn = time_fn(lambda: mod(x), 10)
        print(f"ablate={ab:3d}  ms={med:.4f}  differs_from_full={not torch.equal(ref, y)}  max|y|={float(y.float().abs().max()):.3f}")
