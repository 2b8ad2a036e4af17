#!/usr/bin/env python3
"""Where do non-deterministic runs differ? (race hunting, development tool)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import recnext_amd

dev = torch.device("cuda:0")
n, c, h, level = 256, 64, 56, 4
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
torch.manual_seed(0)
mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level).to(dev).eval()
x = torch.randn(n, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    y0 = mod(x).clone()
    for r in range(reps):
        y = mod(x)
        if not torch.equal(y, y0):
            d = (y.float() - y0.float()).abs()
            idx = torch.nonzero(d > 0)
            print(f"run {r}: {idx.shape[0]} elements differ, max {float(d.max()):.4g}")
            print("  images:", sorted(set(idx[:, 0].tolist()))[:10], " channels:", sorted(set(idx[:, 1].tolist()))[:40])
            print("  rows:", sorted(set(idx[:, 2].tolist())), " cols:", sorted(set(idx[:, 3].tolist())))
print("done")
