#!/usr/bin/env python3
"""Launch one RecConv2d block shape a few times (target for rocprofv3 --kernel-trace / --pmc runs)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import recnext_amd

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="256,64,56,56,4")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()
n, c, h, w, level = map(int, args.shape.split(","))
dev = torch.device("cuda:0")
dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level).to(dev).eval()
x = torch.randn(n, c, h, w, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(args.iters):
        y = mod(x)
torch.cuda.synchronize()
print("done", float(y.float().abs().mean()))
