#!/usr/bin/env python3
"""Launch a list of RecConv2d block shapes in one process (target for one rocprofv3 --kernel-trace run; tools/trace_by_grid.py
then averages the kernel durations per (kernel, grid size), which separates the shapes).
usage: run_shapes.py --shapes "64,256,14,14,2;128,256,14,14,2" [--dtype bf16] [--iters 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import recnext_amd

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", required=True)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
for spec in args.shapes.split(";"):
    n, c, h, w, level = map(int, spec.split(","))
    mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level).to(dev).eval()
    x = torch.randn(n, c, h, w, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(args.iters):
            y = mod(x)
    torch.cuda.synchronize()
    print(spec, recnext_amd.ops.recconv2d_plan(n, c, h, w, level, 5, "bilinear", dtype), float(y.float().abs().mean()), flush=True)
