#!/usr/bin/env python3
"""Time rcx_dwconv2d_mult2_fwd (the Downsample depthwise 7x7 stride-2 conv) on the RecNeXt-M3 stage transitions (development tool)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from recnext_amd import ops

dev = torch.device("cuda:0")
shapes = [(256, 64, 56, 56), (256, 128, 28, 28), (256, 256, 14, 14)]
if len(sys.argv) > 1:                                     # "NxCxHxW,..."
    shapes = [tuple(int(v) for v in sh.split("x")) for sh in sys.argv[1].split(",")]
for n, c, h, wd in shapes:
    x = torch.randn(n, c, h, wd, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = ops.pack_dw_weight(torch.randn(2 * c, 1, 7, 7, device=dev) * 0.1)
    b = ops.pack_bias(torch.randn(2 * c, device=dev))
    for _ in range(3):
        ops.dwconv2d_mult2(x, w, b, k=7, stride=2)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.dwconv2d_mult2(x, w, b, k=7, stride=2)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    alg = x.numel() * 2 + x.numel() // 2 * 2 + 2 * c * 49 * 2
    print(json.dumps({"shape": [n, c, h, wd], "us": round(ms * 1e3, 1), "alg_GBs": round(alg / ms / 1e6, 1), "frac_8TBs": round(alg / ms / 1e6 / 8e3, 4)}))
