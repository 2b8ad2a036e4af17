#!/usr/bin/env python3
"""Training forward (the inference kernel + the saved float32 pyramid) against the inference forward, per stage shape, under
rocprofv3: python tools/bench_fwd_train.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import recnext_amd
from recnext_amd import ops

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for c, h, level in [(64, 56, 4), (128, 28, 3), (256, 14, 2), (512, 7, 1)]:
    mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level).to(dev)
    wpack, bpack = mod.packed_params()
    x = torch.randn(n, c, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for _ in range(30):
        ops.recconv2d_forward(x, wpack, bpack, level, 5, "bilinear")
    torch.cuda.synchronize()
    for _ in range(30):
        ops.recconv2d_forward_train(x, wpack, bpack, level, 5, "bilinear")
    torch.cuda.synchronize()
