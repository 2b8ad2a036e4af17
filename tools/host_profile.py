#!/usr/bin/env python3
"""Host-side cost of one RecConv2d block's forward+backward (the training step is launch-bound below batch ~128): cProfile of 200
iterations of the 14x14 / level 2 block.  python tools/host_profile.py [N]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import recnext_amd

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mod = recnext_amd.RecConv2d(256, kernel_size=5, level=2).to(dev).train()
x = torch.randn(n, 256, 14, 14, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
g = torch.randn(n, 256, 14, 14, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = mod(x)
    y.backward(g)
    x.grad = None
    for p in mod.parameters():
        p.grad = None
        p._version                                       # (parameters are not modified here: the pack is cached)


for _ in range(20):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print("wall per fwd+bwd: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

# ---- the backward runs on autograd's thread, which cProfile does not see: call the same functions directly ----
from recnext_amd import ops

wpack, bpack = mod.packed_params()
wflip = mod._wflip
xd = x.detach()
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(200):
    y, saved = ops.recconv2d_forward_train(xd, wpack, bpack, 2, 5, "bilinear")
    gx, gw, gb = ops.recconv2d_backward(xd, g, wpack, saved, 2, 5, "bilinear", need_bias=False, wflip=wflip)
    gwu = ops.unpack_recconv_grads(gw, 4, 256, 5)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
