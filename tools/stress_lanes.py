#!/usr/bin/env python3
"""Determinism stress: run every fused kernel many times on the same input and demand bit-identical results (race hunting)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import recnext_amd
from recnext_amd import ops

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for dtype in (torch.bfloat16, torch.float32):
    for n, c, h, level in [(256, 64, 56, 4), (256, 128, 28, 3), (256, 256, 14, 2), (256, 512, 7, 1), (37, 96, 28, 3), (19, 48, 56, 4)]:
        torch.manual_seed(0)
        mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level).to(dev).eval()
        x = torch.randn(n, c, h, h, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            y0 = mod(x).clone()
            diffs = 0
            for _ in range(reps):
                y = mod(x)
                if not torch.equal(y, y0):
                    diffs += 1
        print(f"recconv {n}x{c}x{h}x{h} L{level} {dtype}: {diffs} / {reps} runs differ", flush=True)
        bad += diffs
    for n, c, h in [(256, 64, 56), (256, 128, 28), (256, 256, 14)]:
        x = torch.randn(n, c, h, h, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
        w = ops.pack_dw_weight(torch.randn(2 * c, 1, 7, 7, device=dev) * 0.1)
        b = ops.pack_bias(torch.randn(2 * c, device=dev))
        y0 = ops.dwconv2d_mult2(x, w, b, k=7, stride=2).clone()
        diffs = sum(0 if torch.equal(ops.dwconv2d_mult2(x, w, b, k=7, stride=2), y0) else 1 for _ in range(reps))
        print(f"down {n}x{c}x{h}x{h} {dtype}: {diffs} / {reps} runs differ", flush=True)
        bad += diffs
# single-step kernels and the linear-attention core
for dtype in (torch.bfloat16, torch.float32):
    for n, c, h in [(256, 64, 56), (256, 128, 28), (256, 256, 14), (32, 128, 64), (32, 64, 128)]:
        x = torch.randn(n, c, h, h, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
        cs = torch.randn(n, c, h // 2, h // 2, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
        w = ops.pack_dw_weight(torch.randn(c, 1, 5, 5, device=dev) * 0.1)
        b = ops.pack_bias(torch.randn(c, device=dev))
        for name, fn in (("upadd", lambda: ops.upadd_dwconv(x, cs, w, b, k=5, mode="nearest")),
                         ("down5", lambda: ops.dwconv2d(x, w, b, k=5, stride=2)),
                         ("conv5", lambda: ops.dwconv2d(x, w, b, k=5, stride=1))):
            y0 = fn().clone()
            diffs = sum(0 if torch.equal(fn(), y0) else 1 for _ in range(reps))
            print(f"{name} {n}x{c}x{h}x{h} {dtype}: {diffs} / {reps} runs differ", flush=True)
            bad += diffs
for b_, c, heads, h in [(256, 64, 2, 28), (256, 128, 4, 14), (256, 512, 16, 4)]:
    d = torch.randn(b_, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    pe = torch.randn_like(d)
    q = torch.randn(b_, h * h, c, device=dev).bfloat16()
    k = torch.randn(b_, h * h, c, device=dev).bfloat16()
    y0 = ops.linear_attention_core(q, k, d, pe, heads).clone()
    diffs = sum(0 if torch.equal(ops.linear_attention_core(q, k, d, pe, heads), y0) else 1 for _ in range(reps))
    print(f"linattn {b_}x{c}x{h}x{h}: {diffs} / {reps} runs differ", flush=True)
    bad += diffs
print("TOTAL differing runs:", bad)
