#!/usr/bin/env python3
"""Time rcx_linear_attention_fwd on the RecNeXt-A3 stage shapes (development tool)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from recnext_amd import ops

dev = torch.device("cuda:0")
for b, c, heads, h in [(256, 64, 2, 28), (256, 128, 4, 14), (256, 256, 8, 7), (256, 512, 16, 4)]:
    n = h * h
    d = torch.randn(b, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    pe = torch.randn_like(d)
    q = torch.randn(b, n, c, device=dev).bfloat16()
    k = torch.randn(b, n, c, device=dev).bfloat16()
    for _ in range(3):
        ops.linear_attention_core(q, k, d, pe, heads)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.linear_attention_core(q, k, d, pe, heads)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    byts = 5 * b * n * c * 2
    print(json.dumps({"B": b, "C": c, "heads": heads, "tokens": n, "us": round(ms * 1e3, 1), "GBs": round(byts / ms / 1e6, 1)}))
