#!/usr/bin/env python3
"""Time the fused stem (rcx_stem_fwd) against the library path it replaces (conv + bias, GELU, conv + bias) at batch 256, 224 x 224 (development tool)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from recnext_amd import ops

dev = torch.device("cuda:0")
REPS = int(os.environ.get("REPS", "20"))
torch.backends.cudnn.benchmark = True
for b, cm, co, hw in [(256, 32, 64, 224), (256, 24, 48, 224), (256, 20, 40, 224), (256, 40, 80, 224), (32, 32, 64, 512)]:
    xs = [torch.randn(b, 3, hw, hw, device=dev).bfloat16().contiguous(memory_format=torch.channels_last) for _ in range(4)]
    w1, b1 = (torch.randn(cm, 3, 3, 3, device=dev) * 0.2).bfloat16().contiguous(memory_format=torch.channels_last), torch.randn(cm, device=dev).bfloat16()
    w2, b2 = (torch.randn(co, cm, 3, 3, device=dev) * 0.1).bfloat16().contiguous(memory_format=torch.channels_last), torch.randn(co, device=dev).bfloat16()
    pack = ops.pack_stem(w1, b1, w2, b2)

    def timed(fn):
        for i in range(3):
            fn(xs[i % 4])
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for i in range(REPS):
            fn(xs[i % 4])
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / REPS * 1e3

    with torch.no_grad():
        t_f = timed(lambda x: ops.stem(x, *pack, cm, co))
        t_l = timed(lambda x: F.conv2d(F.gelu(F.conv2d(x, w1, b1, stride=2, padding=1)), w2, b2, stride=2, padding=1))
    io = b * hw * hw * 3 * 2 + b * (hw // 4) ** 2 * co * 2
    print(json.dumps({"B": b, "CM": cm, "CO": co, "side": hw, "fused_us": round(t_f, 1), "library_us": round(t_l, 1), "in_plus_out_MB": round(io / 1e6, 1),
                      "fused_TBs": round(io / t_f / 1e6, 2)}))
