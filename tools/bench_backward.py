#!/usr/bin/env python3
"""Forward+backward time of one RecConv2d block: HIP autograd function vs the ATen operator chain (development tool)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import recnext_amd
from oracle.torch_eager import EagerRecConv2d

dev = torch.device("cuda:0")
from recnext_amd import build

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dtypes = (torch.bfloat16, torch.float32) if "--bf16-only" not in sys.argv else (torch.bfloat16,)
impls = (("hip", recnext_amd.RecConv2d), ("aten", EagerRecConv2d)) if "--hip-only" not in sys.argv else (("hip", recnext_amd.RecConv2d),)
print(json.dumps({"library_sources_sha256": build.source_fingerprint(), "device": torch.cuda.get_device_name(0)}), flush=True)
for dtype in dtypes:
    for c, h, level in [(64, 56, 4), (128, 28, 3), (256, 14, 2), (512, 7, 1)]:
        res = {"shape": [n, c, h, h], "level": level, "dtype": str(dtype).split(".")[-1]}
        for name, cls in impls:
            torch.manual_seed(0)
            mod = cls(c, kernel_size=5, level=level).to(dev).to(dtype).train()
            x = torch.randn(n, c, h, h, device=dev).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            g = torch.randn(n, c, h, h, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)

            def step():
                y = mod(x)
                y.backward(g)
                x.grad = None
                for p in mod.parameters():
                    p.grad = None

            for _ in range(3):
                step()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                step()
            e.record()
            torch.cuda.synchronize()
            res[name + "_fwd_bwd_ms"] = round(s.elapsed_time(e) / 10, 3)
            if name == "hip":                                   # the inference forward of the same block: the denominator of the ratio
                mod.eval()
                xi = x.detach()
                with torch.no_grad():
                    for _ in range(3):
                        mod(xi)
                    s.record()
                    for _ in range(20):
                        mod(xi)
                    e.record()
                torch.cuda.synchronize()
                res["hip_inference_ms"] = round(s.elapsed_time(e) / 20, 4)
                res["hip_fwd_bwd_over_inference"] = round(res["hip_fwd_bwd_ms"] / res["hip_inference_ms"], 2)
        print(json.dumps(res), flush=True)
