#!/usr/bin/env python3
"""A/B of the 56x56 block's schedules in one process (development tool, GPU): RCX_CPT_CB=32 / 16, vector and matrix cores.
Kernel times come from rocprofv3 around this script; the wall times printed here include the host's launch path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import recnext_amd
from recnext_amd import ops
from oracle import c_oracle


def run(n, c, dtype, iters=30):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    mod = recnext_amd.RecConv2d(c, kernel_size=5, level=4).to(dev).eval()
    x = torch.randn(n, c, 56, 56, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    sd = {k: v.detach().float().cpu().numpy() for k, v in mod.state_dict().items()}
    ref = c_oracle.recconv2d(x[:2].float().cpu().numpy(), sd["down.weight"], [sd[f"convs.{i}.weight"] for i in range(5)], level=4)
    out = {}
    for cb in ("32", "16"):
        os.environ["RCX_CPT_CB"] = cb
        ops.reload_options()
        with torch.no_grad():
            y = mod(x)
            for _ in range(3):
                mod(x)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                mod(x)
            e.record()
            torch.cuda.synchronize()
        err = np.abs(y[:2].float().cpu().numpy() - ref).max()
        out[cb] = (s.elapsed_time(e) / iters * 1e3, err, ops.recconv2d_plan(n, c, 56, 56, 4, 5, "bilinear", dtype), y)
    same = torch.equal(out["32"][3], out["16"][3])
    for cb in ("32", "16"):
        print(f"{n}x{c}x56x56 {str(dtype)[6:]} RCX_CPT_CB={cb}: {out[cb][0]:.1f} us wall, max|err| vs oracle {out[cb][1]:.2e}, plan {out[cb][2]}")
    print("   bitwise equal:", same)


if __name__ == "__main__":
    run(256, 64, torch.bfloat16)
    run(256, 64, torch.float32)
    run(256, 80, torch.bfloat16)
    run(3, 48, torch.float16)
