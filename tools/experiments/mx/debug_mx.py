#!/usr/bin/env python3
"""Matrix-core schedules against the C oracle and beside the vector kernels (development tool, GPU)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import recnext_amd
from recnext_amd import ops
from oracle import c_oracle
from tests.util import bf16_round_np


def rnd(a, dtype):
    return torch.from_numpy(a).to(dtype).float().numpy()


def check(n, c, h, level, dtype, mode="bilinear", bias=False, seed=0):
    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level, mode=mode, bias=bias).to(dev).eval()
    x = torch.randn(n, c, h, h, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    sd = {k: rnd(v.detach().float().cpu().numpy(), dtype) for k, v in mod.state_dict().items()}      # the taps as the 16-bit module holds them
    with torch.no_grad():
        y_vec = mod(x).float().cpu().numpy()                   # float32 parameters: vector kernels, exact taps
        mm = mod.to(dtype)
        mm.matrix_cores = True
        y_mx = mm(x).float().cpu().numpy()
        y_mx2 = mm(x).float().cpu().numpy()
    plan = ops.recconv2d_plan_mx(n, c, h, h, level, 5, mode, dtype)
    xs = x.float().cpu().numpy()
    ref = c_oracle.recconv2d(xs, sd["down.weight"], [sd[f"convs.{i}.weight"] for i in range(level + 1)], level=level, mode=mode,
                             b_down=sd.get("down.bias"), b_convs=[sd[f"convs.{i}.bias"] for i in range(level + 1)] if bias else None)
    err = np.abs(y_mx - ref)
    tol = 1e-2 + 1e-2 * np.abs(ref)
    bad = err > tol
    print(f"{n}x{c}x{h}x{h} L{level} {str(dtype)[6:]} {mode} bias={bias}: plan {plan.split('(')[0]}  max|err| {err.max():.3e} mean {err.mean():.3e} "
          f"worst err/tol {(err / tol).max():.2f} bad {int(bad.sum())}  deterministic {np.array_equal(y_mx, y_mx2)}  |vec - mx| {np.abs(y_vec - y_mx).max():.3e}")
    if bad.any():
        idx = np.argwhere(bad)
        print("   first bad (n, c, y, x):", idx[:8].tolist())
        rows = sorted(set(int(i[2]) for i in idx)); cols = sorted(set(int(i[3]) for i in idx)); chs = sorted(set(int(i[1]) for i in idx))
        print("   rows", rows[:40], "cols", cols[:40], "channels", chs[:40])
    return not bad.any()


def bench(n, c, h, level, dtype, iters=30):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level).to(dev).eval()
    x = torch.randn(n, c, h, h, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    res = {}
    for name, m in (("vector", mod), ("matrix", None)):
        if m is None:
            m = mod.to(dtype)
            m.matrix_cores = True
        with torch.no_grad():
            for _ in range(5):
                m(x)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(iters):
                    m(x)
                e.record()
                torch.cuda.synchronize()
                ts.append(s.elapsed_time(e) / iters * 1e3)
        res[name] = sorted(ts)[1]
    alg = 2 * n * c * h * h * 2 + (level + 2) * c * 25 * 2
    print(f"bench {n}x{c}x{h}x{h} L{level} {str(dtype)[6:]}: vector {res['vector']:.1f} us ({alg / res['vector'] / 1e3 / 8000:.3f})   "
          f"matrix {res['matrix']:.1f} us ({alg / res['matrix'] / 1e3 / 8000:.3f} of 8 TB/s)")


if __name__ == "__main__":
    ok = True
    ok &= check(2, 64, 56, 4, torch.bfloat16)
    ok &= check(3, 64, 56, 4, torch.bfloat16, mode="nearest")
    ok &= check(2, 64, 56, 4, torch.float16)
    ok &= check(2, 48, 56, 4, torch.bfloat16, bias=True)
    ok &= check(3, 80, 56, 4, torch.bfloat16)
    ok &= check(9, 64, 56, 4, torch.bfloat16, seed=3)
    ok &= check(4, 256, 14, 2, torch.bfloat16)
    ok &= check(5, 256, 14, 2, torch.bfloat16, mode="nearest")
    ok &= check(3, 320, 14, 2, torch.float16)
    ok &= check(7, 24, 14, 2, torch.bfloat16, bias=True)
    ok &= check(2, 200, 14, 2, torch.bfloat16, seed=5)
    print("ALL OK" if ok else "FAILURES")
    if "--bench" in sys.argv:
        bench(256, 64, 56, 4, torch.bfloat16)
        bench(256, 256, 14, 2, torch.bfloat16)
        bench(256, 256, 14, 2, torch.float16)
        bench(256, 320, 14, 2, torch.bfloat16)
        bench(128, 256, 14, 2, torch.bfloat16)
