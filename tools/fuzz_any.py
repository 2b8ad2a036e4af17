#!/usr/bin/env python3
"""Randomised parity sweep of rcx_recconv2d_fwd on ARBITRARY planes against the C oracle (development tool): random heights and widths
(even and odd), levels 1 - 4, channel counts that hit whole and ragged 64-channel waves, three dtypes, both modes, bias -- every
schedule (fused kernels with the full and the shorter ladder, nested, split, LDS pyramid, generic ladder) gets hit; the plan is printed.
    python3 tools/fuzz_any.py [cases=200] [seed=0]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import recnext_amd
from oracle import c_oracle
from recnext_amd import ops


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    bad, plans = 0, collections.Counter()
    special = [(56, 56), (28, 28), (14, 14), (112, 112), (64, 64), (32, 32), (128, 128), (200, 336), (100, 168), (50, 84), (96, 96), (48, 48), (24, 24)]
    for it in range(cases):
        if rng.random() < 0.5:
            h, w = special[int(rng.integers(len(special)))]
        else:
            h, w = int(rng.integers(7, 121)), int(rng.integers(7, 121))
            if rng.random() < 0.6:
                h, w = h & ~1, w & ~1
        level = int(rng.integers(1, 5))
        while min(h, w) >> level < 2 and level > 1:
            level -= 1
        c = int(rng.choice([8, 24, 40, 64, 72, 96, 128, 192]))
        n = int(rng.choice([1, 2, 3]))
        if h * w * c * n > 6_000_000:
            c, n = 64 if c >= 64 else c, 1
        mode = "bilinear" if rng.random() < 0.6 else "nearest"
        bias = bool(rng.random() < 0.3)
        dtype = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(3))]
        torch.manual_seed(it)
        mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level, mode=mode, bias=bias).to(dev).eval()
        x = torch.randn(n, c, h, w, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
        sd = {k: v.detach().float().cpu().numpy() for k, v in mod.state_dict().items()}
        with torch.no_grad():
            y = mod(x)
            y2 = mod(x)
        plan = ops.recconv2d_plan(n, c, h, w, level, 5, mode, dtype)
        plans[plan.split("(")[0]] += 1
        ref = c_oracle.recconv2d(x.float().cpu().numpy(), sd["down.weight"], [sd[f"convs.{i}.weight"] for i in range(level + 1)], level=level, mode=mode,
                                 b_down=sd.get("down.bias"), b_convs=[sd[f"convs.{i}.bias"] for i in range(level + 1)] if bias else None)
        got = y.float().cpu().numpy()
        tol = {torch.float32: (1e-4, 1e-4), torch.bfloat16: (1e-2, 1e-2), torch.float16: (2e-3, 2e-3)}[dtype]
        ok = torch.equal(y, y2) and np.allclose(got, ref, atol=tol[0], rtol=tol[1])
        if not ok:
            bad += 1
            print(f"MISMATCH case {it}: {n}x{c}x{h}x{w} L{level} {mode} bias={bias} {dtype} plan={plan} max|err|={np.abs(got - ref).max():.3e} deterministic={torch.equal(y, y2)}", flush=True)
        elif it % 25 == 0:
            print(f"case {it}: {n}x{c}x{h}x{w} L{level} {str(dtype)[6:]} {plan[:70]} ok ({np.abs(got - ref).max():.2e})", flush=True)
    print(f"{cases} cases, {bad} mismatches; schedules hit: {dict(plans)}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
