#!/usr/bin/env python3
"""Randomised parity sweep of the fused schedules against the C oracle (development tool): every plane size of the
register-resident families, channel counts that hit every workgroup width, odd batch sizes, both modes, bias, both dtypes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import c_oracle
from recnext_amd import ops

LEVEL = {7: 1, 14: 2, 28: 3, 56: 4, 16: 1, 32: 2, 64: 3, 128: 4}


def run(cases, seed, verbose=True):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    bad = 0
    saved = {k: os.environ.get(k) for k in ("RCX_LANES_NI", "RCX_LANES_WAVES")}
    try:
        for it in range(cases):
            bad += _one(rng, dev, it, verbose)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return bad


def _one(rng, dev, it, verbose):
    if True:
        h = int(rng.choice([7, 14, 14, 28, 28, 56, 16, 32, 64, 128]))
        c = int(rng.choice([8, 16, 24, 32, 40, 48, 64, 80, 96, 128, 160, 192]))
        n = int(rng.choice([1, 2, 3, 5, 9, 17])) if h < 56 else int(rng.choice([1, 2, 3]))
        if h == 128:
            c, n = int(rng.choice([16, 32, 64])), 1
        level, mode, bias = LEVEL[h], str(rng.choice(["bilinear", "nearest"])), bool(rng.integers(2))
        dtype = torch.bfloat16 if rng.integers(2) else torch.float32
        for key, val in (("RCX_LANES_NI", str(int(rng.choice([0, 1, 2, 4])))), ("RCX_LANES_WAVES", str(int(rng.choice([8, 4, 2, 1]))))):
            os.environ[key] = val
        x = rng.standard_normal((n, c, h, h)).astype(np.float32)
        if dtype == torch.bfloat16:
            u = x.view(np.uint32); x = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).view(np.float32)
        wd = (rng.standard_normal((c, 1, 5, 5)) * 0.2).astype(np.float32)
        wc = [(rng.standard_normal((c, 1, 5, 5)) * 0.2).astype(np.float32) for _ in range(level + 1)]
        bd = rng.standard_normal(c).astype(np.float32) if bias else None
        bc = [rng.standard_normal(c).astype(np.float32) for _ in range(level + 1)] if bias else None
        ref = c_oracle.recconv2d(x, wd, wc, bd, bc, level, mode)
        t = lambda a: torch.from_numpy(a).to(dev)
        wpack, bpack = ops.pack_recconv_params(t(wd), [t(v) for v in wc], None if bd is None else t(bd), None if bc is None else [t(b) for b in bc])
        plan = ops.recconv2d_plan(n, c, h, h, level, 5, mode, dtype)
        got = ops.recconv2d_forward(t(x).to(dtype).contiguous(memory_format=torch.channels_last), wpack, bpack, level, 5, mode).float().cpu().numpy()
        ok = np.abs(got - ref).max() < 1e-4 if dtype == torch.float32 else np.allclose(got, ref, atol=1e-2, rtol=1e-2)
        if verbose and (not ok or it % 25 == 0):
            print(("ok  " if ok else "FAIL"), n, c, h, level, mode, bias, str(dtype).split(".")[-1], os.environ["RCX_LANES_NI"], os.environ["RCX_LANES_WAVES"],
                  plan[:70], float(np.abs(got - ref).max()), flush=True)
        return 0 if ok else 1

if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    failures = run(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("cases", n_cases, "failures", failures)
