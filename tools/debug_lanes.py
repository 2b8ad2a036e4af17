#!/usr/bin/env python3
"""Compare one small RecConv2d block on the GPU against the C oracle and print the error map (development tool)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import c_oracle
from recnext_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="1,8,7,7,1")
ap.add_argument("--mode", default="bilinear")
ap.add_argument("--dtype", default="f32")
ap.add_argument("--bias", type=int, default=0)
args = ap.parse_args()
n, c, h, w, level = map(int, args.shape.split(","))
rng = np.random.default_rng(0)
x = rng.standard_normal((n, c, h, w)).astype(np.float32)
wd = (rng.standard_normal((c, 1, 5, 5)) * 0.2).astype(np.float32)
wc = [(rng.standard_normal((c, 1, 5, 5)) * 0.2).astype(np.float32) for _ in range(level + 1)]
bd = rng.standard_normal(c).astype(np.float32) if args.bias else None
bc = [rng.standard_normal(c).astype(np.float32) for _ in range(level + 1)] if args.bias else None
ref = c_oracle.recconv2d(x, wd, wc, bd, bc, level, args.mode)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
wpack, bpack = ops.pack_recconv_params(t(wd), [t(v) for v in wc], None if bd is None else t(bd), None if bc is None else [t(b) for b in bc])
dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
print("plan:", ops.recconv2d_plan(n, c, h, w, level, 5, args.mode, dtype))
xin = t(x).to(dtype).contiguous(memory_format=torch.channels_last)
got = ops.recconv2d_forward(xin, wpack, bpack, level, 5, args.mode).float().cpu().numpy()
err = np.abs(got - ref)
print("max err", np.nanmax(err), "nan count", int(np.isnan(got).sum()), "of", got.size)
np.set_printoptions(precision=3, suppress=True, linewidth=200)
print("err per channel:", np.nan_to_num(err, nan=9.0).reshape(n, c, -1).max(-1))
print("err map n0 c0:\n", err[0, 0])
print("got n0 c0:\n", got[0, 0])
print("ref n0 c0:\n", ref[0, 0])
