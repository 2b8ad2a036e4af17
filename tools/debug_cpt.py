#!/usr/bin/env python3
"""Debug aid for rcx_cpt.hip: runs the tiled kernel on reduced problems (weights zeroed so that only some stages contribute) and
prints where it departs from the C oracle, folded by position inside a 14x14 tile and by tile.  usage: debug_cpt.py [hw level]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import c_oracle
from recnext_amd import ops

hw = int(sys.argv[1]) if len(sys.argv) > 1 else 56
level = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mode = sys.argv[3] if len(sys.argv) > 3 else "bilinear"
n, c, k = 1, 64, 5
rng = np.random.default_rng(0)
dev = torch.device("cuda:0")


def ident():
    w = np.zeros((c, 1, k, k), np.float32)
    w[:, 0, 2, 2] = 1.0
    return w


def rnd():
    return (rng.standard_normal((c, 1, k, k)) * 0.2).astype(np.float32)


def zero():
    return np.zeros((c, 1, k, k), np.float32)


def run(name, x, wd, wc):
    t = lambda a: torch.from_numpy(a).to(dev)
    wpack, bpack = ops.pack_recconv_params(t(wd), [t(w) for w in wc], None, None)
    xin = t(x).contiguous(memory_format=torch.channels_last)
    got = ops.recconv2d_forward(xin, wpack, bpack, level, k, mode).float().cpu().numpy()
    ref = c_oracle.recconv2d(x, wd, wc, None, None, level, mode)
    err = np.abs(got - ref)
    print(f"== {name}: plan {ops.recconv2d_plan(n, c, hw, hw, level, k, mode, torch.float32)[:30]} max err {err.max():.3e}")
    if err.max() > 1e-4:
        e = err.max(axis=(0, 1))                         # (hw, hw)
        T = hw // 14
        tile = e.reshape(T, 14, T, 14).max(axis=(1, 3))
        print("  per tile (rows = tile row):")
        for r in range(T):
            print("   ", " ".join(f"{v:9.2e}" for v in tile[r]))
        rows = e.max(axis=1)
        cols = e.max(axis=0)
        print("  bad rows:", [i for i in range(hw) if rows[i] > 1e-4])
        print("  bad cols:", [i for i in range(hw) if cols[i] > 1e-4])
        ch = err.max(axis=(0, 2, 3))
        print("  bad channels:", [i for i in range(c) if ch[i] > 1e-4][:40])


x = rng.standard_normal((n, c, hw, hw)).astype(np.float32)
L = level
# convs[j]: j = 0 coarsest ... L-1 = level 1, L = final
run("final conv only (C1 = 0)", x, rnd(), [zero() for _ in range(L)] + [rnd()])
run("final = identity, level-1 conv = identity, rest 0: y = x + up(F1)", x, rnd(), [zero() for _ in range(L - 1)] + [ident(), ident()])
run("y = x + up(conv1(F1))", x, rnd(), [zero() for _ in range(L - 1)] + [rnd(), ident()])
run("levels 1,2: y = x + up(F1 + up(F2))", x, rnd(), [zero() for _ in range(L - 2)] + [ident(), ident(), ident()])
run("levels 1,2 with convs", x, rnd(), [zero() for _ in range(L - 2)] + [rnd(), rnd(), ident()])
run("all identity convs", x, rnd(), [ident() for _ in range(L + 1)])
run("everything random", x, rnd(), [rnd() for _ in range(L + 1)])
