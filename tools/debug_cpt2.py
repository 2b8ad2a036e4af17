#!/usr/bin/env python3
"""rcx_cpt.hip debug: F1[r][c] = 100 r + c (centre-tap down conv of a suitable x), identity level-1 and final convs, so that
y - x = resize(F1) is linear in the source indices and shows which source pixel every output read."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import c_oracle
from recnext_amd import ops

hw = int(sys.argv[1]) if len(sys.argv) > 1 else 28
level = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n, c, k = 1, 64, 5
dev = torch.device("cuda:0")
x = np.zeros((n, c, hw, hw), np.float32)
ii, jj = np.meshgrid(np.arange(hw), np.arange(hw), indexing="ij")
x[:, :] = ((ii // 2) * 100 + (jj // 2)) * ((ii % 2 == 0) & (jj % 2 == 0))
ident = np.zeros((c, 1, k, k), np.float32)
ident[:, 0, 2, 2] = 1
zero = np.zeros((c, 1, k, k), np.float32)
wc = [zero] * (level - 1) + [ident, ident]
t = lambda a: torch.from_numpy(a).to(dev)
wpack, bpack = ops.pack_recconv_params(t(ident), [t(w) for w in wc], None, None)
got = ops.recconv2d_forward(t(x).contiguous(memory_format=torch.channels_last), wpack, bpack, level, k, "bilinear").float().cpu().numpy()
ref = c_oracle.recconv2d(x, ident, wc, None, None, level, "bilinear")
d_got = (got - x)[0, 5]
d_ref = (ref - x)[0, 5]
np.set_printoptions(linewidth=250, precision=2, suppress=True)
print("max err", np.abs(got - ref).max())
for r in [0, 1, 12, 13, 14, 15, 27]:
    print("row", r, "got cols 10..17:", d_got[r, 10:18], " ref:", d_ref[r, 10:18])
