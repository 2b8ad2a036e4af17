#!/usr/bin/env python3
"""Development tool: RecAttn2d's one-launch unit (rcx_recattn2d_fwd) launched repeatedly on the same input, with LDS-dirtying kernels in between; reports where two
launches differ (a full-suite run once failed test_recattn2d_whole_unit_in_one_launch[f16-2x128x4x7] on its determinism check)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import recnext_amd
from recnext_amd import ops

dev = torch.device("cuda:0")
b, c, heads, hw = [int(v) for v in (sys.argv[2:6] if len(sys.argv) > 5 else (2, 128, 4, 7))]
xdt = torch.float16 if (len(sys.argv) <= 6 or sys.argv[6] == "f16") else torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(3 * c + hw)
t = lambda a: torch.from_numpy(a).to(dev)
x = t(rng.standard_normal((b, c, hw, hw)).astype(np.float32)).to(xdt).contiguous(memory_format=torch.channels_last)
wdn, bdn = ops.pack_dw_weight(t((rng.standard_normal((c, 1, 5, 5)) * 0.2).astype(np.float32))), ops.pack_bias(t((rng.standard_normal(c) * 0.1).astype(np.float32)))
wcv, bcv = ops.pack_dw_weight(t((rng.standard_normal((c, 1, 5, 5)) * 0.2).astype(np.float32))), ops.pack_bias(t((rng.standard_normal(c) * 0.1).astype(np.float32)))
wqk16 = t((rng.standard_normal((2 * c, c // 2)) * (2.0 / c) ** 0.5).astype(np.float32)).to(torch.bfloat16).contiguous()
bqk = t((rng.standard_normal(2 * c) * 0.1).astype(np.float32))
wpe, bpe = ops.pack_dw_weight(t((rng.standard_normal((c, 1, 3, 3)) * 0.2).astype(np.float32))), ops.pack_bias(t((rng.standard_normal(c) * 0.1).astype(np.float32)))
mod = recnext_amd.RecConv2d(64, kernel_size=5, level=4).to(dev).train()
xb = torch.randn(8, 64, 56, 56, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
first = ops.recattn2d(x, wdn, bdn, wqk16, bqk, wpe, bpe, wcv, bcv, heads).clone()
bad = 0
for it in range(reps):
    if it % 3 == 0:
        mod(xb).square().sum().backward()
        xb.grad = None
    y = ops.recattn2d(x, wdn, bdn, wqk16, bqk, wpe, bpe, wcv, bcv, heads)
    if not torch.equal(y, first):
        bad += 1
        d = (y.float() - first.float()).abs()
        idx = torch.nonzero(d > 0)
        print(f"launch {it}: {idx.shape[0]} elements differ, max {float(d.max()):.4g}; images {sorted(set(idx[:, 0].tolist()))} channels {sorted(set(idx[:, 1].tolist()))[:40]} "
              f"rows {sorted(set(idx[:, 2].tolist()))} cols {sorted(set(idx[:, 3].tolist()))}", flush=True)
        if bad >= 6:
            break
print(f"{reps} launches of {b} x {c} x {hw} x {hw}, {heads} heads, {xdt}: {bad} differ from the first")
