#!/usr/bin/env python3
"""Random shapes through the matrix-core RecAttn2d entry points against the NumPy / C oracle (development tool, GPU box).

  rcx_recattn_qkcore_fwd        any plane h x w <= 48 x 48, heads in {1, 2, 4, 8, 16} (where rcx_recattn_qkcore_launches says 1 or 2)
  rcx_recattn_down_qkcore_fwd   14 x 14 / 7 x 7 planes
  rcx_recattn2d_fwd             14 x 14 (up to 8 heads) / 7 x 7 (up to 16) planes, nearest
Head dimension 32, or (half of the cases with more than one head) 4 .. 28.
Prints the worst err / tol (tol = 1e-2 + 1e-2 |ref|) per entry point and fails on the first case over 1.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import c_oracle, recconv_np
from recnext_amd import _lib, ops

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
CASES = int(os.environ.get("CASES", "150"))
lib = _lib.load()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bf = lambda a: torch.from_numpy(a).bfloat16().float().numpy()
worst = {"qkcore": 0.0, "down_qkcore": 0.0, "unit": 0.0}
counts = {"qkcore1": 0, "qkcore2": 0, "unsupported": 0, "down_qkcore": 0, "unit": 0}


def params(c):
    w_qk = (rng.standard_normal((2 * c, c // 2, 1, 1)) * (2.0 / c) ** 0.5).astype(np.float32)
    b_qk = (rng.standard_normal(2 * c) * 0.1).astype(np.float32)
    w_pe = (rng.standard_normal((c, 1, 3, 3)) * 0.2).astype(np.float32)
    b_pe = (rng.standard_normal(c) * 0.1).astype(np.float32)
    return w_qk, b_qk, w_pe, b_pe


def check(name, got, ref, what):
    r = float((np.abs(got - ref) / (1e-2 + 1e-2 * np.abs(ref))).max())
    worst[name] = max(worst[name], r)
    if not r <= 1.0:
        raise SystemExit(f"FAIL {name} {what}: err/tol {r}")


for i in range(CASES):
    heads = int(rng.choice([1, 2, 4, 8, 16]))
    d_head = 32 if heads == 1 or rng.random() < 0.5 else int(rng.choice([4, 8, 12, 16, 20, 24, 28]))      # round 5: heads padded to 32 inside the kernels
    c = d_head * heads
    h, w = (int(v) for v in rng.integers(1, 49, 2))
    if rng.random() < 0.3:
        h, w = (int(v) for v in rng.integers(1, 10, 2))            # short planes more often
    b = int(rng.integers(1, 4))
    if b * h * w * c > 6e6:
        b = 1
    n_launch = lib.rcx_recattn_qkcore_launches(b, h, w, c, heads)
    w_qk, b_qk, w_pe, b_pe = params(c)
    if n_launch == 0:
        counts["unsupported"] += 1
        assert not ops.recattn_qkcore_supported(c, heads, h, w)
        continue
    counts[f"qkcore{n_launch}"] += 1
    d = rng.standard_normal((b, c, h, w)).astype(np.float32)
    ref = recconv_np.linear_attention(d.astype(np.float64), w_qk, b_qk, w_pe, b_pe, heads, variant=1)
    has_b = rng.random() < 0.7
    if not has_b:
        ref = ref - b_pe[None, :, None, None]
    got = ops.recattn_qkcore(t(d).contiguous(memory_format=torch.channels_last), t(w_qk[:, :, 0, 0]).to(torch.bfloat16).contiguous(), t(b_qk),
                             ops.pack_dw_weight(t(w_pe)), ops.pack_bias(t(b_pe)) if has_b else None, heads).cpu().numpy()
    check("qkcore", got, ref, (b, c, heads, h, w, n_launch))

for i in range(CASES // 3):
    hw = int(rng.choice([14, 7]))
    heads = int(rng.choice([1, 2, 4, 8] + ([16] if hw == 7 else [])))
    d_head = 32 if heads == 1 or rng.random() < 0.5 else int(rng.choice([8, 16, 20, 24, 28]))
    c = d_head * heads
    b = int(rng.integers(1, 5))
    xdt = torch.bfloat16 if rng.random() < 0.6 else torch.float16
    rnd = bf if xdt == torch.bfloat16 else (lambda a: a.astype(np.float16).astype(np.float32))
    x = rnd(rng.standard_normal((b, c, hw, hw)).astype(np.float32))
    w_dn = (rng.standard_normal((c, 1, 5, 5)) * 0.2).astype(np.float32)
    b_dn = (rng.standard_normal(c) * 0.1).astype(np.float32)
    w_cv = (rng.standard_normal((c, 1, 5, 5)) * 0.2).astype(np.float32)
    b_cv = (rng.standard_normal(c) * 0.1).astype(np.float32)
    w_qk, b_qk, w_pe, b_pe = params(c)
    a_ref = recconv_np.linear_attention(c_oracle.dwconv2d(x, w_dn, b_dn, 2).astype(np.float64), w_qk, b_qk, w_pe, b_pe, heads, variant=1)
    xx = t(x).to(xdt).contiguous(memory_format=torch.channels_last)
    wdn, bdn, wcv, bcv = ops.pack_dw_weight(t(w_dn)), ops.pack_bias(t(b_dn)), ops.pack_dw_weight(t(w_cv)), ops.pack_bias(t(b_cv))
    wpe, bpe = ops.pack_dw_weight(t(w_pe)), ops.pack_bias(t(b_pe))
    wqk16 = t(w_qk[:, :, 0, 0]).to(torch.bfloat16).contiguous()
    assert ops.recattn_down_qkcore_supported(c, heads, hw, hw, xdt)
    got = ops.recattn_down_qkcore(xx, wdn, bdn, wqk16, t(b_qk), wpe, bpe, heads).cpu().numpy()
    check("down_qkcore", got, a_ref, (b, c, heads, hw, str(xdt)))
    counts["down_qkcore"] += 1
    if ops.recattn2d_supported(c, heads, hw, hw, "nearest", xdt):
        ref = c_oracle.dwconv2d(c_oracle.add_resized(x, a_ref.astype(np.float32), "nearest"), w_cv, b_cv, 1)
        got = ops.recattn2d(xx, wdn, bdn, wqk16, t(b_qk), wpe, bpe, wcv, bcv, heads).float().cpu().numpy()
        check("unit", got, ref, (b, c, heads, hw, str(xdt)))
        counts["unit"] += 1
    else:
        raise SystemExit(f"no one-launch unit for {(b, c, heads, hw)}")          # round 5: 16 heads too (two per wave)
print("cases", counts)
print("worst err/tol", {k: round(v, 3) for k, v in worst.items()})
