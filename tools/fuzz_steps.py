#!/usr/bin/env python3
"""Randomised parity sweep of the tiled single-step kernels (rcx_upcpt.hip) against the oracles (development tool): random even planes,
channel counts with whole and ragged 64-channel waves, every dtype pair, both resize modes, bias or none.  RCX_UPADD_CPT=all is set so that
the tiled kernels run wherever they apply.
    python3 tools/fuzz_steps.py [cases=300] [seed=0]"""
import os
import sys

os.environ["RCX_UPADD_CPT"] = "all"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import c_oracle, recconv_np
from recnext_amd import ops

DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
TOL = {"f32": (1e-4, 1e-4), "bf16": (1e-2, 1e-2), "f16": (2e-3, 2e-3)}


def rnd(a, dt):
    return torch.from_numpy(a).to(DT[dt]).float().numpy() if dt != "f32" else a


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    bad, hit = 0, {"upadd": 0, "down5": 0, "down7": 0}
    for it in range(cases):
        kind = ["upadd", "down5", "down7"][int(rng.integers(3))]
        lo = 14 if kind == "down7" else 28
        h, w = int(rng.integers(lo // 2, 76)) * 2, int(rng.integers(lo // 2, 76)) * 2
        c = int(rng.choice([8, 24, 40, 64, 72, 128, 136]))
        n = int(rng.choice([1, 2, 3]))
        if h * w * c * n > 3_000_000:
            c, n = min(c, 40), 1
        xdt = ["f32", "bf16", "f16"][int(rng.integers(3))]
        bias = bool(rng.random() < 0.5)
        x = rnd(rng.standard_normal((n, c, h, w)).astype(np.float32), xdt)
        if kind == "upadd":
            cdt = xdt if rng.random() < 0.5 else "f32"
            mode = "bilinear" if rng.random() < 0.5 else "nearest"
            cs = rnd(rng.standard_normal((n, c, h // 2, w // 2)).astype(np.float32), cdt)
            wt = (rng.standard_normal((c, 1, 5, 5)) * 0.2).astype(np.float32)
            b = rng.standard_normal(c).astype(np.float32) if bias else None
            ref = c_oracle.dwconv2d(c_oracle.add_resized(x, cs, mode), wt, b, 1)
            plan = ops.upadd_dwconv_plan(n, c, h, w, h // 2, w // 2, 5, mode, DT[xdt], DT[cdt])
            y = ops.upadd_dwconv(t(x).to(DT[xdt]), t(cs).to(DT[cdt]), ops.pack_dw_weight(t(wt)), ops.pack_bias(t(b)) if bias else None, k=5, mode=mode)
            odt, desc = xdt, f"{mode} coarse {cdt} {plan[:40]}"
            ok_plan = plan.startswith("upadd_cpt(")
        elif kind == "down5":
            odt = xdt if rng.random() < 0.5 else "f32"
            wt = (rng.standard_normal((c, 1, 5, 5)) * 0.2).astype(np.float32)
            b = rng.standard_normal(c).astype(np.float32) if bias else None
            ref = c_oracle.dwconv2d(x, wt, b, 2)
            y = ops.dwconv2d(t(x).to(DT[xdt]), ops.pack_dw_weight(t(wt)), ops.pack_bias(t(b)) if bias else None, k=5, stride=2, out_dtype=DT[odt])
            desc, ok_plan = f"out {odt}", True
        else:
            odt = xdt
            wt = (rng.standard_normal((2 * c, 1, 7, 7)) * 0.15).astype(np.float32)
            b = rng.standard_normal(2 * c).astype(np.float32) if bias else None
            ref = recconv_np.dwconv2d_mult(x.astype(np.float64), wt, b, stride=2)
            y = ops.dwconv2d_mult2(t(x).to(DT[xdt]), ops.pack_dw_weight(t(wt)), ops.pack_bias(t(b)) if bias else None, k=7, stride=2)
            desc, ok_plan = "", True
        hit[kind] += 1
        got = y.float().cpu().numpy()
        tol = TOL[odt]
        if not (ok_plan and got.shape == ref.shape and np.allclose(got, ref, atol=tol[0], rtol=tol[1])):
            bad += 1
            print(f"MISMATCH case {it}: {kind} {n}x{c}x{h}x{w} x {xdt} bias={bias} {desc} max|err|={np.abs(got - ref).max() if got.shape == ref.shape else 'shape'}", flush=True)
        elif it % 40 == 0:
            print(f"case {it}: {kind} {n}x{c}x{h}x{w} {xdt} {desc} ok ({np.abs(got - ref).max():.2e})", flush=True)
    print(f"{cases} cases, {bad} mismatches; {hit}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
