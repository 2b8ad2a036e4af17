#!/usr/bin/env python3
"""Sweep the plane-schedule knobs (RCX_PLANE_LPP / _B2 / _NT) per block shape (development tool)."""
import argparse
import itertools
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import recnext_amd
from tools.bench_blocks import SHAPES, time_fn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", default="m3")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--lpps", default="4,8,16,32")
    ap.add_argument("--b2s", default="2,4,8,0")      # 0 = whole plane
    ap.add_argument("--nts", default="128,256,512")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    eb = 2 if args.dtype == "bf16" else 4
    for sname in args.sets.split(","):
        for (n, c, h, w, level) in SHAPES[sname]:
            mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level).to(dev).eval()
            x = torch.randn(n, c, h, w, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
            alg = 2 * n * c * h * w * eb + (level + 2) * c * 25 * eb
            results = []
            with torch.no_grad():
                ref = None
                for lpp, b2, nt in itertools.product(args.lpps.split(","), args.b2s.split(","), args.nts.split(",")):
                    os.environ["RCX_PLANE_LPP"] = lpp
                    os.environ["RCX_PLANE_B2"] = str(h if b2 == "0" else b2)
                    os.environ["RCX_PLANE_NT"] = nt
                    try:
                        if not recnext_amd.ops.recconv2d_plan(n, c, h, w, level, 5, "bilinear", dtype).startswith("plane"):
                            continue
                        y = mod(x)
                        torch.cuda.synchronize()
                    except Exception as e:      # noqa: BLE001
                        print("skip", lpp, b2, nt, str(e)[:80])
                        continue
                    if ref is None:
                        ref = y
                    elif not torch.equal(ref, y):
                        print("MISMATCH", lpp, b2, nt, float((ref.float() - y.float()).abs().max()))
                    med, mn = time_fn(lambda: mod(x), args.iters)
                    results.append((med, lpp, b2, nt))
            for k in ("RCX_PLANE_LPP", "RCX_PLANE_B2", "RCX_PLANE_NT"):
                os.environ.pop(k, None)
            results.sort()
            print(json.dumps({"shape": [n, c, h, w, level], "dtype": args.dtype,
                              "best": [{"ms": round(r[0], 4), "lpp": r[1], "b2": r[2], "nt": r[3],
                                        "frac": round(alg / r[0] / 1e6 / 8000, 4)} for r in results[:6]],
                              "worst_ms": round(results[-1][0], 4) if results else None, "n": len(results)}), flush=True)


if __name__ == "__main__":
    main()
