#!/usr/bin/env python3
"""Per-block micro-benchmark of rcx_recconv2d_fwd on the BASELINE stage shapes (development tool).

Prints, per shape and dtype: mean launch duration (HIP events on the launch stream, interleaved rounds),
algorithmic GB/s (2*N*C*H*W*b + weights) and the fraction of the 8 TB/s HBM peak.
``--eager`` adds the reference's operator chain (ATen on the GPU: MIOpen depthwise conv + upsample + add).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import recnext_amd
from recnext_amd import ops

SHAPES = {
    "m3": [(256, 64, 56, 56, 4), (256, 128, 28, 28, 3), (256, 256, 14, 14, 2), (256, 512, 7, 7, 1)],
    "m3_b128": [(128, 64, 56, 56, 4), (128, 128, 28, 28, 3), (128, 256, 14, 14, 2), (128, 512, 7, 7, 1)],   # the training batch: inference forward beside bench_backward.py
    "m1": [(256, 48, 56, 56, 4), (256, 96, 28, 28, 3), (256, 192, 14, 14, 2), (256, 384, 7, 7, 1)],
    "m5": [(256, 80, 56, 56, 4), (256, 160, 28, 28, 3), (256, 320, 14, 14, 2), (256, 640, 7, 7, 1)],
    # the 28x28 block with channel counts that are not multiples of 64 (M0: 80, M1: 96, M2: 112, M5: 160), several batch sizes: the policy
    # of rcx_cpt.hip::cpt28_ragged (RCX_CPT=32 / 64 force one side)
    "ragged28": [(256, 80, 28, 28, 3), (256, 96, 28, 28, 3), (256, 112, 28, 28, 3), (256, 160, 28, 28, 3), (128, 96, 28, 28, 3), (128, 160, 28, 28, 3),
                 (64, 160, 28, 28, 3), (512, 96, 28, 28, 3)],
    # RecNeXt-M3 at 448 x 448 (twice the training resolution): 112 x 112 / level 4 is the split schedule (conv5 stride 2 -> the fused 56 x 56 block ->
    # conv5(x + resize)); its two outer steps are the tiled channel-per-lane kernels of rcx_upcpt.hip since round 3
    "m3_448": [(64, 64, 112, 112, 4), (64, 128, 56, 56, 3), (64, 256, 28, 28, 2), (64, 512, 14, 14, 1)],
    # the 14x14 / level 2 block with more waves than the chip has SIMDs (reload form, RCX_CPL14_RL=0 / 1 pins either form)
    "cpl14_waves": [(256, 256, 14, 14, 2), (256, 320, 14, 14, 2), (320, 256, 14, 14, 2), (512, 256, 14, 14, 2), (512, 320, 14, 14, 2)],
    "m3_512": [(32, 64, 128, 128, 4), (32, 128, 64, 64, 3), (32, 256, 32, 32, 2), (32, 512, 16, 16, 1)],
    # RecNeXt-M3 backbone on a COCO batch (detection/configs/_base_/datasets/coco_instance.py:9-12: 800 x 1344 padded, 2 images per GPU)
    "m3_coco": [(2, 64, 200, 336, 4), (2, 128, 100, 168, 3), (2, 256, 50, 84, 2), (2, 512, 25, 42, 1)],
}


def time_fn(fn, iters, rounds=3):
    best = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        best.append(s.elapsed_time(e) / iters)
    best.sort()
    return best[len(best) // 2], best[0]


def time_fresh(fn, x, iters):
    """As inside a model: x is WRITTEN by another kernel right before every launch (dirty lines in the producer's L2, nothing of it clean
    in the consumer's), and only the launch itself is bracketed.  Stand-alone loops over a read-only x rank some variants the other way
    round (profiles/archive/r03_cpt_cb16.txt, r03_cpt_28_ragged.txt)."""
    x0 = x.clone()
    other = torch.empty(256 * 1024 * 1024 // 4, device=x.device)             # 256 MB: what a model's other kernels leave in the caches
    ts = []
    for _ in range(iters):
        other.add_(1.0)
        x.copy_(x0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        ts.append((s, e))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in ts)
    return v[len(v) // 2], v[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", default="m3")
    ap.add_argument("--dtypes", default="bf16,fp32")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--mode", default="bilinear")
    ap.add_argument("--json", default=None)
    ap.add_argument("--taps", default="io", choices=["io", "fp32"],
                    help="io: the module's parameters have the activations' dtype (model.bfloat16()); "
                         "fp32: float32 parameters with 16-bit activations (exact taps, vector kernels)")
    ap.add_argument("--fresh", action="store_true", help="x rewritten (and 256 MB of other traffic) before every launch, each launch bracketed alone")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = []
    for sname in args.sets.split(","):
        for (n, c, h, w, level) in SHAPES[sname]:
            for dname in args.dtypes.split(","):
                dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "f16": torch.float16}.get(dname, torch.float32)
                eb = 4 if dtype == torch.float32 else 2
                torch.manual_seed(0)
                mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level, mode=args.mode).to(dev).eval()
                mx = args.taps == "io" and dtype != torch.float32
                if mx:
                    mod = mod.to(dtype)
                x = torch.randn(n, c, h, w, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
                with torch.no_grad():
                    for _ in range(3):
                        mod(x)
                    torch.cuda.synchronize()
                    med, mn = time_fresh(lambda: mod(x), x, args.iters) if args.fresh else time_fn(lambda: mod(x), args.iters)
                    alg = 2 * n * c * h * w * eb + (level + 2) * c * 25 * eb
                    row = {"set": sname, "shape": [n, c, h, w], "level": level, "dtype": dname, "timing": "fresh" if args.fresh else "loop",
                           "taps": "io" if mx else "fp32",
                           "plan": ops.recconv2d_plan(n, c, h, w, level, 5, args.mode, dtype),
                           "ms": med, "ms_min": mn, "alg_GBs": alg / med / 1e6, "frac_8TBs": alg / med / 1e6 / 8000}
                    if args.eager:
                        from oracle.torch_eager import EagerRecConv2d
                        ref = EagerRecConv2d(c, 5, False, level, args.mode).to(dev).to(dtype).to(memory_format=torch.channels_last).eval()
                        for _ in range(3):
                            ref(x)
                        torch.cuda.synchronize()
                        row["eager_ms"], _ = time_fn(lambda: ref(x), max(3, args.iters // 4))
                rows.append(row)
                print(json.dumps(row), flush=True)
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
