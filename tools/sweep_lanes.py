#!/usr/bin/env python3
"""Time the register-resident schedule over its launch knobs (development tool): RCX_LANES_WAVES x RCX_LANES_NI."""
import argparse
import itertools
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import recnext_amd
from recnext_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="256,256,14,14,2;256,512,7,7,1")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--waves", default="8,4,2,1")
ap.add_argument("--ni", default="1,2,4,8")
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
dev = torch.device("cuda:0")
dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
for shp in args.shapes.split(";"):
    n, c, h, w, level = map(int, shp.split(","))
    mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level).to(dev).eval()
    x = torch.randn(n, c, h, w, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    alg = 2 * x.numel() * x.element_size() + (level + 2) * c * 25 * x.element_size()
    for wv, ni in itertools.product(args.waves.split(","), args.ni.split(",")):
        os.environ["RCX_LANES_WAVES"], os.environ["RCX_LANES_NI"] = wv, ni
        with torch.no_grad():
            for _ in range(3):
                mod(x)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.iters):
                mod(x)
            e.record()
            torch.cuda.synchronize()
        ms = s.elapsed_time(e) / args.iters
        print(json.dumps({"shape": [n, c, h, w, level], "waves": int(wv), "ni": int(ni), "plan": ops.recconv2d_plan(n, c, h, w, level, 5, "bilinear", dtype),
                          "us": round(ms * 1e3, 1), "frac": round(alg / ms / 1e6 / 8e3, 4)}), flush=True)
