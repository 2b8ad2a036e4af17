#!/usr/bin/env python3
"""One step of the down ladder, y = conv5 stride 2 (x) + bias (rcx_dwconv2d_fwd; RecAttn2d's `down` conv): wall time per launch and
fraction of the HBM roofline (x + y bytes over 8 TB/s).  RCX_UPADD_CPT=0: the kernels the tiled channel-per-lane kernel replaced.
    python3 tools/bench_down5.py [N=256] [CxHxW,...] ; OUT_F32=1: float32 output (the split / generic schedules' F_1)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from recnext_amd import ops

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shapes = [(64, 56, 56), (128, 28, 28)]
if len(sys.argv) > 2:
    shapes = [tuple(int(v) for v in sh.split("x")) for sh in sys.argv[2].split(",")]
odt = torch.float32 if os.environ.get("OUT_F32") else torch.bfloat16
for c, h, wd in shapes:
    x = torch.randn(n, c, h, wd, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = ops.pack_dw_weight(torch.randn(c, 1, 5, 5, device=dev) * 0.2)
    b = ops.pack_bias(torch.randn(c, device=dev))
    for _ in range(5):
        y = ops.dwconv2d(x, w, b, k=5, stride=2, out_dtype=odt)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        ops.dwconv2d(x, w, b, k=5, stride=2, out_dtype=odt)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 50 * 1e3
    nbytes = 2 * x.numel() + y.numel() * y.element_size()
    print(json.dumps({"shape": [n, c, h, wd], "out": str(odt)[6:], "us": round(us, 1), "GB/s": round(nbytes / us / 1e3, 1),
                      "frac_of_8TBs": round(nbytes / us / 1e3 / 8000, 3), "tiled": os.environ.get("RCX_UPADD_CPT", "1")}), flush=True)
