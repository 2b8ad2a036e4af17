#!/usr/bin/env python3
"""One fused step y = conv5(x + resize(coarse)) + bias (RecAttn2d's kernel, rcx_upadd_dwconv_fwd) at RecNeXt-A3's stage shapes:
wall time per launch and fraction of the HBM roofline (x + coarse + y bytes over 8 TB/s).  RCX_UPADD_CPL=0 / RCX_UPADD_CPT=0: the round-1
lane kernels instead of the whole-plane (14 x 14) / tiled (56 x 56, 28 x 28; round 3) channel-per-lane ones.  Second argument: extra shapes
"CxHxW,..." (e.g. 64x128x128 is not a multiple of 14 wide and stays on the lane kernels; 64x200x336 is a COCO stage)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from recnext_amd import ops

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shapes = [(64, 56, 56), (128, 28, 28), (256, 14, 14), (512, 7, 7)]
if len(sys.argv) > 2:
    shapes = [tuple(int(v) for v in sh.split("x")) for sh in sys.argv[2].split(",")]
cdt = torch.float32 if os.environ.get("COARSE_F32") else torch.bfloat16
for c, h, wd in shapes:
    for mode in ("nearest", "bilinear"):
        x = torch.randn(n, c, h, wd, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        cs = torch.randn(n, c, (h + 1) // 2, (wd + 1) // 2, device=dev).to(cdt).contiguous(memory_format=torch.channels_last)
        w = ops.pack_dw_weight(torch.randn(c, 1, 5, 5, device=dev) * 0.2)
        b = ops.pack_bias(torch.randn(c, device=dev))
        for _ in range(5):
            ops.upadd_dwconv(x, cs, w, b, k=5, mode=mode)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            ops.upadd_dwconv(x, cs, w, b, k=5, mode=mode)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 50 * 1e3
        nbytes = 2 * 2 * x.numel() + cs.numel() * cs.element_size()
        plan = ops.upadd_dwconv_plan(n, c, h, wd, cs.shape[2], cs.shape[3], 5, mode, torch.bfloat16, cdt)
        print(json.dumps({"shape": [n, c, h, wd], "mode": mode, "plan": plan.split("(")[0], "coarse": str(cdt)[6:], "dtype": "bf16", "us": round(us, 1), "GB/s": round(nbytes / us / 1e3, 1),
                          "frac_of_8TBs": round(nbytes / us / 1e3 / 8000, 3), "cpl": os.environ.get("RCX_UPADD_CPL", "1")}), flush=True)
