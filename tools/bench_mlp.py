#!/usr/bin/env python3
"""Time the fused channel mixer (rcx_channel_mlp_fwd) against the four library launches it replaces, on the stage shapes of a model at batch 256 (development tool)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from recnext_amd import ops

dev = torch.device("cuda:0")
REPS = int(os.environ.get("REPS", "30"))
SHAPES = [(256, 64, 128, 56), (256, 128, 256, 28), (256, 256, 512, 14), (256, 192, 384, 14), (256, 160, 320, 28), (256, 320, 640, 14), (256, 48, 96, 56), (256, 96, 192, 28), (256, 80, 160, 56)]


def timed(fn, n):
    for i in range(3):
        fn(i % n)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for i in range(REPS):
        fn(i % n)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / REPS * 1e3


for b, c, hid, hw in SHAPES:
    npool = max(2, int(700e6 / (b * c * hw * hw * 2 * 2)))
    zs = [torch.randn(b, c, hw, hw, device=dev).bfloat16().contiguous(memory_format=torch.channels_last) for _ in range(npool)]
    xs = [torch.randn(b, c, hw, hw, device=dev).bfloat16().contiguous(memory_format=torch.channels_last) for _ in range(npool)]
    w1, b1 = (torch.randn(hid, c, device=dev) * 0.1).bfloat16(), torch.randn(hid, device=dev).bfloat16()
    w2, b2 = (torch.randn(c, hid, device=dev) * 0.1).bfloat16(), torch.randn(c, device=dev).bfloat16()
    wfrag, bias, hp = ops.pack_channel_mlp(w1, b1, w2, b2, hidden_to=ops.channel_mlp_hidden(b * hw * hw, c, hid, torch.bfloat16))
    m = b * hw * hw

    def lib(i):
        zz = zs[i].permute(0, 2, 3, 1).reshape(m, c)
        o = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(zz, w1, b1)), w2, b2)
        return xs[i] + o.view(b, hw, hw, c).permute(0, 3, 1, 2)

    with torch.no_grad():
        t_f = timed(lambda i: ops.channel_mlp(zs[i], xs[i], wfrag, bias, hp), npool)
        t_l = timed(lib, npool)
    bytes_ = 3 * m * c * 2
    print(json.dumps({"B": b, "C": c, "hidden": hid, "plane": hw, "fused_us": round(t_f, 1), "library_us": round(t_l, 1), "algorithmic_MB": round(bytes_ / 1e6, 1),
                      "fused_TBs": round(bytes_ / t_f / 1e6, 2), "mfma_TFLOPs": round(4 * m * c * hp / t_f / 1e6, 1)}))
