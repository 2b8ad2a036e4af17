#!/usr/bin/env python3
"""Time one RecAttn2d unit (inference, bf16 activations) and its launches on the RecNeXt-A3 stage shapes at batch 256 (development tool; HIP events
on the current stream, fresh inputs rotated through a pool larger than the L2 + MALL).  VERDICT r3 item 4 asks for the unit times of stages 2 / 3."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from recnext_amd import ops
from recnext_amd.recattn import RecAttn2d

dev = torch.device("cuda:0")
REPS = int(os.environ.get("REPS", "30"))


def timed(fn, pool):
    for i in range(3):
        fn(pool[i % len(pool)])
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for i in range(REPS):
        fn(pool[i % len(pool)])
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / REPS * 1e3


for stage, (b, c, h) in enumerate([(256, 64, 56), (256, 128, 28), (256, 256, 14), (256, 512, 7)]):
    heads = 2 ** (stage + 1)
    mod = RecAttn2d(c, num_heads=heads, stage=stage).to(dev).eval()
    npool = max(2, int(600e6 / (b * c * h * h * 2)))
    xs = [torch.randn(b, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last) for _ in range(npool)]
    with torch.no_grad():
        unit = timed(lambda x: mod(x), xs)
        wd, bd, wc, bc, wq, bq, wk, bk, wpe, bpe, wqk16, bqk = mod.packed_params()
        ds = [ops.dwconv2d(x, wd, bd, k=mod.kernel_size, stride=2, out_dtype=torch.float32) for x in xs]
        t_down = timed(lambda x: ops.dwconv2d(x, wd, bd, k=mod.kernel_size, stride=2, out_dtype=torch.float32), xs)
        fused = ops.recattn_qkcore_supported(c, heads, h // 2, h // 2)
        t_core = timed(lambda d: ops.recattn_qkcore(d, wqk16, bqk, wpe, bpe, heads), ds) if fused else None
        one = ops.recattn_down_qkcore_supported(c, heads, h, h, xs[0].dtype)        # stride-2 conv + coarse level in ONE launch (14 x 14 / 7 x 7 planes)
        t_one = timed(lambda x: ops.recattn_down_qkcore(x, wd, bd, wqk16, bqk, wpe, bpe, heads), xs) if one else None
        a = ops.recattn_qkcore(ds[0], wqk16, bqk, wpe, bpe, heads) if fused else torch.randn_like(ds[0])
        t_up = timed(lambda x: ops.upadd_dwconv(x, a, wc, bc, k=mod.kernel_size, mode=mod.mode), xs)
    print(json.dumps({"stage": stage, "B": b, "C": c, "plane": h, "heads": heads, "tokens": ((h + 1) // 2) ** 2, "unit_us": round(unit, 1),
                      "down_us": round(t_down, 1), "qk_core_pe_us": None if t_core is None else round(t_core, 1),
                      "down_qk_core_pe_one_launch_us": None if t_one is None else round(t_one, 1), "upadd_conv_us": round(t_up, 1)}))
