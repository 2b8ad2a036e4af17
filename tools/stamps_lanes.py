#!/usr/bin/env python3
"""Phase timeline of the banded register-resident kernel from the diagnostic (-DRCX_STAMPS) build (development tool).

    make -C recnext_amd/csrc diag && RCX_LIBRARY=recnext_amd/lib/librecnext_amd_diag.so python tools/stamps_lanes.py --shape 256,64,56,56,4
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import recnext_amd
from recnext_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="256,64,56,56,4")
ap.add_argument("--dtype", default="bf16")
args = ap.parse_args()
n, c, h, w, level = map(int, args.shape.split(","))
dev = torch.device("cuda:0")
dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
_lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
raw.rcx_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
assert raw.rcx_debug_set_stamp_buffer(buf.data_ptr()) == 0
mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level).to(dev).eval()
x = torch.randn(n, c, h, w, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(3):
        mod(x)
    torch.cuda.synchronize()
    buf.zero_()
    mod(x)
    torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(256, 64).astype(np.float64)
names = {0: "start", 1: "pass 1 done", 2: "levels>=1 done", 3: "end of first image"}
for s in range(4):
    names[8 + 3 * s] = f"p1 band{s} staged"
    names[9 + 3 * s] = f"p1 band{s} barrier passed"
    names[10 + 3 * s] = f"p1 band{s} computed"
    names[24 + 4 * s] = f"p2 band{s} staged"
    names[25 + 4 * s] = f"p2 band{s} barrier passed"
    names[26 + 4 * s] = f"p2 band{s} y(b-2) stored"
    names[27 + 4 * s] = f"p2 band{s} computed"
names.update({32: "whole: start", 33: "whole: taps + x written to LDS", 34: "whole: barrier passed", 35: "whole: computed", 36: "whole: 2nd barrier", 37: "whole: y stored"})
if (st[:, 32] > 0).any():                      # whole-plane kernel: relative to the earliest workgroup's start
    t_first = st[st[:, 32] > 0, 32].min()
    print("workgroup starts after the first one (kilo-ticks): median %.3f  p90 %.3f  max %.3f" % tuple(
        np.percentile((st[st[:, 32] > 0, 32] - t_first) / 1000.0, q) for q in (50, 90, 100)))
    st[:, 0] = st[:, 32]
valid = st[:, 0] > 0
rel = (st[valid] - st[valid, 0:1]) / 1000.0
print("workgroups sampled:", int(valid.sum()), "(kilo-ticks of the 100 MHz-class s_memtime counter since the kernel-local start; median over workgroups)")
order = sorted(names, key=lambda k: np.median(rel[:, k][st[valid, k] > 0]) if (st[valid, k] > 0).any() else 1e18)
for k in order:
    col = rel[:, k][st[valid, k] > 0]
    if len(col):
        print(f"  {names[k]:28s} median {np.median(col):9.3f}   p10 {np.percentile(col, 10):9.3f}  p90 {np.percentile(col, 90):9.3f}")
