#!/usr/bin/env python3
"""Phase timeline of the fused kernel from the diagnostic (-DRCX_STAMPS) build (development tool).

    make -C recnext_amd/csrc diag && RCX_LIBRARY=recnext_amd/lib/librecnext_amd_diag.so python tools/stamps.py --shape 256,64,56,56,4
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
lib = _lib.load()
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
names = {0: "start", 1: "pass1 done", 2: "ladder done", 3: "before C_1", 4: "up done", 5: "end"}
for b in range(8):
    names[8 + 3 * b] = f"p2 band{b} stage"
    names[9 + 3 * b] = f"p2 band{b} conv"
names.update({40: "b2 after stage_write", 41: "b2 after issue", 42: "b2 after conv (thread0)"})
valid = st[:, 0] > 0
t0 = st[valid, 0:1]
rel = (st[valid] - t0) / 1000.0    # kilo-cycles (s_memtime ticks are shader cycles)
print("workgroups sampled:", int(valid.sum()), " (kilo-cycles since kernel-local start, median over workgroups)")
for k in sorted(names):
    col = rel[:, k]
    col = col[st[valid, k] > 0]
    if len(col):
        print(f"  {names[k]:18s} median {np.median(col):8.2f}   p10 {np.percentile(col,10):8.2f}  p90 {np.percentile(col,90):8.2f}")
