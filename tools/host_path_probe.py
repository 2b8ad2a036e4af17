#!/usr/bin/env python3
"""Where the host-side microseconds of one RecConv2d.forward call go (development tool): time.perf_counter around the pieces of the
Python path, the device idle-free (a tiny 7x7 block so that the queue never fills).
    python3 tools/host_path_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import recnext_amd
from recnext_amd import _lib, ops

dev = torch.device("cuda:0")
mod = recnext_amd.RecConv2d(512, kernel_size=5, level=1).to(dev).eval().bfloat16()
x = torch.randn(8, 512, 7, 7, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)


def per_call(fn, it=20000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    dt = (time.perf_counter() - t0) / it * 1e6
    torch.cuda.synchronize()
    return dt


with torch.no_grad():
    lib = _lib.load()
    wpack, bpack = mod.packed_params()
    n, c, h, w = x.shape
    y = torch.empty_like(x)
    st = torch.cuda.current_stream(dev).cuda_stream
    rows = [
        ("module call mod(x)", lambda: mod(x)),
        ("  mod.forward(x) (no nn.Module.__call__ hooks)", lambda: mod.forward(x)),
        ("    packed_params()", lambda: mod.packed_params()),
        ("    ops.recconv2d_forward", lambda: ops.recconv2d_forward(x, wpack, bpack, 1, 5, "bilinear")),
        ("      _nhwc(x)", lambda: ops._nhwc(x)),
        ("      _empty_nhwc", lambda: ops._empty_nhwc(n, c, h, w, x.dtype, x.device)),
        ("      workspace_bytes (ctypes)", lambda: lib.rcx_recconv2d_fwd_workspace_bytes(n, c, h, w, 1, 5, 1)),
        ("      with torch.cuda.device", lambda: torch.cuda.device(x.device).__enter__()),
        ("      _stream", lambda: ops._stream(x.device)),
        ("      the ctypes launch alone", lambda: lib.rcx_recconv2d_fwd(x.data_ptr(), y.data_ptr(), wpack.data_ptr(), None, None, 0, n, c, h, w, 1, 5, 0, 1, st)),
        ("      4 x data_ptr()", lambda: (x.data_ptr(), y.data_ptr(), wpack.data_ptr(), x.data_ptr())),
    ]
    for name, fn in rows:
        print(f"{per_call(fn):7.2f} us  {name}")
