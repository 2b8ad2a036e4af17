"""ctypes front-end of the C oracle (oracle/recconv_c.c) -- TEST INFRASTRUCTURE ONLY.

Loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product.
Arrays are logical NCHW numpy float32 (like the reference's tensors); they are handed to the C code
as NHWC, the layout the product kernels consume.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librcx_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "recconv_c.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        _lib.rcx_oracle_recconv2d_nhwc_f32.argtypes = [fp, fp, fp, fp, fp, fp] + [ctypes.c_int] * 8
        _lib.rcx_oracle_recconv2d_nhwc_f32.restype = ctypes.c_int
        _lib.rcx_oracle_dwconv2d_nhwc_f32.argtypes = [fp, fp, fp, fp] + [ctypes.c_int] * 6
        _lib.rcx_oracle_dwconv2d_nhwc_f32.restype = None
        _lib.rcx_oracle_add_resized_nhwc_f32.argtypes = [fp, fp, fp] + [ctypes.c_int] * 7
        _lib.rcx_oracle_add_resized_nhwc_f32.restype = None
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def to_nhwc(x):
    return np.ascontiguousarray(np.transpose(np.asarray(x, dtype=np.float32), (0, 2, 3, 1)))


def to_nchw(x):
    return np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2)))


def recconv2d_nhwc(x_nhwc, w_down, w_convs, b_down=None, b_convs=None, level=None, mode="bilinear", threads=0):
    """x_nhwc: (N,H,W,C) float32 contiguous. Returns (N,H,W,C)."""
    x_nhwc = _f32(x_nhwc)
    n, h, w, c = x_nhwc.shape
    w_down = _f32(w_down)
    w_convs = _f32(np.stack([np.asarray(a) for a in w_convs]))
    k = w_down.shape[-1]
    if level is None:
        level = w_convs.shape[0] - 1
    assert w_convs.shape[0] == level + 1
    b_down = _f32(b_down)
    b_convs = None if b_convs is None else _f32(np.stack([np.asarray(a) for a in b_convs]))
    y = np.empty_like(x_nhwc)
    rc = lib().rcx_oracle_recconv2d_nhwc_f32(_p(x_nhwc), _p(y), _p(w_down), _p(b_down), _p(w_convs), _p(b_convs),
                                             n, c, h, w, level, k, 0 if mode == "bilinear" else 1, threads)
    if rc != 0:
        raise ValueError("rcx_oracle_recconv2d_nhwc_f32: bad arguments")
    return y


def recconv2d(x, w_down, w_convs, b_down=None, b_convs=None, level=None, mode="bilinear", threads=0):
    """Logical-NCHW convenience wrapper."""
    return to_nchw(recconv2d_nhwc(to_nhwc(x), w_down, w_convs, b_down, b_convs, level, mode, threads))


def dwconv2d(x, w, b=None, stride=1):
    xh = to_nhwc(x)
    n, h, wd, c = xh.shape
    w = _f32(w)
    k = w.shape[-1]
    p = k // 2
    ho, wo = (h + 2 * p - k) // stride + 1, (wd + 2 * p - k) // stride + 1
    out = np.empty((n, ho, wo, c), dtype=np.float32)
    b = _f32(b)
    lib().rcx_oracle_dwconv2d_nhwc_f32(_p(xh), _p(out), _p(w), _p(b), n, c, h, wd, k, stride)
    return to_nchw(out)


def add_resized(base, src, mode="bilinear"):
    bh, sh = to_nhwc(base), to_nhwc(src)
    n, ho, wo, c = bh.shape
    _, hi, wi, _ = sh.shape
    out = np.empty_like(bh)
    lib().rcx_oracle_add_resized_nhwc_f32(_p(bh), _p(sh), _p(out), n, c, hi, wi, ho, wo, 0 if mode == "bilinear" else 1)
    return to_nchw(out)
