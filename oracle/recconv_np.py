"""NumPy restatement of the RecConv2d / RecAttn2d hot path -- TEST INFRASTRUCTURE ONLY.

This file is the *oracle*: a CPU restatement of the reference's algorithm used by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg as the
checker.  Nothing under ``recnext_amd/`` (the product) may import it.

Parity status: PINNED against golden vectors generated in the build container by
importing the reference modules themselves (``tests/golden/make_golden.py``; the
reference ships no tests or fixtures of its own for this path -- SURVEY.md section 8c).

Reference anchors (paths relative to the upstream repository root):
  * control flow and conv<->level pairing ........ model/recnext.py:24-34
  * depthwise conv parameters (pad k//2, groups=C) . model/recnext.py:13-22
  * resize semantics (size=, mode=) ............... model/recnext.py:33, model/recattn.py:67
  * RecAttn2d block ............................... model/recattn.py:54-67
  * linear attention 1 / 2 ........................ model/recattn.py:8-28 / 31-51
  * BN folding .................................... model/recnext.py:75-97
The arithmetic of conv / interpolate itself lives in PyTorch ATen (third party, not in
the reference tree, version unpinned by requirements.txt:1); its published semantics
(cross-correlation with zero padding; bilinear align_corners=False; legacy "nearest")
are restated below and checked against the ATen build in this image by the golden
vectors.

All arrays are logical NCHW like the reference's tensors.
"""
import numpy as np

__all__ = [
    "down_size", "ladder_sizes", "dwconv2d", "dwconv2d_mult", "bilinear_axis_table", "nearest_axis_table",
    "resize", "recconv2d", "recconv2d_trace", "fold_bn", "linear_attention", "recattn2d",
]


def down_size(h, k):
    """Output extent of the stride-2, pad k//2 depthwise conv (model/recnext.py:21)."""
    p = k // 2
    return (h + 2 * p - k) // 2 + 1


def ladder_sizes(h, w, level, k):
    """[(H_0,W_0), ..., (H_level,W_level)] -- the sizes recorded at model/recnext.py:28."""
    out = [(h, w)]
    for _ in range(level):
        h, w = down_size(h, k), down_size(w, k)
        out.append((h, w))
    return out


def dwconv2d(x, w, b=None, stride=1):
    """Depthwise cross-correlation, zero padding k//2 (nn.Conv2d(groups=C), model/recnext.py:13-22).

    x: (N,C,H,W); w: (C,1,k,k) or (C,k,k); b: (C,) or None.
    """
    x = np.asarray(x)
    w = np.asarray(w).reshape(w.shape[0], w.shape[-2], w.shape[-1])
    n, c, h, wd = x.shape
    k = w.shape[-1]
    p = k // 2
    ho = (h + 2 * p - k) // stride + 1
    wo = (wd + 2 * p - k) // stride + 1
    xp = np.zeros((n, c, h + 2 * p, wd + 2 * p), dtype=x.dtype)
    xp[:, :, p:p + h, p:p + wd] = x
    out = np.zeros((n, c, ho, wo), dtype=x.dtype)
    for u in range(k):
        for v in range(k):
            win = xp[:, :, u:u + stride * (ho - 1) + 1:stride, v:v + stride * (wo - 1) + 1:stride]
            out += win * w[None, :, u, v, None, None].astype(x.dtype)
    if b is not None:
        out += np.asarray(b, dtype=x.dtype)[None, :, None, None]
    return out


def dwconv2d_mult(x, w, b=None, stride=1, mult=2):
    """nn.Conv2d(C, mult*C, k, stride, padding=k//2, groups=C): output channel o reads input channel o // mult
    (Downsample.token_mixer, model/recnext.py:165)."""
    x = np.asarray(x)
    return dwconv2d(np.repeat(x, mult, axis=1), w, b, stride)


def bilinear_axis_table(n_in, n_out):
    """ATen upsample_bilinear2d, align_corners=False, per axis.

    Returns (i0, i1, lam) with out[d] = (1-lam[d])*v[i0[d]] + lam[d]*v[i1[d]].
    Computed in float32 like ATen's area_pixel_compute_source_index.
    """
    scale = np.float32(n_in) / np.float32(n_out)
    d = np.arange(n_out, dtype=np.float32)
    src = scale * (d + np.float32(0.5)) - np.float32(0.5)
    src = np.maximum(src, np.float32(0.0))
    i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
    i1 = i0 + (i0 < n_in - 1)
    lam = (src - i0.astype(np.float32)).astype(np.float32)
    return i0, i1, lam


def nearest_axis_table(n_in, n_out):
    """ATen upsample_nearest2d (legacy 'nearest', not 'nearest-exact'): i = min(floor(d*in/out), in-1)."""
    scale = np.float32(n_in) / np.float32(n_out)
    d = np.arange(n_out, dtype=np.float32)
    return np.minimum(np.floor(d * scale).astype(np.int64), n_in - 1)


def resize(x, size, mode):
    """F.interpolate(x, size=size, mode=mode) for mode in {'bilinear','nearest'} (model/recnext.py:33)."""
    x = np.asarray(x)
    ho, wo = size
    hi, wi = x.shape[2:]
    if mode == "nearest":
        iy = nearest_axis_table(hi, ho)
        ix = nearest_axis_table(wi, wo)
        return x[:, :, iy][:, :, :, ix]
    if mode != "bilinear":
        raise ValueError(f"unsupported mode {mode!r}")
    y0, y1, ly = bilinear_axis_table(hi, ho)
    x0, x1, lx = bilinear_axis_table(wi, wo)
    ly = ly.astype(x.dtype)[None, None, :, None]
    lx = lx.astype(x.dtype)[None, None, None, :]
    top = x[:, :, y0]
    bot = x[:, :, y1]
    # ATen order: w00*v00 + w01*v01 + w10*v10 + w11*v11 with w = (1-ly|ly)*(1-lx|lx)
    one = np.asarray(1, dtype=x.dtype)
    return ((one - ly) * ((one - lx) * top[:, :, :, x0] + lx * top[:, :, :, x1])
            + ly * ((one - lx) * bot[:, :, :, x0] + lx * bot[:, :, :, x1]))


def recconv2d_trace(x, w_down, w_convs, b_down=None, b_convs=None, level=None, mode="bilinear"):
    """RecConv2d.forward with every intermediate kept (model/recnext.py:24-34).

    w_convs: sequence of level+1 arrays (C,1,k,k); convs[0] pairs with the COARSEST feature,
    convs[level-1] with F_1, convs[level] is the final full-resolution conv
    (zip(self.convs, reversed(features)), model/recnext.py:32-34).
    Returns dict with F (list F_0..F_level), U (dict l -> U_l) and y.
    """
    x = np.asarray(x)
    if level is None:
        level = len(w_convs) - 1
    assert len(w_convs) == level + 1
    if b_convs is None:
        b_convs = [None] * (level + 1)
    feats = [x]
    for _ in range(level):                                   # down ladder, shared weight (:27-29)
        feats.append(dwconv2d(feats[-1], w_down, b_down, stride=2))
    ups = {}
    u = None                                                 # "x = 0" (:31)
    for j, l in enumerate(range(level, 0, -1)):              # coarsest first (:32-33)
        t = feats[l] if u is None else feats[l] + u
        cv = dwconv2d(t, w_convs[j], b_convs[j], stride=1)
        u = resize(cv, feats[l - 1].shape[2:], mode)
        ups[l] = u
    t = feats[0] if u is None else feats[0] + u              # level == 0 is legal: convs[0](x)
    y = dwconv2d(t, w_convs[level], b_convs[level], stride=1)  # (:34)
    return {"F": feats, "U": ups, "y": y}


def recconv2d(x, w_down, w_convs, b_down=None, b_convs=None, level=None, mode="bilinear"):
    return recconv2d_trace(x, w_down, w_convs, b_down, b_convs, level, mode)["y"]


def fold_bn(w, conv_bias, gamma, beta, mean, var, eps=1e-5):
    """ConvNorm.fuse (model/recnext.py:75-97): w' = w*g/sqrt(var+eps); b' = beta - mean*g/sqrt(var+eps) (+ scaled conv bias)."""
    s = gamma / np.sqrt(var + eps)
    b = beta - s * mean
    if conv_bias is not None:
        b = b + s * conv_bias
    return w * s.reshape(-1, *([1] * (w.ndim - 1))), b


def _elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def linear_attention(x, w_qk, b_qk, w_pe, b_pe, num_heads, variant=1):
    """LinearAttention1/2 after BN folding (model/recattn.py:8-28, 31-51).

    x: (B,C,h,w); w_qk: (2C, C/2, 1, 1) grouped 1x1 conv with groups=2; w_pe: (C,1,3,3) depthwise.
    """
    x = np.asarray(x)
    b, c, h, w = x.shape
    n = h * w
    s = n ** -0.5
    hd = c // num_heads
    xf = x.reshape(b, c, n)
    half = c // 2
    wq = np.asarray(w_qk).reshape(2 * c, half)
    qk = np.empty((b, 2 * c, n), dtype=x.dtype)
    for g in range(2):                                        # groups=2: out [g*C,(g+1)*C) sees in [g*C/2,(g+1)*C/2)
        qk[:, g * c:(g + 1) * c] = np.einsum("oi,bin->bon", wq[g * c:(g + 1) * c], xf[:, g * half:(g + 1) * half])
    qk = qk + np.asarray(b_qk, dtype=x.dtype)[None, :, None]
    qk = _elu(qk) + 1.0
    qk = qk.reshape(b, 2, num_heads, hd, n)
    q, k = qk[:, 0], qk[:, 1]
    v = xf.reshape(b, num_heads, hd, n)
    qt = np.swapaxes(q, -1, -2)                               # (b,heads,n,hd)
    vt = np.swapaxes(v, -1, -2)
    if variant == 1:
        kv = (k * s) @ (vt * s)                               # (b,heads,hd,hd)
        out = qt @ kv / (qt @ k.mean(axis=-1, keepdims=True) + 1e-6)
    else:
        a = qt @ k                                            # (b,heads,n,n)
        a = a / (a.mean(axis=-1, keepdims=True) + 1e-6)
        out = (a * s) @ (vt * s)
    out = np.swapaxes(out, -1, -2).reshape(b, c, h, w)
    return out + dwconv2d(x, w_pe, b_pe, stride=1)


def recattn2d(x, w_down, b_down, attn, w_conv, b_conv, mode="nearest"):
    """RecAttn2d.forward after fuse (model/recattn.py:66-67): conv(x + resize(LA(down(x)), size(x))).

    attn: callable applied to the stride-2 depthwise output (the linear attention), or None.
    """
    d = dwconv2d(x, w_down, b_down, stride=2)
    if attn is not None:
        d = attn(d)
    return dwconv2d(x + resize(d, x.shape[2:], mode), w_conv, b_conv, stride=1)
