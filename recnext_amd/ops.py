"""Tensor-level front-end of the C ABI: torch supplies device memory and the HIP stream, nothing else.

Every function here requires CUDA(ROCm) tensors and the built HIP library; there is no CPU or
PyTorch-operator fallback (calling with a CPU tensor raises).
"""
import torch

from . import _lib

_DT = {torch.float32: _lib.DTYPE_F32, torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16}


_D16_CHECKED = set()            # device indices whose D16-hi zero-fill has been verified (rcx_selftest_d16)


def selftest_d16(device):
    """Run the library's start-up probe on `device` (once per process and device; raises RcxError if the hardware does not zero the other half
    of a D16 "hi" load's destination, which the bf16 load paths rely on -- include/recnext_amd.h).  Returns the failure mask (0 = fine)."""
    dev = torch.device(device)
    src = (torch.arange(64, device=dev, dtype=torch.int32) * 517 + 0x1234).to(torch.int16)        # 64 distinct 16-bit patterns
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.load().rcx_selftest_d16(src.data_ptr(), flag.data_ptr(), _stream(dev))
    _lib.check(rc, "rcx_selftest_d16")
    mask = int(flag.item())
    if mask:
        raise _lib.RcxError(f"D16 self-test failed on {dev} (mask {mask:#x}: bit 0 global_load_short_d16_hi, 1 buffer_load_short_d16_hi, 2 "
                            "ds_read_u16_d16_hi): this device does not zero the other half of a D16-hi load; the 16-bit kernels would return garbage")
    _D16_CHECKED.add(dev.index if dev.index is not None else torch.cuda.current_device())
    return mask


def _dt(t):
    try:
        code = _DT[t.dtype]
    except KeyError:
        raise TypeError(f"recnext_amd kernels take float32, bfloat16 or float16 tensors, got {t.dtype}") from None
    if code != _lib.DTYPE_F32 and t.is_cuda:
        idx = t.device.index
        if idx not in _D16_CHECKED:                      # once per device, before its first 16-bit launch
            if torch.cuda.is_current_stream_capturing():
                # the probe reads its flag back (a synchronisation), which a capturing stream cannot do -- and a captured launch that was never
                # probed would run unguarded on every replay: refuse, loudly, instead of skipping the guard
                raise _lib.RcxError(f"the first 16-bit recnext_amd launch on {t.device} happens while a HIP graph is being captured: call "
                                    "recnext_amd.ops.selftest_d16(device) (or run one eager forward) before the capture")
            selftest_d16(t.device)
            _D16_CHECKED.add(idx)
    return code


def _require_gpu(t, name):
    if not t.is_cuda:
        raise _lib.RcxError(f"{name} is on {t.device}: the recnext_amd token mixers only run as HIP kernels on a GPU "
                            "(no CPU fallback exists in the product path)")


def _nhwc(t, name="x"):
    """Logical N x C x H x W tensor whose storage is N x H x W x C contiguous (torch.channels_last)."""
    if t.dim() != 4:
        raise ValueError(f"{name} must be 4-D (N,C,H,W), got shape {tuple(t.shape)}")
    _require_gpu(t, name)
    n, c, h, w = t.shape
    want = (h * w * c, 1, w * c, c)
    ok = all(sz == 1 or st == wt for sz, st, wt in zip(t.shape, t.stride(), want))
    if not ok:
        t = t.contiguous(memory_format=torch.channels_last)
        if not all(sz == 1 or st == wt for sz, st, wt in zip(t.shape, t.stride(), want)):
            t = t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)   # degenerate-stride corner (C==1 / H==W==1)
    return t


def _empty_nhwc(n, c, h, w, dtype, device):
    return torch.empty((n, h, w, c), dtype=dtype, device=device).permute(0, 3, 1, 2)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device):
    """The current stream of `device` as a hipStream_t (an integer): the raw getter where this PyTorch has it (0.3 us against 2 us)."""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on(device):
    """Device guard for a launch: nothing when `device` is already the current device (the usual case; saves ~2 us a call)."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


def down_size(h, k):
    p = k // 2
    return (h + 2 * p - k) // 2 + 1


def pack_dw_weight(w, out=None):
    """(C,1,k,k) or (C,k,k) f32/bf16 parameter -> float32 (k,k,C) on the same device."""
    _require_gpu(w, "weight")
    c, k = w.shape[0], w.shape[-1]
    w = w.detach().contiguous()
    if out is None:
        out = torch.empty(k * k * c, dtype=torch.float32, device=w.device)
    with _on(w.device):
        _lib.check(_lib.load().rcx_pack_dw_weight(w.data_ptr(), out.data_ptr(), c, k, _dt(w), _stream(w.device)),
                   "rcx_pack_dw_weight")
    return out


def pack_bias(b, out=None):
    _require_gpu(b, "bias")
    b = b.detach().contiguous()
    if out is None:
        out = torch.empty(b.numel(), dtype=torch.float32, device=b.device)
    with _on(b.device):
        _lib.check(_lib.load().rcx_pack_bias(b.data_ptr(), out.data_ptr(), b.numel(), _dt(b), _stream(b.device)),
                   "rcx_pack_bias")
    return out


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


def pack_recconv_params(w_down, w_convs, b_down=None, b_convs=None, with_flipped=False):
    """-> (wpack (level+2, k*k*C) f32, bpack (level+2, C) f32 | None): [down, convs[0], ..., convs[level]].

    One launch for the whole block (rcx_pack_recconv_params).  with_flipped=True also returns the pack with every k x k
    rotated by 180 degrees -- the taps of the backward's transposed convolutions -- as a third value."""
    ws = [w_down] + list(w_convs)
    c, k = w_down.shape[0], w_down.shape[-1]
    _require_gpu(w_down, "weight")
    for w in ws:
        if tuple(w.shape[-2:]) != (k, k) or w.shape[0] != c:
            raise ValueError("all RecConv2d weights must share (C,1,k,k)")
    dev = w_down.device
    wpack = torch.empty((len(ws), k * k * c), dtype=torch.float32, device=dev)
    bs, bpack = None, None
    if b_down is not None:
        bs = [b_down] + list(b_convs)
        bpack = torch.empty((len(bs), c), dtype=torch.float32, device=dev)
    if any(p.dtype != w_down.dtype or p.device != dev for p in ws + (bs or [])):
        # parameters of mixed dtype: one launch per tensor
        for i, w in enumerate(ws):
            pack_dw_weight(w, wpack[i])
        for i, b in enumerate(bs or []):
            pack_bias(b, bpack[i])
        return (wpack, bpack, wpack.view(len(ws), k, k, c).flip(1, 2).reshape(len(ws), -1).contiguous()) if with_flipped else (wpack, bpack)
    ws = [w.detach().contiguous() for w in ws]
    bs = [b.detach().contiguous() for b in bs] if bs else None
    wflip = torch.empty_like(wpack) if with_flipped else None
    with _on(dev):
        rc = _lib.load().rcx_pack_recconv_params(_ptr_array(ws), _ptr_array(bs) if bs else None, wpack.data_ptr(),
                                                 wflip.data_ptr() if wflip is not None else None,
                                                 bpack.data_ptr() if bpack is not None else None,
                                                 len(ws), c, k, _dt(w_down), _stream(dev))
    _lib.check(rc, "rcx_pack_recconv_params")
    return (wpack, bpack, wflip) if with_flipped else (wpack, bpack)


def unpack_recconv_grads(gwpack, count, c, k):
    """gwpack (count, k*k*C) f32 -> (count, C, 1, k, k) f32, each [i] contiguous in the parameter's layout; one launch."""
    out = torch.empty((count, c, 1, k, k), dtype=torch.float32, device=gwpack.device)
    with _on(gwpack.device):
        rc = _lib.load().rcx_unpack_recconv_grads(gwpack.data_ptr(), _ptr_array([out[i] for i in range(count)]), count, c, k,
                                                  _stream(gwpack.device))
    _lib.check(rc, "rcx_unpack_recconv_grads")
    return out


def reload_options():
    """Re-read the RCX_* environment switches (the library reads them once): for tests and A/B tools that flip one inside a process."""
    _lib.load().rcx_reload_options()


def recconv2d_plan(n, c, h, w, level, k, mode, dtype):
    return _lib.load().rcx_recconv2d_fwd_plan(n, c, h, w, level, k, _lib.MODES[mode], _DT[dtype]).decode()


def recconv2d_forward(x, wpack, bpack, level, k, mode="bilinear"):
    """RecConv2d.forward (model/recnext.py:24-34) on the HIP kernels. Returns a channels_last tensor like x."""
    x = _nhwc(x)
    n, c, h, w = x.shape
    if mode not in _lib.MODES:
        raise ValueError(f"mode must be 'bilinear' or 'nearest', got {mode!r}")
    if wpack.numel() != (level + 2) * k * k * c or wpack.dtype != torch.float32:
        raise ValueError("wpack must be float32 of (level+2)*k*k*C elements (see pack_recconv_params)")
    lib = _lib.load()
    dt = _dt(x)
    y = _empty_nhwc(n, c, h, w, x.dtype, x.device)
    nbytes = lib.rcx_recconv2d_fwd_workspace_bytes(n, c, h, w, level, k, dt)    # 0 on the fused schedules: nothing to allocate
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    with _on(x.device):
        rc = lib.rcx_recconv2d_fwd(x.data_ptr(), y.data_ptr(), wpack.data_ptr(),
                                   bpack.data_ptr() if bpack is not None else None,
                                   ws.data_ptr() if ws is not None else None, nbytes, n, c, h, w, level, k, _lib.MODES[mode], dt,
                                   _stream(x.device))
    _lib.check(rc, "rcx_recconv2d_fwd")
    return y


def dwconv2d(x, w_kkc, bias=None, k=5, stride=1, out_dtype=None):
    """Depthwise conv (pad k//2, stride 1|2) with packed float32 (k,k,C) weights."""
    x = _nhwc(x)
    n, c, h, w = x.shape
    out_dtype = out_dtype or x.dtype
    p = k // 2
    ho, wo = (h + 2 * p - k) // stride + 1, (w + 2 * p - k) // stride + 1
    y = _empty_nhwc(n, c, ho, wo, out_dtype, x.device)
    with _on(x.device):
        rc = _lib.load().rcx_dwconv2d_fwd(x.data_ptr(), y.data_ptr(), w_kkc.data_ptr(),
                                          bias.data_ptr() if bias is not None else None,
                                          n, c, h, w, k, stride, _dt(x), _DT[out_dtype], _stream(x.device))
    _lib.check(rc, "rcx_dwconv2d_fwd")
    return y


def upadd_dwconv_plan(n, c, h, w, hc, wc, k, mode, x_dtype, coarse_dtype=None, out_dtype=None):
    """Which kernel upadd_dwconv would run for these extents and types (rcx_upadd_dwconv_fwd_plan); coarse_dtype None = no coarse plane."""
    return _lib.load().rcx_upadd_dwconv_fwd_plan(n, c, h, w, hc, wc, k, _lib.MODES[mode], _DT[x_dtype],
                                                 _DT[coarse_dtype] if coarse_dtype is not None else _lib.DTYPE_F32,
                                                 _DT[out_dtype or x_dtype], 1 if coarse_dtype is not None else 0).decode()


def upadd_dwconv(x, coarse, w_kkc, bias=None, k=5, mode="nearest", out_dtype=None):
    """dwconv_k(x + resize(coarse -> size(x), mode)); coarse may be None."""
    x = _nhwc(x)
    n, c, h, w = x.shape
    out_dtype = out_dtype or x.dtype
    hc = wc = 0
    cdt = _lib.DTYPE_F32
    if coarse is not None:
        coarse = _nhwc(coarse, "coarse")
        if coarse.shape[0] != n or coarse.shape[1] != c:
            raise ValueError("coarse must share N and C with x")
        hc, wc = coarse.shape[2:]
        cdt = _dt(coarse)
    y = _empty_nhwc(n, c, h, w, out_dtype, x.device)
    with _on(x.device):
        rc = _lib.load().rcx_upadd_dwconv_fwd(x.data_ptr(), coarse.data_ptr() if coarse is not None else None, y.data_ptr(),
                                              w_kkc.data_ptr(), bias.data_ptr() if bias is not None else None,
                                              n, c, h, w, hc, wc, k, _lib.MODES[mode], _dt(x), cdt, _DT[out_dtype],
                                              _stream(x.device))
    _lib.check(rc, "rcx_upadd_dwconv_fwd")
    return y


def dwconv2d_mult2(x, w_kkc, bias=None, k=7, stride=2):
    """nn.Conv2d(C, 2C, k, stride, padding=k//2, groups=C) with packed float32 (k,k,2C) weights."""
    x = _nhwc(x)
    n, c, h, w = x.shape
    p = k // 2
    ho, wo = (h + 2 * p - k) // stride + 1, (w + 2 * p - k) // stride + 1
    y = _empty_nhwc(n, 2 * c, ho, wo, x.dtype, x.device)
    with _on(x.device):
        rc = _lib.load().rcx_dwconv2d_mult2_fwd(x.data_ptr(), y.data_ptr(), w_kkc.data_ptr(),
                                                bias.data_ptr() if bias is not None else None,
                                                n, c, h, w, k, stride, _dt(x), _stream(x.device))
    _lib.check(rc, "rcx_dwconv2d_mult2_fwd")
    return y


def dwconv2d_backward(x, gy, w_kkc, k, stride, need_input_grad=True, need_bias=False):
    """Backward of dwconv2d: -> (gx like x | None, gw (k,k,C) float32, gb (C) float32 | None). Deterministic."""
    x = _nhwc(x)
    n, c, h, w = x.shape
    gy = _nhwc(gy.to(torch.float32), "grad_output")
    lib = _lib.load()
    wflip = w_kkc.view(k, k, c).flip(0, 1).contiguous()
    gx = _empty_nhwc(n, c, h, w, x.dtype, x.device) if need_input_grad else None
    gw = torch.empty(k * k * c, dtype=torch.float32, device=x.device)
    gb = torch.empty(c, dtype=torch.float32, device=x.device) if need_bias else None
    nbytes = lib.rcx_dwconv2d_bwd_workspace_bytes(c, k)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=x.device)
    with _on(x.device):
        rc = lib.rcx_dwconv2d_bwd(x.data_ptr(), gy.data_ptr(), w_kkc.data_ptr(), wflip.data_ptr(),
                                  gx.data_ptr() if gx is not None else None, gw.data_ptr(), gb.data_ptr() if gb is not None else None,
                                  ws.data_ptr(), nbytes, n, c, h, w, k, stride, _dt(x), _stream(x.device))
    _lib.check(rc, "rcx_dwconv2d_bwd")
    return gx, gw, gb


def upadd_dwconv_backward(x, coarse, gy, w_kkc, k=5, mode="nearest", need_input_grad=True, need_coarse_grad=True, need_bias=False):
    """Backward of upadd_dwconv with a coarse plane (rcx_upadd_dwconv_bwd; model/recattn.py:67): -> (gx like x | None, gcoarse float32 like coarse | None,
    gw (k,k,C) float32, gb (C) float32 | None).  coarse must be float32; gy float32 or (where the library takes it) x's own 16-bit dtype.  Deterministic."""
    x = _nhwc(x)
    coarse = _nhwc(coarse, "coarse")
    if coarse.dtype != torch.float32:
        raise TypeError("upadd_dwconv_backward takes the coarse plane in float32")
    n, c, h, w = x.shape
    hc, wc = coarse.shape[2:]
    lib = _lib.load()
    want = lib.rcx_upadd_dwconv_bwd_gy_dtype(n, c, h, w, hc, wc, k, _dt(x))
    gy = _nhwc(gy if (gy.dtype == torch.float32 or (_DT.get(gy.dtype) == want and gy.dtype == x.dtype)) else gy.to(torch.float32), "grad_output")
    wflip = w_kkc.view(k, k, c).flip(0, 1).contiguous()
    gx = _empty_nhwc(n, c, h, w, x.dtype, x.device) if need_input_grad else None
    gc = _empty_nhwc(n, c, hc, wc, torch.float32, x.device) if need_coarse_grad else None
    gw = torch.empty(k * k * c, dtype=torch.float32, device=x.device)
    gb = torch.empty(c, dtype=torch.float32, device=x.device) if need_bias else None
    nbytes = lib.rcx_upadd_dwconv_bwd_workspace_bytes(n, c, h, w, hc, wc, k)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=x.device)
    with _on(x.device):
        rc = lib.rcx_upadd_dwconv_bwd(x.data_ptr(), coarse.data_ptr(), gy.data_ptr(), _DT[gy.dtype], w_kkc.data_ptr(), wflip.data_ptr(),
                                      gx.data_ptr() if gx is not None else None, gc.data_ptr() if gc is not None else None, gw.data_ptr(),
                                      gb.data_ptr() if gb is not None else None, ws.data_ptr(), nbytes, n, c, h, w, hc, wc, k, _lib.MODES[mode], _dt(x),
                                      _stream(x.device))
    _lib.check(rc, "rcx_upadd_dwconv_bwd")
    return gx, gc, gw, gb


def dwconv2d_mult2_backward(x, gy, w_kkc, k, need_input_grad=True, need_bias=False):
    """Backward of dwconv2d_mult2 (stride 2): -> (gx like x | None, gw (k,k,2C) float32, gb (2C) float32 | None)."""
    x = _nhwc(x)
    n, c, h, w = x.shape
    gy = _nhwc(gy.to(torch.float32), "grad_output")
    lib = _lib.load()
    gx = _empty_nhwc(n, c, h, w, x.dtype, x.device) if need_input_grad else None
    gw = torch.empty(k * k * 2 * c, dtype=torch.float32, device=x.device)
    gb = torch.empty(2 * c, dtype=torch.float32, device=x.device) if need_bias else None
    nbytes = lib.rcx_dwconv2d_bwd_workspace_bytes(2 * c, k)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=x.device)
    with _on(x.device):
        rc = lib.rcx_dwconv2d_mult2_bwd(x.data_ptr(), gy.data_ptr(), w_kkc.data_ptr(), gx.data_ptr() if gx is not None else None,
                                        gw.data_ptr(), gb.data_ptr() if gb is not None else None, ws.data_ptr(), nbytes,
                                        n, c, h, w, k, _dt(x), _stream(x.device))
    _lib.check(rc, "rcx_dwconv2d_mult2_bwd")
    return gx, gw, gb


def linear_attention_core(qpre, kpre, v, pe, heads):
    """Everything after the qk projection of LinearAttention1/2 (model/recattn.py:21-28, :44-51).

    qpre, kpre: (B, n, C) pre-activations; v, pe: N x C x h x w channels_last (viewed as (B, n, C)); returns channels_last like v.
    """
    v = _nhwc(v, "v")
    pe = _nhwc(pe, "pe")
    b, c, h, w = v.shape
    n = h * w
    for t, name in ((qpre, "qpre"), (kpre, "kpre")):
        _require_gpu(t, name)
        if tuple(t.shape) != (b, n, c) or not t.is_contiguous() or t.dtype != v.dtype:
            raise ValueError(f"{name} must be a contiguous ({b}, {n}, {c}) tensor of {v.dtype}, got {tuple(t.shape)} {t.dtype}")
    if pe.shape != v.shape or pe.dtype != v.dtype:
        raise ValueError("pe must match v")
    out = _empty_nhwc(b, c, h, w, v.dtype, v.device)
    with _on(v.device):
        rc = _lib.load().rcx_linear_attention_fwd(qpre.data_ptr(), kpre.data_ptr(), v.data_ptr(), pe.data_ptr(), out.data_ptr(),
                                                  b, n, c, heads, _dt(v), _stream(v.device))
    _lib.check(rc, "rcx_linear_attention_fwd")
    return out


FUSE_PE = __import__("os").environ.get("RCX_ATTN_FUSE_PE", "1") != "0"      # A/B switch, read at import


def linear_attention_core_fuses_pe(c, heads):
    """Whether linear_attention_core_pe has a kernel for this head size (a multiple of four, at most 64)."""
    d = c // heads
    return c % heads == 0 and d % 4 == 0 and d <= 64 and FUSE_PE


def linear_attention_core_pe(qpre, kpre, v, w_pe_kkc, b_pe, heads):
    """linear_attention_core with pe = dwconv3x3(v) + bias computed inside the kernel (rcx_linear_attention_pe_fwd): w_pe_kkc / b_pe are
    the float32 packs of pack_dw_weight / pack_bias of LinearAttention.pe (BatchNorm folded).  None if the library has no such kernel for
    this configuration (the long sequences that run on the matrix cores: measured slower there)."""
    v = _nhwc(v, "v")
    b, c, h, w = v.shape
    n = h * w
    for t, name in ((qpre, "qpre"), (kpre, "kpre")):
        _require_gpu(t, name)
        if tuple(t.shape) != (b, n, c) or not t.is_contiguous() or t.dtype != v.dtype:
            raise ValueError(f"{name} must be a contiguous ({b}, {n}, {c}) tensor of {v.dtype}, got {tuple(t.shape)} {t.dtype}")
    if w_pe_kkc.dtype != torch.float32 or w_pe_kkc.numel() != 9 * c:
        raise ValueError("w_pe_kkc must be the float32 (3, 3, C) pack of pack_dw_weight")
    out = _empty_nhwc(b, c, h, w, v.dtype, v.device)
    with _on(v.device):
        rc = _lib.load().rcx_linear_attention_pe_fwd(qpre.data_ptr(), kpre.data_ptr(), v.data_ptr(), w_pe_kkc.data_ptr(),
                                                     b_pe.data_ptr() if b_pe is not None else None, out.data_ptr(),
                                                     b, h, w, c, heads, _dt(v), _stream(v.device))
    if rc == _lib.ERR_UNSUPPORTED:       # (RCX_ATTN_SCALAR=1 pins the kernel without this form): the caller runs the two steps
        return None
    _lib.check(rc, "rcx_linear_attention_pe_fwd")
    return out


def recattn_qkcore_supported(c, heads, h, w):
    """Whether recattn_qkcore has a kernel for this coarse plane (rcx_recattn_qkcore_launches: 1/2/4/8/16 heads of 32 channels, or of 4 .. 28 in fours when heads is even; one launch for
    planes of at most 64 tokens that fit the LDS, two launches otherwise)."""
    return _lib.load().rcx_recattn_qkcore_launches(1, h, w, c, heads) > 0


def recattn_qkcore(d, wqk_bf16, bqk, w_pe_kkc, b_pe, heads):
    """RecAttn2d's coarse level on the matrix cores (rcx_recattn_qkcore_fwd; one launch up to 64 tokens, two above): d float32 N x C x h x w channels_last; wqk_bf16 (2C, C/2) bfloat16, bqk (2C)
    float32, the pe packs as linear_attention_core_pe.  Returns the attention output + pe, float32 channels_last like d."""
    d = _nhwc(d, "d")
    b, c, h, w = d.shape
    if d.dtype != torch.float32:
        raise ValueError("d must be float32 (the coarse chain of RecAttn2d)")
    if wqk_bf16.dtype != torch.bfloat16 or tuple(wqk_bf16.shape) != (2 * c, c // 2) or not wqk_bf16.is_contiguous():
        raise ValueError(f"wqk_bf16 must be a contiguous ({2 * c}, {c // 2}) bfloat16 tensor")
    if bqk.dtype != torch.float32 or bqk.numel() != 2 * c:
        raise ValueError("bqk must be float32 of 2C elements")
    out = _empty_nhwc(b, c, h, w, torch.float32, d.device)
    lib = _lib.load()
    need = lib.rcx_recattn_qkcore_workspace_bytes(b, h, w, c, heads)               # the k^T v partial sums of the two-launch form (> 64 tokens)
    ws = torch.empty(need, dtype=torch.uint8, device=d.device) if need else None
    with _on(d.device):
        rc = lib.rcx_recattn_qkcore_fwd(d.data_ptr(), wqk_bf16.data_ptr(), bqk.data_ptr(), w_pe_kkc.data_ptr(),
                                        b_pe.data_ptr() if b_pe is not None else None, out.data_ptr(),
                                        ws.data_ptr() if ws is not None else None, need, b, h, w, c, heads, _stream(d.device))
    _lib.check(rc, "rcx_recattn_qkcore_fwd")
    return out


def recattn_down_qkcore_supported(c, heads, h, w, x_dtype):
    """Whether RecAttn2d's stride-2 conv + coarse level run as ONE launch from x (rcx_recattn_down_qkcore_fwd: the 14 x 14 and 7 x 7 planes of 16-bit
    activations, 32-wide heads)."""
    return x_dtype in _DT and _lib.load().rcx_recattn_down_qkcore_supported(1, h, w, c, heads, _DT[x_dtype]) > 0


def recattn_down_qkcore(x, w_down_kkc, b_down, wqk_bf16, bqk, w_pe_kkc, b_pe, heads):
    """a = LinearAttention(ConvNorm_k5s2(x)) of RecAttn2d.forward (model/recattn.py:61-66) in one launch: x N x C x H x W channels_last, 16-bit; the conv's
    packs as dwconv2d takes them, the rest as recattn_qkcore.  Returns float32 N x C x ceil(H/2) x ceil(W/2), channels_last."""
    x = _nhwc(x, "x")
    b, c, h, w = x.shape
    if wqk_bf16.dtype != torch.bfloat16 or tuple(wqk_bf16.shape) != (2 * c, c // 2) or not wqk_bf16.is_contiguous():
        raise ValueError(f"wqk_bf16 must be a contiguous ({2 * c}, {c // 2}) bfloat16 tensor")
    if bqk.dtype != torch.float32 or bqk.numel() != 2 * c:
        raise ValueError("bqk must be float32 of 2C elements")
    out = _empty_nhwc(b, c, (h + 1) // 2, (w + 1) // 2, torch.float32, x.device)
    with _on(x.device):
        rc = _lib.load().rcx_recattn_down_qkcore_fwd(x.data_ptr(), w_down_kkc.data_ptr(), b_down.data_ptr() if b_down is not None else None,
                                                     wqk_bf16.data_ptr(), bqk.data_ptr(), w_pe_kkc.data_ptr(),
                                                     b_pe.data_ptr() if b_pe is not None else None, out.data_ptr(), b, h, w, c, heads, _dt(x), _stream(x.device))
    _lib.check(rc, "rcx_recattn_down_qkcore_fwd")
    return out


def recattn2d_supported(c, heads, h, w, mode, dtype):
    """Whether RecAttn2d.forward is ONE launch for this plane (rcx_recattn2d_fwd: 14 x 14 / 7 x 7, 1 .. 8 heads of 32, nearest, 16-bit)."""
    return dtype in _DT and mode in _lib.MODES and _lib.load().rcx_recattn2d_fwd_supported(1, h, w, c, heads, _lib.MODES[mode], _DT[dtype]) > 0


def recattn2d(x, w_down_kkc, b_down, wqk_bf16, bqk, w_pe_kkc, b_pe, w_conv_kkc, b_conv, heads, mode="nearest"):
    """RecAttn2d.forward (eval, BatchNorms folded; model/recattn.py:54-67) in one launch: x N x C x H x W channels_last, 16-bit -> y like x."""
    x = _nhwc(x, "x")
    b, c, h, w = x.shape
    if wqk_bf16.dtype != torch.bfloat16 or tuple(wqk_bf16.shape) != (2 * c, c // 2) or not wqk_bf16.is_contiguous():
        raise ValueError(f"wqk_bf16 must be a contiguous ({2 * c}, {c // 2}) bfloat16 tensor")
    if bqk.dtype != torch.float32 or bqk.numel() != 2 * c:
        raise ValueError("bqk must be float32 of 2C elements")
    y = torch.empty_like(x, memory_format=torch.channels_last)
    p = lambda t: t.data_ptr() if t is not None else None
    with _on(x.device):
        rc = _lib.load().rcx_recattn2d_fwd(x.data_ptr(), y.data_ptr(), w_down_kkc.data_ptr(), p(b_down), wqk_bf16.data_ptr(), bqk.data_ptr(),
                                           w_pe_kkc.data_ptr(), p(b_pe), w_conv_kkc.data_ptr(), p(b_conv), b, h, w, c, heads, _lib.MODES[mode], _dt(x),
                                           _stream(x.device))
    _lib.check(rc, "rcx_recattn2d_fwd")
    return y


def _mlp_acc_unit(i, h):
    """Hidden unit (within a 32-unit tile) that accumulator register i of lane half h holds after the first product (rcx_mlp.hip acc_row)."""
    return (i & 3) + 8 * (i >> 2) + 4 * h


def channel_mlp_hidden(m, c, hidden, dtype):
    """The hidden width rcx_channel_mlp_fwd would run M tokens of C channels and `hidden` units at (the next multiple of 32, 64 or 128 it has a kernel
    for: zero units change nothing), or 0 when it has none (bf16 only)."""
    if dtype != torch.bfloat16:
        return 0
    lib = _lib.load()
    for mult in (32, 64, 128):
        hp = -(-int(hidden) // mult) * mult
        if hp <= 1.125 * hidden + 31 and lib.rcx_channel_mlp_supported(int(m), int(c), hp, _DT[dtype]) > 0:
            return hp
    return 0


def channel_mlp_supported(m, c, hidden, dtype):
    return channel_mlp_hidden(m, c, hidden, dtype) > 0


def pack_channel_mlp(w1, b1, w2, b2, hidden_to=None):
    """The two BN-folded 1x1 convs of a channel mixer -> (wfrag, bias, H) for channel_mlp: w1 (H0, C[, 1, 1]), b1 (H0) | None, w2 (C, H0[, 1, 1]), b2 (C) | None;
    hidden_to: the padded hidden width H (channel_mlp_hidden; default the next multiple of 32).

    The hidden layer is padded with zero units to H = a multiple of 32, C to whole k-steps / output tiles with zero columns / rows; the weights are rounded to
    bf16 (they already are bf16 in a bf16 model) and laid out fragment by fragment in the order the kernel's lanes read them (rcx_mlp.hip):
      W1 fragment (ht, ks), lane (h, m), element j = W1[32 ht + m][16 ks + 8 h + j]
      W2 fragment (ht, ct, q), lane (h, m), element j = 0.5 W2[32 ct + m][32 ht + unit(8 q + j, h)]      unit = _mlp_acc_unit: the order the first product leaves in the registers
    stored hidden tile by hidden tile: [W1 (ht, 0 .. KS1-1)] [W2 (ht, ct, q) for ct, q] for ht = 0 .. H/32 - 1, 1 KB (64 lanes x 8 bf16) per fragment.
    """
    w1 = w1.detach().reshape(w1.shape[0], -1)
    w2 = w2.detach().reshape(w2.shape[0], -1)
    h0, c = w1.shape
    if tuple(w2.shape) != (c, h0):
        raise ValueError(f"w2 must be ({c}, {h0}), got {tuple(w2.shape)}")
    dev = w1.device
    hp = -(-h0 // 32) * 32 if hidden_to is None else int(hidden_to)
    if hp % 32 or hp < h0:
        raise ValueError(f"hidden_to={hidden_to} must be a multiple of 32 and at least {h0}")
    ks1, ht, ct = -(-c // 16), hp // 32, -(-c // 32)
    if c > 128 and ct % 2:                  # the streamed kernels write y in halves of two output tiles: one more tile of zero rows (rcx_mlp.hip mlp_shape)
        ct += 1
    w1p = torch.zeros(32 * ht, 16 * ks1, dtype=torch.bfloat16, device=dev)
    w1p[:h0, :c] = w1.to(torch.bfloat16)
    f1 = w1p.view(ht, 32, ks1, 2, 8).permute(0, 2, 3, 1, 4)                    # [ht, ks, h, m, j]
    w2p = torch.zeros(32 * ct, 32 * ht, dtype=torch.bfloat16, device=dev)
    w2p[:c, :h0] = (0.5 * w2.float()).to(torch.bfloat16)           # the kernel's hidden activations are 2 gelu(.) (rcx_gelu.h gelu2x_batch); halving a bf16 is exact
    unit = torch.tensor([[[_mlp_acc_unit(8 * q + j, h) for j in range(8)] for h in range(2)] for q in range(2)], device=dev)      # [q, h, j]
    f2 = w2p.view(ct, 32, ht, 32)[:, :, :, unit]                               # [ct, m, ht, q, h, j]
    f2 = f2.permute(0, 2, 3, 4, 1, 5)                                          # [ct, ht, q, h, m, j]
    # hidden-tile-major: chunk ht = its KS1 W1 fragments, then its 2 CT W2 fragments (ct, q) -- what one step of the kernels' hidden loop reads (and what the
    # large shapes stream through LDS one chunk at a time)
    f1 = f1.reshape(ht, ks1 * 512)
    f2 = f2.reshape(ct, ht, 2, 512).permute(1, 0, 2, 3).reshape(ht, ct * 2 * 512)
    wfrag = torch.cat([f1, f2], dim=1).reshape(-1).contiguous()
    bias = torch.zeros(32 * (ht + ct), dtype=torch.float32, device=dev)
    if b1 is not None:
        bias[:h0] = b1.detach().float()
    if b2 is not None:
        bias[32 * ht:32 * ht + c] = b2.detach().float()
    return wfrag, bias, 32 * ht


def _check_pack(t, dtype, numel, device, name):
    """A derived pack handed to a raw launch: the kernels read its whole extent, so a wrong size / dtype / device is refused here, not found by the GPU."""
    if not torch.is_tensor(t) or t.dtype != dtype or t.numel() != numel or not t.is_contiguous() or t.device != device:
        got = f"{tuple(t.shape)} {t.dtype} on {t.device}" if torch.is_tensor(t) else type(t).__name__
        raise ValueError(f"{name} must be a contiguous {dtype} tensor of {numel} elements on {device}, got {got}")


def channel_mlp(z, x, wfrag, bias, hidden):
    """y = x + W2 gelu(W1 z + b1) + b2 in one launch (rcx_channel_mlp_fwd; model/recnext.py:157-158): z, x N x C x H x W channels_last bf16 -> y like x."""
    z = _nhwc(z, "z")
    x = _nhwc(x, "x")
    if z.shape != x.shape or z.dtype != x.dtype:
        raise ValueError("z and x must have the same shape and dtype")
    n, c, h, w = x.shape
    lib = _lib.load()
    if wfrag.dtype != torch.bfloat16 or wfrag.numel() * 2 != lib.rcx_channel_mlp_pack_bytes(c, hidden) or not wfrag.is_contiguous():
        raise ValueError("wfrag is not the pack of pack_channel_mlp for this (C, H)")
    ct = -(-c // 32)
    ct += 1 if c > 128 and ct % 2 else 0
    _check_pack(bias, torch.float32, hidden + 32 * ct, x.device, "bias")     # the kernel copies 32 (H/32 + CT) floats into LDS unconditionally
    _check_pack(wfrag, torch.bfloat16, wfrag.numel(), x.device, "wfrag")
    if z.device != x.device:
        raise ValueError("z and x must be on the same device")
    y = _empty_nhwc(n, c, h, w, x.dtype, x.device)
    with _on(x.device):
        rc = lib.rcx_channel_mlp_fwd(z.data_ptr(), x.data_ptr(), y.data_ptr(), wfrag.data_ptr(), bias.data_ptr(), n * h * w, c, hidden, _dt(x), _stream(x.device))
    _lib.check(rc, "rcx_channel_mlp_fwd")
    return y


def stem_supported(n, h, w, cm, co, dtype):
    """Whether rcx_stem_fwd has a kernel for a stem with CM intermediate and CO output channels on an N x 3 x H x W input (bf16 only)."""
    return dtype == torch.bfloat16 and _lib.load().rcx_stem_supported(int(n), int(h), int(w), int(cm), int(co), _DT[dtype]) > 0


def pack_stem(w1, b1, w2, b2):
    """The stem's two BN-folded 3x3 stride-2 convs -> (w1p, b1p, w2frag, b2p) for stem(): w1 (CM, 3, 3, 3), b1 (CM), w2 (CO, CM, 3, 3), b2 (CO).

    w1p: bf16 matrix-core fragments of the first conv (K = (dy, dx, c) padded to 32), b1p padded to a multiple of 32.  w2frag: bf16 matrix-core fragments, fragment (mt, ks) with ks = tap (KC / 16) + cg (KC = CM rounded up to 16):
    lane (h, m), element j = w2[32 mt + m][16 cg + 8 h + j][tap] (zeros past CO / CM).  b2p: zero padded to a multiple of 32."""
    cm, co = w1.shape[0], w2.shape[0]
    if tuple(w1.shape) != (cm, 3, 3, 3) or tuple(w2.shape) != (co, cm, 3, 3):
        raise ValueError(f"stem weights must be (CM, 3, 3, 3) and (CO, CM, 3, 3), got {tuple(w1.shape)} and {tuple(w2.shape)}")
    dev = w1.device
    kc, mt = -(-cm // 16) * 16, -(-co // 32)
    m1 = -(-cm // 32)
    # first conv: K = (dy, dx, c) = 27 padded to 32; fragment (m1, ks), lane (h, m), element j = w1[32 m1 + m][k = 16 ks + 8 h + j]
    w1k = torch.zeros(32 * m1, 32, dtype=torch.bfloat16, device=dev)
    w1k[:cm, :27] = w1.detach().permute(0, 2, 3, 1).reshape(cm, 27).to(torch.bfloat16)                # (CM, dy, dx, c)
    w1p = w1k.view(m1, 32, 2, 2, 8).permute(0, 2, 3, 1, 4).reshape(-1).contiguous()                   # [m1, ks, h, m, j]
    b1p = torch.zeros(32 * m1, dtype=torch.float32, device=dev)
    if b1 is not None:
        b1p[:cm] = b1.detach().float()
    wp = torch.zeros(32 * mt, kc, 9, dtype=torch.bfloat16, device=dev)
    wp[:co, :cm] = w2.detach().reshape(co, cm, 9).to(torch.bfloat16)
    f = wp.view(mt, 32, kc // 16, 2, 8, 9).permute(0, 5, 2, 3, 1, 4)                         # [mt, tap, cg, h, m, j]
    w2frag = f.reshape(-1).contiguous()
    b2p = torch.zeros(32 * mt, dtype=torch.float32, device=dev)
    if b2 is not None:
        b2p[:co] = b2.detach().float()
    return w1p, b1p, w2frag, b2p


def stem(x, w1p, b1p, w2frag, b2p, cm, co):
    """RecNextStem.forward in one launch (rcx_stem_fwd; model/recnext.py:134-146): x N x 3 x H x W channels_last bf16 -> N x CO x ceil(H/4) x ceil(W/4)."""
    x = _nhwc(x, "x")
    n, c, h, w = x.shape
    if c != 3:
        raise ValueError(f"the stem takes 3 input channels, got {c}")
    h2, w2_ = -(-(-(-h // 2)) // 2), -(-(-(-w // 2)) // 2)
    m1, mt = -(-cm // 32), -(-co // 32)
    _check_pack(w1p, torch.bfloat16, m1 * 2 * 512, x.device, "w1p")           # 2 KB per 32 intermediate channels (K = 27 padded to 32)
    _check_pack(b1p, torch.float32, 32 * m1, x.device, "b1p")
    _check_pack(w2frag, torch.bfloat16, _lib.load().rcx_stem_pack_bytes(cm, co) // 2, x.device, "w2frag")
    _check_pack(b2p, torch.float32, 32 * mt, x.device, "b2p")
    y = _empty_nhwc(n, co, h2, w2_, x.dtype, x.device)
    with _on(x.device):
        rc = _lib.load().rcx_stem_fwd(x.data_ptr(), y.data_ptr(), w1p.data_ptr(), b1p.data_ptr(), w2frag.data_ptr(), b2p.data_ptr(), n, h, w, cm, co, _dt(x), _stream(x.device))
    _lib.check(rc, "rcx_stem_fwd")
    return y


def linear_attention_core_backward(qpre, kpre, v, gout, heads):
    """Gradients of linear_attention_core with respect to qpre, kpre (B, n, C) and v (N x C x h x w); dL/dpe = gout."""
    v = _nhwc(v, "v")
    gout = _nhwc(gout, "grad_output")
    b, c, h, w = v.shape
    n = h * w
    if gout.dtype != v.dtype:
        gout = gout.to(v.dtype)
    gq = torch.empty_like(qpre)
    gk = torch.empty_like(kpre)
    gv = _empty_nhwc(b, c, h, w, v.dtype, v.device)
    with _on(v.device):
        rc = _lib.load().rcx_linear_attention_bwd(qpre.data_ptr(), kpre.data_ptr(), v.data_ptr(), gout.data_ptr(),
                                                  gq.data_ptr(), gk.data_ptr(), gv.data_ptr(), b, n, c, heads, _dt(v), _stream(v.device))
    _lib.check(rc, "rcx_linear_attention_bwd")
    return gq, gk, gv


class LinearAttentionCoreFn(torch.autograd.Function):
    """linear_attention_core with its HIP backward (rcx_linear_attention_bwd): forward and backward both through the C ABI."""

    @staticmethod
    def forward(ctx, qpre, kpre, v, pe, heads):
        out = linear_attention_core(qpre, kpre, v, pe, heads)
        ctx.save_for_backward(qpre, kpre, v)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, gout):
        qpre, kpre, v = ctx.saved_tensors
        gq, gk, gv = linear_attention_core_backward(qpre, kpre, v, gout, ctx.heads)
        return gq, gk, gv, gout, None


def recconv2d_forward_train(x, wpack, bpack, level, k, mode="bilinear"):
    """Training forward: same result as recconv2d_forward, plus the saved fp32 pyramid the backward needs."""
    x = _nhwc(x)
    n, c, h, w = x.shape
    lib = _lib.load()
    dt = _dt(x)
    y = _empty_nhwc(n, c, h, w, x.dtype, x.device)
    nbytes = lib.rcx_recconv2d_train_saved_bytes(n, c, h, w, level, k)
    saved = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=x.device)
    with _on(x.device):
        rc = lib.rcx_recconv2d_fwd_train(x.data_ptr(), y.data_ptr(), wpack.data_ptr(),
                                         bpack.data_ptr() if bpack is not None else None, saved.data_ptr(), nbytes,
                                         n, c, h, w, level, k, _lib.MODES[mode], dt, _stream(x.device))
    _lib.check(rc, "rcx_recconv2d_fwd_train")
    return y, saved


def recconv2d_backward(x, gy, wpack, saved, level, k, mode="bilinear", need_bias=False, wflip=None, param_grads=None):
    """-> (gx like x, gwpack (level+2, k*k*C) f32, gbpack (level+2, C) f32 | None). Deterministic.
    wflip: the flipped pack from pack_recconv_params(with_flipped=True), if the caller has it (else it is made here).
    param_grads=(gws, gbs | None): level+2 preallocated contiguous (C,1,k,k) / (C) tensors of ONE dtype; the final reduction writes the parameters'
    gradients straight into them (no packed gradient, no unpack launch, no dtype copy) and (gx, None, None) is returned.
    gy may be float32 or, where the library reads it as it is (rcx_recconv2d_bwd_gy_dtype: the 56x56 / level 4 and 28x28 / level 3 blocks), x's own
    16-bit dtype; anything else is converted to float32 here."""
    x = _nhwc(x)
    n, c, h, w = x.shape
    lib = _lib.load()
    want = lib.rcx_recconv2d_bwd_gy_dtype(n, c, h, w, level, k, _dt(x))
    gy = _nhwc(gy if (gy.dtype == torch.float32 or (_DT.get(gy.dtype) == want and gy.dtype == x.dtype)) else gy.to(torch.float32), "grad_output")
    if wflip is None:
        wflip = wpack.view(level + 2, k, k, c).flip(1, 2).contiguous()
    gx = _empty_nhwc(n, c, h, w, x.dtype, x.device)
    gw = gb = None
    gw_ptrs = gb_ptrs = None
    grad_dt = 0
    if param_grads is not None:
        gws, gbs = param_grads
        ts = list(gws) + (list(gbs) if gbs is not None else [])
        if len(gws) != level + 2 or (gbs is not None and len(gbs) != level + 2) or any(t.dtype != ts[0].dtype or not t.is_contiguous() or t.device != x.device for t in ts) \
                or any(t.numel() != c * k * k for t in gws) or (gbs is not None and any(t.numel() != c for t in gbs)):
            raise ValueError("param_grads: level+2 contiguous (C,1,k,k) [and (C)] tensors of one dtype on x's device")
        grad_dt = _DT[ts[0].dtype]
        gw_ptrs = _ptr_array(list(gws))
        gb_ptrs = _ptr_array(list(gbs)) if gbs is not None else None
    else:
        gw = torch.empty_like(wpack)
        gb = torch.empty((level + 2, c), dtype=torch.float32, device=x.device) if need_bias else None
    nbytes = lib.rcx_recconv2d_bwd_workspace_bytes(n, c, h, w, level, k)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=x.device)
    with _on(x.device):
        rc = lib.rcx_recconv2d_bwd(x.data_ptr(), gy.data_ptr(), _DT[gy.dtype], wpack.data_ptr(), wflip.data_ptr(), saved.data_ptr(),
                                   gx.data_ptr(), gw.data_ptr() if gw is not None else None, gb.data_ptr() if gb is not None else None,
                                   gw_ptrs, gb_ptrs, grad_dt,
                                   ws.data_ptr(), nbytes, n, c, h, w, level, k, _lib.MODES[mode], _dt(x), _stream(x.device))
    _lib.check(rc, "rcx_recconv2d_bwd")
    return gx, gw, gb
