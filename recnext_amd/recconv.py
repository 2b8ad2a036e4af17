"""Drop-in ``RecConv2d`` -- same constructor, parameter names and forward contract as the reference
block (model/recnext.py:8-34), body executed by the gfx950 HIP kernels behind the C ABI.

    RecConv2d(in_channels, kernel_size=5, bias=False, level=2, mode='bilinear')
    state_dict keys: down.weight, convs.{0..level}.weight  (+ .bias when bias=True), each (C,1,k,k) / (C,)

``convs[0]`` pairs with the coarsest level and ``convs[level]`` is the full-resolution conv, exactly as
``zip(self.convs, reversed(features))`` does at model/recnext.py:32-34.  ``level=0`` is legal.

Forward and backward accept float32, bfloat16 or float16 CUDA tensors (logical N x C x H x W; channels_last
storage is consumed zero-copy, anything else is converted once) and returns a channels_last tensor
of the same shape and dtype.  All arithmetic is float32 inside the kernels.  There is no CPU path.
"""

import torch
import torch.nn as nn

from . import ops


class RecConv2d(nn.Module):
    def __init__(self, in_channels, kernel_size=5, bias=False, level=2, mode="bilinear"):
        super().__init__()
        if kernel_size % 2 != 1:
            raise ValueError("RecConv2d kernel_size must be odd")
        if mode not in ("bilinear", "nearest"):
            raise ValueError("RecConv2d mode must be 'bilinear' or 'nearest'")
        self.level = level
        self.mode = mode
        self.kernel_size = kernel_size
        self.in_channels = in_channels
        # nn.Conv2d objects are kept purely as parameter containers: identical names, shapes and
        # default initialisation (kaiming-uniform) to the reference, so its checkpoints load as-is.
        kwargs = dict(in_channels=in_channels, out_channels=in_channels, groups=in_channels,
                      kernel_size=kernel_size, padding=kernel_size // 2, bias=bias)
        self.down = nn.Conv2d(stride=2, **kwargs)
        self.convs = nn.ModuleList([nn.Conv2d(**kwargs) for _ in range(level + 1)])
        self._pack_key = None
        self._pack = None
        self._wflip = None                                  # the pack with every k x k flipped (the backward's taps)
        # Optional per-channel affine applied to the block's OUTPUT (y*scale + shift), folded into
        # convs[level] when the packs are built: used to absorb the eval-mode BatchNorm that follows the
        # token mixer in MetaNeXtBlock (model/recnext.py:153,158) -- SURVEY.md section 8f row 2.
        # kept as plain float32 tensors (not buffers): .to(bfloat16) must not round them, and they follow
        # the parameters' device when the packs are built
        self.fold_scale = None
        self.fold_shift = None

    @torch.no_grad()
    def fold_output_affine(self, scale, shift):
        """Absorb y -> y*scale + shift (per channel) into the final conv; composes with an earlier fold."""
        scale = scale.detach().float().clone()
        shift = shift.detach().float().clone()
        if self.fold_scale is not None:
            shift = shift + scale * self.fold_shift.to(scale.device)
            scale = scale * self.fold_scale.to(scale.device)
        self.fold_scale, self.fold_shift = scale, shift
        self._pack_key = None

    def _params(self):
        # straight out of the registries: nn.Module.__getattr__ costs ~0.3 us a lookup and this runs on every forward (12 lookups); a
        # parametrised weight (torch.nn.utils.parametrize) is not in _parameters and takes the attribute path
        try:
            mods = self._modules
            convs = [mods["down"], *mods["convs"]._modules.values()]
            ws = [cv._parameters["weight"] for cv in convs]
            bs = [cv._parameters["bias"] for cv in convs]
            if any(w is None for w in ws):
                raise KeyError("weight")
        except KeyError:
            ws = [self.down.weight] + [cv.weight for cv in self.convs]
            bs = [self.down.bias] + [cv.bias for cv in self.convs]
        return ws, (bs if bs[0] is not None else None)

    def _plist(self):
        """The parameters in nn.Module.parameters() order (down.weight, [down.bias], convs.0.weight, ...) without its module-tree
        traversal -- that generator costs ~35 us a call, twice per forward, on a training step the host's launches already bound."""
        out = []
        for cv in (self.down, *self.convs):
            out.append(cv.weight)
            if cv.bias is not None:
                out.append(cv.bias)
        return out

    def packed_params(self):
        """(wpack, bpack) float32 tap-major copies, rebuilt only when a parameter changed."""
        ws, bs = self._params()
        allp = ws + (bs or [])
        key = tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in allp)
        if self.fold_scale is not None:
            key += ((self.fold_scale.data_ptr(), self.fold_scale._version, self.fold_shift._version),)
        if key != self._pack_key:
            wpack, bpack, self._wflip = ops.pack_recconv_params(ws[0], ws[1:], bs[0] if bs else None, bs[1:] if bs else None,
                                                                with_flipped=True)
            if self.fold_scale is not None:
                c, k = self.in_channels, self.kernel_size
                if bpack is None:
                    bpack = torch.zeros((self.level + 2, c), dtype=torch.float32, device=wpack.device)
                sc = self.fold_scale.to(wpack.device)
                wpack[-1].view(k * k, c).mul_(sc)                    # tap-major (k*k, C): scale each channel's taps
                bpack[-1].mul_(sc).add_(self.fold_shift.to(wpack.device))
            self._pack = (wpack, bpack)
            self._pack_key = key
        return self._pack

    def forward(self, x):
        # Schedule choice.  Autograd needs the per-level schedule (it keeps the float32 pyramid for the backward), so ANY call
        # with grad mode on and something that requires grad takes it -- including a model in eval() called outside
        # torch.no_grad(), which is legal (the block has no train/eval distinction) but several times slower than the fused
        # inference kernels: warn once so that a forgotten no_grad() does not pass for a slow kernel.
        plist = self._plist() if torch.is_grad_enabled() else ()
        if plist and (x.requires_grad or any(p.requires_grad for p in plist)):
            if self.fold_scale is not None:
                raise RuntimeError("this RecConv2d carries a folded output affine (fold_token_mixer_norms / fold_output_affine), an "
                                   "inference-only transform: wrap the call in torch.no_grad() (or torch.inference_mode())")
            if self.in_channels % 4:
                raise NotImplementedError(f"the HIP backward of RecConv2d needs a channel count that is a multiple of 4, got "
                                          f"{self.in_channels}; run inference under torch.no_grad()")
            if not self.training and not x.requires_grad:
                _warn_eval_with_grad()
            return _RecConv2dFn.apply(x, self, *plist)
        wpack, bpack = self.packed_params()
        return ops.recconv2d_forward(x, wpack, bpack, self.level, self.kernel_size, self.mode)

    def extra_repr(self):
        return (f"{self.in_channels}, kernel_size={self.kernel_size}, level={self.level}, mode={self.mode!r}, "
                f"bias={self.down.bias is not None}")


_warned = False


def _warn_eval_with_grad():
    global _warned
    if not _warned:
        _warned = True
        import warnings
        warnings.warn("recnext_amd.RecConv2d: module is in eval() but grad mode is on and its parameters require grad, so the "
                      "training schedule (saved float32 pyramid, one launch per level) runs instead of the fused inference "
                      "kernel; wrap inference in torch.no_grad()", stacklevel=3)


class _RecConv2dFn(torch.autograd.Function):
    """Autograd wrapper: forward and backward both run on the HIP kernels (rcx_recconv2d_fwd_train / rcx_recconv2d_bwd).

    The training forward uses the per-level schedule, which leaves the fp32 pyramid in a buffer the backward reads;
    parameter gradients come back in the packed (k,k,C) layout and are permuted to the reference's (C,1,k,k).
    """

    @staticmethod
    def forward(ctx, x, module, *params):
        if module.fold_scale is not None:
            raise RuntimeError("RecConv2d.fold_output_affine is an inference transform; it cannot be trained through")
        wpack, bpack = module.packed_params()
        y, saved = ops.recconv2d_forward_train(x, wpack, bpack, module.level, module.kernel_size, module.mode)
        ctx.module = module
        ctx.save_for_backward(x, wpack, saved, module._wflip)
        ctx.has_bias = bpack is not None
        ctx.param_dtypes = [p.dtype for p in params]
        ctx.param_strides = [p.stride() for p in params]
        return y

    @staticmethod
    def backward(ctx, grad_out):
        x, wpack, saved, wflip = ctx.saved_tensors
        m = ctx.module
        k, c, L = m.kernel_size, m.in_channels, m.level
        # the parameters' gradients are written by the backward's final reduction itself, in the parameters' layout and dtype (one dtype per module)
        dts = set(ctx.param_dtypes)
        if len(dts) == 1:
            dt = ctx.param_dtypes[0]
            gws = torch.empty((L + 2, c, 1, k, k), dtype=dt, device=x.device)
            gbs = torch.empty((L + 2, c), dtype=dt, device=x.device) if ctx.has_bias else None
            gx, _, _ = ops.recconv2d_backward(x, grad_out, wpack, saved, L, k, m.mode, wflip=wflip,
                                              param_grads=([gws[i] for i in range(L + 2)], [gbs[i] for i in range(L + 2)] if ctx.has_bias else None))
            gw, gb = gws, gbs
        else:                                                        # mixed parameter dtypes: packed float32 gradients, then a copy each
            gx, gw, gb = ops.recconv2d_backward(x, grad_out, wpack, saved, L, k, m.mode, need_bias=ctx.has_bias, wflip=wflip)
            gw = ops.unpack_recconv_grads(gw, L + 2, c, k)                       # (L+2, C, 1, k, k), each [i] contiguous
        grads = []
        # parameter order of nn.Module.parameters(): down.weight, [down.bias], convs.0.weight, [convs.0.bias], ...
        for i in range(L + 2):
            grads.append(gw[i])
            if ctx.has_bias:
                grads.append(gb[i])
        grads = [g if g.dtype == dt_ else g.to(dt_) for g, dt_ in zip(grads, ctx.param_dtypes)]
        # a (C, 1, k, k) parameter of a channels_last model has strides (k*k, 1, k, 1), the same memory as the contiguous (k*k, k*k, k, 1): hand the
        # gradient back with the parameter's own strides, so that DDP's bucket views take it without a copy (its "grad strides do not match bucket
        # view strides" warning, GPUTEST_r04).  Only size-1 dimensions may differ: no data moves.
        grads = [g.as_strided(g.shape, ps) if g.stride() != ps and all(n == 1 or a == b for n, a, b in zip(g.shape, g.stride(), ps)) else g
                 for g, ps in zip(grads, ctx.param_strides)]
        return (gx if ctx.needs_input_grad[0] else None, None, *grads)
