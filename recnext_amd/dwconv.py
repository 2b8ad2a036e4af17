"""HIP stand-in for ``Downsample.token_mixer`` + ``Downsample.norm`` (model/recnext.py:165-166, :170):
a depthwise k x k stride-2 conv with channel multiplier 2 (``nn.Conv2d(C, 2C, 7, padding=3, groups=C, stride=2)``)
followed by an eval-mode BatchNorm, which is a per-channel affine and is folded into the packed weights.
In a training step the BatchNorm works on batch statistics and is not folded: the conv (forward and backward) runs on HIP
through ``DwConvMult2Fn`` and the norm stays a PyTorch module.  ``models.use_hip_downsample`` attaches the wrapper to a
``Downsample`` WITHOUT registering it as a child module: ``token_mixer`` and ``norm`` stay the Downsample's own direct children,
so the state_dict keys are the reference's (``downsample.token_mixer.weight``, ``downsample.norm.*``) before and after.  ``DwConvFn`` is the plain depthwise conv (stride 1|2) with HIP forward and backward, used by RecAttn2d.
"""
import torch
import torch.nn as nn

from . import ops


class DwConvFn(torch.autograd.Function):
    """Depthwise k x k conv (pad k//2, stride 1|2): rcx_dwconv2d_fwd / rcx_dwconv2d_bwd."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride):
        k = weight.shape[-1]
        wp = ops.pack_dw_weight(weight.float())
        bp = ops.pack_bias(bias.float()) if bias is not None else None
        ctx.save_for_backward(x, wp)
        ctx.k, ctx.stride, ctx.has_bias, ctx.wdtype = k, stride, bias is not None, weight.dtype
        return ops.dwconv2d(x, wp, bp, k=k, stride=stride)

    @staticmethod
    def backward(ctx, gy):
        x, wp = ctx.saved_tensors
        k, c = ctx.k, x.shape[1]
        gx, gw, gb = ops.dwconv2d_backward(x, gy, wp, k, ctx.stride, need_input_grad=ctx.needs_input_grad[0], need_bias=ctx.has_bias)
        gw = gw.view(k, k, c).permute(2, 0, 1).unsqueeze(1).to(ctx.wdtype)
        return gx, gw, (gb.to(ctx.wdtype) if gb is not None else None), None


class UpAddDwConvFn(torch.autograd.Function):
    """conv_k(x + interpolate(a, size=x.shape[2:], mode)) -- the last line of RecAttn2d.forward (model/recattn.py:67) -- as ONE HIP launch each way for the
    resize, the add and the conv: rcx_upadd_dwconv_fwd / rcx_upadd_dwconv_bwd (no ATen resize or add in the training step).  The coarse plane is taken
    in float32 (a quarter of x's pixels), as the inference path keeps it."""

    @staticmethod
    def forward(ctx, x, a, weight, bias, mode):
        k = weight.shape[-1]
        wp = ops.pack_dw_weight(weight.float())
        bp = ops.pack_bias(bias.float()) if bias is not None else None
        a32 = a if a.dtype == torch.float32 else a.float()
        ctx.save_for_backward(x, a32, wp)
        ctx.k, ctx.mode, ctx.has_bias, ctx.wdtype, ctx.adtype = k, mode, bias is not None, weight.dtype, a.dtype
        return ops.upadd_dwconv(x, a32, wp, bp, k=k, mode=mode)

    @staticmethod
    def backward(ctx, gy):
        x, a32, wp = ctx.saved_tensors
        k, c = ctx.k, x.shape[1]
        gx, ga, gw, gb = ops.upadd_dwconv_backward(x, a32, gy, wp, k, ctx.mode, need_input_grad=ctx.needs_input_grad[0],
                                                   need_coarse_grad=ctx.needs_input_grad[1], need_bias=ctx.has_bias)
        gw = gw.view(k, k, c).permute(2, 0, 1).unsqueeze(1).to(ctx.wdtype)
        if ga is not None and ga.dtype != ctx.adtype:
            ga = ga.to(ctx.adtype)
        return gx, ga, gw, (gb.to(ctx.wdtype) if gb is not None else None), None


class DwConvMult2Fn(torch.autograd.Function):
    """nn.Conv2d(C, 2C, k, stride=2, padding=k//2, groups=C): rcx_dwconv2d_mult2_fwd / rcx_dwconv2d_mult2_bwd."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        k = weight.shape[-1]
        wp = ops.pack_dw_weight(weight.float())
        bp = ops.pack_bias(bias.float()) if bias is not None else None
        ctx.save_for_backward(x, wp)
        ctx.k, ctx.has_bias, ctx.wdtype = k, bias is not None, weight.dtype
        return ops.dwconv2d_mult2(x, wp, bp, k=k, stride=2)

    @staticmethod
    def backward(ctx, gy):
        x, wp = ctx.saved_tensors
        k, c = ctx.k, x.shape[1]
        gx, gw, gb = ops.dwconv2d_mult2_backward(x, gy, wp, k, need_input_grad=ctx.needs_input_grad[0], need_bias=ctx.has_bias)
        gw = gw.view(k, k, 2 * c).permute(2, 0, 1).unsqueeze(1).to(ctx.wdtype)
        return gx, gw, (gb.to(ctx.wdtype) if gb is not None else None)


class DownsampleDwConv(nn.Module):
    def __init__(self, conv, norm=None):
        super().__init__()
        if conv.groups != conv.in_channels or conv.out_channels != 2 * conv.in_channels:
            raise ValueError("DownsampleDwConv wraps nn.Conv2d(C, 2C, k, groups=C)")
        if conv.kernel_size[0] != conv.kernel_size[1] or conv.kernel_size[0] % 2 != 1 or conv.padding[0] != conv.kernel_size[0] // 2:
            raise ValueError("DownsampleDwConv needs an odd square kernel with padding k//2")
        self.token_mixer = conv          # same attribute names as the reference's Downsample
        self.norm = norm
        self._pack_key = None
        self._pack = None

    def _tensors(self):
        t = [self.token_mixer.weight, self.token_mixer.bias]
        if self.norm is not None:
            t += [self.norm.weight, self.norm.bias, self.norm.running_mean, self.norm.running_var]
        return [x for x in t if x is not None]

    @torch.no_grad()
    def packed_params(self):
        key = tuple((t.data_ptr(), t._version, t.dtype, str(t.device)) for t in self._tensors())
        if key != self._pack_key:
            conv, bn = self.token_mixer, self.norm
            w = conv.weight.float()
            b = conv.bias.float() if conv.bias is not None else torch.zeros(conv.out_channels, device=w.device)
            if bn is not None:
                s = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
                b = bn.bias.float() - s * bn.running_mean.float() + s * b
                w = w * s.view(-1, 1, 1, 1)
            self._pack = (ops.pack_dw_weight(w.contiguous()), ops.pack_bias(b.contiguous()))
            self._pack_key = key
        return self._pack

    def forward(self, x):
        conv = self.token_mixer
        training = self.norm.training if self.norm is not None else conv.training     # the flags of the modules that own the state
        if training or (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))):
            if conv.stride[0] != 2 or conv.kernel_size[0] not in (3, 5, 7) or x.shape[1] % 2:
                raise NotImplementedError("the HIP backward of the multiplier-2 conv covers stride 2, k in {3,5,7}, even channel counts")
            y = DwConvMult2Fn.apply(x, conv.weight, conv.bias)
            return self.norm(y) if self.norm is not None else y
        w, b = self.packed_params()
        k, stride = self.token_mixer.kernel_size[0], self.token_mixer.stride[0]
        return ops.dwconv2d_mult2(x, w, b, k=k, stride=stride)
