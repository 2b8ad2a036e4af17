"""HIP stand-in for ``Downsample.token_mixer`` + ``Downsample.norm`` (model/recnext.py:165-166, :170):
a depthwise k x k stride-2 conv with channel multiplier 2 (``nn.Conv2d(C, 2C, 7, padding=3, groups=C, stride=2)``)
followed by an eval-mode BatchNorm, which is a per-channel affine and is folded into the packed weights.
Inference only; parameters stay in the wrapped modules, so state_dict keys do not change.
"""
import torch
import torch.nn as nn

from . import ops


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
        if self.training or (torch.is_grad_enabled() and x.requires_grad):
            raise NotImplementedError("DownsampleDwConv is inference-only (eval mode, no autograd)")
        w, b = self.packed_params()
        k, stride = self.token_mixer.kernel_size[0], self.token_mixer.stride[0]
        return ops.dwconv2d_mult2(x, w, b, k=k, stride=stride)
