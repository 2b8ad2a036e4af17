"""Host-side building blocks around the token mixers (plain PyTorch-ROCm modules, no kernels here).

Parameter names follow the reference so its checkpoints load unchanged:
``ConvNorm`` = {conv, norm} (model/recnext.py:56-97), ``NormLinear`` = {norm, linear} (:100-122).
"""
import torch
import torch.nn as nn


def _bn_affine(norm):
    scale = norm.weight / torch.sqrt(norm.running_var + norm.eps)
    return scale, norm.bias - scale * norm.running_mean


class ConvNorm(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0, dilation=1, groups=1,
                 bias=False, bn_weight_init=1):
        super().__init__()
        self.add_module("conv", nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias=bias))
        self.add_module("norm", nn.BatchNorm2d(out_channels))
        nn.init.constant_(self.norm.weight, bn_weight_init)
        nn.init.constant_(self.norm.bias, 0)

    @torch.no_grad()
    def fuse(self):
        """Eval-mode BN folded into a biased conv: w' = w*g/sqrt(var+eps), b' = beta - mean*g/sqrt(var+eps)."""
        conv = self.conv
        scale, shift = _bn_affine(self.norm)
        if conv.bias is not None:
            shift = shift + scale * conv.bias
        out = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, stride=conv.stride, padding=conv.padding,
                        dilation=conv.dilation, groups=conv.groups, bias=True, device=conv.weight.device,
                        dtype=conv.weight.dtype)
        out.weight.copy_(conv.weight * scale.view(-1, 1, 1, 1))
        out.bias.copy_(shift)
        return out


class NormLinear(nn.Sequential):
    def __init__(self, in_channels, out_channels, bias=True, std=0.02):
        super().__init__()
        self.add_module("norm", nn.BatchNorm1d(in_channels))
        self.add_module("linear", nn.Linear(in_channels, out_channels, bias=bias))
        nn.init.trunc_normal_(self.linear.weight, std=std)
        if bias:
            nn.init.constant_(self.linear.bias, 0)

    @torch.no_grad()
    def fuse(self):
        lin = self.linear
        scale, shift = _bn_affine(self.norm)
        out = nn.Linear(lin.in_features, lin.out_features, bias=True, device=lin.weight.device, dtype=lin.weight.dtype)
        out.weight.copy_(lin.weight * scale.view(1, -1))
        b = lin.weight @ shift
        out.bias.copy_(b if lin.bias is None else b + lin.bias)
        return out


class DropPath(nn.Module):
    """Stochastic depth (timm.layers.DropPath semantics); identity in eval mode."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if not self.training or self.drop_prob == 0.0:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask.div_(keep)


class PointwiseLinear(nn.Module):
    """A BN-folded 1x1 conv evaluated as ``F.linear`` on the (N*H*W, C) view of a channels_last tensor.

    Same parameters (``weight`` (Cout,Cin,1,1), ``bias``) and same function as the ``nn.Conv2d`` it replaces; the
    only difference is the library call: on ROCm the GEMM path (hipBLASLt) applies the bias in its epilogue, whereas
    the MIOpen conv path launches a separate elementwise kernel for it.  Host-side plumbing, inference only.

    ``pad(out_to=, in_to=)`` makes the GEMM run on zero-padded copies of the parameters (extra output channels with
    zero weights and zero bias, extra input channels with zero weights): the GEMM library's tiles for RecNeXt-A's
    hidden widths (1.875 x dim: 120, 240, 480) are far slower than for the next multiple of 64.  The parameters and
    the ``state_dict`` stay as they are; the padded copies follow them (rebuilt when a parameter changes).
    """

    def __init__(self, conv):
        super().__init__()
        if conv.kernel_size != (1, 1) or conv.groups != 1 or conv.stride != (1, 1) or conv.padding != (0, 0):
            raise ValueError("PointwiseLinear replaces a dense 1x1, stride-1 convolution")
        self.weight, self.bias = conv.weight, conv.bias
        self.in_channels, self.out_channels = conv.in_channels, conv.out_channels
        self.out_padded, self.in_padded = self.out_channels, self.in_channels
        self._pad_key, self._pad_w, self._pad_b = None, None, None

    def pad(self, out_to=None, in_to=None):
        out_to = self.out_channels if out_to is None else int(out_to)
        in_to = self.in_channels if in_to is None else int(in_to)
        if out_to < self.out_channels or in_to < self.in_channels:
            raise ValueError("padding cannot shrink a layer")
        self.out_padded, self.in_padded, self._pad_key = out_to, in_to, None
        return self

    def _operands(self):
        w, b = self.weight, self.bias
        if self.out_padded == self.out_channels and self.in_padded == self.in_channels:
            return w.view(self.out_channels, self.in_channels), b
        key = (w.data_ptr(), w._version, w.dtype, w.device, None if b is None else (b.data_ptr(), b._version))
        if key != self._pad_key:
            with torch.no_grad():
                wp = w.new_zeros(self.out_padded, self.in_padded)
                wp[:self.out_channels, :self.in_channels] = w.view(self.out_channels, self.in_channels)
                bp = None
                if b is not None:
                    bp = b.new_zeros(self.out_padded)
                    bp[:self.out_channels] = b
            self._pad_key, self._pad_w, self._pad_b = key, wp, bp
        return self._pad_w, self._pad_b

    def forward(self, x):
        n, c, h, w = x.shape
        if c != self.in_padded:
            raise ValueError(f"expected {self.in_padded} input channels, got {c}")
        wt, b = self._operands()
        x2 = x.permute(0, 2, 3, 1)                       # a view when x is channels_last
        y2 = torch.nn.functional.linear(x2.reshape(n * h * w, c), wt, b)
        return y2.view(n, h, w, self.out_padded).permute(0, 3, 1, 2)


def _is_exact_gelu(m):
    return type(m) is nn.GELU and getattr(m, "approximate", "none") == "none"


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class FusedChannelMlp:
    """The channel mixer [PointwiseLinear, GELU, PointwiseLinear] of a block and the residual add around it as ONE HIP launch (ops.channel_mlp; inference,
    bf16, channels_last).  Not a Module: it reads the two layers' own parameters (state_dict untouched) and rebuilds its fragment pack when one changes."""

    def __init__(self, fc1, fc2):
        self.fc1, self.fc2 = fc1, fc2
        self._key, self._pack = None, None

    def supported(self, x):
        from . import ops
        n, c, h, w = x.shape
        return x.is_cuda and x.dtype == torch.bfloat16 and ops.channel_mlp_hidden(n * h * w, c, self.fc1.out_channels, x.dtype) > 0

    def usable(self, seq, t, x):
        """May this call take the fused launch?  Only if it still describes the mixer ``seq`` as it is NOW (a layer or the activation replaced after
        use_fused_mlp -- a fusion / quantisation pass, nn.DataParallel's replicas, ``seq[1] = nn.ReLU()`` -- is followed, not ignored), the operands
        agree in dtype and device with the weights, and nothing on the way needs a gradient: ops.channel_mlp is a raw launch with no grad_fn, so
        an eval-mode forward with grad enabled (input gradients, a frozen-BN fine-tune) must keep the autograd path."""
        if len(seq) != 3 or seq[0] is not self.fc1 or seq[2] is not self.fc2 or not _is_exact_gelu(seq[1]):
            return False
        if t.dtype != x.dtype or t.device != x.device or self.fc1.weight.device != x.device or self.fc2.weight.device != x.device:
            return False
        if _needs_grad(t, x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias):
            return False
        return self.supported(x)

    def packs(self):
        return [] if self._pack is None else [p for p in self._pack if torch.is_tensor(p)]

    def _operands(self, hidden_to):
        from . import ops
        ts = [self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias]
        key = (hidden_to,) + tuple(None if t is None else (t.data_ptr(), t._version, t.dtype, t.device) for t in ts)
        if key != self._key:
            self._pack = ops.pack_channel_mlp(*ts, hidden_to=hidden_to)
            self._key = key
        return self._pack

    def __call__(self, z, x):
        from . import ops
        n, c, h, w = x.shape
        wfrag, bias, hidden = self._operands(ops.channel_mlp_hidden(n * h * w, c, self.fc1.out_channels, x.dtype))
        return ops.channel_mlp(z, x, wfrag, bias, hidden)


class FusedStem:
    """RecNextStem's [3x3 stride-2 conv, GELU, 3x3 stride-2 conv] (BatchNorms folded) as ONE HIP launch (ops.stem; inference, bf16, channels_last).  Not a Module: it
    reads the two convs' own parameters (state_dict untouched) and rebuilds its packs when one changes."""

    def __init__(self, conv1, conv2):
        self.conv1, self.conv2 = conv1, conv2
        self._key, self._pack = None, None

    def supported(self, x):
        from . import ops
        n, c, h, w = x.shape
        return x.is_cuda and x.dtype == torch.bfloat16 and c == 3 and ops.stem_supported(n, h, w, self.conv1.out_channels, self.conv2.out_channels, x.dtype)

    def usable(self, seq, x):
        """As FusedChannelMlp.usable: the stem must still be [conv1, exact GELU, conv2] with these very layers, on x's device, with no gradient wanted."""
        if len(seq) != 3 or seq[0] is not self.conv1 or seq[2] is not self.conv2 or not _is_exact_gelu(seq[1]):
            return False
        if self.conv1.weight.device != x.device or self.conv2.weight.device != x.device:
            return False
        if _needs_grad(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias):
            return False
        return self.supported(x)

    def packs(self):
        return [] if self._pack is None else [p for p in self._pack if torch.is_tensor(p)]

    def __call__(self, x):
        from . import ops
        ts = [self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias]
        key = tuple(None if t is None else (t.data_ptr(), t._version, t.dtype, t.device) for t in ts)
        if key != self._key:
            self._pack = ops.pack_stem(*ts)
            self._key = key
        return ops.stem(x, *self._pack, self.conv1.out_channels, self.conv2.out_channels)
