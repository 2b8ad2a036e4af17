"""The reference's CPU path, restated on the same ATen operators -- TEST INFRASTRUCTURE ONLY.

The reference's hot path *is* three ATen calls per level (depthwise ``conv2d``, ``interpolate``,
``add``; model/recnext.py:24-34, model/recattn.py:54-67).  The Python files of the reference cannot
travel to the GPU box, so this module restates the token mixers as functional code over the very
same operators.  It is what ``bench.py`` times as ``cpu_baseline`` (kind "port": same library
kernels -- mkldnn depthwise conv, upsample_bilinear2d -- the reference would run on those cores) and
what the whole-model tests use as the slow-but-trusted token mixer.  The product never imports it.

Pinned by tests/test_oracle.py against the golden vectors captured from the imported reference.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def recconv2d_eager(x, w_down, w_convs, b_down=None, b_convs=None, mode="bilinear"):
    """Functional RecConv2d.forward (model/recnext.py:24-34) on ATen ops."""
    c = x.shape[1]
    k = w_down.shape[-1]
    level = len(w_convs) - 1
    pyramid = [x]
    while len(pyramid) <= level:                      # shared stride-2 depthwise conv (:27-29)
        pyramid.append(F.conv2d(pyramid[-1], w_down, b_down, stride=2, padding=k // 2, groups=c))
    carry = None
    for j in range(level):                            # coarsest level first (:32-33)
        fine, coarse = pyramid[level - j - 1], pyramid[level - j]
        t = coarse if carry is None else coarse + carry
        t = F.conv2d(t, w_convs[j], None if b_convs is None else b_convs[j], padding=k // 2, groups=c)
        carry = F.interpolate(t, size=fine.shape[2:], mode=mode)
    t = x if carry is None else x + carry
    return F.conv2d(t, w_convs[level], None if b_convs is None else b_convs[level], padding=k // 2, groups=c)


class EagerRecConv2d(nn.Module):
    """State-dict compatible with the reference block: ``down.weight``, ``convs.{i}.weight`` (+ ``.bias``)."""

    def __init__(self, in_channels, kernel_size=5, bias=False, level=2, mode="bilinear"):
        super().__init__()
        self.level, self.mode = level, mode
        mk = lambda s: nn.Conv2d(in_channels, in_channels, kernel_size, stride=s, padding=kernel_size // 2,
                                 groups=in_channels, bias=bias)
        self.down = mk(2)
        self.convs = nn.ModuleList([mk(1) for _ in range(level + 1)])

    def forward(self, x):
        has_b = self.down.bias is not None
        return recconv2d_eager(x, self.down.weight, [cv.weight for cv in self.convs],
                               self.down.bias, [cv.bias for cv in self.convs] if has_b else None, self.mode)


def upadd_dwconv_eager(x, coarse, w, b=None, mode="nearest"):
    """conv(x + interpolate(coarse, size(x))) -- the tail of RecAttn2d.forward (model/recattn.py:67)."""
    k = w.shape[-1]
    return F.conv2d(x + F.interpolate(coarse, size=x.shape[2:], mode=mode), w, b, padding=k // 2, groups=x.shape[1])


def dwconv_eager(x, w, b=None, stride=1):
    k = w.shape[-1]
    return F.conv2d(x, w, b, stride=stride, padding=k // 2, groups=x.shape[1])


class EagerConvNorm(nn.Sequential):
    """{conv, norm} pair with the reference's parameter names and its own fuse() (model/recattn.py:70-111)."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0, groups=1):
        super().__init__()
        self.add_module("conv", nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, 1, groups, bias=False))
        self.add_module("norm", nn.BatchNorm2d(out_channels))

    @torch.no_grad()
    def fuse(self):
        g = self.norm.weight / (self.norm.running_var + self.norm.eps) ** 0.5          # :91
        conv = nn.Conv2d(self.conv.in_channels, self.conv.out_channels, self.conv.kernel_size, stride=self.conv.stride,
                         padding=self.conv.padding, groups=self.conv.groups, bias=True,
                         device=self.conv.weight.device, dtype=self.conv.weight.dtype)
        conv.weight.copy_(self.conv.weight * g[:, None, None, None])                    # :97
        conv.bias.copy_(self.norm.bias - g * self.norm.running_mean)                    # :92
        return conv


class EagerLinearAttention(nn.Module):
    """LinearAttention1 / LinearAttention2 (model/recattn.py:8-28, :31-51) on ATen operators, line for line."""

    def __init__(self, dim, num_heads, variant=1):
        super().__init__()
        self.num_heads, self.head_dim, self.variant = num_heads, dim // num_heads, variant
        self.qk = EagerConvNorm(dim, dim * 2, kernel_size=1, groups=2)
        self.pe = EagerConvNorm(dim, dim, kernel_size=3, padding=1, groups=dim)

    def forward(self, x):
        b, c, h, w = x.shape
        n = h * w
        s = n ** -0.5
        qk = F.elu(self.qk(x)) + 1.0                                                    # :21 / :44
        q, k = qk.reshape(b, 2, self.num_heads, self.head_dim, n).unbind(dim=1)
        v_t = x.reshape(b, self.num_heads, self.head_dim, n).transpose(-1, -2)
        q_t = q.transpose(-1, -2)
        if self.variant == 1:
            kv = (k * s) @ (v_t * s)                                                    # :25
            out = q_t @ kv / (q_t @ k.mean(dim=-1, keepdim=True) + 1e-6)                # :26
        else:
            a = q_t @ k                                                                 # :47
            a = a / (a.mean(dim=-1, keepdim=True) + 1e-6)                               # :48
            out = (a * s) @ (v_t * s)                                                   # :49
        return out.transpose(-1, -2).reshape(b, c, h, w) + self.pe(x)                   # :28 / :51


class EagerRecAttn2d(nn.Module):
    """RecAttn2d (model/recattn.py:54-67) on ATen ops with the reference's parameter names.  Imports nothing from the product."""

    def __init__(self, dim, num_heads, kernel_size=5, stage=1, mode="nearest"):
        super().__init__()
        self.mode = mode
        self.down = nn.Sequential(
            EagerConvNorm(dim, dim, kernel_size=kernel_size, padding=kernel_size // 2, stride=2, groups=dim),
            EagerLinearAttention(dim=dim, num_heads=num_heads, variant=2 if stage >= 3 else 1),       # :59
        )
        self.conv = EagerConvNorm(dim, dim, kernel_size=kernel_size, padding=kernel_size // 2, groups=dim)

    def forward(self, x):
        return self.conv(x + F.interpolate(self.down(x), size=x.shape[2:], mode=self.mode))           # :67


def eager_token_mixer(family):
    """token_mixer factory for recnext_amd.models.RecNext hosting the ATen restatements (CPU baseline / tests)."""
    if family == "m":
        return lambda dim, stage: EagerRecConv2d(dim, level=4 - stage, kernel_size=5)
    return lambda dim, stage: EagerRecAttn2d(dim, num_heads=2 ** (stage + 1), stage=stage)
