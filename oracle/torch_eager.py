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


class EagerRecAttn2d(nn.Module):
    """RecAttn2d (model/recattn.py:54-67) on ATen ops with the reference's parameter names.

    Built from the same host-side plumbing modules (ConvNorm, LinearAttention) as the product skeleton;
    only the token-mixer arithmetic differs (ATen here, HIP kernels there).
    """

    def __init__(self, dim, num_heads, kernel_size=5, stage=1, mode="nearest"):
        super().__init__()
        from recnext_amd.layers import ConvNorm
        from recnext_amd.recattn import LinearAttention
        self.mode = mode
        self.down = nn.Sequential(
            ConvNorm(dim, dim, kernel_size=kernel_size, padding=kernel_size // 2, stride=2, groups=dim),
            LinearAttention(dim=dim, num_heads=num_heads, variant=2 if stage >= 3 else 1),
        )
        self.conv = ConvNorm(dim, dim, kernel_size=kernel_size, padding=kernel_size // 2, groups=dim)

    def forward(self, x):
        return self.conv(x + F.interpolate(self.down(x), size=x.shape[2:], mode=self.mode))


def eager_token_mixer(family):
    """token_mixer factory for recnext_amd.models.RecNext hosting the ATen restatements (CPU baseline / tests)."""
    if family == "m":
        return lambda dim, stage: EagerRecConv2d(dim, level=4 - stage, kernel_size=5)
    return lambda dim, stage: EagerRecAttn2d(dim, num_heads=2 ** (stage + 1), stage=stage)
