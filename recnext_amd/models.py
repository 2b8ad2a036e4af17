"""timm-free model skeleton that hosts the token mixers: RecNeXt-M0..M5 (RecConv2d) and A0..A5 (RecAttn2d).

Restates the *structure* of model/recnext.py:125-287 and model/recattn.py:139-300 (stem, four stages of
MetaNeXt blocks, strided-depthwise Downsample, two-headed classifier) with the reference's module
and parameter names, so ``state_dict`` keys -- and therefore released checkpoints -- are interchangeable.
Everything here is ordinary PyTorch-ROCm plumbing (1x1 convs, BN, GELU run on MIOpen/hipBLASLt);
the only arithmetic owned by this repository is inside ``token_mixer``.

``token_mixer`` factories are injectable so tests / the CPU baseline can host the oracle's ATen
restatement in the same skeleton; the default is always the HIP module.
"""
import torch
import torch.nn as nn

from .layers import ConvNorm, DropPath, NormLinear
from .recattn import RecAttn2d
from .recconv import RecConv2d

# model/recnext.py:365-407 and model/recattn.py:378-420
CONFIGS = {
    "recnext_m0": dict(family="m", embed_dim=(40, 80, 160, 320), depth=(2, 2, 9, 1)),
    "recnext_m1": dict(family="m", embed_dim=(48, 96, 192, 384), depth=(3, 3, 15, 2)),
    "recnext_m2": dict(family="m", embed_dim=(56, 112, 224, 448), depth=(3, 3, 15, 2)),
    "recnext_m3": dict(family="m", embed_dim=(64, 128, 256, 512), depth=(3, 3, 13, 2)),
    "recnext_m4": dict(family="m", embed_dim=(64, 128, 256, 512), depth=(5, 5, 25, 4), drop_path=0.2),
    "recnext_m5": dict(family="m", embed_dim=(80, 160, 320, 640), depth=(7, 7, 35, 2), drop_path=0.3),
    "recnext_a0": dict(family="a", embed_dim=(40, 80, 160, 320), depth=(2, 2, 9, 1)),
    "recnext_a1": dict(family="a", embed_dim=(48, 96, 192, 384), depth=(3, 3, 15, 2)),
    "recnext_a2": dict(family="a", embed_dim=(56, 112, 224, 448), depth=(3, 3, 15, 2)),
    "recnext_a3": dict(family="a", embed_dim=(64, 128, 256, 512), depth=(3, 3, 13, 2), mlp_ratio=1.875),
    "recnext_a4": dict(family="a", embed_dim=(64, 128, 256, 512), depth=(5, 5, 25, 4), mlp_ratio=1.875, drop_path=0.2),
    "recnext_a5": dict(family="a", embed_dim=(80, 160, 320, 640), depth=(7, 7, 35, 2), mlp_ratio=1.875, drop_path=0.3),
}


def default_token_mixer(family):
    if family == "m":   # model/recnext.py:152
        return lambda dim, stage: RecConv2d(dim, level=4 - stage, kernel_size=5)
    if family == "a":   # model/recattn.py:166
        return lambda dim, stage: RecAttn2d(dim, num_heads=2 ** (stage + 1), stage=stage)
    raise ValueError(f"unknown family {family!r}")


def channel_mlp(dim, hidden, act_layer):
    return nn.Sequential(ConvNorm(dim, int(hidden), kernel_size=1), act_layer(), ConvNorm(int(hidden), dim, kernel_size=1))


class RecNextStem(nn.Module):
    def __init__(self, in_channels, out_channels, act_layer=nn.GELU):
        super().__init__()
        self.stem = nn.Sequential(ConvNorm(in_channels, out_channels // 2, kernel_size=3, stride=2, padding=1), act_layer(),
                                  ConvNorm(out_channels // 2, out_channels, kernel_size=3, stride=2, padding=1))

    def forward(self, x):
        fused = self.__dict__.get("_fused_stem")
        if fused is not None and not self.training and fused.usable(self.stem, x):
            return fused(x)                             # one launch (use_fused_stem)
        return self.stem(x)


class MetaNeXtBlock(nn.Module):
    """x + drop_path(channel_mixer([norm](token_mixer(x)))); the M family has the BatchNorm, the A family does not."""

    def __init__(self, dim, mlp_ratio, act_layer, stage, drop_path, family, token_mixer):
        super().__init__()
        self.token_mixer = token_mixer(dim, stage)
        if family == "m":
            self.norm = nn.BatchNorm2d(dim)
        self.channel_mixer = channel_mlp(dim, dim * mlp_ratio, act_layer)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self._has_norm = family == "m"

    def forward(self, x):
        t = self.token_mixer(x)
        if self._has_norm:
            t = self.norm(t)
        fused = self.__dict__.get("_fused_mlp")
        if fused is not None and not self.training and fused.usable(self.channel_mixer, t, x):
            return fused(t, x)                          # x + channel_mixer(t) in one launch (use_fused_mlp)
        return x + self.drop_path(self.channel_mixer(t))


class Downsample(nn.Module):
    def __init__(self, in_channels, mlp_ratio, act_layer):
        super().__init__()
        out_channels = in_channels * 2
        self.token_mixer = nn.Conv2d(in_channels, out_channels, kernel_size=7, padding=3, groups=in_channels, stride=2)
        self.norm = nn.BatchNorm2d(out_channels)
        self.channel_mixer = channel_mlp(out_channels, out_channels * mlp_ratio, act_layer)
        object.__setattr__(self, "_hip", None)          # set by use_hip_downsample; never a registered child (state_dict keys stay)

    def _hip_path(self):
        """The HIP reroute if it still wraps THIS module's current children.  A child replaced after use_hip_downsample
        (SyncBatchNorm.convert_sync_batchnorm, a fusion pass, `m.norm = ...`) must be followed: the old module's parameters
        are in neither state_dict nor the optimizer any more.  Whole-model pickles from before the reroute existed have no attribute."""
        hip = self.__dict__.get("_hip")
        if hip is not None and (hip.token_mixer is not self.token_mixer or hip.norm is not self.norm):
            hip = None
            if isinstance(self.token_mixer, nn.Conv2d) and type(self.norm) is nn.BatchNorm2d:
                from .dwconv import DownsampleDwConv
                try:
                    hip = DownsampleDwConv(self.token_mixer, self.norm)
                except ValueError:                       # not the depthwise C -> 2C conv any more: the PyTorch operators
                    hip = None
            object.__setattr__(self, "_hip", hip)
        return hip

    def forward(self, x):
        hip = self._hip_path()
        x = hip(x) if hip is not None else self.norm(self.token_mixer(x))
        fused = self.__dict__.get("_fused_mlp")
        if fused is not None and not self.training and fused.usable(self.channel_mixer, x, x):
            return fused(x, x)
        return x + self.channel_mixer(x)


class RecNextClassifier(nn.Module):
    def __init__(self, dim, num_classes, distillation=False, drop=0.0):
        super().__init__()
        self.head_drop = nn.Dropout(drop)
        self.head = NormLinear(dim, num_classes) if num_classes > 0 else nn.Identity()
        self.head_dist = NormLinear(dim, num_classes) if num_classes > 0 else nn.Identity()
        self.distillation = distillation
        self.num_classes = num_classes

    def forward(self, x):
        x = self.head_drop(x)
        a, b = self.head(x), self.head_dist(x)
        if self.training and self.distillation:
            return a, b
        return (a + b) / 2

    @torch.no_grad()
    def fuse(self):
        if self.num_classes <= 0:
            return nn.Identity()
        a, b = self.head.fuse(), self.head_dist.fuse()
        a.weight.copy_((a.weight + b.weight) / 2)
        a.bias.copy_((a.bias + b.bias) / 2)
        return a


class RecNextStage(nn.Module):
    def __init__(self, in_channels, out_channels, depth, mlp_ratio, act_layer, downsample, stage, drop_path, family, token_mixer):
        super().__init__()
        self.downsample = Downsample(in_channels, mlp_ratio, act_layer) if downsample else nn.Identity()
        self.blocks = nn.Sequential(*[MetaNeXtBlock(out_channels, mlp_ratio, act_layer, stage, drop_path, family, token_mixer)
                                      for _ in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


class RecNext(nn.Module):
    def __init__(self, family="m", in_chans=3, embed_dim=(48,), depth=(2,), mlp_ratio=2, global_pool="avg", num_classes=1000,
                 act_layer=nn.GELU, distillation=False, drop_rate=0.0, drop_path=0.0, token_mixer=None):
        super().__init__()
        token_mixer = token_mixer or default_token_mixer(family)
        self.family = family
        self.global_pool = global_pool
        self.embed_dim = tuple(embed_dim)
        self.num_classes = num_classes
        self.stem = RecNextStem(in_chans, embed_dim[0], act_layer)
        stages, prev = [], embed_dim[0]
        for i, (dim, d) in enumerate(zip(embed_dim, depth)):
            stages.append(RecNextStage(prev, dim, d, mlp_ratio, act_layer, downsample=i != 0, stage=i, drop_path=drop_path,
                                       family=family, token_mixer=token_mixer))
            prev = dim
        self.stages = nn.Sequential(*stages)
        self.num_features = embed_dim[-1]
        self.head_drop = nn.Dropout(drop_rate)
        self.head = RecNextClassifier(embed_dim[-1], num_classes, distillation)

    def forward_features(self, x):
        return self.stages(self.stem(x))

    def forward_head(self, x):
        if self.global_pool == "avg":
            x = x.mean((2, 3))
        return self.head(self.head_drop(x))

    def forward(self, x):
        return self.forward_head(self.forward_features(x))


def create_model(name, distillation=False, token_mixer=None, **overrides):
    """``timm.create_model`` stand-in for the twelve registered names (model/recnext.py:365-407, model/recattn.py:378-420)."""
    cfg = dict(CONFIGS[name])
    if distillation:
        cfg["drop_path"] = 0.0                      # drop_path applies to the non-distilled recipe only
    cfg.update(overrides)
    return RecNext(distillation=distillation, token_mixer=token_mixer, **cfg)


def replace_batchnorm(net):
    """utils.replace_batchnorm (utils.py:227-234): swap every child that knows how to ``fuse`` itself, recursively.

    Plain ``nn.BatchNorm2d`` children (MetaNeXtBlock.norm, Downsample.norm) have no ``fuse`` and stay.
    """
    for name, child in list(net.named_children()):
        if hasattr(child, "fuse"):
            fused = child.fuse()
            setattr(net, name, fused)
            replace_batchnorm(fused)
        else:
            replace_batchnorm(child)
    return net


@torch.no_grad()
def fold_token_mixer_norms(net):
    """Inference-only: absorb each MetaNeXtBlock's eval-mode ``norm`` (a per-channel affine that
    ``replace_batchnorm`` leaves in place, utils.py:227-234) into its HIP token mixer's final conv.

    x + mlp(norm(mixer(x)))  ==  x + mlp(mixer'(x))  with  convs[level]' = scale*convs[level] + shift.
    Returns the number of BatchNorm layers removed (21 for RecNeXt-M3).  SURVEY.md section 8f row 2.
    """
    n = 0
    for m in net.modules():
        if isinstance(m, MetaNeXtBlock) and m._has_norm and isinstance(m.norm, nn.BatchNorm2d) \
                and isinstance(m.token_mixer, RecConv2d):
            if m.norm.training:
                raise RuntimeError("fold_token_mixer_norms needs eval mode (running statistics)")
            bn = m.norm
            scale = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
            shift = bn.bias.float() - scale * bn.running_mean.float()
            m.token_mixer.fold_output_affine(scale, shift)
            m.norm = nn.Identity()
            m._has_norm = False
            n += 1
    return n


@torch.no_grad()
def use_hip_downsample(net):
    """Run each Downsample's depthwise 7x7 stride-2 conv (C -> 2C) on HIP: in eval mode fused with
    the BatchNorm after it into one kernel, in a training step with a HIP backward and the norm on batch statistics.
    ``token_mixer`` and ``norm`` remain the Downsample's direct children (same state_dict keys as the reference, so checkpoints
    load and save unchanged in either order); only the forward is rerouted.  Returns the number of layers rerouted."""
    from .dwconv import DownsampleDwConv
    n = 0
    for m in net.modules():
        if isinstance(m, Downsample) and m._hip is None and isinstance(m.token_mixer, nn.Conv2d) and isinstance(m.norm, nn.BatchNorm2d):
            object.__setattr__(m, "_hip", DownsampleDwConv(m.token_mixer, m.norm))
            n += 1
    return n


def use_linear_pointwise(net):
    """Inference-only plumbing: evaluate the (BN-folded) 1x1 convs of every channel mixer through the GEMM library
    (bias in the epilogue) instead of the conv library (bias as a separate kernel).  Returns the number replaced."""
    from .layers import PointwiseLinear
    n = 0
    for m in net.modules():
        if isinstance(m, (MetaNeXtBlock, Downsample)):
            seq = m.channel_mixer
            for i, sub in enumerate(seq):
                if isinstance(sub, nn.Conv2d) and sub.kernel_size == (1, 1) and sub.groups == 1 and sub.bias is not None:
                    seq[i] = PointwiseLinear(sub)
                    n += 1
    return n


def use_fused_stem(net):
    """Inference-only, after ``replace_batchnorm``: evaluate RecNextStem ([3x3 stride-2 conv + bias, exact GELU, 3x3 stride-2 conv + bias]) as ONE HIP launch where
    rcx_stem_fwd has a kernel (bf16 on a GPU; decided per call) -- the 112 x 112 intermediate never reaches memory.  The convs, their parameters and the state_dict
    are untouched.  Returns the number of stems given the fused path (0 or 1)."""
    from .layers import FusedStem
    n = 0
    for m in net.modules():
        if isinstance(m, RecNextStem) and m.__dict__.get("_fused_stem") is None:
            seq = m.stem
            if len(seq) == 3 and all(isinstance(seq[i], nn.Conv2d) and seq[i].kernel_size == (3, 3) and seq[i].stride == (2, 2) and seq[i].padding == (1, 1)
                                     and seq[i].groups == 1 and seq[i].dilation == (1, 1) for i in (0, 2)) \
                    and isinstance(seq[1], nn.GELU) and getattr(seq[1], "approximate", "none") == "none" and seq[0].in_channels == 3:
                object.__setattr__(m, "_fused_stem", FusedStem(seq[0], seq[2]))
                n += 1
    return n


def use_fused_mlp(net):
    """Inference-only, after ``use_linear_pointwise``: evaluate ``x + channel_mixer(t)`` of every MetaNeXtBlock / Downsample whose mixer is
    [1x1 conv, exact GELU, 1x1 conv] as ONE HIP launch where rcx_channel_mlp_fwd has a kernel (bf16; the 56 x 56 and 28 x 28 stages, where the two GEMMs
    are memory-bound and the hidden tensor's round trips are most of their time) -- other blocks, dtypes and devices keep the GEMM library, decided per
    call.  The layers, their parameters and the state_dict are untouched.  Returns the number of blocks given the fused path."""
    from .layers import FusedChannelMlp, PointwiseLinear
    n = 0
    for m in net.modules():
        if isinstance(m, (MetaNeXtBlock, Downsample)) and m.__dict__.get("_fused_mlp") is None:
            seq = m.channel_mixer
            if len(seq) == 3 and isinstance(seq[0], PointwiseLinear) and isinstance(seq[2], PointwiseLinear) and isinstance(seq[1], nn.GELU) \
                    and getattr(seq[1], "approximate", "none") == "none":            # (drop_path is the identity in eval mode, the only mode the fused path runs in)
                object.__setattr__(m, "_fused_mlp", FusedChannelMlp(seq[0], seq[2]))
                n += 1
    return n


def padded_hidden_width(hidden, max_growth=0.125):
    """The hidden width the channel mixers' GEMMs run at: the next multiple of 64 (else of 16) when that adds at most ``max_growth``, else unchanged."""
    for mult in (64, 16):
        hp = -(-hidden // mult) * mult
        if hp != hidden and hp <= hidden * (1 + max_growth):
            return hp
    return hidden


def pad_mlp_hidden(net):
    """Inference-only plumbing, after ``use_linear_pointwise``: run the two GEMMs of a channel mixer at a zero-padded hidden width (RecNeXt-A's
    1.875 x dim -- 120 / 240 / 480 -- become 128 / 256 / 512).  The same function: a padded hidden channel is act(0 + 0) = 0 and meets zero weights
    (checked: the activation must map 0 to 0).  Parameters and ``state_dict`` are untouched (``PointwiseLinear.pad``).  Returns the mixers changed."""
    from .layers import PointwiseLinear
    n = 0
    for m in net.modules():
        if isinstance(m, (MetaNeXtBlock, Downsample)):
            seq = m.channel_mixer
            if len(seq) != 3 or not isinstance(seq[0], PointwiseLinear) or not isinstance(seq[2], PointwiseLinear):
                continue
            with torch.no_grad():
                if float(seq[1](torch.zeros(1)).abs().max()) != 0.0:
                    continue
            hp = padded_hidden_width(seq[0].out_channels)
            if hp != seq[0].out_padded or hp != seq[2].in_padded:
                seq[0].pad(out_to=hp)
                seq[2].pad(in_to=hp)
                n += hp != seq[0].out_channels
    return n


def token_mixer_shapes(name, resolution=224):
    """[(C, H, W, level|None, count)] of every token mixer call in one forward pass (SURVEY 8 model tables)."""
    cfg = CONFIGS[name]
    out = []
    side = resolution // 4
    for s, (dim, d) in enumerate(zip(cfg["embed_dim"], cfg["depth"])):
        out.append((dim, side, side, 4 - s if cfg["family"] == "m" else None, d))
        side = (side + 1) // 2 if s < 3 else side      # Downsample: k7 s2 p3 -> ceil(side/2)
    return out


def token_mixer_algorithmic_bytes(name, resolution=224, elem_bytes=2, k=5):
    """Compulsory traffic of all RecConv2d blocks for ONE image: 2*C*H*W*b + (level+2)*C*k*k*b per block."""
    total = 0
    for (c, h, w, level, count) in token_mixer_shapes(name, resolution):
        if level is None:
            raise ValueError("defined for the M family")
        total += count * (2 * c * h * w * elem_bytes + (level + 2) * c * k * k * elem_bytes)
    return total
