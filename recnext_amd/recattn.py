"""Drop-in ``RecAttn2d`` (A-series token mixer, model/recattn.py:54-67).

    y = conv( x + interpolate( LinearAttention( down(x) ), size=x.shape[2:], mode ) )

``down[0]`` (depthwise k x k, stride 2) and ``conv`` (depthwise k x k) are ``ConvNorm`` pairs in the
reference; in eval mode their BatchNorm is a per-channel affine, which is folded into the packed
weights here whether or not ``replace_batchnorm`` has already replaced the pair by a biased
``nn.Conv2d`` (model/recattn.py:89-111).  The HIP kernels cover the two depthwise convs and the fused
nearest-resize + add + conv, the `pe` depthwise 3x3 and the linear-attention core (q/k activation,
k v^T, normaliser, + pe: ``rcx_linear_attention_fwd``); only the grouped 1x1
`qk` projection -- two plain GEMMs -- goes through the GEMM library.  There is no PyTorch-operator path: CPU tensors and head sizes the
core does not take raise (the reference's formulation lives in oracle/torch_eager.py).  In eval mode everything between the stride-2 conv
and the final conv -- the quarter-size tensors d, q, k, pe, the attention output -- stays float32 whatever x's type, so a bf16 / float16 run
rounds once, at the store of y (north_star's flat 1e-2 against the float32 reference).  Parameter names and shapes equal the reference's.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .dwconv import DwConvFn as _DwConvFn
from .dwconv import UpAddDwConvFn as _UpAddDwConvFn
from .layers import ConvNorm


class LinearAttention(nn.Module):
    """LinearAttention1 (variant 1, model/recattn.py:8-28) and LinearAttention2 (variant 2, :31-51).

    The two are the same function (the reference asserts it, lsnet/model/recattn.py:481-501); variant 2
    materialises the n x n map and is what stage >= 3 uses.
    """

    def __init__(self, dim, num_heads, variant=1):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.variant = variant
        self.qk = ConvNorm(dim, dim * 2, kernel_size=1, groups=2)
        self.pe = ConvNorm(dim, dim, kernel_size=3, padding=1, groups=dim)

    def forward(self, x):
        """The training-step form (RecAttn2d.forward's autograd branch calls it; the inference form is fused into RecAttn2d.forward): the
        projection as two GEMMs + the module's BatchNorm, then everything after it -- activation, k v^T, normaliser, + pe -- in one HIP
        kernel with a HIP backward (rcx_linear_attention_fwd / _bwd).  Like RecConv2d there is no PyTorch-operator path: a CPU tensor or a
        head size the HIP core does not take raises (the reference's own formulation lives in oracle/torch_eager.py, for the tests)."""
        b, c, h, w = x.shape
        n = h * w
        if not x.is_cuda:
            raise RuntimeError("recnext_amd.LinearAttention runs on the GPU only (HIP kernels); the CPU formulation is oracle/torch_eager.py")
        if c % 4 or not head_dim_supported(self.head_dim) or x.dtype not in ops._DT:
            raise NotImplementedError(f"LinearAttention: the HIP core takes head sizes up to 32, or multiples of 4 up to 64, and channel counts that "
                                      f"are multiples of 4; got dim {c}, {self.num_heads} heads, {x.dtype}")
        # Under autocast the GEMMs answer in the autocast type while x stays float32: the core takes one type, x's
        qkpre = self._qk_gpu(x).to(x.dtype)                 # (b, 2c, h, w), channels_last storage
        tok = qkpre.permute(0, 2, 3, 1).reshape(b, n, 2 * c)
        qpre, kpre = tok[..., :c].contiguous(), tok[..., c:].contiguous()
        pe = _conv_norm_train(self.pe, x, 1).to(x.dtype)
        return ops.LinearAttentionCoreFn.apply(qpre, kpre, x.contiguous(memory_format=torch.channels_last), pe, self.num_heads)

    def _qk_gpu(self, x):
        """The grouped 1x1 `qk` conv as two GEMMs on the token-major view (same function; the GEMM library's forward and
        backward are far faster than the grouped-conv path), then the module's own BatchNorm if it has not been fused."""
        b, c, h, w = x.shape
        m = self.qk
        conv = m if isinstance(m, nn.Conv2d) else m.conv
        tok = x.permute(0, 2, 3, 1).reshape(b * h * w, c)
        wq, wk = conv.weight[:c, :, 0, 0], conv.weight[c:, :, 0, 0]
        bq = None if conv.bias is None else conv.bias[:c]
        bk = None if conv.bias is None else conv.bias[c:]
        y = torch.cat((F.linear(tok[:, :c // 2], wq, bq), F.linear(tok[:, c // 2:], wk, bk)), dim=1)
        y = y.view(b, h, w, 2 * c).permute(0, 3, 1, 2)
        return y if isinstance(m, nn.Conv2d) else m.norm(y)


def head_dim_supported(d):
    """The HIP core's own rule for the head size (rcx_linear_attention_fwd): any size up to 32, or a multiple of 4 up to 64."""
    return d <= 32 or (d <= 64 and d % 4 == 0)


def _conv_norm_train(m, x, stride):
    """ConvNorm in a training step: HIP depthwise conv (+ autograd), then the module's own BatchNorm (batch statistics)."""
    if isinstance(m, nn.Conv2d):
        return _DwConvFn.apply(x, m.weight, m.bias, stride)
    return m.norm(_DwConvFn.apply(x, m.conv.weight, m.conv.bias, stride))


def _upadd_conv_norm_train(m, x, a, mode):
    """ConvNorm(x + interpolate(a, size(x), mode)) in a training step: resize, add and conv in one HIP launch with a HIP backward (rcx_upadd_dwconv_fwd / _bwd),
    then the module's own BatchNorm (batch statistics)."""
    if isinstance(m, nn.Conv2d):
        return _UpAddDwConvFn.apply(x, a, m.weight, m.bias, mode)
    return m.norm(_UpAddDwConvFn.apply(x, a, m.conv.weight, m.conv.bias, mode))


def _folded(m):
    """(weight, bias) of a ConvNorm in eval mode or of its fused nn.Conv2d."""
    if isinstance(m, nn.Conv2d):
        return m.weight, m.bias
    # in float32 whatever the parameters' type: a module in bfloat16 would otherwise round the scale, the product and the shift again
    # (measured on RecNeXt-A3's mixers, round 3: 1.4 x the mean error of the reference's own bfloat16 run; the reference folds in float32,
    # utils.py:227-234, before any cast)
    conv, norm = m.conv, m.norm
    s = norm.weight.float() / torch.sqrt(norm.running_var.float() + norm.eps)
    b = norm.bias.float() - s * norm.running_mean.float()
    if conv.bias is not None:
        b = b + s * conv.bias.float()
    return conv.weight.float() * s[:, None, None, None], b


class RecAttn2d(nn.Module):
    def __init__(self, dim, num_heads, kernel_size=5, stage=1, mode="nearest"):
        super().__init__()
        self.mode = mode
        self.kernel_size = kernel_size
        variant = 2 if stage >= 3 else 1                      # model/recattn.py:59
        self.down = nn.Sequential(
            ConvNorm(dim, dim, kernel_size=kernel_size, padding=kernel_size // 2, stride=2, groups=dim),
            LinearAttention(dim=dim, num_heads=num_heads, variant=variant),
        )
        self.conv = ConvNorm(dim, dim, kernel_size=kernel_size, padding=kernel_size // 2, groups=dim)
        self._pack_key = None
        self._pack = None

    def _tensors(self):
        # read out of the registries: this runs on every forward, and nn.Module.__getattr__ costs ~0.3 us a look-up (40 of them here)
        try:
            mods = self._modules
            down = mods["down"]._modules
            la = down["1"]._modules
            out = []
            for m in (down["0"], mods["conv"], la["qk"], la["pe"]):
                if isinstance(m, nn.Conv2d):
                    out += [m._parameters["weight"], m._parameters["bias"]]
                else:
                    cv, bn = m._modules["conv"], m._modules["norm"]
                    out += [cv._parameters["weight"], bn._parameters["weight"], bn._parameters["bias"], bn._buffers["running_mean"], bn._buffers["running_var"]]
        except KeyError:             # a parametrized weight (torch.nn.utils.parametrize, weight_norm) or a wrapped ConvNorm: the attribute path
            out = []
            for m in (self.down[0], self.conv, self.down[1].qk, self.down[1].pe):
                if isinstance(m, nn.Conv2d):
                    out += [m.weight, m.bias]
                else:
                    out += [m.conv.weight, m.norm.weight, m.norm.bias, m.norm.running_mean, m.norm.running_var]
        return [t for t in out if t is not None]

    def packed_params(self):
        key = tuple((t.data_ptr(), t._version, t.dtype, t.device) for t in self._tensors())
        if key != self._pack_key:
            with torch.no_grad():
                wd, bd = _folded(self.down[0])
                wc, bc = _folded(self.conv)
                la = self.down[1]
                wqk, bqk = _folded(la.qk)                       # (2C, C/2, 1, 1): rows [0,C) = q from channels [0,C/2), rows [C,2C) = k
                wpe, bpe = _folded(la.pe)
                c = wqk.shape[0] // 2
                zeros = lambda b_, n_: torch.zeros(n_, device=wd.device) if b_ is None else b_.float()
                self._pack = (ops.pack_dw_weight(wd.float()), None if bd is None else ops.pack_bias(bd.float()),
                              ops.pack_dw_weight(wc.float()), None if bc is None else ops.pack_bias(bc.float()),
                              wqk[:c, :, 0, 0].float().contiguous(), zeros(bqk, 2 * c)[:c].contiguous(),        # float32 GEMM operands: the coarse chain is float32
                              wqk[c:, :, 0, 0].float().contiguous(), zeros(bqk, 2 * c)[c:].contiguous(),
                              ops.pack_dw_weight(wpe.float()), None if bpe is None else ops.pack_bias(bpe.float()),
                              wqk[:, :, 0, 0].float().to(torch.bfloat16).contiguous(), zeros(bqk, 2 * c).contiguous())     # the one-launch form's operands
            self._pack_key = key
        return self._pack

    def forward(self, x):
        if self.training or (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))):
            # training step (engine.py:48-64): BatchNorm uses batch statistics, so nothing is folded; the depthwise convs, the
            # linear-attention core and their gradients run on HIP, the qk GEMMs and the norms on PyTorch-ROCm operators (autograd)
            if x.shape[1] % 4:
                raise NotImplementedError("the HIP depthwise backward needs a channel count that is a multiple of 4")
            d = _conv_norm_train(self.down[0], x, 2)
            a = self.down[1](d)
            return _upadd_conv_norm_train(self.conv, x, a, self.mode)           # conv(x + resize(a)), :67: one HIP launch each way
        wd, bd, wc, bc, wq, bq, wk, bk, wpe, bpe, wqk16, bqk = self.packed_params()
        k = self.kernel_size
        la = self.down[1]
        if x.shape[1] % la.num_heads or not head_dim_supported(x.shape[1] // la.num_heads):
            raise NotImplementedError(f"RecAttn2d: the HIP attention core takes head sizes up to 32, or multiples of 4 up to 64; got dim {x.shape[1]}, {la.num_heads} heads")
        if k == 5 and ops.recattn2d_supported(x.shape[1], la.num_heads, x.shape[2], x.shape[3], self.mode, x.dtype):
            # the 14 x 14 (and, up to 8 heads, 7 x 7) stage of a 16-bit run: the WHOLE unit in one launch (rcx_recattn2d_fwd, round 4), :61-67
            return ops.recattn2d(x, wd, bd, wqk16, bqk, wpe, bpe, wc, bc, la.num_heads, self.mode)
        if k == 5 and ops.recattn_down_qkcore_supported(x.shape[1], la.num_heads, x.shape[2], x.shape[3], x.dtype):
            # the 14 x 14 / 7 x 7 stages of a 16-bit run: the stride-2 conv, the projection, the core and pe in ONE launch (d stays in LDS), :61-66
            a = ops.recattn_down_qkcore(x, wd, bd, wqk16, bqk, wpe, bpe, la.num_heads)
            return ops.upadd_dwconv(x, a, wc, bc, k=k, mode=self.mode)              # conv(x + resize(.)), :67
        # the coarse chain in float32 (quarter-size tensors: ~1/4 of x's bytes per tensor even at twice the element size)
        d = ops.dwconv2d(x, wd, bd, k=k, stride=2, out_dtype=torch.float32)        # ConvNorm(dw k5 s2), :61
        b, c, h, w = d.shape
        if x.dtype != torch.float32 and ops.recattn_qkcore_supported(c, la.num_heads, h, w):
            # 16-bit activations, short sequences (the 14 x 14 and 7 x 7 stages): projection + core + pe in ONE launch, one wave per (image, head),
            # bf16 operands on the matrix cores with float32 accumulation (rcx_recattn_qkcore_fwd, round 4); :21-27 / :44-50
            a = ops.recattn_qkcore(d, wqk16, bqk, wpe, bpe, la.num_heads)
            return ops.upadd_dwconv(x, a, wc, bc, k=k, mode=self.mode)              # conv(x + resize(.)), :67
        tok = d.permute(0, 2, 3, 1).reshape(b * h * w, c)                           # NHWC storage viewed token-major, no copy
        qpre = F.linear(tok[:, :c // 2], wq, bq).view(b, h * w, c)                  # grouped 1x1 conv = two GEMMs, :21 / :44
        kpre = F.linear(tok[:, c // 2:], wk, bk).view(b, h * w, c)
        a = None
        if ops.linear_attention_core_fuses_pe(c, la.num_heads):                      # pe = ConvNorm(dw 3x3)(d) inside the core kernel (round 3)
            a = ops.linear_attention_core_pe(qpre, kpre, d, wpe, bpe, la.num_heads)  # :22-27 / :45-50
        if a is None:
            pe = ops.dwconv2d(d, wpe, bpe, k=3, stride=1)                           # ConvNorm(dw 3x3), :27 / :50
            a = ops.linear_attention_core(qpre, kpre, d, pe, la.num_heads)          # :22-27 / :45-50
        return ops.upadd_dwconv(x, a, wc, bc, k=k, mode=self.mode)                  # conv(x + resize(.)), :67
