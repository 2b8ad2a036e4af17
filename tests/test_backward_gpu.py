"""GPU: the HIP backward of RecConv2d against autograd through the ATen restatement (the reference's own operators).

Bar: float32 gradients agree to 1e-4 relative to the gradient's max-abs (both sides accumulate in float32 in different
orders); bfloat16 inputs: the HIP path keeps float32 intermediates, compared against the float32 autograd on the
bf16-rounded input at 1e-2.
"""
import os

import numpy as np
import pytest
import torch

import recnext_amd
from oracle.torch_eager import EagerRecConv2d

pytestmark = pytest.mark.gpu

CASES = [
    # n, c, h, w, level, mode, bias
    (2, 8, 7, 7, 1, "bilinear", False),
    (2, 8, 14, 14, 2, "bilinear", True),
    (1, 16, 28, 28, 3, "bilinear", False),
    (1, 8, 25, 13, 2, "bilinear", True),
    (2, 12, 9, 9, 0, "bilinear", True),
    (1, 8, 14, 14, 2, "nearest", False),
    (1, 8, 16, 16, 1, "nearest", True),
    (1, 4, 5, 3, 3, "bilinear", True),
    (2, 64, 56, 56, 4, "bilinear", False),
]


def _pair(c, level, mode, bias, dev):
    torch.manual_seed(5)
    ref = EagerRecConv2d(c, 5, bias, level, mode).to(dev)
    ours = recnext_amd.RecConv2d(c, 5, bias, level, mode).to(dev)
    ours.load_state_dict(ref.state_dict(), strict=True)
    return ref, ours


# the shapes at which rcx_recconv2d_bwd switches kernels: the one-launch backward with its two-wave split (< 512 planes per
# wave set: batch 128 x 256 channels), the nested launch inside the deeper blocks, the tiled weight gradients at batch 128 x 256 channels
# ... and (round 6) the tiled backward of the fine levels past 512 partial rows per weight-gradient launch (N * H / 14: its buffers are sized per launch)
DISPATCH_CASES = [(128, 256, 14, 2), (128, 512, 7, 1), (128, 128, 28, 3), (64, 64, 56, 4), (160, 64, 56, 4), (288, 128, 28, 3)]


@pytest.mark.parametrize("case", DISPATCH_CASES, ids=lambda c: "x".join(map(str, c)))
def test_fp32_gradients_at_dispatch_sizes_match_aten_autograd(case):
    """Full training-batch sizes: input, weight and bias gradients of the HIP backward against autograd through the reference's
    operator chain (oracle/torch_eager.py on the same device), float32 -- not just HIP kernels against each other."""
    n, c, h, level = case
    dev = torch.device("cuda:0")
    ref, ours = _pair(c, level, "bilinear", True, dev)
    torch.manual_seed(11)
    x = torch.randn(n, c, h, h, device=dev)
    gy = torch.randn(n, c, h, h, device=dev)
    xr = x.clone().requires_grad_(True)
    xo = x.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yr = ref(xr)
    yo = ours(xo)
    assert _rel(yo, yr.detach()) < 1e-5
    yr.backward(gy)
    yo.backward(gy)
    assert _rel(xo.grad, xr.grad) < 1e-4
    for (name, pr), (_, po) in zip(ref.named_parameters(), ours.named_parameters()):
        assert po.grad is not None and po.grad.shape == pr.shape, name
        assert _rel(po.grad, pr.grad) < 2e-4, (name, _rel(po.grad, pr.grad))      # sums over 128 x 196 .. 3136 products, two summation orders


def _rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_fp32_gradients_match_aten_autograd(case):
    n, c, h, w, level, mode, bias = case
    dev = torch.device("cuda:0")
    ref, ours = _pair(c, level, mode, bias, dev)
    x = torch.randn(n, c, h, w, device=dev)
    gy = torch.randn(n, c, h, w, device=dev)
    xr = x.clone().requires_grad_(True)
    xo = x.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yr = ref(xr)
    yo = ours(xo)
    assert _rel(yo, yr.detach()) < 1e-5
    yr.backward(gy)
    yo.backward(gy)
    assert _rel(xo.grad, xr.grad) < 1e-4
    for (name, pr), (_, po) in zip(ref.named_parameters(), ours.named_parameters()):
        assert po.grad is not None and po.grad.shape == pr.shape, name
        if pr.grad is None:                                         # level 0: the ladder parameters are unused
            assert name.startswith("down.") and float(po.grad.abs().max()) == 0.0
        else:
            assert _rel(po.grad, pr.grad) < 1e-4, name


@pytest.mark.parametrize("name", __import__("tests.util", fromlist=["grad_cases"]).grad_cases())
def test_fp32_gradients_match_pinned_reference_fixtures(name):
    """HIP backward against gradients produced by the imported reference block (tests/golden/grad_recconv_*.npz)."""
    from tests.util import load_grad
    d, m = load_grad(name)
    dev = torch.device("cuda:0")
    mod = recnext_amd.RecConv2d(m["C"], m["k"], m["bias"], m["level"], m["mode"])
    sd = {"down.weight": d["w_down"], **{f"convs.{i}.weight": w for i, w in enumerate(d["w_convs"])}}
    if m["bias"]:
        sd.update({"down.bias": d["b_down"], **{f"convs.{i}.bias": b for i, b in enumerate(d["b_convs"])}})
    mod.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    mod = mod.to(dev).train()
    x = torch.from_numpy(d["x"]).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = mod(x)
    assert float((y.detach().cpu() - torch.from_numpy(d["y"])).abs().max()) < 1e-4
    y.backward(torch.from_numpy(d["gy"]).to(dev))
    rel = lambda a, b: float(np.abs(a.detach().cpu().numpy() - b).max() / (np.abs(b).max() + 1e-12))
    assert rel(x.grad, d["gx"]) < 1e-4
    assert rel(mod.down.weight.grad, d["gw_down"]) < 1e-4
    for j, cv in enumerate(mod.convs):
        assert rel(cv.weight.grad, d["gw_convs"][j]) < 1e-4, j
    if m["bias"]:
        assert rel(mod.down.bias.grad, d["gb_down"]) < 1e-4
        for j, cv in enumerate(mod.convs):
            assert rel(cv.bias.grad, d["gb_convs"][j]) < 1e-4, j


def test_bf16_input_gradients():
    dev = torch.device("cuda:0")
    ref, ours = _pair(16, 2, "bilinear", False, dev)
    x = torch.randn(2, 16, 14, 14, device=dev).bfloat16()
    gy = torch.randn(2, 16, 14, 14, device=dev).bfloat16()
    xr = x.float().requires_grad_(True)
    ref(xr).backward(gy.float())
    xo = x.clone().requires_grad_(True)
    yo = ours(xo)
    assert yo.dtype == torch.bfloat16
    yo.backward(gy)
    assert xo.grad.dtype == torch.bfloat16
    assert _rel(xo.grad.float(), xr.grad) < 1e-2
    for (name, pr), (_, po) in zip(ref.named_parameters(), ours.named_parameters()):
        assert _rel(po.grad, pr.grad) < 1e-3, name               # parameter grads are float32 on both sides


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", [(3, 64, 56, 4), (2, 128, 28, 3), (2, 64, 14, 2), (2, 64, 7, 1), (2, 16, 9, 2)], ids=lambda c: "x".join(map(str, c)))
def test_16bit_module_gets_16bit_parameter_gradients(case, dtype):
    """A module cast to a 16-bit type (`.to(torch.bfloat16)`): the backward's final reduction writes the parameters' gradients in the parameters' own (C,1,k,k) layout
    and dtype (rcx_recconv2d_bwd gw_out / gb_out; no unpack launch, no copies), and dL/dy is read in bfloat16 where the library takes it.  Against float32 autograd
    through the ATen chain on the same (rounded) values."""
    n, c, hw, level = case
    dev = torch.device("cuda:0")
    ref, ours = _pair(c, level, "bilinear", True, dev)
    ours = ours.to(dtype)
    ref.load_state_dict({k: v.float() for k, v in ours.state_dict().items()})          # the rounded parameters
    torch.manual_seed(21)
    x = torch.randn(n, c, hw, hw, device=dev).to(dtype)
    gy = torch.randn(n, c, hw, hw, device=dev).to(dtype)
    xr = x.float().clone().requires_grad_(True)
    ref(xr).backward(gy.float())
    xo = x.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ours(xo).backward(gy.contiguous(memory_format=torch.channels_last))
    assert xo.grad.dtype == dtype and _rel(xo.grad.float(), xr.grad) < 1e-2
    for (name, pr), (_, po) in zip(ref.named_parameters(), ours.named_parameters()):
        assert po.grad is not None and po.grad.dtype == dtype and po.grad.shape == pr.shape and po.grad.stride() == po.stride(), name
        assert _rel(po.grad.float(), pr.grad) < 1e-2, (name, _rel(po.grad.float(), pr.grad))


def test_backward_is_deterministic_and_supports_a_training_step():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    mod = recnext_amd.RecConv2d(32, level=3).to(dev)
    x = torch.randn(4, 32, 28, 28, device=dev)
    tgt = torch.randn(4, 32, 28, 28, device=dev)
    grads = []
    for _ in range(2):
        mod.zero_grad(set_to_none=True)
        torch.nn.functional.mse_loss(mod(x), tgt).backward()
        grads.append([p.grad.clone() for p in mod.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*grads))
    opt = torch.optim.SGD(mod.parameters(), lr=0.5)
    with torch.no_grad():
        before = float(torch.nn.functional.mse_loss(mod(x), tgt))
    for _ in range(5):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(mod(x), tgt)
        loss.backward()
        opt.step()
    with torch.no_grad():
        after = float(torch.nn.functional.mse_loss(mod(x), tgt))     # the packs must have been rebuilt after each step
    assert after < before


def test_inference_and_training_forward_agree():
    dev = torch.device("cuda:0")
    mod = recnext_amd.RecConv2d(64, level=4).to(dev)
    x = torch.randn(2, 64, 56, 56, device=dev)
    with torch.no_grad():
        y_inf = mod(x)                                            # fused plane schedule
    y_trn = mod(x.clone().requires_grad_(True))                   # per-level schedule with saved pyramid
    assert _rel(y_trn.detach(), y_inf) < 1e-5


# ---- depthwise conv backward (rcx_dwconv2d_bwd) and RecAttn2d in a training step ----
@pytest.mark.parametrize("case", [(2, 16, 14, 14, 5, 1, True), (2, 16, 14, 14, 5, 2, False), (1, 64, 28, 28, 5, 2, True), (2, 64, 56, 56, 5, 1, False), (2, 64, 56, 56, 5, 2, True), (3, 40, 28, 28, 5, 2, True),
                                  (2, 8, 9, 12, 5, 2, True), (2, 12, 7, 7, 3, 1, True), (1, 8, 10, 10, 7, 2, False),
                                  (2, 40, 28, 28, 5, 1, True), (1, 64, 56, 56, 5, 2, False), (3, 128, 28, 28, 5, 1, False)],     # the tiled weight-gradient kernels
                         ids=lambda c: "x".join(map(str, c)))
def test_dwconv_backward_matches_aten_autograd(case):
    from recnext_amd.dwconv import DwConvFn as _DwConvFn
    n, c, h, w, k, stride, bias = case
    dev = torch.device("cuda:0")
    torch.manual_seed(c + h + k)
    conv = torch.nn.Conv2d(c, c, k, stride=stride, padding=k // 2, groups=c, bias=bias).to(dev)
    x = torch.randn(n, c, h, w, device=dev, requires_grad=True)
    g = torch.randn_like(conv(x))
    conv(x).backward(g)
    ref = [x.grad.clone(), conv.weight.grad.clone()] + ([conv.bias.grad.clone()] if bias else [])
    x.grad = None; conv.zero_grad()
    y = _DwConvFn.apply(x, conv.weight, conv.bias, stride)
    assert float((y - conv(x)).abs().max()) < 1e-4
    y.backward(g)
    got = [x.grad, conv.weight.grad] + ([conv.bias.grad] if bias else [])
    for a, b in zip(got, ref):
        assert _rel(a, b) < 1e-4


@pytest.mark.parametrize("stage,dim,hw", [(0, 16, 14), (1, 32, 14), (3, 64, 7)])
def test_recattn2d_training_step_matches_aten(stage, dim, hw):
    """Train-mode RecAttn2d (BatchNorm on batch statistics): HIP depthwise convs + autograd against the ATen operator chain."""
    from oracle.torch_eager import EagerRecAttn2d
    from recnext_amd.recattn import RecAttn2d
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    ref = EagerRecAttn2d(dim, num_heads=2 ** (stage + 1), stage=stage).to(dev).train()
    ours = RecAttn2d(dim, num_heads=2 ** (stage + 1), stage=stage).to(dev).train()
    ours.load_state_dict(ref.state_dict(), strict=True)
    x = torch.randn(4, dim, hw, hw, device=dev)
    xr = x.clone().requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    g = torch.randn_like(x)
    yr = ref(xr); yr.backward(g)
    yo = ours(xo); yo.backward(g)
    assert _rel(yo, yr) < 1e-4
    assert _rel(xo.grad, xr.grad) < 2e-3
    scale = max(float(p.grad.abs().max()) for p in ref.parameters())
    for (name, pr), (_, po) in zip(ref.named_parameters(), ours.named_parameters()):
        assert po.grad is not None, name
        assert float((po.grad - pr.grad).abs().max()) < 2e-3 * float(pr.grad.abs().max()) + 1e-5 * scale, name
    # running statistics moved identically
    for (name, br), (_, bo) in zip(ref.named_buffers(), ours.named_buffers()):
        assert torch.allclose(br.float(), bo.float(), atol=1e-5, rtol=1e-4), name


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("mode", ["nearest", "bilinear"])
@pytest.mark.parametrize("case", [(2, 64, 56, 56, 28, 28), (3, 128, 28, 28, 14, 14), (2, 40, 56, 56, 28, 28), (2, 16, 14, 14, 7, 7), (3, 32, 7, 7, 4, 4),
                                  (1, 8, 25, 13, 13, 7)], ids=lambda c: "x".join(map(str, c)))
def test_upadd_dwconv_backward_matches_aten_autograd(case, mode, dtype):
    """conv5(x + interpolate(a, size(x), mode)) -- RecAttn2d's last line (model/recattn.py:67) -- as recnext_amd.dwconv.UpAddDwConvFn (one HIP launch each
    way for resize + add + conv) against autograd through the ATen operators in float32: gx, ga, gw, gb.  56x56 / 28x28 with an exact 2x coarse plane
    take the tiled adjoint kernels (and a 16-bit gy as it is); the others the per-step kernels."""
    from recnext_amd.dwconv import UpAddDwConvFn
    n, c, h, w, hc, wc = case
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(c, c, 5, padding=2, groups=c, bias=True).to(dev)
    x = torch.randn(n, c, h, w, device=dev).to(dtype)
    a = torch.randn(n, c, hc, wc, device=dev).to(dtype)
    gy = torch.randn(n, c, h, w, device=dev).to(dtype)
    xr, ar = x.float().clone().requires_grad_(True), a.float().clone().requires_grad_(True)
    yr = conv(xr + torch.nn.functional.interpolate(ar, size=(h, w), mode=mode))
    yr.backward(gy.float())
    gwr, gbr = conv.weight.grad.clone(), conv.bias.grad.clone()
    conv.weight.grad = conv.bias.grad = None
    xo = x.detach().clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ao = a.detach().clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yo = UpAddDwConvFn.apply(xo, ao, conv.weight, conv.bias, mode)
    assert yo.dtype == dtype
    yo.backward(gy.contiguous(memory_format=torch.channels_last))
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    assert _rel(yo.float(), yr) < tol
    assert xo.grad.dtype == dtype and ao.grad.dtype == dtype
    assert _rel(xo.grad.float(), xr.grad) < tol
    assert _rel(ao.grad.float(), ar.grad) < tol
    assert _rel(conv.weight.grad, gwr) < 2e-4 and _rel(conv.bias.grad, gbr) < 2e-4      # float32 sums either way (inputs are the same rounded values)


@pytest.mark.parametrize("case", [(2, 16, 14, 14, 7), (1, 64, 56, 56, 7), (3, 6, 9, 12, 7), (2, 8, 10, 10, 5), (2, 8, 7, 7, 3),
                                  (2, 32, 28, 28, 7), (1, 24, 56, 56, 7), (2, 72, 28, 28, 7)],    # the tiled weight-gradient kernel: both plane sizes, ragged 64-channel blocks
                         ids=lambda c: "x".join(map(str, c)))
def test_downsample_conv_backward_matches_aten_autograd(case):
    """nn.Conv2d(C, 2C, k, stride=2, groups=C) + train-mode BatchNorm through DownsampleDwConv against ATen autograd."""
    from recnext_amd.dwconv import DownsampleDwConv
    n, c, h, w, k = case
    dev = torch.device("cuda:0")
    torch.manual_seed(c * 7 + h)
    conv = torch.nn.Conv2d(c, 2 * c, k, stride=2, padding=k // 2, groups=c).to(dev)
    bn = torch.nn.BatchNorm2d(2 * c).to(dev).train()
    import copy
    ours = DownsampleDwConv(copy.deepcopy(conv), copy.deepcopy(bn)).train()
    x = torch.randn(n, c, h, w, device=dev)
    xr, xo = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    yr = bn(conv(xr))
    g = torch.randn_like(yr)
    yr.backward(g)
    yo = ours(xo)
    yo.backward(g)
    assert _rel(yo, yr) < 1e-4
    assert _rel(xo.grad, xr.grad) < 1e-3
    assert _rel(ours.token_mixer.weight.grad, conv.weight.grad) < 1e-3
    assert _rel(ours.token_mixer.bias.grad + 1.0, conv.bias.grad + 1.0) < 1e-3      # analytically zero under train-mode BN
    assert _rel(ours.norm.weight.grad, bn.weight.grad) < 1e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "f16", "f32"])
@pytest.mark.parametrize("case", [(3, 64, 56), (2, 128, 28), (5, 256, 14), (2, 40, 56), (1, 104, 28)], ids=lambda c: "x".join(map(str, c)))
def test_downsample_input_gradient_tile_kernel(case, dtype):
    """The input gradient of nn.Conv2d(C, 2C, 7, stride 2, groups = C) on the 56 / 28 / 14 planes (k_bwd_down7m2: a lane per output channel, the lane pair's
    shares added by DPP) against ATen autograd in float32 on the same (rounded) x, for every I/O type and ragged 64-channel blocks."""
    from recnext_amd import ops
    n, c, h = case
    dev = torch.device("cuda:0")
    torch.manual_seed(c + h)
    conv = torch.nn.Conv2d(c, 2 * c, 7, stride=2, padding=3, groups=c, bias=False).to(dev)
    x = torch.randn(n, c, h, h, device=dev).to(dtype)
    gy = torch.randn(n, 2 * c, h // 2, h // 2, device=dev)
    xr = x.float().clone().requires_grad_(True)
    conv(xr).backward(gy)
    wp = ops.pack_dw_weight(conv.weight.detach().float())
    gx, gw, _ = ops.dwconv2d_mult2_backward(x.contiguous(memory_format=torch.channels_last), gy.contiguous(memory_format=torch.channels_last), wp, 7)
    assert gx.dtype == dtype and gx.shape == x.shape
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    assert _rel(gx.float(), xr.grad) < tol
    assert _rel(gw.view(7, 7, 2 * c).permute(2, 0, 1), conv.weight.grad[:, 0]) < 2e-4


def _la_reference(qpre, kpre, v, pe, heads):
    """model/recattn.py:21-27 on token-major (B, n, C) tensors in float32 (LinearAttention1; LinearAttention2 is the same function)."""
    b, n, c = qpre.shape
    d = c // heads
    q = (torch.nn.functional.elu(qpre) + 1.0).view(b, n, heads, d).permute(0, 2, 1, 3)     # b, h, n, d
    k = (torch.nn.functional.elu(kpre) + 1.0).view(b, n, heads, d).permute(0, 2, 1, 3)
    vv = v.view(b, n, heads, d).permute(0, 2, 1, 3)
    s = n ** -0.5
    kv = (k.transpose(-1, -2) * s) @ (vv * s)                                               # b, h, d, d
    out = q @ kv / (q @ k.mean(dim=2, keepdim=True).transpose(-1, -2) + 1e-6)
    return out.permute(0, 2, 1, 3).reshape(b, n, c) + pe


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", [(2, 16, 14, 14, 2), (3, 64, 28, 28, 2), (2, 128, 14, 14, 4), (2, 256, 7, 7, 8), (1, 512, 4, 4, 16), (2, 24, 5, 9, 2),
                                  (1, 128, 9, 9, 2)], ids=lambda c: "x".join(map(str, c)))
def test_linear_attention_core_backward(case, dtype):
    """rcx_linear_attention_bwd against PyTorch autograd of the reference formulation (float32, on the same rounded inputs):
    the A3 stage shapes (head dimension 32), a head dimension of 12 and one of 64."""
    from recnext_amd import ops
    b, c, h, w, heads = case
    dev = torch.device("cuda:0")
    torch.manual_seed(zlib_seed(case))
    n = h * w
    mk = lambda: (torch.randn(b, n, c, device=dev) * 0.7).to(dtype)
    qpre, kpre, gout = mk(), mk(), mk()
    v = torch.randn(b, c, h, w, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    pe = torch.randn(b, c, h, w, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    tokv = lambda t: t.permute(0, 2, 3, 1).reshape(b, n, c)
    ref_in = [t.float().clone().requires_grad_(True) for t in (qpre, kpre, tokv(v), tokv(pe))]
    _la_reference(*ref_in, heads).backward(gout.float())
    ours_in = [t.clone().requires_grad_(True) for t in (qpre, kpre, v, pe)]
    out = ops.LinearAttentionCoreFn.apply(*ours_in, heads)
    out.backward(gout.view(b, h, w, c).permute(0, 3, 1, 2))
    tol = 2e-4 if dtype == torch.float32 else (2e-2 if dtype == torch.bfloat16 else 4e-3)
    got = [ours_in[0].grad, ours_in[1].grad, tokv(ours_in[2].grad), tokv(ours_in[3].grad)]
    for name, g, r in zip(("qpre", "kpre", "v", "pe"), got, ref_in):
        assert g.dtype == dtype
        assert _rel(g.float(), r.grad) < tol, name
    # deterministic
    ours2 = [t.clone().requires_grad_(True) for t in (qpre, kpre, v, pe)]
    ops.LinearAttentionCoreFn.apply(*ours2, heads).backward(gout.view(b, h, w, c).permute(0, 3, 1, 2))
    assert torch.equal(ours2[0].grad, ours_in[0].grad) and torch.equal(ours2[2].grad, ours_in[2].grad)


def zlib_seed(obj):
    import zlib
    return zlib.crc32(repr(obj).encode())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 64, 56, 4), (3, 128, 28, 3), (2, 192, 28, 3), (2, 256, 14, 2), (3, 100, 14, 2), (2, 512, 7, 1)],
                         ids=lambda v: "x".join(map(str, v)))
def test_fused_training_forward_leaves_the_same_pyramid(shape, dtype, monkeypatch):
    """rcx_recconv2d_fwd_train on the channel-per-lane kernels (one launch that also writes F_l and C_l) against the per-step schedule
    (RCX_TRAIN_FUSED=0): same output, same saved float32 pyramid up to summation order -- the backward reads it either way."""
    from recnext_amd import ops
    n, c, hw, level = shape
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level, bias=True).to(dev)
    wpack, bpack = mod.packed_params()
    x = torch.randn(n, c, hw, hw, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    y1, s1 = ops.recconv2d_forward_train(x, wpack, bpack, level, 5, "bilinear")
    monkeypatch.setenv("RCX_TRAIN_FUSED", "0")
    y0, s0 = ops.recconv2d_forward_train(x, wpack, bpack, level, 5, "bilinear")
    assert s0.numel() == s1.numel()
    off, h = 0, hw                                            # the planes F_l, C_l (l = 1 .. level), each padded to 256 bytes
    for l in range(1, level + 1):
        h = (h + 1) // 2
        nb = 4 * n * c * h * h
        for name in ("F", "C"):
            a0, a1 = s0[off:off + nb].view(torch.float32), s1[off:off + nb].view(torch.float32)
            assert torch.isfinite(a1).all(), (name, l)
            assert float((a1 - a0).abs().max()) < 2e-5 * max(1.0, float(a0.abs().max())), (name, l)
            off += (nb + 255) // 256 * 256
    assert off == s0.numel()
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert float((y1.float() - y0.float()).abs().max()) < tol * max(1.0, float(y0.float().abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("mode", ["bilinear", "nearest"])
@pytest.mark.parametrize("shape", [(2, 512, 7, 1), (3, 40, 7, 1), (2, 64, 7, 1), (2, 256, 14, 2), (3, 100, 14, 2), (1, 192, 14, 2),
                                   (2, 128, 28, 3), (1, 96, 28, 3), (2, 64, 56, 4), (1, 40, 56, 4)],
                         ids=lambda s: "x".join(map(str, s)))
def test_fused_backward_matches_the_per_step_backward(shape, mode, dtype, monkeypatch):
    """rcx_recconv2d_bwd on the channel-per-lane backward kernels (rcx_cplbwd.hip: the block's whole adjoint in one launch, one
    partial row per image, then the batch reduction) against the per-step schedule (RCX_BWD_FUSED=0): same gx, gW, gb up to the
    float32 summation order (and, for 16-bit I/O, one rounding of gx).  The 28x28 / level 3 and 56x56 / level 4 blocks hand their
    14x14-and-below part to the same kernel (float32 in and out)."""
    from recnext_amd import ops
    n, c, hw, level = shape
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level, bias=True, mode=mode).to(dev)
    wpack, bpack = mod.packed_params()
    x = torch.randn(n, c, hw, hw, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(n, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    _, saved = ops.recconv2d_forward_train(x, wpack, bpack, level, 5, mode)
    assert "RCX_BWD_FUSED" not in os.environ and "RCX_WGRAD_CPL" not in os.environ
    gx1, gw1, gb1 = ops.recconv2d_backward(x, gy, wpack, saved, level, 5, mode, need_bias=True)
    monkeypatch.setenv("RCX_BWD_FUSED", "0")
    monkeypatch.setenv("RCX_WGRAD_CPL", "0")                  # and the tiled weight-gradient kernel of the 56x56 / 28x28 planes
    gx0, gw0, gb0 = ops.recconv2d_backward(x, gy, wpack, saved, level, 5, mode, need_bias=True)
    tol = 3e-5 if dtype == torch.float32 else 1e-2
    for name, a1, a0, t in (("gx", gx1.float(), gx0.float(), tol), ("gw", gw1, gw0, 3e-5), ("gb", gb1, gb0, 3e-5)):
        assert torch.isfinite(a1).all(), name
        assert float((a1 - a0).abs().max()) < t * max(1.0, float(a0.abs().max())), (name, float((a1 - a0).abs().max()), float(a0.abs().max()))


def test_adjoint_pieces_against_host_loops(tmp_path):
    """tools/cplbwd_probe.hip: the register-plane pieces of the fused backward (tap-pair weight gradients for both strides, the
    stride-2 adjoint, the resize adjoint) each against plain double-precision loops on the host, at 4x4, 7x7 and 14x14."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cplbwd_probe")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", os.path.join(root, "tools", "cplbwd_probe.hip"),
                    "-o", exe], check=True, timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-2000:] + out.stderr[-500:]
