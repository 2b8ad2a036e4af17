"""The fused channel mixer (rcx_channel_mlp_fwd: x + W2 gelu(W1 z + b1) + b2 in one launch) against the float64 formula on the same bf16 inputs and
weights, and against the library path it replaces (two GEMMs + GELU + add in bf16).  Reference: model/recnext.py:125-132, :157-158, :169-171."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _reference(z, x, w1, b1, w2, b2):
    """float64 on the CPU: the operands as the kernel sees them (bf16 values), exact erf GELU, no intermediate rounding."""
    z64, x64 = z.double().cpu(), x.double().cpu()
    n, c, h, w = z64.shape
    zz = z64.permute(0, 2, 3, 1).reshape(-1, c)
    hid = zz @ w1.double().cpu().t() + b1.double().cpu()
    hid = 0.5 * hid * (1.0 + torch.erf(hid / math.sqrt(2.0)))
    out = hid @ w2.double().cpu().t() + b2.double().cpu()
    return x64 + out.reshape(n, h, w, c).permute(0, 3, 1, 2)


@pytest.mark.parametrize("case", [(2, 64, 128, 56, 56), (3, 128, 256, 28, 28), (2, 64, 120, 9, 11), (1, 48, 96, 56, 56), (2, 40, 80, 13, 7), (2, 56, 112, 28, 28),
                                  (2, 96, 192, 28, 28), (1, 80, 160, 56, 56), (1, 80, 150, 5, 5), (1, 64, 128, 1, 1), (1, 128, 240, 3, 33),
                                  (3, 256, 512, 14, 14), (2, 256, 480, 14, 14), (1, 256, 512, 3, 5), (5, 256, 512, 16, 16), (3, 192, 384, 14, 14), (2, 128, 256, 64, 64), (2, 160, 320, 28, 28), (1, 160, 300, 9, 7), (3, 320, 640, 14, 14), (1, 320, 600, 5, 3), (9, 320, 640, 7, 9)],
                         ids=lambda c: "x".join(map(str, c)))
def test_fused_channel_mlp_against_float64_and_the_gemm_path(case):
    from recnext_amd import ops
    n, c, hid, h, w = case
    g = torch.Generator(device="cpu").manual_seed(c * 1000 + hid + h)
    rb = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16)
    z = rb(n, c, h, w).to(dev()).contiguous(memory_format=torch.channels_last)
    x = rb(n, c, h, w).to(dev()).contiguous(memory_format=torch.channels_last)
    w1, b1 = rb(hid, c, sc=(2.0 / c) ** 0.5).to(dev()), rb(hid, sc=0.3).to(dev())
    w2, b2 = rb(c, hid, sc=(1.0 / hid) ** 0.5).to(dev()), rb(c, sc=0.3).to(dev())
    hp = ops.channel_mlp_hidden(n * h * w, c, hid, torch.bfloat16)
    assert hp >= hid and hp % 32 == 0 and hp - hid < 64
    wfrag, bias, hp2 = ops.pack_channel_mlp(w1, b1, w2, b2, hidden_to=hp)
    assert hp2 == hp
    y = ops.channel_mlp(z, x, wfrag, bias, hp)
    assert y.shape == x.shape and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y, ops.channel_mlp(z, x, wfrag, bias, hp)), "not deterministic"
    ref = _reference(z, x, w1, b1, w2, b2)
    err = (y.double().cpu() - ref).abs()
    tol = 1e-2 + 1e-2 * ref.abs()                                    # north_star's bf16 bar
    print(f"\n{case}: worst err / tol {float((err / tol).max()):.3f}, max |ref| {float(ref.abs().max()):.2f}")
    assert bool((err <= tol).all())
    # the path it replaces, all in bf16 (four launches; it rounds the hidden layer twice and the output twice): the fused kernel, which rounds the hidden
    # layer once (after the GELU) and the output once, is at least as close to the float64 result
    zz = z.permute(0, 2, 3, 1).reshape(-1, c)
    lib = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(zz, w1, b1)), w2, b2)
    lib = x + lib.view(n, h, w, c).permute(0, 3, 1, 2)
    lib_err = (lib.double().cpu() - ref).abs()
    print(f"    mean |err| fused {float(err.mean()):.2e} / library {float(lib_err.mean()):.2e}; max {float(err.max()):.2e} / {float(lib_err.max()):.2e}")
    assert float(err.mean()) <= 1.05 * float(lib_err.mean()) + 1e-5 and float(err.max()) <= 1.25 * float(lib_err.max()) + 1e-3


@pytest.mark.parametrize("case", [(2, 64, 128, 28, 28), (2, 256, 512, 14, 14)], ids=lambda c: "x".join(map(str, c)))
def test_fused_channel_mlp_gelu_tails_are_exact(case):
    """Hidden pre-activations of magnitude 1e2 .. 1e3: past its clamp the kernel's erf is exactly +-1 (rcx_gelu.h), so gelu(v) is exactly v or 0 and
    the error does not grow with |v|.  W2 picks out single hidden units, so the output shows gelu of one pre-activation each."""
    from recnext_amd import ops
    n, c, hid, h, w = case
    g = torch.Generator(device="cpu").manual_seed(7)
    z = (torch.randn(n, c, h, w, generator=g)).to(torch.bfloat16).to(dev()).contiguous(memory_format=torch.channels_last)
    x = torch.zeros_like(z)
    w1 = (torch.randn(hid, c, generator=g) * 40.0).to(torch.bfloat16).to(dev())          # pre-activations ~ N(0, (40 sqrt(c))^2): |v| up to ~1e3
    b1 = torch.zeros(hid).to(torch.bfloat16).to(dev())
    w2 = torch.zeros(c, hid)
    w2[torch.arange(c), torch.arange(c)] = 1.0                                            # y[:, j] = gelu(hidden unit j)
    w2, b2 = w2.to(torch.bfloat16).to(dev()), torch.zeros(c).to(torch.bfloat16).to(dev())
    hp = ops.channel_mlp_hidden(n * h * w, c, hid, torch.bfloat16)
    wfrag, bias, _ = ops.pack_channel_mlp(w1, b1, w2, b2, hidden_to=hp)
    y = ops.channel_mlp(z, x, wfrag, bias, hp).float()
    pre = (z.float().permute(0, 2, 3, 1).reshape(-1, c) @ w1.float().t())[:, :c].reshape(n, h, w, c).permute(0, 3, 1, 2)
    assert float(pre.abs().max()) > 300.0
    neg = pre < -6.0
    assert bool(neg.any()) and float(y[neg].abs().max()) == 0.0, float(y[neg].abs().max())        # exactly zero, not -1.7e-6 |v|
    pos = pre > 6.0
    ref = pre.double().cpu()
    assert bool(((y.double().cpu() - ref)[pos.cpu()].abs() <= 8e-3 * ref[pos.cpu()].abs() + 1e-6).all())          # gelu(v) = v there: only the bf16 roundings (hidden layer, output) remain


def test_fused_channel_mlp_rejects_what_it_has_no_kernel_for():
    from recnext_amd import _lib, ops
    assert not ops.channel_mlp_supported(1024, 512, 1024, torch.bfloat16)         # the 7 x 7 stage: the GEMM library
    assert not ops.channel_mlp_supported(1024, 64, 128, torch.float32) and not ops.channel_mlp_supported(1024, 64, 128, torch.float16)
    assert not ops.channel_mlp_supported(1024, 60, 128, torch.bfloat16)           # C % 8
    z = torch.zeros(1, 512, 4, 4, device=dev(), dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wfrag = torch.zeros(_lib.load().rcx_channel_mlp_pack_bytes(512, 1024) // 2, device=dev(), dtype=torch.bfloat16)
    bias = torch.zeros(32 * (32 + 16), device=dev())
    with pytest.raises(_lib.RcxError, match="no kernel"):
        ops.channel_mlp(z, z.clone(), wfrag, bias, 1024)


@pytest.mark.parametrize("name", ["recnext_m3", "recnext_a3", "recnext_m1", "recnext_m5"])
def test_model_with_fused_mlp_matches_the_gemm_path(name):
    """build_inference_model(fused_mlp=True) against the same weights on the GEMM-library path: the logits agree within bf16 noise, the state_dict is the same,
    and the fused path is what ran (the stage-0 / stage-1 blocks)."""
    from recnext_amd import models
    from recnext_amd.speed import build_inference_model, synthetic_batch
    a = build_inference_model(name, dev(), torch.bfloat16, seed=0, fused_mlp=False)
    b = build_inference_model(name, dev(), torch.bfloat16, seed=0, fused_mlp=True)
    assert list(a.state_dict()) == list(b.state_dict())
    fused = [m for m in b.modules() if m.__dict__.get("_fused_mlp") is not None]
    assert len(fused) == sum(models.CONFIGS[name]["depth"]) + 3
    x = synthetic_batch(4, 224, dev(), torch.bfloat16, seed=1)
    blk = b.stages[0].blocks[0]
    assert blk._fused_mlp.supported(torch.empty(4, blk.channel_mixer[0].in_channels, 56, 56, device=dev(), dtype=torch.bfloat16))
    with torch.no_grad():
        ya, yb = a(x).float(), b(x).float()
    scale = float(ya.abs().max())
    assert float((ya - yb).abs().max()) < 0.05 * scale + 0.02, (float((ya - yb).abs().max()), scale)
