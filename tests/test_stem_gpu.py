"""The fused stem (rcx_stem_fwd: conv3x3 s2 + bias -> exact GELU -> conv3x3 s2 + bias in one launch, the intermediate only in LDS) against the float64 chain on
the same bf16 operands with the intermediate rounded to bf16 (what the library path and the kernel both do), and against the library path itself.
Reference: model/recnext.py:134-146."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _reference(x, w1, b1, w2, b2):
    x64, w164, w264 = x.double().cpu(), w1.double().cpu(), w2.double().cpu()
    h = F.gelu(F.conv2d(x64, w164, b1.double().cpu(), stride=2, padding=1))
    h = h.to(torch.bfloat16).double()                      # the intermediate is a bf16 tensor in the library path, a bf16 LDS tile in the kernel
    return F.conv2d(h, w264, b2.double().cpu(), stride=2, padding=1)


@pytest.mark.parametrize("case", [(2, 32, 64, 224, 224), (1, 24, 48, 224, 224), (2, 20, 40, 64, 96), (1, 28, 56, 37, 51), (1, 40, 80, 128, 128), (3, 32, 64, 8, 8),
                                  (1, 32, 64, 1, 1), (1, 32, 64, 17, 16), (2, 32, 64, 512, 512)], ids=lambda c: "x".join(map(str, c)))
def test_fused_stem_against_float64_and_the_conv_library(case):
    from recnext_amd import ops
    n, cm, co, h, w = case
    g = torch.Generator(device="cpu").manual_seed(cm * 100 + h)
    rb = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(dev())
    x = rb(n, 3, h, w).contiguous(memory_format=torch.channels_last)
    w1, b1 = rb(cm, 3, 3, 3, sc=(2.0 / 27) ** 0.5), rb(cm, sc=0.3)
    w2, b2 = rb(co, cm, 3, 3, sc=(2.0 / (9 * cm)) ** 0.5), rb(co, sc=0.3)
    assert ops.stem_supported(n, h, w, cm, co, torch.bfloat16)
    pack = ops.pack_stem(w1, b1, w2, b2)
    y = ops.stem(x, *pack, cm, co)
    h2, w2_ = -(-(-(-h // 2)) // 2), -(-(-(-w // 2)) // 2)
    assert tuple(y.shape) == (n, co, h2, w2_) and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y, ops.stem(x, *pack, cm, co)), "not deterministic"
    ref = _reference(x, w1, b1, w2, b2)
    err = (y.double().cpu() - ref).abs()
    tol = 1e-2 + 1e-2 * ref.abs()                         # north_star's bf16 bar
    lib = F.conv2d(F.gelu(F.conv2d(x, w1, b1, stride=2, padding=1)), w2, b2, stride=2, padding=1)
    lib_err = (lib.double().cpu() - ref).abs()
    print(f"\n{case}: worst err / tol {float((err / tol).max()):.3f}; mean |err| fused {float(err.mean()):.2e} / library {float(lib_err.mean()):.2e}")
    assert bool((err <= tol).all())
    assert float(err.mean()) <= 1.05 * float(lib_err.mean()) + 1e-5


def test_fused_stem_rejects_what_it_has_no_kernel_for():
    from recnext_amd import ops
    assert not ops.stem_supported(1, 224, 224, 32, 64, torch.float32) and not ops.stem_supported(1, 224, 224, 30, 60, torch.bfloat16)
    assert not ops.stem_supported(1, 224, 224, 32, 128, torch.bfloat16)


@pytest.mark.parametrize("name", ["recnext_m3", "recnext_a0", "recnext_m5"])
def test_model_with_fused_stem_matches_the_conv_library(name):
    from recnext_amd.speed import build_inference_model, synthetic_batch
    a = build_inference_model(name, dev(), torch.bfloat16, seed=0, fused_stem=False)
    b = build_inference_model(name, dev(), torch.bfloat16, seed=0, fused_stem=True)
    assert list(a.state_dict()) == list(b.state_dict())
    assert b.stem.__dict__.get("_fused_stem") is not None and a.stem.__dict__.get("_fused_stem") is None
    x = synthetic_batch(4, 224, dev(), torch.bfloat16, seed=1)
    assert b.stem._fused_stem.supported(x)
    with torch.no_grad():
        sa, sb = a.stem(x).float(), b.stem(x).float()
        ya, yb = a(x).float(), b(x).float()
    assert float((sa - sb).abs().max()) < 0.02 * float(sa.abs().max()) + 0.02
    scale = float(ya.abs().max())
    assert float((ya - yb).abs().max()) < 0.05 * scale + 0.02, (float((ya - yb).abs().max()), scale)
