"""CPU: the oracle (numpy, C and ATen restatements) against the golden vectors captured from the reference."""
import numpy as np
import pytest
import torch

from oracle import c_oracle, recconv_np, torch_eager
from tests.util import GOLDEN, grad_cases, load_grad, load_recattn, load_recconv, recattn_cases, recconv_cases

import os

TOL = 5e-6   # float32 round-off between summation orders on N(0,1) data


def _args(d, m):
    b = m["bias"]
    return (d["w_down"], list(d["w_convs"]), d["b_down"] if b else None, list(d["b_convs"]) if b else None)


@pytest.mark.parametrize("name", recconv_cases())
def test_numpy_oracle_matches_reference(name):
    d, m = load_recconv(name)
    if d["x"].size > 200_000:
        pytest.skip("numpy oracle kept to small cases; C oracle covers the large ones")
    wd, wc, bd, bc = _args(d, m)
    tr = recconv_np.recconv2d_trace(d["x"].astype(np.float64), wd, wc, bd, bc, m["level"], m["mode"])
    assert np.abs(tr["y"] - d["y"]).max() < TOL
    for l in range(1, m["level"] + 1):
        if f"F{l}" in d:
            assert np.abs(tr["F"][l] - d[f"F{l}"]).max() < TOL
            assert np.abs(tr["U"][l] - d[f"U{l}"]).max() < TOL


@pytest.mark.parametrize("name", recconv_cases())
def test_c_oracle_matches_reference(name):
    d, m = load_recconv(name)
    wd, wc, bd, bc = _args(d, m)
    y = c_oracle.recconv2d(d["x"], wd, wc, bd, bc, m["level"], m["mode"])
    assert np.abs(y - d["y"]).max() < TOL
    # bf16 target: same oracle on bf16-rounded inputs and weights
    from tests.util import bf16_round_np as r
    y2 = c_oracle.recconv2d(r(d["x"]), r(wd), [r(w) for w in wc], None if bd is None else r(bd),
                            None if bc is None else [r(b) for b in bc], m["level"], m["mode"])
    assert np.abs(y2 - d["y_bf16in_f32"]).max() < TOL


@pytest.mark.parametrize("name", recconv_cases())
def test_aten_restatement_matches_reference(name):
    d, m = load_recconv(name)
    wd, wc, bd, bc = _args(d, m)
    t = torch.from_numpy
    y = torch_eager.recconv2d_eager(t(d["x"]), t(wd), [t(w) for w in wc], None if bd is None else t(bd),
                                    None if bc is None else [t(b) for b in bc], m["mode"]).numpy()
    assert np.abs(y - d["y"]).max() < 1e-6
    mod = torch_eager.EagerRecConv2d(m["C"], m["k"], m["bias"], m["level"], m["mode"])
    sd = {"down.weight": t(wd), **{f"convs.{i}.weight": t(w) for i, w in enumerate(wc)}}
    if bd is not None:
        sd.update({"down.bias": t(bd), **{f"convs.{i}.bias": t(b) for i, b in enumerate(bc)}})
    mod.load_state_dict(sd, strict=True)          # the reference's exact key set
    with torch.no_grad():
        assert np.abs(mod(t(d["x"])).numpy() - d["y"]).max() < 1e-6


def test_c_oracle_pieces_match_numpy():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 6, 9, 11)).astype(np.float32)
    w = rng.standard_normal((6, 1, 5, 5)).astype(np.float32)
    b = rng.standard_normal(6).astype(np.float32)
    for stride in (1, 2):
        assert np.abs(c_oracle.dwconv2d(x, w, b, stride) - recconv_np.dwconv2d(x.astype(np.float64), w, b, stride)).max() < TOL
    small = rng.standard_normal((2, 6, 5, 6)).astype(np.float32)
    for mode in ("bilinear", "nearest"):
        ref = x + recconv_np.resize(small.astype(np.float64), (9, 11), mode)
        assert np.abs(c_oracle.add_resized(x, small, mode) - ref).max() < TOL


def test_interp_tables_match_aten():
    t = np.load(os.path.join(GOLDEN, "interp_tables.npz"))
    for key in t.files:
        kind, a, b = key.split("_")
        a, b = int(a), int(b)
        if kind == "bilinear":
            i0, i1, lam = recconv_np.bilinear_axis_table(a, b)
            mat = np.zeros((a, b), np.float32)
            for d in range(b):
                mat[i0[d], d] += 1 - lam[d]
                mat[i1[d], d] += lam[d]
            assert np.abs(mat - t[key]).max() < 1e-6, key
        else:
            assert (recconv_np.nearest_axis_table(a, b) == t[key]).all(), key


def test_ladder_sizes_224_and_512():
    # every 224 stage ends 7 -> 4; 512 is all-even
    assert [s[0] for s in recconv_np.ladder_sizes(56, 56, 4, 5)] == [56, 28, 14, 7, 4]
    assert [s[0] for s in recconv_np.ladder_sizes(128, 128, 4, 5)] == [128, 64, 32, 16, 8]
    assert recconv_np.ladder_sizes(25, 13, 2, 5) == [(25, 13), (13, 7), (7, 4)]


@pytest.mark.parametrize("name", recattn_cases())
def test_recattn_oracle_matches_reference(name):
    d, m = load_recattn(name)
    x = d["x"].astype(np.float64)
    attn = lambda t: recconv_np.linear_attention(t, d["w_qk"], d["b_qk"], d["w_pe"], d["b_pe"], m["heads"], m["variant"])
    dn = recconv_np.dwconv2d(x, d["w_down"], d["b_down"], 2)
    assert np.abs(dn - d["down_out"]).max() < TOL
    assert np.abs(attn(dn) - d["attn_out"]).max() < TOL
    y = recconv_np.recattn2d(x, d["w_down"], d["b_down"], attn, d["w_conv"], d["b_conv"], "nearest")
    assert np.abs(y - d["y"]).max() < TOL
    # LA1 == LA2 algebraically (lsnet/model/recattn.py:481-501 asserts 1e-4)
    other = recconv_np.linear_attention(dn, d["w_qk"], d["b_qk"], d["w_pe"], d["b_pe"], m["heads"], 3 - m["variant"])
    assert np.abs(other - d["attn_out"]).max() < 1e-4


def test_channel_multiplier_dwconv_matches_aten():
    # Downsample.token_mixer: nn.Conv2d(C, 2C, 7, padding=3, groups=C, stride=2)  (model/recnext.py:165)
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(6, 12, 7, padding=3, groups=6, stride=2)
    x = torch.randn(2, 6, 11, 9)
    with torch.no_grad():
        ref = conv(x).numpy()
    got = recconv_np.dwconv2d_mult(x.numpy().astype(np.float64), conv.weight.detach().numpy(), conv.bias.detach().numpy(), stride=2)
    assert np.abs(got - ref).max() < TOL


def _load_eager_recattn(d, m):
    """EagerRecAttn2d (oracle/torch_eager.py) carrying a fixture's fused weights: identity BatchNorm with the fused bias as beta."""
    from oracle.torch_eager import EagerRecAttn2d
    mod = EagerRecAttn2d(m["dim"], num_heads=m["heads"], stage=m["stage"]).eval()
    with torch.no_grad():
        for cn, wk_, bk_ in ((mod.down[0], "w_down", "b_down"), (mod.conv, "w_conv", "b_conv"),
                             (mod.down[1].qk, "w_qk", "b_qk"), (mod.down[1].pe, "w_pe", "b_pe")):
            cn.conv.weight.copy_(torch.from_numpy(d[wk_]))
            cn.norm.weight.fill_(1.0); cn.norm.bias.copy_(torch.from_numpy(d[bk_]))
            cn.norm.running_mean.zero_(); cn.norm.running_var.fill_(1.0 - cn.norm.eps)
    return mod


@pytest.mark.parametrize("name", recattn_cases())
def test_eager_recattn_restatement_matches_reference(name):
    """oracle/torch_eager.py's own ConvNorm / LinearAttention / RecAttn2d (nothing imported from the product) against the
    imported reference's outputs, unfused and after fuse(); includes the four RecNeXt-A3 token-mixer shapes (config 4)."""
    from recnext_amd.models import replace_batchnorm
    d, m = load_recattn(name)
    mod = _load_eager_recattn(d, m)
    assert not any(type(s).__module__.startswith("recnext_amd") for s in mod.modules())
    x = torch.from_numpy(d["x"])
    with torch.no_grad():
        assert float((mod(x) - torch.from_numpy(d["y"])).abs().max()) < 2e-5
        assert float((mod.down(x) - torch.from_numpy(d["attn_out"])).abs().max()) < 2e-5
        replace_batchnorm(mod)                                    # utils.py:227-234 through EagerConvNorm.fuse
        assert isinstance(mod.conv, torch.nn.Conv2d)
        assert float((mod(x) - torch.from_numpy(d["y"])).abs().max()) < 2e-5


@pytest.mark.parametrize("name", grad_cases())
def test_eager_recconv_autograd_matches_reference_gradients(name):
    """Autograd through the ATen restatement reproduces the gradients the imported reference block produced
    (tests/golden/make_golden.py::gen_recconv_grads): this is what the HIP backward is compared with on the GPU."""
    from oracle.torch_eager import recconv2d_eager
    d, m = load_grad(name)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).requires_grad_(True)
    x, wd, wc = t(d["x"]), t(d["w_down"]), [t(w) for w in d["w_convs"]]
    bd = t(d["b_down"]) if m["bias"] else None
    bc = [t(b) for b in d["b_convs"]] if m["bias"] else None
    y = recconv2d_eager(x, wd, wc, bd, bc, m["mode"])
    assert float((y.detach() - torch.from_numpy(d["y"])).abs().max()) < 1e-5
    y.backward(torch.from_numpy(d["gy"]))
    rel = lambda a, b: float(np.abs(a.numpy() - b).max() / (np.abs(b).max() + 1e-12))
    assert rel(x.grad, d["gx"]) < 1e-5
    assert rel(wd.grad, d["gw_down"]) < 1e-5
    for j, w in enumerate(wc):
        assert rel(w.grad, d["gw_convs"][j]) < 1e-5
    if m["bias"]:
        assert rel(bd.grad, d["gb_down"]) < 1e-5
        for j, b in enumerate(bc):
            assert rel(b.grad, d["gb_convs"][j]) < 1e-5
