"""GPU parity of the matrix-core schedules (rcx_recconv2d_fwd_mx: rcx_cpt_kernel.h MX, rcx_cpl14mx.hip), through the C ABI.

These run the 5x5 convs as 4 x 4 x 4 products with operands in the activations' 16-bit type, so they are held to north_star's bar --
allclose(atol = rtol = 1e-2) against the float32 oracle on the rounded inputs and taps -- not to the half-ulp bar of the vector
kernels, and next to it to the reference's OWN bf16 run of the same inputs (fixture y_ref_bf16, tests/golden/make_golden.py).
"""
import numpy as np
import pytest
import torch

import recnext_amd
from recnext_amd import ops
from oracle import c_oracle
from tests.util import bf16_round_np, load_recconv

pytestmark = pytest.mark.gpu

ATOL = RTOL = 1e-2


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rounded(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).float().numpy()


def mx_module(c, level, mode, bias, dtype, seed):
    torch.manual_seed(seed)
    mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level, mode=mode, bias=bias).to(dev()).eval().to(dtype)
    mod.matrix_cores = True
    return mod


def oracle_of(mod, x, level, mode, bias):
    sd = {k: v.detach().float().cpu().numpy() for k, v in mod.state_dict().items()}           # the 16-bit parameters as they are
    return c_oracle.recconv2d(x.float().cpu().numpy(), sd["down.weight"], [sd[f"convs.{i}.weight"] for i in range(level + 1)], level=level,
                              mode=mode, b_down=sd.get("down.bias"), b_convs=[sd[f"convs.{i}.bias"] for i in range(level + 1)] if bias else None)


@pytest.mark.parametrize("name", ["l4_56x56", "l2_14x14", "l4_56x56_c40", "l2_14x14_nearest"])
def test_matrix_cores_against_reference_golden_and_its_own_bf16_run(name):
    d, m = load_recconv(name)
    mod = recnext_amd.RecConv2d(m["C"], kernel_size=m["k"], bias=m["bias"], level=m["level"], mode=m["mode"])
    sd = {"down.weight": bf16_round_np(d["w_down"]), **{f"convs.{i}.weight": bf16_round_np(w) for i, w in enumerate(d["w_convs"])}}
    mod.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=True)
    mod = mod.to(dev()).eval().bfloat16()
    mod.matrix_cores = True
    x = torch.from_numpy(d["x"]).to(dev()).bfloat16().contiguous(memory_format=torch.channels_last)
    n, c, h, w = x.shape
    plan = ops.recconv2d_plan_mx(n, c, h, w, m["level"], m["k"], m["mode"], torch.bfloat16)
    assert plan.startswith("cpt_mx(") or plan.startswith("cpl14_mx("), plan
    with torch.no_grad():
        y = mod(x).float().cpu().numpy()
    want = d["y_bf16in_f32"]
    err = np.abs(y - want)
    tol = ATOL + RTOL * np.abs(want)
    line = f"{name}: {plan.split('(')[0]} max|err| {err.max():.3e} mean {err.mean():.3e} worst err/tol {(err / tol).max():.2f}"
    if "y_ref_bf16" in d:
        e_ref = np.abs(d["y_ref_bf16"] - want)
        line += f" | the reference's own bf16 run: max {e_ref.max():.3e} mean {e_ref.mean():.3e} worst err/tol {(e_ref / tol).max():.2f}"
        assert err.mean() <= 1.25 * e_ref.mean() + 1e-4, line
    print(line)
    assert (err <= tol).all(), line


CASES = [
    # (N, C, H, level, dtype, mode, bias): both kernels, ragged channel blocks, image counts that are not multiples of four, both modes
    (2, 64, 56, 4, torch.bfloat16, "bilinear", False),
    (3, 64, 56, 4, torch.bfloat16, "nearest", False),
    (2, 64, 56, 4, torch.float16, "bilinear", False),
    (2, 48, 56, 4, torch.bfloat16, "bilinear", True),
    (3, 80, 56, 4, torch.bfloat16, "bilinear", False),
    (4, 256, 14, 2, torch.bfloat16, "bilinear", False),
    (5, 256, 14, 2, torch.bfloat16, "nearest", False),
    (3, 320, 14, 2, torch.float16, "bilinear", False),
    (7, 24, 14, 2, torch.bfloat16, "bilinear", True),
    (2, 200, 14, 2, torch.bfloat16, "bilinear", False),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}x{c[1]}x{c[2]}-L{c[3]}-{str(c[4])[6:]}-{c[5]}{'-bias' if c[6] else ''}")
def test_matrix_cores_against_c_oracle(case):
    n, c, h, level, dtype, mode, bias = case
    mod = mx_module(c, level, mode, bias, dtype, seed=n + c)
    x = torch.randn(n, c, h, h, device=dev()).to(dtype).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = mod(x)
        assert torch.equal(mod(x), y), "not deterministic"
    assert y.dtype == dtype and y.shape == x.shape
    want = oracle_of(mod, x, level, mode, bias)
    got = y.float().cpu().numpy()
    err = np.abs(got - want)
    tol = ATOL + RTOL * np.abs(want)
    print(f"max|err| {err.max():.3e} mean {err.mean():.3e} worst err/tol {(err / tol).max():.2f}")
    assert (err <= tol).all(), (float(err.max()), float((err / tol).max()))


@pytest.mark.parametrize("case", [(256, 64, 56, 4), (256, 256, 14, 2), (256, 80, 56, 4), (130, 320, 14, 2)], ids=lambda c: "x".join(map(str, c)))
def test_matrix_cores_full_size_properties(case):
    """BASELINE sizes: determinism, batch shard == full batch bit for bit (images never mix: the four images of a matrix block are
    independent columns of the product), exact homogeneity under x4, and three images against the oracle."""
    n, c, h, level = case
    mod = mx_module(c, level, "bilinear", False, torch.bfloat16, seed=1)
    x = torch.randn(n, c, h, h, device=dev()).bfloat16().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = mod(x)
        assert torch.equal(mod(x), y), "not deterministic"
        lo = n // 3
        assert torch.equal(mod(x[lo:lo + 5]), y[lo:lo + 5]), "batch shard differs from full batch"
        assert torch.equal(mod(x * 4), y * 4), "not homogeneous"
    assert torch.isfinite(y.float()).all()
    idx = [0, n // 2, n - 1]
    want = oracle_of(mod, x[idx], level, "bilinear", False)
    assert np.allclose(y[idx].float().cpu().numpy(), want, atol=ATOL, rtol=RTOL)


def test_matrix_cores_are_opt_in_and_need_16_bit_taps():
    mod = recnext_amd.RecConv2d(64, kernel_size=5, level=4).to(dev()).eval()
    assert mod.packed_params() and mod.packed_mx(torch.bfloat16) is None            # default: off
    mod.matrix_cores = True
    assert mod.packed_mx(torch.bfloat16) is None                                     # float32 parameters: exact taps, vector kernels
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert mod.packed_mx(torch.bfloat16) is not None                             # autocast: the conv casts its weight (engine.py:48)
    mod = mod.bfloat16()
    mod.packed_params()
    assert mod.packed_mx(torch.bfloat16) is not None and mod.packed_mx(torch.float32) is None
    # the vector kernels' answer stays within half an ulp of the float32 forward; the matrix-core answer within the tolerance of both
    x = torch.randn(2, 64, 56, 56, device=dev()).bfloat16().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y_mx = mod(x).float()
        mod.matrix_cores = False
        y_vec = mod(x).float()
    assert not torch.equal(y_mx, y_vec)
    assert torch.allclose(y_mx, y_vec, atol=2e-2, rtol=2e-2)
