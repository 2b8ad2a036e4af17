"""GPU: randomised parity sweep of the fused schedules (tools/fuzz_lanes.py) with a fixed seed: every plane size of the
register-resident families, channel counts that select every workgroup width, odd batch sizes, launch-knob overrides."""
import os
import sys

import pytest

from tests.util import rcx_env

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_configurations_match_the_c_oracle():
    import fuzz_lanes
    assert fuzz_lanes.run(60, seed=2024, verbose=False) == 0


def test_random_backward_configurations_fused_against_per_step():
    """Randomised sweep of rcx_recconv2d_bwd on the channel-per-lane backward kernels (whole-block launch, nested launch, tiled weight
    gradients, two-wave split) against the per-step schedule with every one of them switched off: batch sizes that do and do not
    fill whole XCD rounds, channel counts around the 64-lane blocks, both resize modes, all three I/O types."""
    import numpy as np
    import torch
    import recnext_amd
    from recnext_amd import ops
    rng = np.random.default_rng(77)
    dev = torch.device("cuda:0")
    knobs = ("RCX_BWD_FUSED", "RCX_WGRAD_CPL")
    assert not any(k in os.environ for k in knobs)
    for it in range(24):
        hw, level = [(7, 1), (14, 2), (28, 3), (56, 4)][it % 4]
        n = int(rng.integers(1, 6)) if hw >= 28 else int(rng.choice([1, 2, 3, 4, 8, 16]))
        c = int(rng.choice([4, 8, 36, 60, 64, 68, 128, 132, 200, 256])) if hw < 56 else int(rng.choice([4, 32, 64, 72]))
        mode = "nearest" if rng.integers(0, 2) else "bilinear"
        dtype = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 3))]
        torch.manual_seed(100 + it)
        mod = recnext_amd.RecConv2d(c, kernel_size=5, level=level, bias=bool(rng.integers(0, 2)), mode=mode).to(dev)
        wpack, bpack = mod.packed_params()
        x = torch.randn(n, c, hw, hw, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(n, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
        _, saved = ops.recconv2d_forward_train(x, wpack, bpack, level, 5, mode)
        gx1, gw1, gb1 = ops.recconv2d_backward(x, gy, wpack, saved, level, 5, mode, need_bias=True)
        with rcx_env(**{k: "0" for k in knobs}):
            gx0, gw0, gb0 = ops.recconv2d_backward(x, gy, wpack, saved, level, 5, mode, need_bias=True)
        tol = 3e-5 if dtype == torch.float32 else 1e-2
        tag = (it, n, c, hw, level, mode, str(dtype))
        for name, a1, a0, t in (("gx", gx1.float(), gx0.float(), tol), ("gw", gw1, gw0, 3e-5), ("gb", gb1, gb0, 3e-5)):
            assert torch.isfinite(a1).all(), (name, tag)
            err, ref = float((a1 - a0).abs().max()), float(a0.abs().max())
            assert err < t * max(1.0, ref), (name, tag, err, ref)
