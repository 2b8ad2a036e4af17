"""The ctypes stub printed in INTEGRATION.md is real code: extract it, run it, compare with the shipped module."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "ctypes.CDLL" in b]
    assert len(stub) == 1
    return stub[0]


def test_stub_is_present_and_binds_declared_symbols():
    src = _stub_source()
    header = open(os.path.join(ROOT, "include", "recnext_amd.h")).read()
    for sym in set(re.findall(r"_rcx\.(rcx_[a-z0-9_]+)", src)):
        assert sym + "(" in header, sym
    compile(src, "INTEGRATION.md", "exec")


@pytest.mark.gpu
def test_stub_matches_shipped_module_on_gpu():
    import torch
    import recnext_amd
    from recnext_amd import _lib
    ns = {}
    exec(compile(_stub_source().replace("/path/to/librecnext_amd.so", _lib.LIB_PATH), "INTEGRATION.md", "exec"), ns)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ours = recnext_amd.RecConv2d(64, kernel_size=5, level=3).to(dev).eval()
    stub = ns["RecConv2d"](64, kernel_size=5, level=3).to(dev).eval()
    stub.load_state_dict(ours.state_dict(), strict=True)
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.randn(4, 64, 28, 28, device=dev).to(dtype)
        with torch.no_grad():
            assert torch.equal(stub(x), ours(x))
