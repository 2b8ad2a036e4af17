"""ctypes binding of librecnext_amd.so (include/recnext_amd.h).

The library is built in-tree by ``recnext_amd.build`` (``recnext_amd/lib/librecnext_amd.so``).  There is
no fallback: if it is missing or fails to load, every entry point raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RCX_LIBRARY points development tools at a diagnostic build of the same ABI (tools/stamps.py)
LIB_PATH = os.environ.get("RCX_LIBRARY") or os.path.join(_HERE, "lib", "librecnext_amd.so")

DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
MODE_BILINEAR, MODE_NEAREST = 0, 1
MODES = {"bilinear": MODE_BILINEAR, "nearest": MODE_NEAREST}
MAX_LEVEL = 8
ERR_BAD_ARG, ERR_UNSUPPORTED, ERR_WORKSPACE = -1, -2, -3     # include/recnext_amd.h (rcx_status); positive = hipError_t
ABI_VERSION = 7

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t

# name -> (restype, argtypes); every symbol include/recnext_amd.h declares
SIGNATURES = {
    "rcx_selftest_d16": (_i, [_vp, _vp, _vp]),
    "rcx_abi_version": (_i, []),
    "rcx_last_error": (ctypes.c_char_p, []),
    "rcx_reload_options": (None, []),
    "rcx_recconv2d_fwd_plan": (ctypes.c_char_p, [_i] * 8),
    "rcx_pack_dw_weight": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "rcx_pack_bias": (_i, [_vp, _vp, _i, _i, _vp]),
    "rcx_pack_recconv_params": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rcx_unpack_recconv_grads": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "rcx_recconv2d_fwd_workspace_bytes": (_sz, [_i] * 7),
    "rcx_recconv2d_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _sz] + [_i] * 8 + [_vp]),
    "rcx_time_next_launch": (_i, [_vp, _vp]),
    "rcx_launch_events_pending": (_i, []),
    "rcx_recconv2d_train_saved_bytes": (_sz, [_i] * 6),
    "rcx_recconv2d_bwd_workspace_bytes": (_sz, [_i] * 6),
    "rcx_recconv2d_fwd_train": (_i, [_vp, _vp, _vp, _vp, _vp, _sz] + [_i] * 8 + [_vp]),
    "rcx_recconv2d_bwd_gy_dtype": (_i, [_i] * 7),
    "rcx_upadd_dwconv_bwd_workspace_bytes": (_sz, [_i] * 7),
    "rcx_upadd_dwconv_bwd_gy_dtype": (_i, [_i] * 8),
    "rcx_upadd_dwconv_bwd": (_i, [_vp, _vp, _vp, _i] + [_vp] * 7 + [_sz] + [_i] * 9 + [_vp]),
    "rcx_recconv2d_bwd": (_i, [_vp, _vp, _i] + [_vp] * 8 + [_i, _vp, _sz] + [_i] * 8 + [_vp]),
    "rcx_dwconv2d_fwd": (_i, [_vp, _vp, _vp, _vp] + [_i] * 8 + [_vp]),
    "rcx_dwconv2d_mult2_fwd": (_i, [_vp, _vp, _vp, _vp] + [_i] * 7 + [_vp]),
    "rcx_upadd_dwconv_fwd": (_i, [_vp, _vp, _vp, _vp, _vp] + [_i] * 11 + [_vp]),
    "rcx_upadd_dwconv_fwd_plan": (ctypes.c_char_p, [_i] * 12),
    "rcx_linear_attention_pe_fwd": (_i, [_vp] * 6 + [_i] * 6 + [_vp]),
    "rcx_recattn_qkcore_launches": (_i, [_i] * 5),
    "rcx_recattn_qkcore_workspace_bytes": (_sz, [_i] * 5),
    "rcx_recattn_qkcore_fwd": (_i, [_vp] * 7 + [_sz] + [_i] * 5 + [_vp]),
    "rcx_recattn_down_qkcore_supported": (_i, [_i] * 6),
    "rcx_recattn_down_qkcore_fwd": (_i, [_vp] * 8 + [_i] * 6 + [_vp]),
    "rcx_stem_supported": (_i, [_i] * 6),
    "rcx_stem_pack_bytes": (_sz, [_i] * 2),
    "rcx_stem_fwd": (_i, [_vp] * 6 + [_i] * 6 + [_vp]),
    "rcx_channel_mlp_supported": (_i, [_i] * 4),
    "rcx_channel_mlp_pack_bytes": (_sz, [_i] * 2),
    "rcx_channel_mlp_fwd": (_i, [_vp] * 5 + [_i] * 4 + [_vp]),
    "rcx_recattn2d_fwd_supported": (_i, [_i] * 7),
    "rcx_recattn2d_fwd": (_i, [_vp] * 10 + [_i] * 7 + [_vp]),
    "rcx_dwconv2d_bwd_workspace_bytes": (ctypes.c_size_t, [_i, _i]),
    "rcx_dwconv2d_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t] + [_i] * 7 + [_vp]),
    "rcx_dwconv2d_mult2_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t] + [_i] * 6 + [_vp]),
    "rcx_linear_attention_fwd": (_i, [_vp, _vp, _vp, _vp, _vp] + [_i] * 5 + [_vp]),
    "rcx_linear_attention_bwd": (_i, [_vp] * 7 + [_i] * 5 + [_vp]),
}

_lib = None


class RcxError(RuntimeError):
    pass


def load():
    """Load (once) and return the shared library; raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RcxError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or `make -C recnext_amd/csrc`). There is no fallback path.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        got = lib.rcx_abi_version()
        if got != ABI_VERSION:
            raise RcxError(f"librecnext_amd ABI {got} != binding ABI {ABI_VERSION}; rebuild the library")
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().rcx_last_error().decode("utf-8", "replace")
        raise RcxError(f"{what} failed (code {rc}): {msg}")
