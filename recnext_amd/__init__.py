"""recnext_amd -- MI355X (gfx950) native token mixers for RecNeXt.

Only the hot path of the reference lives here: the RecConv2d / RecAttn2d blocks as hand-written HIP
kernels behind a C ABI (include/recnext_amd.h), the drop-in nn.Modules that call them, the timm-free
model skeleton that hosts them, and the throughput harness.
"""
from . import _lib, ops                                   # noqa: F401
from .recconv import RecConv2d                            # noqa: F401

__all__ = ["RecConv2d", "ops"]
