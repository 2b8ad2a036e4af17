"""Shared helpers for the parity tests."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def recconv_cases():
    return sorted(os.path.basename(p)[len("recconv_"):-4] for p in glob.glob(os.path.join(GOLDEN, "recconv_*.npz")))


def recattn_cases():
    return sorted(os.path.basename(p)[len("recattn_"):-4] for p in glob.glob(os.path.join(GOLDEN, "recattn_*.npz")))


def grad_cases():
    return sorted(os.path.basename(p)[len("grad_recconv_"):-4] for p in glob.glob(os.path.join(GOLDEN, "grad_recconv_*.npz")))


def load_grad(name):
    d = np.load(os.path.join(GOLDEN, f"grad_recconv_{name}.npz"))
    return d, json.loads(str(d["meta"]))


def load_recconv(name):
    d = np.load(os.path.join(GOLDEN, f"recconv_{name}.npz"))
    meta = json.loads(str(d["meta"]))
    return d, meta


def load_recattn(name):
    d = np.load(os.path.join(GOLDEN, f"recattn_{name}.npz"))
    return d, json.loads(str(d["meta"]))


def bf16_round_np(a):
    """float32 -> nearest-even bfloat16 -> float32, in numpy."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)


import contextlib


@contextlib.contextmanager
def rcx_env(**switches):
    """Set RCX_* environment switches (value None: unset) for the body and make the library re-read them on both sides."""
    from recnext_amd import ops
    old = {k: os.environ.get(k) for k in switches}
    try:
        for k, v in switches.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        ops.reload_options()
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        ops.reload_options()
