"""In-tree build of librecnext_amd.so for gfx950 (hipcc cross-compiles without a GPU)."""
import hashlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")


def source_fingerprint():
    """sha256 over the kernel sources the library is built from (csrc/*.hip, *.h, Makefile and the public header), in name order.  A PMC
    traffic profile records it (tools/profile_summary.py); bench.py quotes a profile's HBM bytes only for the sources it was measured on."""
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) or f == "Makefile")
    paths = [os.path.join(CSRC, f) for f in files] + [os.path.join(os.path.dirname(_HERE), "include", "recnext_amd.h")]
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build_library(force=False, jobs=4, verbose=False):
    cmd = ["make", "-C", CSRC, f"-j{jobs}"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building librecnext_amd.so failed")
    from . import _lib
    return _lib.LIB_PATH


if __name__ == "__main__":
    print(build_library(verbose=True))
