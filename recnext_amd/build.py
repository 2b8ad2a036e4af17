"""In-tree build of librecnext_amd.so for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")


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
