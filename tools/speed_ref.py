#!/usr/bin/env python3
"""The reference's operator chain on recnext_amd.speed's loop (speed_gpu.py:11-27): the measurement leg that stands beside the product's.

    python tools/speed_ref.py --model recnext_m3 --batch-size 8 --device cpu --threads 8 --dtype fp32
    python tools/speed_ref.py --model recnext_m3 --batch-size 256 --device cuda

The token mixers are the reference's ATen operators restated in oracle/torch_eager.py (through bench.py's reference_model, the
cpu_baseline leg of the headline benchmark); everything else of the model is the same skeleton.  Test infrastructure: nothing
in the recnext_amd package imports this.
"""
import argparse
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

from recnext_amd import speed


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="recnext_m1")
    ap.add_argument("--resolution", default=224, type=int)
    ap.add_argument("--batch-size", default=8, type=int)
    ap.add_argument("--dtype", default="fp32", choices=sorted(speed.DTYPES))
    ap.add_argument("--device", default="cpu", choices=["cuda", "cpu"])
    ap.add_argument("--threads", default=0, type=int, help="--device cpu: torch thread count (0 = leave)")
    ap.add_argument("--t0", default=speed.T0, type=float)
    ap.add_argument("--t1", default=speed.T1, type=float)
    args = ap.parse_args(argv)
    spec = importlib.util.spec_from_file_location("_rcx_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    if args.device == "cpu" and args.threads:
        torch.set_num_threads(args.threads)
    dtype = speed.DTYPES[args.dtype]
    net = bench.reference_model(args.model, args.device, dtype)
    with torch.no_grad():
        return speed.throughput(args.model + "[ref]", net, args.device, args.batch_size, args.resolution, dtype, args.t0, args.t1)


if __name__ == "__main__":
    main()
