#!/usr/bin/env python3
"""Is the inference forward launch-bound?  Eager launches against one HIP graph of the same forward (development tool).

    python3 tools/graph_probe.py [--model recnext_m3] [--batch 256] [--steps 30]

Prints images/s of the eager loop (what bench.py times), of `torch.cuda.CUDAGraph` replays of the same forward on the same input
buffer, the GPU-busy time of a step (sum of kernel durations is rocprofv3's business; here: the graph's replay time is the floor with
no host in the loop), and checks that the replayed output equals the eager output bit for bit.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from recnext_amd.speed import build_inference_model, synthetic_batch, tune_gemms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="recnext_m3")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.backends.cudnn.benchmark = True
    net = build_inference_model(args.model, dev, dtype, seed=0)
    x = synthetic_batch(args.batch, 224, dev, dtype, seed=0)
    tune_gemms(net, x)
    out = {"model": args.model, "batch": args.batch, "dtype": args.dtype, "steps": args.steps}
    with torch.no_grad():
        for _ in range(10):
            y_eager = net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            net(x)
        torch.cuda.synchronize()
        out["eager_ms_per_step"] = (time.perf_counter() - t0) / args.steps * 1e3
        # host time of a step with nothing waiting on the device: how long Python + the launch calls alone take
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        net(x)
        out["host_ms_one_step_enqueue"] = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()

        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                net(x)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            y_graph = net(x)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            g.replay()
        torch.cuda.synchronize()
        out["graph_ms_per_step"] = (time.perf_counter() - t0) / args.steps * 1e3
        out["graph_equals_eager"] = bool(torch.equal(y_graph, y_eager))
    out["eager_images_per_s"] = args.batch / out["eager_ms_per_step"] * 1e3
    out["graph_images_per_s"] = args.batch / out["graph_ms_per_step"] * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
