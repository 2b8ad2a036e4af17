#!/usr/bin/env python3
"""Is the inference forward launch-bound?  Eager launches against one HIP graph of the same forward (development tool).

    python3 tools/graph_probe.py [--model recnext_m3] [--batches 1,8,32,256] [--steps 30]

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
    ap.add_argument("--batches", default="256")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    from recnext_amd.graph import GraphedInference
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.backends.cudnn.benchmark = True
    net = build_inference_model(args.model, dev, dtype, seed=0)
    for batch in [int(b) for b in args.batches.split(",")]:
        x = synthetic_batch(batch, 224, dev, dtype, seed=0)
        tune_gemms(net, x)
        out = {"model": args.model, "batch": batch, "dtype": args.dtype, "steps": args.steps}
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
            t0 = time.perf_counter()
            net(x)
            out["host_ms_one_step_enqueue"] = (time.perf_counter() - t0) * 1e3
            torch.cuda.synchronize()
            y_eager = y_eager.clone()
            run = GraphedInference(net)
            y_graph = run(x)
            for _ in range(5):
                run(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run(x)
            torch.cuda.synchronize()
            out["graph_ms_per_step"] = (time.perf_counter() - t0) / args.steps * 1e3
            out["graph_equals_eager"] = bool(torch.equal(y_graph, y_eager))
        out["eager_images_per_s"] = batch / out["eager_ms_per_step"] * 1e3
        out["graph_images_per_s"] = batch / out["graph_ms_per_step"] * 1e3
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}), flush=True)


if __name__ == "__main__":
    main()
