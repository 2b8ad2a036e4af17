#!/usr/bin/env python3
"""The other BASELINE.json configurations next to the headline one (development tool; bench.py stays the contract):
  cfg1  RecNeXt-M0 fp32 batch 1 on the host CPU (reference operator chain restated in oracle/torch_eager.py), 1 and all threads
  cfg2  RecNeXt-M1 bf16 224x224 batch 256, 1 GPU          cfg4  RecNeXt-A3 bf16 224x224 batch 256 (RecAttn2d, nearest)
  cfg5  RecNeXt-M3 bf16 512x512 batch 32                   (cfg3 is the 8-GPU run of bench.py --model recnext_m5)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from recnext_amd import models
from recnext_amd.speed import build_inference_model, synthetic_batch


def gpu_rate(name, batch, res, steps=20, warm=5):
    net = build_inference_model(name, "cuda:0", torch.bfloat16, seed=0)
    x = synthetic_batch(batch, res, "cuda:0", torch.bfloat16, seed=0)
    with torch.no_grad():
        for _ in range(warm):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"model": name, "resolution": res, "batch": batch, "dtype": "bf16", "ms_per_step": round(dt * 1e3, 3), "images_per_s": round(batch / dt, 1)}


def cpu_latency(threads):
    from oracle.torch_eager import eager_token_mixer
    torch.set_num_threads(threads)
    net = build_inference_model("recnext_m0", "cpu", torch.float32, token_mixer=eager_token_mixer("m"))
    x = synthetic_batch(1, 224, "cpu", torch.float32)
    with torch.no_grad():
        for _ in range(5):
            net(x)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            net(x)
            ts.append(time.perf_counter() - t0)
    ts.sort()
    return {"model": "recnext_m0", "device": "cpu", "threads": threads, "host_cpus": os.cpu_count(), "batch": 1, "dtype": "fp32",
            "median_ms": round(ts[len(ts) // 2] * 1e3, 2), "images_per_s": round(1.0 / ts[len(ts) // 2], 1)}


if __name__ == "__main__":
    torch.backends.cudnn.benchmark = True
    which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["cfg2", "cfg4", "cfg5", "cfg1"]
    if "cfg2" in which:
        print(json.dumps({"config": "cfg2", **gpu_rate("recnext_m1", 256, 224)}), flush=True)
    if "cfg4" in which:
        print(json.dumps({"config": "cfg4", **gpu_rate("recnext_a3", 256, 224)}), flush=True)
    if "cfg5" in which:
        print(json.dumps({"config": "cfg5", **gpu_rate("recnext_m3", 32, 512)}), flush=True)
    if "m5" in which:
        print(json.dumps({"config": "cfg3 (1 of 8 GPUs)", **gpu_rate("recnext_m5", 256, 224)}), flush=True)
    if "cfg1" in which:
        print(json.dumps({"config": "cfg1", **cpu_latency(1)}), flush=True)
        print(json.dumps({"config": "cfg1", **cpu_latency(min(os.cpu_count(), 128))}), flush=True)
