#!/usr/bin/env python3
"""Does PyTorch's TunableOp (GEMM solution autotuning) speed up the FFN GEMMs of the skeleton? (development probe)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from recnext_amd.speed import build_inference_model, synthetic_batch

dev = "cuda:0"
net = build_inference_model("recnext_m3", dev, torch.bfloat16, seed=0)
x = synthetic_batch(256, 224, dev, torch.bfloat16, seed=0)


def rate(tag):
    with torch.no_grad():
        for _ in range(5):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            net(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print(f"{tag}: {dt * 1e3:.3f} ms/step  {256 / dt:.0f} img/s", flush=True)


rate("default GEMM selection")
import torch.cuda.tunable as tn
tn.enable(True)
tn.tuning_enable(True)
tn.set_max_tuning_duration(20)
tn.set_max_tuning_iterations(10)
t0 = time.perf_counter()
with torch.no_grad():
    net(x)
torch.cuda.synchronize()
print(f"tuning pass: {time.perf_counter() - t0:.1f} s", flush=True)
tn.tuning_enable(False)
rate("TunableOp selections")
