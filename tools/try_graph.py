#!/usr/bin/env python3
"""Eager launches vs one captured hipGraph of the whole RecNeXt-M3 forward (development tool)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from recnext_amd.speed import build_inference_model, synthetic_batch

dev = "cuda:0"
net = build_inference_model("recnext_m3", dev, torch.bfloat16, seed=0)
x = synthetic_batch(256, 224, dev, torch.bfloat16, seed=0)
with torch.no_grad():
    for _ in range(10):
        net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        net(x)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 30
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            net(x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        y = net(x)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 30
print(f"eager {eager*1e3:.3f} ms/step ({256/eager:.0f} img/s)   graph {graph*1e3:.3f} ms/step ({256/graph:.0f} img/s)")
