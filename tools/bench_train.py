#!/usr/bin/env python3
"""Training step (forward + backward + AdamW) of RecNeXt-M3 at 224x224: HIP token mixers vs the ATen operator chain.

engine.py:38-71 style step under bf16 autocast, channels_last, synthetic data; one GPU (development tool, not bench.py).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle.torch_eager import eager_token_mixer
from recnext_amd import models

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="recnext_m3")
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--which", default="hip,aten")
ap.add_argument("--graph", action="store_true", help="capture the whole step (forward, backward, AdamW) in one HIP graph and replay it: "
                "the eager step is bound by the host's ~2000 launches below batch ~128")
args = ap.parse_args()
dev = torch.device("cuda:0")
for which in args.which.split(","):
    torch.manual_seed(0)
    fam = models.CONFIGS[args.model]["family"]
    net = models.create_model(args.model, token_mixer=None if which == "hip" else eager_token_mixer(fam))
    net = net.to(dev).to(memory_format=torch.channels_last).train()
    if which == "hip":
        models.use_hip_downsample(net)                         # Downsample depthwise conv: HIP forward + backward
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, capturable=args.graph)
    x = torch.randn(args.batch, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 1000, (args.batch,), device=dev)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(x)
            if isinstance(out, tuple):
                out = out[0]
            loss = torch.nn.functional.cross_entropy(out.float(), y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    if args.graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            static_loss = step()
        eager_step = step

        def step():
            graph.replay()
            return static_loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.steps):
        loss = step()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / args.steps
    print(json.dumps({"model": args.model, "token_mixers": which, "batch": args.batch, "hip_graph": bool(args.graph), "ms_per_step": round(ms, 2),
                      "images_per_s": round(args.batch / ms * 1e3, 1), "loss": round(float(loss), 4)}), flush=True)
