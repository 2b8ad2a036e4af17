#!/usr/bin/env python3
"""Where the host's time goes in a training step (the A3 step is bound by it at batch 128): cProfile of the main thread over a
few steps (the backward runs on autograd's own thread and shows up only as run_backward).  python tools/host_profile_step.py [model] [batch]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from recnext_amd import models

name = sys.argv[1] if len(sys.argv) > 1 else "recnext_a3"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
net = models.create_model(name).to(dev).to(memory_format=torch.channels_last).train()
models.use_hip_downsample(net)
opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
x = torch.randn(batch, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
y = torch.randint(0, 1000, (batch,), device=dev)


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(x)
        if isinstance(out, tuple):
            out = out[0]
        loss = torch.nn.functional.cross_entropy(out.float(), y)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
