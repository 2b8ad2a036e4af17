"""Worker script of the multi-rank tests (started by recnext_amd.launch.spawn_ranks, i.e. torch.distributed.run).

    dist_worker.py gloo-sum OUT          CPU: all-reduce the rank numbers over gloo, rank 0 writes the result
    dist_worker.py ddp-step OUT          GPU: one engine.py-style DDP training step (main.py:310-313) of a small RecNeXt with the
                                         HIP token mixers over RCCL ("nccl"), this rank's shard of a fixed global batch; rank 0
                                         writes the loss and every gradient after DDP's all-reduce
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from recnext_amd import dist as rdist  # noqa: E402


def tiny(token_mixer=None):
    from recnext_amd import models
    torch.manual_seed(11)
    return models.RecNext(family="m", embed_dim=(8, 16, 32, 64), depth=(1, 1, 1, 1), num_classes=10, token_mixer=token_mixer)


def global_batch(n=8):
    g = torch.Generator().manual_seed(77)
    return torch.randn(n, 3, 64, 64, generator=g), torch.randint(0, 10, (n,), generator=g)


def ddp_step(r, net, x, tgt, autocast_dtype=None):
    """Forward + backward of this rank's shard under DDP; returns (mean loss over the global batch, grads by name)."""
    lo, hi = rdist.shard_bounds(x.shape[0], r.rank, r.world)
    xs = x[lo:hi].to(r.device).contiguous(memory_format=torch.channels_last)
    ts = tgt[lo:hi].to(r.device)
    ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[r.local_rank]) if r.world > 1 else net
    with torch.autocast("cuda", dtype=autocast_dtype, enabled=autocast_dtype is not None):
        loss = torch.nn.functional.cross_entropy(ddp(xs), ts)
    loss.backward()                                             # DDP's bucketed all-reduce (RCCL over xGMI) fires in here
    total = rdist.sum_over_ranks(r, float(loss) * (hi - lo)) / x.shape[0]
    return total, {k: p.grad.detach().float().cpu() for k, p in net.named_parameters()}


def main():
    what, out = sys.argv[1], sys.argv[2]
    if what == "gloo-sum":
        r = rdist.init("cpu")
        t = torch.tensor([float(r.rank + 1)])
        dist.all_reduce(t)
        if r.is_main:
            torch.save({"sum": float(t), "world": r.world, "gathered": rdist.gather_over_ranks(r, r.rank * 2.0)}, out)
        else:
            rdist.gather_over_ranks(r, r.rank * 2.0)
        rdist.finish(r)
        return
    if what == "ddp-step":
        r = rdist.init("cuda")                                  # backend "nccl" = RCCL
        net = tiny().to(r.device).to(memory_format=torch.channels_last).train()
        for m in net.modules():                                 # SyncBN is not what the reference uses by default: freeze BN statistics
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                m.eval()
        x, tgt = global_batch()
        loss, grads = ddp_step(r, net, x, tgt)
        if r.is_main:
            torch.save({"loss": loss, "grads": grads, "world": r.world}, out)
        rdist.finish(r)
        return
    raise SystemExit(f"unknown job {what!r}")


if __name__ == "__main__":
    main()
