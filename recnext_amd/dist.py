"""One-process-per-GPU plumbing for the batch-sharded (data-parallel) path.

The token mixers never mix samples, so N GPUs = N independent replicas over
contiguous batch shards.  The only collectives are the ones the reference has too: a barrier around the
timed region (utils.py:218-223) and a scalar reduction afterwards (utils.py:35-41); on GPUs the backend
string "nccl" is RCCL over xGMI.  Rendezvous is env:// on 127.0.0.1 as the launcher sets it.
"""
import os

import torch
import torch.distributed as dist


class Ranks:
    def __init__(self, rank=0, world=1, local_rank=0, device=None):
        self.rank, self.world, self.local_rank, self.device = rank, world, local_rank, device

    @property
    def is_main(self):
        return self.rank == 0


def init(device_type="cuda", backend=None):
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (no-op for one process)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if device_type == "cuda":
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if device_type == "cuda" else "gloo")
        kw = {"device_id": device} if device_type == "cuda" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return Ranks(rank, world, local_rank, device)


def barrier(r):
    if r.world > 1:
        if r.device.type == "cuda":
            dist.barrier(device_ids=[r.local_rank])
        else:
            dist.barrier()
    if r.device.type == "cuda":
        torch.cuda.synchronize(r.device)


def max_over_ranks(r, value):
    if r.world == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=r.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(r, value):
    if r.world == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=r.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(r, value):
    """[value of rank 0, ..., value of rank world-1] on every rank (one small all-gather)."""
    if r.world == 1:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=r.device)
    out = [torch.empty_like(t) for _ in range(r.world)]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


from .launch import free_port, launcher_command, spawn_ranks  # noqa: E402,F401  (torch-free; bench.py loads it by path)


def shard_bounds(n_global, rank, world):
    """Contiguous shard [lo, hi) of a global batch, sizes differing by at most one (as DistributedSampler pads)."""
    base, rem = divmod(n_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def finish(r):
    if r.world > 1 and dist.is_initialized():
        barrier(r)
        dist.destroy_process_group()


def timed_steps(r, step, steps, warmup):
    """W untimed steps, then exactly K steps bracketed by barrier + device sync; returns MAX-over-ranks seconds."""
    import time
    for _ in range(warmup):
        step()
    barrier(r)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier(r)
    return max_over_ranks(r, time.perf_counter() - t0)
