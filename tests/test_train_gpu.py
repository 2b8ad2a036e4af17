"""GPU: an engine.py-style training step (engine.py:38-71) of a small RecNeXt with the HIP token mixers, wrapped in
DistributedDataParallel over RCCL ("nccl" backend, one process = world size 1 on the test box) as main.py:310-313 does.
Gradients of the whole model are compared with the same skeleton hosting the ATen token mixers.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from recnext_amd import models
from oracle.torch_eager import eager_token_mixer

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tiny(token_mixer=None):
    torch.manual_seed(11)
    return models.RecNext(family="m", embed_dim=(8, 16, 32, 64), depth=(1, 1, 1, 1), num_classes=10, token_mixer=token_mixer)


def test_ddp_training_step_matches_aten_token_mixers():
    dev = torch.device("cuda:0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ref = _tiny(eager_token_mixer("m")).to(dev).train()
        net = _tiny().to(dev).to(memory_format=torch.channels_last).train()
        net.load_state_dict(ref.state_dict(), strict=True)
        ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0])
        x = torch.randn(4, 3, 64, 64, device=dev)
        tgt = torch.randint(0, 10, (4,), device=dev)
        loss_ref = torch.nn.functional.cross_entropy(ref(x), tgt)
        loss_ref.backward()
        loss = torch.nn.functional.cross_entropy(ddp(x.contiguous(memory_format=torch.channels_last)), tgt)
        loss.backward()                                         # DDP's bucketed all-reduce (RCCL) fires in here
        assert abs(float(loss) - float(loss_ref)) < 1e-4
        # Compare against the overall gradient scale: some gradients are analytically zero (a conv bias feeding a
        # train-mode BatchNorm) and consist of round-off on both sides, so a per-tensor relative error is meaningless.
        scale = max(float(p.grad.abs().max()) for p in ref.parameters())
        worst, worst_name = 0.0, None
        for (name, pr), (_, po) in zip(ref.named_parameters(), net.named_parameters()):
            assert po.grad is not None, name
            err = float((po.grad - pr.grad).abs().max())
            tol = 2e-3 * float(pr.grad.abs().max()) + 1e-5 * scale
            if err / tol > worst:
                worst, worst_name = err / tol, name
        assert worst < 1.0, (worst_name, worst)
        opt = torch.optim.SGD(ddp.parameters(), lr=0.05)
        opt.step()
        with torch.no_grad():
            assert torch.isfinite(ddp(x.contiguous(memory_format=torch.channels_last))).all()
    finally:
        dist.destroy_process_group()


def test_ddp_training_step_under_fp16_autocast():
    """The reference's own mixed-precision recipe (engine.py:48: autocast float16; main.py:321: loss scaler) around the HIP token
    mixers: float16 activations reach RecConv2d, its float32 parameters get float32 gradients through the HIP backward, the
    scaler unscales and steps.  Loss and gradients against the same skeleton hosting the ATen token mixers under the same autocast."""
    dev = torch.device("cuda:0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ref = _tiny(eager_token_mixer("m")).to(dev).train()
        net = _tiny().to(dev).to(memory_format=torch.channels_last).train()
        net.load_state_dict(ref.state_dict(), strict=True)
        ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0])
        x = torch.randn(4, 3, 64, 64, device=dev)
        tgt = torch.randint(0, 10, (4,), device=dev)
        with torch.autocast("cuda", dtype=torch.float16):
            loss_ref = torch.nn.functional.cross_entropy(ref(x), tgt)
        loss_ref.backward()
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
        opt = torch.optim.SGD(ddp.parameters(), lr=0.05)
        with torch.autocast("cuda", dtype=torch.float16):
            loss = torch.nn.functional.cross_entropy(ddp(x.contiguous(memory_format=torch.channels_last)), tgt)
        scaler.scale(loss).backward()                           # DDP's bucketed all-reduce (RCCL) fires in here
        assert abs(float(loss) - float(loss_ref)) < 2e-2
        scaler.unscale_(opt)
        scale = max(float(p.grad.abs().max()) for p in ref.parameters())
        for (name, pr), (_, po) in zip(ref.named_parameters(), net.named_parameters()):
            assert po.grad is not None and po.grad.dtype == torch.float32 and torch.isfinite(po.grad).all(), name
            assert float((po.grad - pr.grad).abs().max()) < 5e-2 * float(pr.grad.abs().max()) + 2e-3 * scale, name
        scaler.step(opt)
        scaler.update()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            assert torch.isfinite(ddp(x.contiguous(memory_format=torch.channels_last))).all()
    finally:
        dist.destroy_process_group()
