"""GPU, more than one rank: DDP's gradient all-reduce over RCCL with the HIP token mixers (main.py:310-313, utils.py:202-224).

Needs two GPUs in the box (skipped otherwise; the round-end 8-GPU node runs it).  Two ranks, each with half of a fixed global
batch, must end with the gradients one rank computes on the whole batch (DDP averages; BatchNorm statistics frozen so that the
shards do not couple through batch statistics, as in the reference without SyncBN).
"""
import os
import sys

import pytest
import torch

from recnext_amd import launch

pytestmark = pytest.mark.gpu

WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_two_rank_rccl_ddp_step_equals_one_rank_on_the_whole_batch(tmp_path):
    out2, out1 = str(tmp_path / "w2.pt"), str(tmp_path / "w1.pt")
    assert launch.spawn_ranks(2, WORKER, ["ddp-step", out2]) == 0
    assert launch.spawn_ranks(1, WORKER, ["ddp-step", out1]) == 0
    a, b = torch.load(out2), torch.load(out1)
    assert a["world"] == 2 and b["world"] == 1
    assert abs(a["loss"] - b["loss"]) < 1e-5
    scale = max(float(g.abs().max()) for g in b["grads"].values())
    for k, g1 in b["grads"].items():
        err = float((a["grads"][k] - g1).abs().max())
        assert err <= 1e-3 * float(g1.abs().max()) + 1e-5 * scale, (k, err)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` as the driver invokes it: the GPU-free parent starts two ranks and relays their exit code."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "32"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and len(rec["per_rank_images_per_s"]) == 2
    assert rec["scaling"] == "weak" and rec["config"]["global_batch"] == 64
