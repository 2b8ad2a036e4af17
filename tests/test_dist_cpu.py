"""CPU, world_size 2 over gloo: the one-process-per-GPU plumbing (sharding, barrier-bracketed timing, reductions).

The HIP kernels cannot run here, so the replicas host the oracle's ATen token mixer; what is under test is
recnext_amd.dist and the property the scaling path relies on: batch shards computed by independent ranks,
concatenated, equal the full batch (images never mix).
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from recnext_amd import dist as rdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from oracle.torch_eager import EagerRecConv2d
    r = rdist.init(device_type="cpu")
    assert (r.rank, r.world) == (rank, world)
    torch.manual_seed(0)                                   # identical weights on every rank
    mod = EagerRecConv2d(8, level=2).eval()
    g = torch.Generator().manual_seed(123)
    x = torch.randn(5, 8, 14, 14, generator=g)             # identical global batch; 5 does not divide by 2
    lo, hi = rdist.shard_bounds(x.shape[0], r.rank, r.world)
    calls = {"n": 0}

    def step():
        calls["n"] += 1
        with torch.no_grad():
            return mod(x[lo:hi])

    elapsed = rdist.timed_steps(r, step, steps=3, warmup=2)
    assert calls["n"] == 5 and elapsed > 0
    total = rdist.sum_over_ranks(r, hi - lo)
    assert total == x.shape[0]
    assert rdist.max_over_ranks(r, float(rank)) == world - 1
    torch.save({"lo": lo, "hi": hi, "y": step(), "elapsed": elapsed}, os.path.join(out_dir, f"rank{rank}.pt"))
    if rank == 0:
        with torch.no_grad():
            torch.save(mod(x), os.path.join(out_dir, "full.pt"))
    rdist.finish(r)


def test_two_process_gloo_shards_equal_full_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(tmp_path / f"rank{i}.pt") for i in range(world)]
    full = torch.load(tmp_path / "full.pt")
    assert [(p["lo"], p["hi"]) for p in parts] == [(0, 3), (3, 5)]
    assert torch.equal(torch.cat([p["y"] for p in parts]), full)
    assert parts[0]["elapsed"] == parts[1]["elapsed"]       # both ranks report the MAX


@pytest.mark.parametrize("n,world", [(256, 8), (10, 4), (3, 8), (2048, 8)])
def test_shard_bounds_partition_the_batch(n, world):
    spans = [rdist.shard_bounds(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    sizes = [hi - lo for lo, hi in spans]
    assert max(sizes) - min(sizes) <= 1


def test_single_process_is_a_noop():
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    r = rdist.init(device_type="cpu")
    assert r.world == 1 and r.is_main
    assert rdist.max_over_ranks(r, 2.5) == 2.5
    rdist.barrier(r)
    rdist.finish(r)


def test_spawn_ranks_starts_fresh_children_under_torch_distributed_run(tmp_path):
    """The self-launch path of `python bench.py --gpus N`: recnext_amd.launch (torch-free) starts N ranks with
    torch.distributed.run on 127.0.0.1 and relays the exit code; here two CPU ranks over gloo."""
    import sys
    from recnext_amd import launch
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")
    cmd = launch.launcher_command(2, worker, ["gloo-sum", "x"], port=1234)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    out = tmp_path / "sum.pt"
    assert launch.spawn_ranks(2, worker, ["gloo-sum", str(out)]) == 0
    rec = torch.load(out)
    assert rec == {"sum": 3.0, "world": 2, "gathered": [0.0, 2.0]}
    assert launch.spawn_ranks(2, worker, ["no-such-job", str(out)]) != 0          # a failing rank is reported


def test_bench_parent_decides_to_spawn_before_importing_torch(monkeypatch):
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.maybe_spawn_ranks(["--gpus", "1", "--steps", "2"]) is None
    assert bench.maybe_spawn_ranks(["--steps", "2"]) is None
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert bench.maybe_spawn_ranks(["--gpus", "4"]) is None                        # already under a launcher
    monkeypatch.delenv("WORLD_SIZE")
    calls = []
    import recnext_amd.launch as real
    monkeypatch.setattr(real, "spawn_ranks", lambda n, script, argv, env=None: calls.append((n, script, argv)) or 7)
    monkeypatch.setattr(importlib.util, "spec_from_file_location",
                        lambda name, path: type("S", (), {"loader": type("L", (), {"exec_module": staticmethod(lambda m: None)})()})())
    monkeypatch.setattr(importlib.util, "module_from_spec", lambda spec: real)
    with pytest.raises(SystemExit) as e:
        bench.maybe_spawn_ranks(["--gpus=8", "--steps", "2"])
    assert e.value.code == 7 and calls == [(8, os.path.join(root, "bench.py"), ["--gpus=8", "--steps", "2"])]


def test_speed_harness_reference_leg_on_the_host_cores(capsys):
    """recnext_amd.speed's T0/T1 loop and output line (speed_gpu.py:11-27, :39-51) on the CPU reference leg, which lives outside the
    package (tools/speed_ref.py); the product's own entry point refuses to run without a GPU."""
    import importlib.util
    from recnext_amd import speed
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_speed_ref", os.path.join(root, "tools", "speed_ref.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rate = ref.main(["--model", "recnext_m0", "--batch-size", "2", "--resolution", "64", "--device", "cpu",
                     "--dtype", "fp32", "--threads", "2", "--t0", "0.2", "--t1", "0.5"])
    out = capsys.readouterr().out.strip().splitlines()[-1].split()
    assert out[0] == "recnext_m0[ref]" and out[1] == "cpu" and out[3:] == ["images/s", "@", "batch", "size", "2"]
    assert abs(float(out[2]) - rate) < 1e-6 * rate and rate > 0
    if not torch.cuda.is_available():
        with pytest.raises(SystemExit):
            speed.main(["--model", "recnext_m0"])
    assert "oracle" not in open(os.path.join(root, "recnext_amd", "speed.py")).read().replace("oracle/torch_eager.py", "")
