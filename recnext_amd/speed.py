"""Throughput harness -- the MI355X counterpart of the reference's speed_gpu.py (speed_gpu.py:11-27, :39-51).

    python -m recnext_amd.speed --model recnext_m3 --resolution 224 --batch-size 256 [--dtype bf16]

Same procedure: build the registered model, fold BatchNorm (utils.replace_batchnorm), eval mode, random
input created once on the device, T0 seconds of warm-up, then iterate with a device synchronise per
iteration until T1 seconds have accumulated; print ``name device images/s @ batch size B``.
Additions: ``--dtype`` (the reference runs fp32 only) and channels_last storage, which is what the HIP
token mixers consume zero-copy; ``--gpus N`` (one rank per GPU on batch shards, whole-job images/s;
recnext_amd.launch starts the ranks).  The product has no CPU mode and exactly one implementation: timing the REFERENCE's operator
chain (on the GPU or on the host cores) is tools/speed_ref.py, outside this package, on the same loop (``throughput`` below).
"""
import argparse
import os
import time

import torch

from . import models

T0 = 5
T1 = 10

DTYPES = {"fp32": torch.float32, "f32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16, "f16": torch.float16}


def build_inference_model(name, device, dtype=torch.bfloat16, token_mixer=None, seed=0, fold_mixer_norm=True,
                          hip_downsample=True, linear_pointwise=True, pad_hidden=True, fused_mlp=True, fused_stem=True):
    """create_model -> replace_batchnorm -> device/eval, as speed_gpu.py:47-50 (plus dtype + channels_last).

    ``fold_mixer_norm`` additionally absorbs the BatchNorm after each HIP token mixer into the mixer's last
    conv (same function, one kernel less per block); it is a no-op for other token mixers.  ``hip_downsample``
    runs the three strided depthwise Downsample convs (+ their BatchNorm) on the HIP kernels as well.  ``pad_hidden`` lets the
    channel mixers' GEMMs run at a zero-padded hidden width (``models.pad_mlp_hidden``: RecNeXt-A's 120 / 240 / 480 -> 128 / 256 / 512).
    ``fused_mlp`` (bf16): ``x + channel_mixer(.)`` of the blocks rcx_channel_mlp_fwd has a kernel for as one HIP launch (``models.use_fused_mlp``).
    ``fused_stem`` (bf16): the stem's two convs and GELU as one HIP launch (``models.use_fused_stem``).
    """
    torch.manual_seed(seed)
    net = models.create_model(name, num_classes=1000, token_mixer=token_mixer)
    models.replace_batchnorm(net)
    net = net.eval()
    if fold_mixer_norm:
        models.fold_token_mixer_norms(net)
    if hip_downsample and token_mixer is None and torch.device(device).type == "cuda":
        models.use_hip_downsample(net)
    net = net.to(device=device, dtype=dtype).eval()
    if torch.device(device).type == "cuda":
        net = net.to(memory_format=torch.channels_last)
        if linear_pointwise:
            models.use_linear_pointwise(net)
            if pad_hidden:
                models.pad_mlp_hidden(net)
            if fused_mlp and dtype == torch.bfloat16:
                models.use_fused_mlp(net)
        if fused_stem and dtype == torch.bfloat16:
            models.use_fused_stem(net)
    return net


def synthetic_batch(batch_size, resolution, device, dtype=torch.bfloat16, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(batch_size, 3, resolution, resolution, generator=g).to(device=device, dtype=dtype)
    return x.contiguous(memory_format=torch.channels_last) if torch.device(device).type == "cuda" else x


@torch.no_grad()
def tune_gemms(model, inputs):
    """PyTorch-ROCm plumbing, the GEMM counterpart of ``cudnn.benchmark``: let TunableOp pick the GEMM-library solution for
    every 1x1-conv / linear shape of the skeleton during one forward pass, then freeze the choices (same GEMMs, faster tiles;
    about 5 s).  Returns True if tuning ran."""
    try:
        import tempfile
        import torch.cuda.tunable as tn
    except ImportError:        # an older PyTorch without TunableOp: keep the default selection
        return False
    tn.enable(True)
    try:
        tn.set_filename(os.path.join(tempfile.gettempdir(), "recnext_amd_tunableop.csv"), True)   # results file: not the cwd
        tn.set_max_tuning_duration(20)
        tn.set_max_tuning_iterations(10)
        tn.tuning_enable(True)
        with torch.no_grad():
            model(inputs)
        torch.cuda.synchronize()
        return True
    except Exception:
        return False
    finally:
        tn.tuning_enable(False)


def throughput(name, model, device, batch_size, resolution=224, dtype=torch.bfloat16, t0=T0, t1=T1, ranks=None, quiet=False, graph=False):
    """speed_gpu.py:11-27.  With `ranks` (recnext_amd.dist.Ranks, world > 1) every rank runs the loop on its own shard of
    `batch_size` images per GPU and rank 0 reports the whole-job rate (sum over ranks).  `graph`: the same forward replayed as one
    HIP graph (recnext_amd.graph.GraphedInference) -- what small batches, bound by the host's launches, want."""
    cuda = torch.device(device).type == "cuda"
    sync = torch.cuda.synchronize if cuda else (lambda: None)
    inputs = synthetic_batch(batch_size, resolution, device, dtype, seed=ranks.rank if ranks else 0)
    if cuda:
        torch.cuda.empty_cache()
        tune_gemms(model, inputs)
    if graph:
        from .graph import GraphedInference
        model = GraphedInference(model)
    sync()
    start = time.time()
    model(inputs)
    while time.time() - start < t0:
        model(inputs)
    timing = []
    sync()
    while sum(timing) < t1:
        start = time.time()
        model(inputs)
        sync()
        timing.append(time.time() - start)
    rate = batch_size / (sum(timing) / len(timing))
    if ranks is not None and ranks.world > 1:
        from . import dist as rdist
        rate = rdist.sum_over_ranks(ranks, rate)
        batch_size *= ranks.world
    if not quiet and (ranks is None or ranks.is_main):
        print(name, device if ranks is None or ranks.world == 1 else f"{ranks.world}x{torch.device(device).type}", rate,
              "images/s @ batch size", batch_size)
    return rate


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="recnext_m1", type=str)
    ap.add_argument("--resolution", default=224, type=int)
    ap.add_argument("--batch-size", default=2048, type=int)
    ap.add_argument("--dtype", default="bf16", choices=sorted(DTYPES))
    ap.add_argument("--gpus", default=1, type=int, help="ranks (one per GPU); --batch-size is per GPU")
    ap.add_argument("--graph", action="store_true", help="replay the forward as one HIP graph (small batches are bound by the host's launches)")
    ap.add_argument("--t0", default=T0, type=float)
    ap.add_argument("--t1", default=T1, type=float)
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        ap.error("the HIP token mixers have no CPU path (tools/speed_ref.py --device cpu times the reference's CPU baseline)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import sys
        from . import launch              # children are fresh processes; this parent has not touched the GPU
        raise SystemExit(launch.spawn_ranks(args.gpus, os.path.abspath(__file__), list(argv if argv is not None else sys.argv[1:])))
    from . import dist as rdist
    ranks = rdist.init("cuda")
    device = str(ranks.device)
    dtype = DTYPES[args.dtype]
    net = build_inference_model(args.model, device, dtype)
    rdist.barrier(ranks)
    with torch.no_grad():                 # speed_gpu.py:40 switches autograd off for the whole process; scoped here
        rate = throughput(args.model, net, device, args.batch_size, args.resolution, dtype, args.t0, args.t1, ranks=ranks, graph=args.graph)
    rdist.finish(ranks)
    return rate


if __name__ == "__main__":
    if __package__ in (None, ""):         # started by path (a child rank of --gpus N): make the package importable
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from recnext_amd.speed import main as _main
        _main()
    else:
        main()
