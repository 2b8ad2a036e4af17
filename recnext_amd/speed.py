"""Throughput harness -- the MI355X counterpart of the reference's speed_gpu.py (speed_gpu.py:11-27, :39-51).

    python -m recnext_amd.speed --model recnext_m3 --resolution 224 --batch-size 256 [--dtype bf16]

Same procedure: build the registered model, fold BatchNorm (utils.replace_batchnorm), eval mode, random
input created once on the device, T0 seconds of warm-up, then iterate with a device synchronise per
iteration until T1 seconds have accumulated; print ``name device images/s @ batch size B``.
Additions: ``--dtype`` (the reference runs fp32 only) and channels_last storage, which is what the HIP
token mixers consume zero-copy.  There is no CPU mode here: the product path is GPU-only.
"""
import argparse
import os
import time

import torch

from . import models

T0 = 5
T1 = 10

DTYPES = {"fp32": torch.float32, "f32": torch.float32, "bf16": torch.bfloat16}


def build_inference_model(name, device, dtype=torch.bfloat16, token_mixer=None, seed=0, fold_mixer_norm=True,
                          hip_downsample=True, linear_pointwise=True):
    """create_model -> replace_batchnorm -> device/eval, as speed_gpu.py:47-50 (plus dtype + channels_last).

    ``fold_mixer_norm`` additionally absorbs the BatchNorm after each HIP token mixer into the mixer's last
    conv (same function, one kernel less per block); it is a no-op for other token mixers.  ``hip_downsample``
    runs the three strided depthwise Downsample convs (+ their BatchNorm) on the HIP kernels as well.
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


def throughput(name, model, device, batch_size, resolution=224, dtype=torch.bfloat16, t0=T0, t1=T1):
    inputs = synthetic_batch(batch_size, resolution, device, dtype)
    torch.cuda.empty_cache()
    tune_gemms(model, inputs)
    torch.cuda.synchronize()
    start = time.time()
    while time.time() - start < t0:
        model(inputs)
    timing = []
    torch.cuda.synchronize()
    while sum(timing) < t1:
        start = time.time()
        model(inputs)
        torch.cuda.synchronize()
        timing.append(time.time() - start)
    rate = batch_size / (sum(timing) / len(timing))
    print(name, device, rate, "images/s @ batch size", batch_size)
    return rate


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="recnext_m1", type=str)
    ap.add_argument("--resolution", default=224, type=int)
    ap.add_argument("--batch-size", default=2048, type=int)
    ap.add_argument("--dtype", default="bf16", choices=sorted(DTYPES))
    args = ap.parse_args(argv)
    torch.autograd.set_grad_enabled(False)
    device = "cuda:0"
    net = build_inference_model(args.model, device, DTYPES[args.dtype])
    throughput(args.model, net, device, args.batch_size, args.resolution, DTYPES[args.dtype])


if __name__ == "__main__":
    main()
