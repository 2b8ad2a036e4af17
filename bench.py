#!/usr/bin/env python3
"""Headline benchmark: images/sec of RecNeXt-M3 224x224 bf16 with the HIP RecConv2d token mixers.

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts its own N ranks, see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Called with --gpus N > 1 and no WORLD_SIZE in the environment, the still GPU-free parent starts N child ranks with
torch.distributed.run (fresh processes) and exits with their return code.

One "step" = one forward pass of the BN-folded, channels_last RecNeXt-M3 over one synthetic batch of
256 images per GPU (weak scaling: batch-sharded, no collective in the timed region -- SURVEY 8e).
Rank 0 prints ONE JSON line.  Besides the contract keys it carries

  roofline      the fused RecConv2d forward of the dominant block shape: ALGORITHMIC bytes
                (2*N*C*H*W*b + (level+2)*C*k*k*b) / its mean duration measured with HIP
                events on the launch stream inside the timed region, against the 8 TB/s HBM peak.
  token_mixers  the same accounting summed over all 21 RecConv2d calls of the model.
  cpu_baseline  the reference's CPU path (oracle/torch_eager.py: the same ATen operators on the host
                cores) timed on a bounded sample of the same workload at 1 / 8 / 32 / all-physical-core
                threads, best reported with its thread count; plus BASELINE config 1 (RecNeXt-M0, batch 1,
                fp32); rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
VALU_PEAK_TFS = 157.3          # float32 vector peak (same table): the secondary ceiling of SURVEY 8d


def measure_copy_ceiling(torch, device, gib=1.0, reps=5):
    """The copy ceiling of THIS device ("state both the nominal and the measured copy ceiling"): a device-to-device copy of
    1 GiB, read + written bytes over its HIP-event time, best of `reps` after one warm-up.  Runs before the timed region."""
    n = int(gib * (1 << 30)) // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=device).normal_()
    b = torch.empty_like(a)
    best = 0.0
    for i in range(reps + 1):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        b.copy_(a)
        e.record()
        torch.cuda.synchronize(device)
        if i:
            best = max(best, 2 * a.numel() * 2 / (s.elapsed_time(e) * 1e-3) / 1e9)
    del a, b
    torch.cuda.empty_cache()
    return best


def recconv_flops(n, c, h, w, level, k):
    """Algorithmic flops of one RecConv2d forward: 2 k^2 C (sum_{l>=1} H_l W_l + sum_{l>=0} H_l W_l) for the stride-2 ladder
    and the level + 1 stride-1 convs, + 8 per resized element (bilinear) + 1 per added element."""
    hs, ws = [h], [w]
    for _ in range(level):
        hs.append((hs[-1] + 1) // 2)
        ws.append((ws[-1] + 1) // 2)
    areas = [a * b for a, b in zip(hs, ws)]
    conv = 2 * k * k * c * (sum(areas[1:]) + sum(areas))
    resize_add = 9 * c * sum(areas[:-1])
    return n * (conv + resize_add)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="recnext_m3")
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--resolution", type=int, default=224)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="total budget of the cpu_baseline leg")
    return ap.parse_args()


class MixerTimers:
    """HIP events around token-mixer calls (recorded on the stream the kernels are launched on): RecConv2d is one kernel
    launch; RecAttn2d (A family) is a unit of three or four HIP kernels (or four and two GEMMs) and is reported as such.

    Every event pair costs the stream ~2.5 us (the marker packets keep the next kernel from starting early): 40 pairs a step are
    2.7 % of RecNeXt-M3's step (tools/graph_probe.py: 63.8 k img/s without any, 62.1 k with all).  So the warm-up steps bracket
    EVERY mixer (the per-kernel table, and which kernel dominates), and the timed region times only the dominant kernel's
    launches (`only`), which is what the roofline object is computed from.

    Where the dominant kernel is ONE fused launch (plans "cpt(...)" / "cpl(...)": `exact`), the timed region does not bracket the call: it hands
    the event pair to the library (rcx_time_next_launch), which has the command processor record them AT the kernel's start and end
    (hipExtLaunchKernelGGL) -- the duration rocprofv3 reports, without the dispatch gaps a bracket includes (+2.5 .. 4 us on a ~105 us kernel:
    0.2325 against rocprofv3's 0.2414 in round 4's first record) and without the markers' cost to the step."""

    def __init__(self, net, torch, RecConv2d):
        self.torch = torch
        self.records = []          # (module_key, start, end)
        self.enabled = False
        self.only = None           # None: every mixer; else the set of (C, H, W, level, k) keys whose launches are bracketed
        self.exact = set()         # keys (subset of `only`) whose one-kernel launches are timed by events attached to the dispatch
        self.pool = []             # event pairs created (recorded once) before the timed region, for the exact launches
        self.lib = None
        self.keys = {}
        for name, m in net.named_modules():
            if isinstance(m, RecConv2d):
                self.keys[m] = name
                m.register_forward_pre_hook(self._pre)
                m.register_forward_hook(self._post)
        self._open = {}

    @staticmethod
    def shape_key(m, shape):
        return (shape[1], shape[2], shape[3], getattr(m, "level", None), m.kernel_size)

    def prepare_exact(self, keys, pairs, lib):
        """Event pairs for `pairs` launches of the one-kernel plans `keys`: created here (an event exists once it has been recorded), outside the
        timed region."""
        self.exact, self.lib = set(keys), lib
        self.pool = []
        for _ in range(pairs if keys else 0):
            s, e = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            s.record(); e.record()
            self.pool.append((s, e))
        self.torch.cuda.synchronize()

    def _pre(self, m, args):
        if self.enabled and (self.only is None or self.shape_key(m, args[0].shape) in self.only):
            if self.pool and self.shape_key(m, args[0].shape) in self.exact:
                s, e = self.pool.pop()
                self.lib.rcx_time_next_launch(s.cuda_event, e.cuda_event)
                self._open[m] = (s, tuple(args[0].shape), e)
                return
            e = self.torch.cuda.Event(enable_timing=True)
            e.record()
            self._open[m] = (e, tuple(args[0].shape), None)

    def _post(self, m, args, out):
        if m in self._open:
            s, shape, e = self._open.pop(m)
            if e is None:
                e = self.torch.cuda.Event(enable_timing=True)
                e.record()
            elif self.lib.rcx_launch_events_pending():        # the schedule did not take the pair (cannot happen for an `exact` key): refuse to report a stale pair
                raise RuntimeError("bench.py: the timed launch did not consume its event pair")
            self.records.append((m, shape, s, e))

    def summarize(self, elem_bytes, plan_of):
        """Per block shape, and per kernel instantiation (what rocprofv3 --stats aggregates by)."""
        by_shape = {}
        for m, shape, s, e in self.records:
            n = shape[0]
            key = self.shape_key(m, shape)
            ent = by_shape.setdefault(key, {"ms": 0.0, "calls": 0, "N": n})
            ent["ms"] += s.elapsed_time(e)
            ent["calls"] += 1
        shapes, kernels = [], {}
        for (c, h, w, level, k), ent in by_shape.items():
            n = ent["N"]
            avg_ms = ent["ms"] / ent["calls"]
            if level is None:                 # RecAttn2d: compulsory bytes of its depthwise pieces, 3.5 * S per block
                alg = int(3.5 * n * c * h * w * elem_bytes)
                plan = "recattn2d(conv5 stride 2 + qk projection / attention core / pe [rcx_recattn_qkcore_fwd: 1-2 launches; or 2 GEMMs + k_linattn_core4] + conv5(x + resize))"
            else:
                alg = 2 * n * c * h * w * elem_bytes + (level + 2) * c * k * k * elem_bytes
                plan = plan_of(n, c, h, w, level, k)
            shapes.append({"C": c, "H": h, "W": w, "level": level, "k": k, "N": n, "calls": ent["calls"], "plan": plan,
                           "total_ms": ent["ms"], "avg_ms": avg_ms, "algorithmic_bytes": alg,
                           "achieved_GBs": alg / (avg_ms * 1e-3) / 1e9})
            kn = kernel_name(plan, elem_bytes)
            kk = kernels.setdefault(kn, {"kernel": kn, "total_ms": 0.0, "launches": 0, "algorithmic_bytes": 0, "shapes": []})
            kk["total_ms"] += ent["ms"]
            kk["launches"] += ent["calls"]
            kk["algorithmic_bytes"] += alg * ent["calls"]
            kk["shapes"].append(f"{n}x{c}x{h}x{w}_L{level}" if level is not None else f"{n}x{c}x{h}x{w}_attn")
        shapes.sort(key=lambda r: -r["total_ms"])
        kernels = sorted(kernels.values(), key=lambda r: -r["total_ms"])
        for kk in kernels:
            kk["avg_launch_ms"] = kk["total_ms"] / kk["launches"]
            kk["achieved_GBs"] = kk["algorithmic_bytes"] / (kk["total_ms"] * 1e-3) / 1e9
        return shapes, kernels


def kernel_name(plan, elem_bytes):
    """Device-kernel name rocprofv3 reports for a plan string such as 'plane(cb=64,whole-plane,nt=512,lds=...)'."""
    t = "unsigned short" if elem_bytes == 2 else "float"
    if plan.startswith("plane(cb="):
        lpp = int(plan[len("plane(cb="):].split(",")[0]) // 2
        kern = "k_recconv_whole" if "whole-plane" in plan else "k_recconv_plane"
        return f"rcx::{kern}<{lpp}, {t}>"
    if plan.startswith("recattn2d("):
        return "RecAttn2d token mixer: " + plan[len("recattn2d("):-1] + " (one unit, several launches)"
    if plan.startswith("split("):
        return "rcx split schedule: " + plan[len("split("):-1]
    if plan.startswith("lanes(k_recconv_lanes"):
        kern = plan[len("lanes("):plan.index(">")]
        return f"rcx::lanes::{kern}, {t}>"
    if plan.startswith("cpt(k_recconv_cpt"):
        kern = plan[len("cpt("):plan.index(">")]
        # template arguments after the type: training, levels (4 / 3 = the full ladder of a 56 / 28 plane; one less: "levels-1"; the 64 x 64 block on
        # 16-pixel tiles: 3), staged rows (diagnostic build), pixels per tile side (14; "ts=16" in the plan: 16)
        ts = 14
        if kern.endswith(", ts=16"):
            kern, ts = kern[:-len(", ts=16")], 16
        t_ = int(kern[len("k_recconv_cpt<"):].split(",")[0])
        lv = 3 if ts == 16 else (4 if t_ == 4 else 3) - (1 if ">,levels-1," in plan else 0)
        return f"rcx::cpt::{kern}, {t}, false, {lv}, 0, {ts}>"
    if plan.startswith("cpl(k_recconv_cpl"):
        kern = plan[len("cpl("):plan.index(">")]
        ns = "cpl14"                                                          # rcx_cpl14.hip
        if kern.startswith("k_recconv_cpl14<"):                              # template arguments after the type: x through LDS (A/B variant), levels, reload form
            xl, rl = kern.endswith(", XL"), kern.endswith(", RL")
            return (f"rcx::{ns}::{kern[:-4] if xl or rl else kern}, {t}, {'true' if xl else 'false'}, {1 if '>,levels-1,' in plan else 2}, "
                    f"{'true' if rl else 'false'}>")
        return f"rcx::{ns}::{kern}, {t}>"
    return "rcx::k_conv_generic<...> (one launch per ladder step)"


def _physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False) or 0
    except Exception:
        n = 0
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n or avail, avail))


def _cpu_rate(net, x, seconds, torch, max_iters=200):
    with torch.no_grad():
        net(x)                                                   # warm-up (mkldnn primitive creation for this thread count)
        t0 = time.perf_counter()
        iters = 0
        while iters < 1 or (time.perf_counter() - t0 < seconds and iters < max_iters):
            net(x)
            iters += 1
        return x.shape[0] * iters / (time.perf_counter() - t0), iters


def reference_model(model_name, device, dtype):
    """The model with the REFERENCE's token mixers (ATen depthwise conv2d + interpolate + add, oracle/torch_eager.py), BN-folded,
    eval: what the cpu_baseline leg times, and what `python -m recnext_amd.speed --impl ref` times."""
    from oracle.torch_eager import eager_token_mixer
    from recnext_amd import models
    from recnext_amd.speed import build_inference_model
    return build_inference_model(model_name, device, dtype, token_mixer=eager_token_mixer(models.CONFIGS[model_name]["family"]))


def cpu_baseline(model_name, resolution, seconds, torch):
    """Reference CPU path on the host cores: same skeleton, ATen token mixers (oracle/torch_eager.py), fp32, the loop of
    speed_gpu.py:11-27.  torch's default thread count (= every logical CPU of a shared host) oversubscribes badly, so the
    thread count is swept and the best is reported."""
    from recnext_amd.speed import synthetic_batch
    bs = 8
    phys = _physical_cores()
    counts = sorted({c for c in (1, 8, 32, phys) if c <= phys})
    default_threads = torch.get_num_threads()
    net = reference_model(model_name, "cpu", torch.float32)
    x = synthetic_batch(bs, resolution, "cpu", torch.float32)
    share = seconds * 0.7 / len(counts)
    sweep = {}
    for c in counts:
        torch.set_num_threads(c)
        sweep[c], _ = _cpu_rate(net, x, share, torch)
    best = max(sweep, key=sweep.get)
    # BASELINE config 1: RecNeXt-M0 forward, 224x224, batch 1, fp32 (SURVEY 8d "Config 1")
    m0 = reference_model("recnext_m0", "cpu", torch.float32)
    x1 = synthetic_batch(1, 224, "cpu", torch.float32)
    cfg1 = {}
    for c in sorted({1, min(8, phys)}):
        torch.set_num_threads(c)
        rate, _ = _cpu_rate(m0, x1, seconds * 0.15, torch, max_iters=60)
        cfg1[c] = {"images_per_s": round(rate, 2), "ms_per_image": round(1e3 / rate, 2)}
    torch.set_num_threads(default_threads)
    return {"value": sweep[best], "unit": "images/s", "cores": best, "kind": "port",
            "host_cpus": os.cpu_count(), "physical_cores": phys,
            "thread_sweep_images_per_s": {str(c): round(v, 2) for c, v in sweep.items()},
            "config1_recnext_m0_batch1_fp32": {str(c): v for c, v in cfg1.items()},
            "sample": f"forward passes of {model_name} {resolution}x{resolution}, batch {bs}, fp32 (BN-folded), token mixers = "
                      f"reference ATen ops (depthwise conv2d + interpolate + add) restated in oracle/torch_eager.py; "
                      f"{share:.1f} s per thread count in {counts}, best = {best} threads; config 1 (M0, batch 1) timed beside it"}


def load_traffic(kernel, fingerprint=None, profiles_dir=None, metric=None):
    """(HBM bytes per launch of `kernel`, source file, note) from the committed PMC profiles (profiles/*traffic*.json).  Not measured in this
    run -- PMC collection needs rocprofv3 around the process -- so a profile counts only if it was taken on THESE kernel sources: it records
    the sha256 of recnext_amd/csrc (recnext_amd/build.py::source_fingerprint) and a file without that field, or with another value, is stale:
    (None, None, reason).  The newest matching round that lists the kernel wins."""
    import glob
    import importlib.util
    if fingerprint is None:
        spec = importlib.util.spec_from_file_location("_rcx_build", os.path.join(ROOT, "recnext_amd", "build.py"))   # torch-free
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        fingerprint = mod.source_fingerprint()
    best, stale = (None, None), 0
    for p in sorted(glob.glob(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), "*traffic*.json"))):
        try:
            doc = json.load(open(p))
        except (OSError, ValueError):
            continue
        if doc.get("library_sources_sha256") != fingerprint:
            stale += 1
            continue
        for rec in doc.get("kernels", []):
            if rec.get("kernel") == kernel and rec.get("hbm_bytes_per_launch") is not None:
                # a multi-launch unit's record is a mean over the units of ONE model's forward: it is that model's only (its name does not say which)
                if rec.get("composite") and metric is not None and rec.get("bench_metric") != metric:
                    continue
                best = (rec["hbm_bytes_per_launch"], os.path.relpath(p, ROOT))
    if best[0] is None:
        return None, None, (f"no PMC profile of this kernel taken on the current kernel sources (sha256 {fingerprint[:12]}...; {stale} profile file(s) "
                            f"are from other sources): re-run tools/collect_profiles.sh")
    return best[0], best[1], None


def maybe_spawn_ranks(argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves.  Runs before torch is
    imported, so this process never touches the GPU; the ranks are fresh children, and we exit with their return code."""
    if "WORLD_SIZE" in os.environ:
        return
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    import importlib.util
    spec = importlib.util.spec_from_file_location("_rcx_launch", os.path.join(ROOT, "recnext_amd", "launch.py"))   # torch-free
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.exit(mod.spawn_ranks(n, os.path.abspath(__file__), argv))


def count_fused_channel_mixers(net, x):
    """(blocks whose x + channel_mixer(.) ran as rcx_channel_mlp_fwd in one forward of `net` on `x`, blocks in all)."""
    import torch
    from recnext_amd import models
    hits, handles = [0, 0], []

    def hook(mod, inputs, output):
        fused = mod.__dict__.get("_fused_mlp")
        hits[1] += 1
        if fused is not None and not mod.training and fused.supported(output):
            hits[0] += 1

    for m in net.modules():
        if isinstance(m, (models.MetaNeXtBlock, models.Downsample)):
            handles.append(m.register_forward_hook(hook))
    with torch.no_grad():
        net(x)
    for hnd in handles:
        hnd.remove()
    return hits[0], hits[1]


def main():
    maybe_spawn_ranks(sys.argv[1:])
    args = parse_args()
    import torch
    import recnext_amd
    from recnext_amd import models, ops
    from recnext_amd.speed import build_inference_model, synthetic_batch

    from recnext_amd import dist as rdist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: launch with "
                         "`python bench.py --gpus N` or torch.distributed.run --nproc-per-node N")
    r = rdist.init("cuda")                                  # backend "nccl" == RCCL over xGMI
    world, rank, device = r.world, r.rank, r.device

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    elem = 2 if args.dtype == "bf16" else 4
    torch.backends.cudnn.benchmark = True
    net = build_inference_model(args.model, device, dtype, seed=0)       # identical weights on every rank
    x = synthetic_batch(args.batch, args.resolution, device, dtype, seed=rank)   # this rank's shard of the global batch
    from recnext_amd.recattn import RecAttn2d
    timers = MixerTimers(net, torch, (recnext_amd.RecConv2d, RecAttn2d))

    from recnext_amd.speed import tune_gemms
    gemm_tuned = tune_gemms(net, x)                       # before the warm-up steps, outside the timed region
    # how many blocks run their channel mixer + residual as the fused HIP launch AT THIS INPUT (decided per call from the tensor's shape and dtype)
    n_fused_mlp, n_blocks = count_fused_channel_mixers(net, x)
    _fs = getattr(net, "stem", None)
    _fs = _fs.__dict__.get("_fused_stem") if _fs is not None else None
    fused_stem = _fs is not None and _fs.supported(x)
    copy_gbs = measure_copy_ceiling(torch, device) if rank == 0 else None
    plan_of = None
    with torch.no_grad():
        # warm-up: the first step builds the packs and sets the kernels' attributes; the others are bracketed mixer by mixer
        skip = 1 if args.warmup < 4 else 2                     # lazy one-time work (packs, kernel attributes, library solution look-ups) is over by then
        for i in range(args.warmup):
            timers.enabled = i >= skip
            net(x)
        timers.enabled = False
        torch.cuda.synchronize(device)
        plan_of = lambda n, c, h, w, level, k: ops.recconv2d_plan(n, c, h, w, level, k, "bilinear", dtype)
        survey_steps = max(args.warmup - skip, 0)
        survey = timers.summarize(elem, plan_of) if survey_steps else None
        if survey:                                            # timed region: only the launches of the kernel with the most time in a step
            dom_name = survey[1][0]["kernel"]
            timers.only = {(rr["C"], rr["H"], rr["W"], rr["level"], rr["k"]) for rr in survey[0] if kernel_name(rr["plan"], elem) == dom_name}
            one_kernel = {(rr["C"], rr["H"], rr["W"], rr["level"], rr["k"]) for rr in survey[0]
                          if kernel_name(rr["plan"], elem) == dom_name and rr["level"] is not None and rr["plan"].startswith(("cpt(", "cpl("))}
            per_step = sum(rr["calls"] for rr in survey[0] if (rr["C"], rr["H"], rr["W"], rr["level"], rr["k"]) in one_kernel) // max(survey_steps, 1)
            timers.prepare_exact(one_kernel, per_step * args.steps + 8, recnext_amd._lib.load())
        timers.records = []
        rdist.barrier(r)
        timers.enabled = True
        t0 = time.perf_counter()
        for _ in range(args.steps):
            net(x)
        torch.cuda.synchronize(device)
        own = time.perf_counter() - t0                        # this rank's K steps, before waiting for the others
        rdist.barrier(r)
        elapsed = rdist.max_over_ranks(r, time.perf_counter() - t0)
        per_rank = [args.batch * args.steps / t for t in rdist.gather_over_ranks(r, own)]
        timers.enabled = False

    if rank == 0:
        images = world * args.batch * args.steps
        value = images / elapsed
        timed_shapes, timed_kernels = timers.summarize(elem, plan_of)
        dom = timed_kernels[0]                                # the kernel instantiation with the most time in the step, bracketed in the timed region
        per_shape, per_kernel = survey if survey else (timed_shapes, timed_kernels)
        table_steps = survey_steps if survey else args.steps
        traffic, traffic_source, traffic_note = load_traffic(dom["kernel"], metric=f"images/sec RecNeXt-{args.model.split('_')[1].upper()} {args.resolution}x{args.resolution} {args.dtype}")
        dom_shape = next(rr for rr in timed_shapes if kernel_name(rr["plan"], elem) == dom["kernel"])
        if dom_shape["level"] is None:                        # RecAttn2d (A family): a unit of several launches, no single-kernel flop count
            dom_flops = dom_tfs = None
        else:
            dom_flops = recconv_flops(dom_shape["N"], dom_shape["C"], dom_shape["H"], dom_shape["W"], dom_shape["level"], dom_shape["k"])
            dom_tfs = dom_flops / (dom["avg_launch_ms"] * 1e-3) / 1e12
        mixer_ms_per_step = sum(rr["total_ms"] for rr in per_shape) / table_steps
        mixer_bytes = models.token_mixer_algorithmic_bytes(args.model, args.resolution, elem) * args.batch \
            if models.CONFIGS[args.model]["family"] == "m" else None
        rnd = lambda d: {k: (round(v, 5) if isinstance(v, float) else v) for k, v in d.items()}
        out = {
            "metric": f"images/sec RecNeXt-{args.model.split('_')[1].upper()} {args.resolution}x{args.resolution} {args.dtype}",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "rccl_ranks": world, "per_rank_images_per_s": [round(v, 1) for v in per_rank],
            "config": {"workload": f"{args.model} forward, BN-folded, channels_last, {args.resolution}x{args.resolution}, "
                                   f"batch {args.batch}/GPU, random-init weights, HIP token mixers"
                                   + (f", HIP channel mixers on {n_fused_mlp} of {n_blocks} blocks (x + mlp in one launch where rcx_channel_mlp_fwd has a kernel "
                                      "for the shape; the rest: GEMM library)" if n_fused_mlp else "")
                                   + (", HIP stem (one launch)" if fused_stem else "")
                                   + (", GEMM solutions picked by TunableOp" if gemm_tuned else ""),
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world} (batch-sharded replicas, no collective in the timed region)"},
            "roofline": {"bound": "hbm", "achieved": dom["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": dom["achieved_GBs"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "traffic_note": traffic_note,
                         # the measured copy ceiling of this device beside the nominal peak, and the vector-ALU ceiling
                         "peak_measured": copy_gbs, "frac_of_measured": dom["achieved_GBs"] / copy_gbs if copy_gbs else None,
                         "valu": None if dom_flops is None else
                                 {"flops_per_launch": dom_flops, "achieved_TFs": dom_tfs, "peak_TFs": VALU_PEAK_TFS, "frac": dom_tfs / VALU_PEAK_TFS,
                                  "note": "algorithmic float32 flops of the block (2 k^2 C (sum_l>=1 + sum_l>=0 H_l W_l) + 9 per resized-and-added "
                                          "element) against the vector peak; the matrix-core schedules do the same flops on the other pipe"},
                         "kernel": dom["kernel"], "shapes": dom["shapes"], "avg_launch_ms": dom["avg_launch_ms"],
                         "algorithmic_bytes_per_launch": dom["algorithmic_bytes"] / dom["launches"],
                         "launches_timed": dom["launches"],
                         "timing": ("HIP events recorded by the command processor at the kernel's start and end (hipExtLaunchKernelGGL through "
                                    "rcx_time_next_launch), every launch of this kernel inside the timed region") if timers.exact else
                                   "HIP events recorded on the launch stream around each launch of this unit inside the timed region (dispatch gaps included)",
                         "note": "algorithmic bytes = 2*N*C*H*W*b + (level+2)*C*k*k*b per launch (SURVEY 8d)"},
            "token_mixers": {"measured_over": (f"the last {survey_steps} warm-up steps (every mixer bracketed; the timed region brackets only the "
                                               "dominant kernel, an event pair costs the stream ~2.5 us)") if survey else "the timed region",
                             "ms_per_step": mixer_ms_per_step, "share_of_step": mixer_ms_per_step / (elapsed / args.steps * 1e3),
                             "algorithmic_bytes_per_step": mixer_bytes,
                             "achieved_GBs": (mixer_bytes / (mixer_ms_per_step * 1e-3) / 1e9) if mixer_bytes else None,
                             "frac_of_hbm_peak": (mixer_bytes / (mixer_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if mixer_bytes else None,
                             "images_per_s_mixers_only": args.batch / (mixer_ms_per_step * 1e-3),
                             "per_kernel": [rnd({k: v for k, v in kk.items()}) for kk in per_kernel],
                             "per_shape": [rnd(rr) for rr in per_shape]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.model, args.resolution, args.cpu_seconds, torch)
        print(json.dumps(out), flush=True)

    rdist.finish(r)


if __name__ == "__main__":
    main()
