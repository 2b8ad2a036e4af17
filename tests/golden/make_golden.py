#!/usr/bin/env python3
"""Generate the committed golden vectors by IMPORTING the reference (build container only).

    python tests/golden/make_golden.py [--reference /root/reference]

The reference is pure Python on top of torch; it cannot travel to the GPU box, so this
script runs it HERE (torch CPU) on seeded inputs and stores inputs + expected outputs as
small ``.npz`` fixtures next to this file.  ``timm`` is not installed, so the five names
the reference imports from it are provided by an in-memory shim;
``RecConv2d`` / ``RecAttn2d`` themselves only need torch.

Fixtures are data only (inputs, weights, expected outputs, integer known-answers).
"""
import argparse
import copy
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))


def install_timm_shim():
    registry = {}

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            if not self.training or self.p == 0.0:
                return x
            keep = 1 - self.p
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x * mask / keep

    def register_model(fn):
        registry[fn.__name__] = fn
        return fn

    def create_model(name, **kw):
        return registry[name](**kw)

    def build_model_with_cfg(cls, variant, pretrained, feature_cfg=None, **kw):
        return cls(**kw)

    timm = types.ModuleType("timm")
    layers = types.ModuleType("timm.layers")
    models = types.ModuleType("timm.models")
    layers.trunc_normal_ = nn.init.trunc_normal_
    layers.DropPath = DropPath
    models.register_model = register_model
    models.create_model = create_model
    models.build_model_with_cfg = build_model_with_cfg
    models.generate_default_cfgs = lambda d: d
    timm.layers, timm.models, timm.create_model = layers, models, create_model
    sys.modules.update({"timm": timm, "timm.layers": layers, "timm.models": models})
    return registry


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def np32(t):
    return t.detach().to(torch.float32).contiguous().numpy()


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


# (name, N, C, H, W, level, k, mode, bias, seed, keep_intermediates)
RECCONV_CASES = [
    ("l1_7x7", 2, 8, 7, 7, 1, 5, "bilinear", False, 0, True),
    ("l2_14x14", 2, 8, 14, 14, 2, 5, "bilinear", False, 1, True),
    ("l3_28x28", 1, 8, 28, 28, 3, 5, "bilinear", False, 2, True),
    ("l4_56x56", 1, 8, 56, 56, 4, 5, "bilinear", False, 3, True),
    ("l1_16x16_even", 1, 8, 16, 16, 1, 5, "bilinear", False, 4, True),
    ("l4_128x128_even", 1, 8, 128, 128, 4, 5, "bilinear", False, 5, False),
    ("l2_25x13_bias", 1, 8, 25, 13, 2, 5, "bilinear", True, 6, True),
    ("l2_14x14_k3", 1, 8, 14, 14, 2, 3, "bilinear", False, 7, True),
    ("l2_14x14_k7", 1, 8, 14, 14, 2, 7, "bilinear", False, 8, True),
    ("l2_14x14_nearest", 1, 8, 14, 14, 2, 5, "nearest", False, 9, True),
    ("l0_9x9", 1, 8, 9, 9, 0, 5, "bilinear", False, 10, True),
    # channel counts of the real configs (non power-of-two slabs) and a nearest/bias/odd mix
    ("l3_28x28_c96", 1, 96, 28, 28, 3, 5, "bilinear", False, 11, False),
    ("l4_56x56_c40", 1, 40, 56, 56, 4, 5, "bilinear", False, 12, False),
    ("l2_15x22_nearest_bias", 2, 24, 15, 22, 2, 5, "nearest", True, 13, False),
    ("l1_7x7_c512", 1, 512, 7, 7, 1, 5, "bilinear", False, 14, False),
    ("l5_40x40", 1, 16, 40, 40, 5, 5, "bilinear", False, 15, False),
    ("l3_5x3_tiny", 1, 8, 5, 3, 3, 5, "bilinear", True, 16, True),
]


def gen_recconv(ref, out):
    for (name, n, c, h, w, level, k, mode, bias, seed, keep) in RECCONV_CASES:
        torch.manual_seed(seed)
        mod = ref.RecConv2d(c, kernel_size=k, bias=bias, level=level, mode=mode).eval()
        x = torch.randn(n, c, h, w)
        with torch.no_grad():
            y = mod(x)
            # bf16 target: fp32 forward on bf16-rounded inputs / weights
            modr = ref.RecConv2d(c, kernel_size=k, bias=bias, level=level, mode=mode).eval()
            modr.load_state_dict({kk: bf16_round(v) for kk, v in mod.state_dict().items()})
            y_r = modr(bf16_round(x))
            # the reference's own bf16 forward (informational)
            y_bf = copy.deepcopy(mod).to(torch.bfloat16)(x.to(torch.bfloat16)).to(torch.float32)
        rec = {
            "x": np32(x), "y": np32(y), "y_bf16in_f32": np32(y_r),
            "w_down": np32(mod.down.weight),
            "w_convs": np.stack([np32(cv.weight) for cv in mod.convs]),
            "meta": np.array(json.dumps(dict(N=n, C=c, H=h, W=w, level=level, k=k, mode=mode,
                                             bias=bias, seed=seed))),
        }
        if bias:
            rec["b_down"] = np32(mod.down.bias)
            rec["b_convs"] = np.stack([np32(cv.bias) for cv in mod.convs])
        if keep:
            rec["y_ref_bf16"] = np32(y_bf)
            with torch.no_grad():
                feats, cur = [], x
                for _ in range(level):
                    cur = mod.down(cur)
                    feats.append(cur)
                u = 0
                sizes = [x.shape[2:]] + [f.shape[2:] for f in feats]
                for j, l in enumerate(range(level, 0, -1)):
                    u = nn.functional.interpolate(mod.convs[j](feats[l - 1] + u), size=sizes[l - 1], mode=mode)
                    rec[f"U{l}"] = np32(u)
                for l, f in enumerate(feats, 1):
                    rec[f"F{l}"] = np32(f)
        np.savez(os.path.join(out, f"recconv_{name}.npz"), **rec)
        print("recconv", name, tuple(y.shape), f"bf16-vs-f32 maxabs={float((y_bf - y).abs().max()):.4f}")


# gradients of the reference block itself (autograd through model/recnext.py:24-34): (name, N, C, H, W, level, k, mode, bias, seed)
RECCONV_GRAD_CASES = [
    ("l1_7x7", 2, 8, 7, 7, 1, 5, "bilinear", False, 40),
    ("l2_14x14", 2, 8, 14, 14, 2, 5, "bilinear", False, 41),
    ("l3_28x28", 1, 8, 28, 28, 3, 5, "bilinear", False, 42),
    ("l2_25x13_nearest_bias", 1, 8, 25, 13, 2, 5, "nearest", True, 43),
    ("l4_56x56_c16", 1, 16, 56, 56, 4, 5, "bilinear", False, 44),
]


def gen_recconv_grads(ref, out):
    for (name, n, c, h, w, level, k, mode, bias, seed) in RECCONV_GRAD_CASES:
        torch.manual_seed(seed)
        mod = ref.RecConv2d(c, kernel_size=k, bias=bias, level=level, mode=mode).train()
        x = torch.randn(n, c, h, w, requires_grad=True)
        gy = torch.randn(n, c, h, w)
        with torch.enable_grad():
            y = mod(x)
            y.backward(gy)
        rec = {
            "x": np32(x), "gy": np32(gy), "y": np32(y), "gx": np32(x.grad),
            "w_down": np32(mod.down.weight), "gw_down": np32(mod.down.weight.grad),
            "w_convs": np.stack([np32(cv.weight) for cv in mod.convs]),
            "gw_convs": np.stack([np32(cv.weight.grad) for cv in mod.convs]),
            "meta": np.array(json.dumps(dict(N=n, C=c, H=h, W=w, level=level, k=k, mode=mode, bias=bias, seed=seed))),
        }
        if bias:
            rec.update({"b_down": np32(mod.down.bias), "gb_down": np32(mod.down.bias.grad),
                        "b_convs": np.stack([np32(cv.bias) for cv in mod.convs]),
                        "gb_convs": np.stack([np32(cv.bias.grad) for cv in mod.convs])})
        np.savez(os.path.join(out, f"grad_recconv_{name}.npz"), **rec)
        print("recconv grads", name, f"|gx|max={float(x.grad.abs().max()):.3f} |gw_down|max={float(mod.down.weight.grad.abs().max()):.3f}")


def randomize_bn(module, gen):
    for m in module.modules():
        if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.3)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) * 0.8 + 0.4)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=gen) * 0.8 + 0.6)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=gen) * 0.2)


def gen_recattn(refa, utils, out):
    # (name, dim, stage, H, W, seed) ; num_heads = 2**(stage+1) as at model/recattn.py:166
    # the last four are the token mixers of RecNeXt-A3 at 224x224 (BASELINE config 4): dim / stage / plane of model/recattn.py:403
    for (name, dim, stage, h, w, seed, n) in [("la1_14x14", 16, 1, 14, 14, 20, 2), ("la2_7x7", 32, 3, 7, 7, 21, 2),
                                              ("la1_9x12", 16, 0, 9, 12, 22, 2),
                                              ("a3s0_56x56", 64, 0, 56, 56, 23, 1), ("a3s1_28x28", 128, 1, 28, 28, 24, 1),
                                              ("a3s2_14x14", 256, 2, 14, 14, 25, 2), ("a3s3_7x7", 512, 3, 7, 7, 26, 2)]:
        torch.manual_seed(seed)
        gen = torch.Generator().manual_seed(seed)
        heads = 2 ** (stage + 1)
        mod = refa.RecAttn2d(dim, num_heads=heads, stage=stage).eval()
        randomize_bn(mod, gen)
        x = torch.randn(n, dim, h, w)
        with torch.no_grad():
            y_unfused = mod(x)
            utils.replace_batchnorm(mod)
            y = mod(x)
            d = mod.down[0](x)                      # stride-2 depthwise (+folded BN)
            a = mod.down[1](d)                      # linear attention output
            # the bf16 targets (round 3): the float32 module on bf16-rounded inputs, and the reference's OWN bf16 run of the same
            # inputs -- its distance from the float32 result is the yardstick for the HIP path's bf16 error (BASELINE config 4)
            xr = bf16_round(x)
            y_r = mod(xr)
            import copy
            y_b = copy.deepcopy(mod).bfloat16()(xr.bfloat16()).float()
        la = mod.down[1]
        rec = {
            "x": np32(x), "y": np32(y), "y_unfused": np32(y_unfused), "down_out": np32(d), "attn_out": np32(a),
            "y_bf16in_f32": np32(y_r), "y_bf16": np32(y_b),
            "w_down": np32(mod.down[0].weight), "b_down": np32(mod.down[0].bias),
            "w_conv": np32(mod.conv.weight), "b_conv": np32(mod.conv.bias),
            "w_qk": np32(la.qk.weight), "b_qk": np32(la.qk.bias),
            "w_pe": np32(la.pe.weight), "b_pe": np32(la.pe.bias),
            "meta": np.array(json.dumps(dict(dim=dim, stage=stage, heads=heads, H=h, W=w, seed=seed,
                                             variant=2 if stage >= 3 else 1))),
        }
        np.savez(os.path.join(out, f"recattn_{name}.npz"), **rec)
        print("recattn", name, tuple(y.shape), f"fuse drift={float((y - y_unfused).abs().max()):.2e}")


def gen_tiny_model(ref, refa, utils, out):
    for fam, modref, cls in (("m", ref, ref.RecNext), ("a", refa, refa.RecNext)):
        torch.manual_seed(30)
        gen = torch.Generator().manual_seed(30)
        net = cls(embed_dim=(8, 16, 32, 64), depth=(1, 1, 1, 1), num_classes=10).eval()
        randomize_bn(net, gen)
        x = torch.randn(2, 3, 64, 64)
        with torch.no_grad():
            logits = net(x)
            sd = {k: np32(v) for k, v in net.state_dict().items() if v.dtype.is_floating_point}
            utils.replace_batchnorm(net)
            logits_fused = net(x)
        rec = {"x": np32(x), "logits": np32(logits), "logits_fused": np32(logits_fused)}
        rec.update({"sd::" + k: v for k, v in sd.items()})
        np.savez(os.path.join(out, f"tiny_model_{fam}.npz"), **rec)
        print("tiny model", fam, np32(logits)[0, :3], f"fuse drift={float((logits - logits_fused).abs().max()):.2e}")


def gen_kats(registry, utils, out):
    kat = {}
    for name in sorted(registry):
        net = registry[name]()
        n_params = sum(p.numel() for p in net.parameters() if p.requires_grad)
        utils.replace_batchnorm(net)
        n_fused = sum(p.numel() for p in net.parameters())
        kat[name] = {"n_parameters": n_params, "n_parameters_fused": n_fused}
        print("kat", name, n_params, n_fused)
    with open(os.path.join(out, "param_counts.json"), "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)


def gen_interp_tables(out):
    rec = {}
    for (n_in, n_out) in [(4, 7), (7, 14), (13, 25), (8, 16), (2, 3), (1, 1), (1, 2), (3, 5), (14, 28), (28, 56)]:
        v = torch.arange(n_in, dtype=torch.float32).view(1, 1, 1, n_in)
        rows = torch.eye(n_in, dtype=torch.float32).view(n_in, 1, 1, n_in)  # one-hot probes -> weight matrix
        wmat = nn.functional.interpolate(rows.expand(n_in, 1, 2, n_in).contiguous(), size=(2, n_out), mode="bilinear")[:, 0, 0]
        near = nn.functional.interpolate(v, size=(1, n_out), mode="nearest")[0, 0, 0]
        rec[f"bilinear_{n_in}_{n_out}"] = np32(wmat)          # (n_in, n_out) interpolation matrix
        rec[f"nearest_{n_in}_{n_out}"] = near.numpy().astype(np.int64)
    np.savez(os.path.join(out, "interp_tables.npz"), **rec)
    print("interp tables written")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=HERE)
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    torch.set_num_threads(4)
    registry = install_timm_shim()
    ref = load_by_path("ref_recnext", os.path.join(args.reference, "model", "recnext.py"))
    refa = load_by_path("ref_recattn", os.path.join(args.reference, "model", "recattn.py"))
    utils = load_by_path("ref_utils", os.path.join(args.reference, "utils.py"))
    gen_interp_tables(args.out)
    gen_recconv(ref, args.out)
    gen_recconv_grads(ref, args.out)
    gen_recattn(refa, utils, args.out)
    gen_tiny_model(ref, refa, utils, args.out)
    gen_kats(registry, utils, args.out)


if __name__ == "__main__":
    main()
