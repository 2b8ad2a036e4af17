"""Model skeleton: parameter-count known answers, state-dict compatibility, BN folding, tiny-model logits.

CPU tests host the oracle's ATen token mixers in the product skeleton (the HIP modules refuse CPU tensors);
GPU tests run the same fixtures through the HIP token mixers.
"""
import json
import os

import numpy as np
import pytest
import torch

from recnext_amd import models
from oracle.torch_eager import eager_token_mixer
from tests.util import GOLDEN

KAT = json.load(open(os.path.join(GOLDEN, "param_counts.json")))


@pytest.mark.parametrize("name", sorted(models.CONFIGS))
def test_parameter_counts_match_reference_logs(name):
    # first JSON line of logs/{normal,distill}/recnext_*.txt (n_parameters) and README "Params" after fusion
    net = models.create_model(name)
    assert sum(p.numel() for p in net.parameters() if p.requires_grad) == KAT[name]["n_parameters"]
    models.replace_batchnorm(net)
    assert sum(p.numel() for p in net.parameters()) == KAT[name]["n_parameters_fused"]


def _tiny(fam, token_mixer=None):
    return models.RecNext(family=fam, embed_dim=(8, 16, 32, 64), depth=(1, 1, 1, 1), num_classes=10, token_mixer=token_mixer)


def _load_tiny(fam):
    d = np.load(os.path.join(GOLDEN, f"tiny_model_{fam}.npz"))
    sd = {k[4:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd::")}
    return d, sd


@pytest.mark.parametrize("fam", ["m", "a"])
def test_tiny_model_logits_cpu_with_eager_token_mixers(fam):
    d, sd = _load_tiny(fam)
    net = _tiny(fam, eager_token_mixer(fam)).eval()
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    x = torch.from_numpy(d["x"])
    with torch.no_grad():
        assert np.abs(net(x).numpy() - d["logits"]).max() < 1e-5
        models.replace_batchnorm(net)
        assert np.abs(net(x).numpy() - d["logits_fused"]).max() < 1e-5
    # replace_batchnorm leaves the plain BatchNorm2d of MetaNeXtBlock / Downsample in place (utils.py:227-234)
    n_bn = sum(isinstance(m, torch.nn.BatchNorm2d) for m in net.modules())
    assert n_bn == (4 + 3 if fam == "m" else 3)


def test_hip_skeleton_has_reference_state_dict_keys():
    for fam in ("m", "a"):
        _, sd = _load_tiny(fam)
        keys = {k for k in _tiny(fam).state_dict() if not k.endswith("num_batches_tracked")}
        assert keys == set(sd)


def test_downsample_reroute_follows_replaced_children():
    """use_hip_downsample keeps direct references to token_mixer and norm; a child replaced afterwards (SyncBatchNorm conversion, a
    fusion pass, `m.norm = ...`) must be the one that runs, and a module pickled before the reroute existed must still run."""
    from recnext_amd.models import Downsample
    m = Downsample(8, 2, torch.nn.GELU).eval()
    net = torch.nn.Sequential(m)
    assert models.use_hip_downsample(net) == 1 and m._hip is not None
    x = torch.randn(2, 8, 8, 8)
    m.norm = torch.nn.GroupNorm(4, 16)                          # not a BatchNorm2d: the reroute is dropped, the new norm runs (on the CPU here)
    want = m.norm(m.token_mixer(x))
    want = want + m.channel_mixer(want)
    assert torch.allclose(m(x), want) and m._hip is None
    m.norm = torch.nn.BatchNorm2d(16).eval()                    # a fresh BatchNorm2d: rerouted again, around the NEW module
    models.use_hip_downsample(net)
    old = m._hip
    m.norm = torch.nn.BatchNorm2d(16).eval()
    assert m._hip_path() is not old and m._hip_path().norm is m.norm and m._hip_path().token_mixer is m.token_mixer
    del m.__dict__["_hip"]                                      # a whole-model pickle from before the attribute existed
    assert m._hip_path() is None


def test_token_mixer_shapes_and_algorithmic_bytes():
    assert models.token_mixer_shapes("recnext_m3") == [(64, 56, 56, 4, 3), (128, 28, 28, 3, 3), (256, 14, 14, 2, 13), (512, 7, 7, 1, 2)]
    # SURVEY 8d / BASELINE.md section 3 (MB per image)
    assert abs(models.token_mixer_algorithmic_bytes("recnext_m3") / 1e6 - 7.395) < 1e-3
    assert abs(models.token_mixer_algorithmic_bytes("recnext_m1") / 1e6 - 5.924) < 1e-3
    assert abs(models.token_mixer_algorithmic_bytes("recnext_m5") / 1e6 - 22.449) < 1e-3
    assert abs(models.token_mixer_algorithmic_bytes("recnext_m3", 512) / 1e6 - 34.527) < 1e-3
    assert abs(models.token_mixer_algorithmic_bytes("recnext_m0", elem_bytes=4) / 1e6 - 6.194) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("fam", ["m", "a"])
def test_tiny_model_logits_gpu_with_hip_token_mixers(fam):
    d, sd = _load_tiny(fam)
    dev = torch.device("cuda:0")
    net = _tiny(fam).eval()
    net.load_state_dict(sd, strict=False)
    net = net.to(dev).to(memory_format=torch.channels_last)
    x = torch.from_numpy(d["x"]).to(dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        assert np.abs(net(x).cpu().numpy() - d["logits"]).max() < 1e-4
        models.replace_batchnorm(net)
        assert np.abs(net(x).cpu().numpy() - d["logits_fused"]).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name,batch", [("recnext_m0", 2), ("recnext_a0", 2), ("recnext_a3", 4), ("recnext_m3", 2), ("recnext_m1", 2), ("recnext_m5", 2)])
def test_full_model_hip_vs_eager_gpu(name, batch):
    """Whole registered model at 224: HIP token mixers vs the ATen restatement, same weights, fp32 and bf16.
    recnext_a3 is BASELINE config 4, recnext_m3 the headline model, recnext_m1 / recnext_m5 configs 2 and 3 (per GPU)."""
    dev = torch.device("cuda:0")
    fam = models.CONFIGS[name]["family"]
    torch.manual_seed(0)
    ref = models.create_model(name, token_mixer=eager_token_mixer(fam)).eval()
    for m in ref.modules():                                       # non-trivial BN statistics
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    net = models.create_model(name).eval()
    net.load_state_dict(ref.state_dict(), strict=True)
    models.replace_batchnorm(ref)
    models.replace_batchnorm(net)
    ref, net = ref.to(dev), net.to(dev).to(memory_format=torch.channels_last)
    x = torch.randn(batch, 3, 224, 224, device=dev)
    with torch.no_grad():
        a, b = ref(x), net(x.contiguous(memory_format=torch.channels_last))
        assert (a - b).abs().max() < 1e-3 * max(1.0, float(a.abs().max()))
        ab = ref.bfloat16()(x.bfloat16()).float()
        bb = net.bfloat16()(x.bfloat16().contiguous(memory_format=torch.channels_last)).float()
    # bf16 end-to-end: both paths round everywhere outside the token mixer; compare loosely to the fp32 logits
    scale = float(a.abs().max())
    assert (bb - a).abs().max() < 0.1 * scale + 0.05
    assert (bb - a).abs().max() <= 1.5 * (ab - a).abs().max() + 0.02 * scale


@pytest.mark.gpu
def test_full_model_at_512_hip_vs_eager_gpu():
    """BASELINE config 5: RecNeXt-M3 on a 512 x 512 input (the detection backbone's resolution, detection/recnext.py:11-36; token-mixer planes
    128 / 64 / 32 / 16: the split schedule and the 16 * 2^k kernels), HIP token mixers against the ATen restatement with the same weights."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ref = models.create_model("recnext_m3", token_mixer=eager_token_mixer("m")).eval()
    for m in ref.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    net = models.create_model("recnext_m3").eval()
    net.load_state_dict(ref.state_dict(), strict=True)
    models.replace_batchnorm(ref)
    models.replace_batchnorm(net)
    ref, net = ref.to(dev), net.to(dev).to(memory_format=torch.channels_last)
    x = torch.randn(2, 3, 512, 512, device=dev)
    with torch.no_grad():
        a, b = ref(x), net(x.contiguous(memory_format=torch.channels_last))
        assert (a - b).abs().max() < 1e-3 * max(1.0, float(a.abs().max()))
        ab = ref.bfloat16()(x.bfloat16()).float()
        bb = net.bfloat16()(x.bfloat16().contiguous(memory_format=torch.channels_last)).float()
    scale = float(a.abs().max())
    assert (bb - a).abs().max() < 0.1 * scale + 0.05
    assert (bb - a).abs().max() <= 1.5 * (ab - a).abs().max() + 0.02 * scale
    # the plans of the four stage shapes at this resolution (what the forward above ran)
    from recnext_amd import ops
    plans = [ops.recconv2d_plan(2, c, hw, hw, lv, 5, "bilinear", torch.bfloat16) for c, hw, lv in ((64, 128, 4), (128, 64, 3), (256, 32, 2), (512, 16, 1))]
    # 128 x 128: split (whose inner 64 x 64 block is the 16-pixel-tile kernel, in float32); 64 x 64: that kernel itself (round 5); 32 x 32 and 16 x 16: lanes
    assert plans[0].startswith("split(") and "ts=16" in plans[0] and plans[1].startswith("cpt(k_recconv_cpt<4, 4, 0, 0, ts=16>") and \
        all(p.startswith("lanes(") for p in plans[2:]), plans


def test_fold_token_mixer_norms_counts_and_is_noop_for_other_mixers():
    net = models.create_model("recnext_m3").eval()
    models.replace_batchnorm(net)
    assert models.fold_token_mixer_norms(net) == 21
    assert sum(isinstance(m, torch.nn.BatchNorm2d) for m in net.modules()) == 3      # the three Downsample norms remain
    eager = models.create_model("recnext_m0", token_mixer=eager_token_mixer("m")).eval()
    assert models.fold_token_mixer_norms(eager) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_folding_the_block_norm_preserves_the_function_gpu(dtype):
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    net = models.create_model("recnext_m0").eval()
    for m in net.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    models.replace_batchnorm(net)
    net = net.to(dev).to(memory_format=torch.channels_last)
    x = torch.randn(2, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ref = net(x)                                            # fp32, norms still separate
        assert models.fold_token_mixer_norms(net) == 14
        got32 = net(x)
        assert (got32 - ref).abs().max() < 1e-4 * max(1.0, float(ref.abs().max()))
        if dtype == torch.bfloat16:
            gotb = net.bfloat16()(x.bfloat16()).float()
            assert (gotb - ref).abs().max() < 0.1 * float(ref.abs().max()) + 0.05


def test_linear_pointwise_is_the_same_function_cpu():
    torch.manual_seed(0)
    net = models.create_model("recnext_m0", token_mixer=eager_token_mixer("m")).eval()
    models.replace_batchnorm(net)
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        ref = net(x)
        keys = set(net.state_dict())
        assert models.use_linear_pointwise(net) == 2 * (14 + 3)
        assert set(net.state_dict()) == keys                      # same parameter names
        assert (net(x) - ref).abs().max() < 1e-5


def test_padded_mlp_hidden_width_is_the_same_function_cpu():
    """pad_mlp_hidden: RecNeXt-A3's channel mixers run their GEMMs at 128 / 256 / 512 instead of 120 / 240 / 480 -- same outputs, same state_dict,
    and the padded operands follow a parameter that changes afterwards."""
    assert [models.padded_hidden_width(h) for h in (120, 240, 480, 960, 150, 1200, 80, 128)] == [128, 256, 512, 960, 160, 1216, 80, 128]
    torch.manual_seed(0)
    net = models.create_model("recnext_a3", token_mixer=eager_token_mixer("a")).eval()
    models.replace_batchnorm(net)
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        ref = net(x)
        keys = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        models.use_linear_pointwise(net)
        assert models.pad_mlp_hidden(net) == 3 + 4 + 14 and models.pad_mlp_hidden(net) == 0          # stages 0 - 2 incl. two Downsample mixers; idempotent
        assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == keys
        fc1, act, fc2 = net.stages[0].blocks[0].channel_mixer
        assert (fc1.out_channels, fc1.out_padded, fc2.in_channels, fc2.in_padded) == (120, 128, 120, 128)
        assert (net(x) - ref).abs().max() < 1e-6
        fc1.weight.mul_(2.0)                                       # an optimizer step / a loaded checkpoint: the padded copy is rebuilt
        changed = net(x)
        assert (changed - ref).abs().max() > 1e-4
        fc1.pad()
        fc2.pad()
        assert (net(x) - changed).abs().max() < 1e-6
        with pytest.raises(ValueError):
            fc1.pad(out_to=64)


def test_use_fused_mlp_marks_the_blocks_and_keeps_the_library_path_on_cpu():
    """use_fused_mlp only attaches the fused path (decided per call: bf16 on a GPU with a kernel for the shape); on the CPU the forward is unchanged."""
    torch.manual_seed(0)
    net = models.create_model("recnext_m0", token_mixer=eager_token_mixer("m")).eval()
    models.replace_batchnorm(net)
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        ref = net(x)
        keys = list(net.state_dict())
        assert models.use_fused_mlp(net) == 0                      # plain convs: nothing to fuse before use_linear_pointwise
        models.use_linear_pointwise(net)
        assert models.use_fused_mlp(net) == 14 + 3 and models.use_fused_mlp(net) == 0
        assert list(net.state_dict()) == keys
        assert not net.stages[0].blocks[0]._fused_mlp.supported(torch.zeros(1, 40, 16, 16))
        assert (net(x) - ref).abs().max() < 1e-5


def test_use_fused_stem_marks_the_stem_and_keeps_the_library_path_on_cpu():
    """use_fused_stem only attaches the fused path (bf16 on a GPU, decided per call); on the CPU the forward is unchanged; pack_stem's fragment layout."""
    from recnext_amd import ops
    torch.manual_seed(0)
    net = models.create_model("recnext_m0", token_mixer=eager_token_mixer("m")).eval()
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        assert models.use_fused_stem(net) == 0                     # ConvNorm pairs: nothing to fuse before replace_batchnorm
        models.replace_batchnorm(net)
        ref = net(x)
        keys = list(net.state_dict())
        assert models.use_fused_stem(net) == 1 and models.use_fused_stem(net) == 0
        assert list(net.state_dict()) == keys and not net.stem._fused_stem.supported(x)
        assert (net(x) - ref).abs().max() < 1e-6
    cm, co = 20, 40
    w1 = torch.arange(cm * 27, dtype=torch.float32).reshape(cm, 3, 3, 3) % 127
    w2 = (torch.arange(co * cm * 9, dtype=torch.float32).reshape(co, cm, 3, 3) * 3) % 113
    w1p, b1p, w2f, b2p = ops.pack_stem(w1, torch.arange(cm, dtype=torch.float32), w2, -torch.arange(co, dtype=torch.float32))
    assert w1p.numel() == 2 * 512 and b1p.numel() == 32 and w2f.numel() == 2 * 18 * 512 and b2p.numel() == 64
    f1 = w1p.float().view(2, 64, 8)                                 # [ks][lane (h, m)][j] = w1[m][k = 16 ks + 8 h + j], k = (dy 3 + dx) 3 + c
    for (ks, lane, j) in [(0, 0, 0), (1, 37, 2), (1, 45, 5), (0, 19, 7), (1, 63, 7)]:
        m, k = lane % 32, 16 * ks + 8 * (lane // 32) + j
        want = w1[m, k % 3, k // 9, (k // 3) % 3] if m < cm and k < 27 else 0
        assert f1[ks, lane, j] == want
    f2 = w2f.float().view(2, 18, 64, 8)                             # [mt][ks = tap 2 + cg][lane][j] = w2[32 mt + m][16 cg + 8 h + j][tap]
    for (mt, ks, lane, j) in [(0, 0, 0, 0), (1, 17, 39, 3), (0, 9, 33, 4), (1, 4, 7, 7), (0, 5, 50, 6)]:
        m, tap, ch = 32 * mt + lane % 32, ks // 2, 16 * (ks % 2) + 8 * (lane // 32) + j
        want = w2[m, ch, tap // 3, tap % 3] if m < co and ch < cm else 0
        assert f2[mt, ks, lane, j] == want
    assert torch.equal(b2p[:co], -torch.arange(co, dtype=torch.float32)) and b2p[co:].abs().sum() == 0


def test_fused_paths_follow_replaced_layers_and_gradients_cpu(monkeypatch):
    """The fused channel-mixer / stem launch is taken only while it still describes the module (a layer or the activation replaced after use_fused_*
    is followed, not ignored), the operands agree in dtype, and nothing on the way wants a gradient: FusedChannelMlp.usable / FusedStem.usable."""
    from recnext_amd.layers import FusedChannelMlp, FusedStem
    torch.manual_seed(0)
    net = models.create_model("recnext_m0", token_mixer=eager_token_mixer("m")).eval()
    models.replace_batchnorm(net)
    models.use_linear_pointwise(net)
    models.use_fused_mlp(net)
    models.use_fused_stem(net)
    monkeypatch.setattr(FusedChannelMlp, "supported", lambda self, x: True)          # isolate the checks from "is there a kernel on this device"
    monkeypatch.setattr(FusedStem, "supported", lambda self, x: True)
    blk = net.stages[0].blocks[0]
    fused, seq = blk._fused_mlp, blk.channel_mixer
    x = torch.randn(1, 40, 8, 8)
    for p in net.parameters():
        p.requires_grad_(False)
    with torch.no_grad():
        assert fused.usable(seq, x, x)
    assert fused.usable(seq, x, x)                                  # grad mode on, but nothing requires grad
    assert not fused.usable(seq, x, x.clone().requires_grad_(True))   # an input gradient is wanted: the autograd path
    seq[0].weight.requires_grad_(True)
    assert not fused.usable(seq, x, x)                              # a parameter gradient is wanted
    with torch.no_grad():
        assert fused.usable(seq, x, x)
        assert not fused.usable(seq, x.double(), x)                 # t and x disagree in dtype
        act = seq[1]
        seq[1] = torch.nn.ReLU()
        assert not fused.usable(seq, x, x)                          # the activation was replaced
        seq[1] = torch.nn.GELU(approximate="tanh")
        assert not fused.usable(seq, x, x)
        seq[1] = act
        assert fused.usable(seq, x, x)
        old = seq[2]
        seq[2] = type(old)(torch.nn.Conv2d(old.in_channels, old.out_channels, 1))
        assert not fused.usable(seq, x, x)                          # a layer was replaced
        seq[2] = old
        stem = net.stem
        img = torch.randn(1, 3, 32, 32)
        assert stem._fused_stem.usable(stem.stem, img)
        stem.stem[1] = torch.nn.ReLU()
        assert not stem._fused_stem.usable(stem.stem, img)
    assert not stem._fused_stem.usable(stem.stem, img.clone().requires_grad_(True))


@pytest.mark.gpu
def test_eval_mode_input_gradients_pass_through_the_fused_blocks_gpu():
    """eval() outside no_grad is legal: with the fused channel mixers and stem attached, an input gradient must be the one the unfused model gives
    (the fused launches have no grad_fn: they may only run when no gradient is wanted)."""
    from recnext_amd.speed import build_inference_model, synthetic_batch
    dev = torch.device("cuda:0")
    nets = [build_inference_model("recnext_m0", dev, torch.bfloat16, seed=3, fold_mixer_norm=False, fused_mlp=f, fused_stem=f) for f in (True, False)]
    for n_ in nets:
        for p in n_.parameters():
            p.requires_grad_(False)
    x = synthetic_batch(2, 64, dev, torch.bfloat16, seed=1)
    grads, outs = [], []
    for n_ in nets:
        xi = x.clone().requires_grad_(True)
        y = n_(xi)
        y.float().square().sum().backward()
        assert xi.grad is not None and torch.isfinite(xi.grad.float()).all() and float(xi.grad.float().abs().max()) > 0
        grads.append(xi.grad.float())
        outs.append(y.float())
    assert torch.equal(outs[0], outs[1]) and torch.equal(grads[0], grads[1])       # the same operators ran: the fused launches stood aside
    with torch.no_grad():
        y_fused = nets[0](x).float()                                 # ... and without a gradient in sight the fused launches run (close, not equal)
    assert float((y_fused - outs[1]).abs().max()) <= 2e-2 * max(1.0, float(outs[1].abs().max()))


def test_channel_mlp_pack_layout_cpu():
    """pack_channel_mlp: every weight lands in the fragment slot the kernel's lane reads it from (rcx_mlp.hip), padding is zeros."""
    from recnext_amd import ops
    c, h0 = 40, 80
    w1 = torch.arange(h0 * c, dtype=torch.float32).reshape(h0, c) % 251
    w2 = (torch.arange(c * h0, dtype=torch.float32).reshape(c, h0) * 7) % 241
    b1, b2 = torch.arange(h0, dtype=torch.float32), -torch.arange(c, dtype=torch.float32)
    wfrag, bias, hp = ops.pack_channel_mlp(w1, b1, w2, b2)
    ks1, ht, ct = 3, 3, 2
    assert hp == 96 and wfrag.dtype == torch.bfloat16 and wfrag.numel() == (ht * ks1 + ct * 2 * ht) * 512 and bias.numel() == 32 * (ht + ct)
    f = wfrag.float().view(-1, 64, 8)
    for (t, ks, lane, j) in [(0, 0, 0, 0), (1, 2, 37, 5), (2, 1, 63, 7), (2, 2, 31, 3)]:
        row, col = 32 * t + lane % 32, 16 * ks + 8 * (lane // 32) + j
        assert f[t * (ks1 + 2 * ct) + ks, lane, j] == (w1[row, col] if row < h0 and col < c else 0)
    for (tc, t, q, lane, j) in [(0, 0, 0, 0, 0), (1, 2, 1, 40, 6), (0, 1, 1, 33, 2), (1, 0, 0, 31, 7)]:
        row, col = 32 * tc + lane % 32, 32 * t + ops._mlp_acc_unit(8 * q + j, lane // 32)
        assert f[t * (ks1 + 2 * ct) + ks1 + 2 * tc + q, lane, j] == (0.5 * w2[row, col] if row < c and col < h0 else 0)      # stored halved: the kernel's hidden layer is 2 gelu(.)
    assert torch.equal(bias[:h0], b1) and bias[h0:96].abs().sum() == 0 and torch.equal(bias[96:96 + c], b2) and bias[96 + c:].abs().sum() == 0


def test_hip_downsample_keeps_the_reference_state_dict_keys():
    """use_hip_downsample reroutes the forward only: keys equal the reference's and strict loads work in both orders."""
    _, sd = _load_tiny("m")
    a = _tiny("m")
    before = dict(a.state_dict())
    assert models.use_hip_downsample(a) == 3
    assert list(a.state_dict()) == list(before)
    assert {k for k in a.state_dict() if not k.endswith("num_batches_tracked")} == set(sd)
    assert "stages.1.downsample.token_mixer.weight" in a.state_dict() and "stages.1.downsample.norm.running_var" in a.state_dict()
    b = _tiny("m")
    b.load_state_dict(a.state_dict(), strict=True)                # transformed -> plain
    c = _tiny("m")
    models.use_hip_downsample(c)
    c.load_state_dict(b.state_dict(), strict=True)                # plain -> transformed
    assert models.use_hip_downsample(c) == 0                      # idempotent
    # the rerouted forward reads the very modules the state_dict names
    ds = c.stages[1].downsample
    assert ds._hip.token_mixer is ds.token_mixer and ds._hip.norm is ds.norm
    import copy
    d = copy.deepcopy(c).stages[1].downsample
    assert d._hip.token_mixer is d.token_mixer and d._hip.norm is d.norm


@pytest.mark.gpu
def test_speed_harness_hip_leg(capsys):
    """recnext_amd.speed on the GPU (the reference's speed_gpu.py loop with the HIP token mixers), short T0/T1."""
    from recnext_amd import speed
    rate = speed.main(["--model", "recnext_m0", "--batch-size", "16", "--t0", "0.3", "--t1", "0.6"])
    out = capsys.readouterr().out.strip().splitlines()[-1].split()
    assert out[0] == "recnext_m0" and out[1].startswith("cuda") and out[-1] == "16" and rate > 100
