"""HIP-graph replay of the inference forward (recnext_amd/graph.py): the HIP token mixers are captured like any other launch and the
replay equals the eager forward bit for bit, per input shape."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["recnext_m0", "recnext_a0", "recnext_a3"])          # a3: the matrix-core RecAttn2d kernels and their workspace inside a capture
def test_graph_replay_equals_eager(name):
    from recnext_amd.graph import GraphedInference
    from recnext_amd.speed import build_inference_model, synthetic_batch
    dev = torch.device("cuda:0")
    net = build_inference_model(name, dev, torch.bfloat16, seed=0)
    run = GraphedInference(net)
    with torch.no_grad():
        for batch, seed in ((2, 0), (5, 1), (2, 2)):                 # two shapes; the first one again with other data (a replay)
            x = synthetic_batch(batch, 224, dev, torch.bfloat16, seed=seed)
            want = net(x).clone()
            got = run(x)
            assert got.shape == want.shape and torch.equal(got, want), (name, batch)
    assert len(run._graphs) == 2


def test_graph_follows_a_weight_change():
    """a graph bakes in the addresses of the parameters and of the packs derived from them; after load_state_dict / an in-place edit
    the replay must use the new weights (the graphs are dropped and captured again), not a mix of old packs and new GEMM weights."""
    from recnext_amd.graph import GraphedInference
    from recnext_amd.speed import build_inference_model, synthetic_batch
    dev = torch.device("cuda:0")
    net = build_inference_model("recnext_m0", dev, torch.bfloat16, seed=0)
    other = build_inference_model("recnext_m0", dev, torch.bfloat16, seed=1)
    run = GraphedInference(net)
    x = synthetic_batch(2, 224, dev, torch.bfloat16, seed=0)
    with torch.no_grad():
        assert torch.equal(run(x), net(x))
        before = net(x).clone()
        net.load_state_dict(other.state_dict())                      # in place: every parameter's version moves
        want = net(x).clone()
        assert not torch.equal(want, before)
        assert torch.equal(run(x), want)
        for m in net.modules():                                      # one in-place edit of one token mixer's taps
            if type(m).__name__ == "RecConv2d":
                m.convs[0].weight.mul_(1.5)
                break
        want2 = net(x).clone()
        assert torch.equal(run(x), want2) and not torch.equal(want2, want)
    assert len(run._graphs) == 1


def test_graph_refuses_training_mode_and_cpu_tensors():
    from recnext_amd.graph import GraphedInference
    from recnext_amd.speed import build_inference_model
    net = build_inference_model("recnext_m0", torch.device("cuda:0"), torch.bfloat16, seed=0)
    run = GraphedInference(net)
    with pytest.raises(RuntimeError):
        run(torch.zeros(1, 3, 224, 224))
    net.train()
    with pytest.raises(RuntimeError):
        run(torch.zeros(1, 3, 224, 224, device="cuda:0", dtype=torch.bfloat16))
