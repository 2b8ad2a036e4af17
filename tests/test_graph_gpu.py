"""HIP-graph replay of the inference forward (recnext_amd/graph.py): the HIP token mixers are captured like any other launch and the
replay equals the eager forward bit for bit, per input shape."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["recnext_m0", "recnext_a0"])
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
