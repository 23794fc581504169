"""hipGraph replay of the frozen backbone pass (tscd_amd/graphs.py) against the eager pass: same pyramid bit for bit, same
BatchNorm running statistics and batch counters, in evaluation mode (folded BatchNorm) and in training mode (batch statistics,
as the reference's frozen backbone keeps them under model.train())."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _backbone(width=0.5, depth=0.33):
    from yolox.models import YOLOPAFPN
    torch.manual_seed(0)
    bb = YOLOPAFPN(depth, width).cuda()
    for p in bb.parameters():
        p.requires_grad_(False)
    return bb


@pytest.mark.parametrize("training", [False, True])
def test_graph_replay_matches_eager_pass(training):
    from tscd_amd import graphs
    graphs.clear()
    bb = _backbone()
    bb.train(training)
    ref = copy.deepcopy(bb)
    g = torch.Generator(device="cuda").manual_seed(1)
    clips = [(torch.rand(4, 3, 128, 192, device="cuda", generator=g) * 255).half() for _ in range(4)]
    assert graphs.ENABLED
    for i, x in enumerate(clips):
        with torch.no_grad():
            want = ref(x)                                  # eager
        got = graphs.frozen_pass(bb, x)                    # first call eager + capture, then replays
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            assert a.shape == b.shape and a.stride() == b.stride()
            assert torch.equal(a, b), (i, float((a.float() - b.float()).abs().max()))
    key = [k for k in graphs._GRAPHS if k[0] == id(bb)]
    assert len(key) == 1                                   # one signature, one graph, three replays
    if training:
        sd, sr = bb.state_dict(), ref.state_dict()
        for k in sd:
            assert torch.equal(sd[k], sr[k]), k            # running_mean / running_var / num_batches_tracked included
        assert int(sd["backbone.stem.conv.bn.num_batches_tracked"]) == len(clips)
    # outputs of one replay survive the next one
    a = graphs.frozen_pass(bb, clips[0])
    keep = [t.clone() for t in a]
    graphs.frozen_pass(bb, clips[1])
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(a, keep))
    # a written parameter invalidates the graph
    with torch.no_grad():
        next(bb.parameters()).mul_(1.0)
    graphs.frozen_pass(bb, clips[0])
    assert graphs._GRAPHS[key[0]].versions == graphs._state_versions(bb, bb.training)
    graphs.clear()


def test_evaluation_after_training_passes_sees_the_new_running_statistics():
    """Training-mode passes (eager or replayed) move the BatchNorm running statistics through raw pointers; the folded scale / shift
    of the evaluation path and a recorded evaluation pass must both notice."""
    from tscd_amd import graphs
    graphs.clear()
    bb = _backbone()
    ref = copy.deepcopy(bb)
    g = torch.Generator(device="cuda").manual_seed(2)
    clips = [(torch.rand(2, 3, 96, 128, device="cuda", generator=g) * 255).half() for _ in range(5)]
    for rounds in range(2):
        bb.eval(); ref.eval()
        for x in clips[:2]:                                  # eval: first call eager + recorded, second replayed
            with torch.no_grad():
                want = ref(x)
            got = graphs.frozen_pass(bb, x)
            assert all(torch.equal(a, b) for a, b in zip(got, want)), rounds
        bb.train(); ref.train()
        for x in clips[2:]:                                  # three training passes: eager + recorded, replayed, replayed
            with torch.no_grad():
                ref(x)
            graphs.frozen_pass(bb, x)
    sd, sr = bb.state_dict(), ref.state_dict()
    assert all(torch.equal(sd[k], sr[k]) for k in sd)
    graphs.clear()


def test_tscd_eval_uses_the_graph_and_matches_eager(monkeypatch):
    import bench
    from tscd_amd import graphs
    graphs.clear()
    dev = torch.device("cuda:0")
    model = bench.build_model(0.33, 0.5, "50").to(dev).eval().half()
    imgs, t0, temb = bench.synth_clip(4, 192, dev)
    outs = []
    for enabled in (True, True, False):
        monkeypatch.setattr(graphs, "ENABLED", enabled)
        with torch.no_grad():
            r = model(imgs.half(), None, temb, lframe=2, gframe=2, resume=False)
        torch.cuda.synchronize()
        outs.append(r)
    assert len(graphs._GRAPHS) == 1
    flat = lambda r: [t for part in r for t in (part if isinstance(part, (list, tuple)) else [part]) if torch.is_tensor(t)]
    for a, b in zip(flat(outs[1]), flat(outs[2])):
        assert torch.equal(a, b)
    graphs.clear()
