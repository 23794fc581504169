"""tscd_amd.optim.FusedSGD (one launch: unscale + Nesterov SGD over all parameter groups + EMA) against torch.optim.SGD (foreach)
followed by the reference's EMA formula (yolox/utils/ema.py:48-60), including a step the GradScaler skips."""
import copy
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 32, 3, 3), (64,), (7,), (33, 5), (256, 256), (1,), (100003,)]
    return [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]


def test_fused_sgd_ema_matches_torch_sgd_and_ema_formula():
    from tscd_amd.optim import FusedSGD
    ps, qs = _params(0), _params(0)
    mk = lambda cls, p, **kw: cls([{"params": p[:2]}, {"params": p[2:5], "weight_decay": 5e-4}, {"params": p[5:], "lr": 0.003}],
                                  lr=0.02, momentum=0.9, nesterov=True, **kw)
    fused, ref = mk(FusedSGD, ps), mk(torch.optim.SGD, qs, foreach=True)
    ema_f = {p: p.detach().clone() for p in ps}
    ema_r = [q.detach().clone() for q in qs]
    updates = [0]

    def decay():
        updates[0] += 1
        return 0.9998 * (1 - math.exp(-updates[0] / 2000))
    fused.attach_ema(ema_f, decay)
    scale = torch.tensor(1024.0, device="cuda")
    g = torch.Generator().manual_seed(1)
    for it in range(6):
        skip = it == 3
        for p, q in zip(ps, qs):
            gr = torch.randn(p.shape, generator=g).cuda()
            if skip and p.numel() == 7:
                gr[3] = float("inf")
            p.grad = gr * 1024.0
            q.grad = gr.clone()
        fused.grad_scale, fused.found_inf = scale, torch.tensor(1.0 if skip else 0.0, device="cuda")
        fused.step()
        if not skip:
            ref.step()
        d = 0.9998 * (1 - math.exp(-(it + 1) / 2000))
        torch._foreach_mul_(ema_r, d)
        torch._foreach_add_(ema_r, [q.detach() for q in qs], alpha=1.0 - d)
        torch.cuda.synchronize()
        for i, (p, q) in enumerate(zip(ps, qs)):
            assert torch.isfinite(p).all()
            # fma contraction and 1/scale against torch's separate multiply / add / divide: a few ulp per step
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-6), (it, i, float((p - q).abs().max()))
            assert torch.allclose(ema_f[p], ema_r[i], rtol=1e-5, atol=1e-6), (it, i)
            mb, mr = fused.state[p]["momentum_buffer"], ref.state[q]["momentum_buffer"]
            assert torch.allclose(mb, mr, rtol=1e-5, atol=2e-6), (it, i, float((mb - mr).abs().max()))
    assert updates[0] == 6
    # state_dict layout = torch.optim.SGD's: loads both ways
    sd = fused.state_dict()
    ref2 = mk(torch.optim.SGD, [torch.nn.Parameter(q.detach().clone()) for q in qs], foreach=True)
    ref2.load_state_dict(sd)
    assert torch.equal(ref2.state[ref2.param_groups[1]["params"][0]]["momentum_buffer"], fused.state[ps[2]]["momentum_buffer"])
    fused2 = mk(FusedSGD, _params(0))
    fused2.load_state_dict(ref.state_dict())
    assert fused2.param_groups[2]["lr"] == 0.003


def test_grad_scaler_drives_the_fused_step_without_sync_and_versions_move():
    from tscd_amd.optim import FusedSGD
    lin = torch.nn.Linear(16, 8).cuda()
    opt = FusedSGD(lin.parameters(), lr=0.1, momentum=0.9, nesterov=True)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 4)
    x = torch.randn(4, 16, device="cuda")
    w0, v0 = lin.weight.detach().clone(), lin.weight._version
    loss = lin(x).square().mean()
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    torch.cuda.synchronize()
    assert not torch.equal(lin.weight, w0) and lin.weight._version > v0
    # an overflowing step leaves the parameters alone and halves the scale
    w1 = lin.weight.detach().clone()
    opt.zero_grad(set_to_none=True)
    loss = (lin(x) * float("inf")).sum()
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    torch.cuda.synchronize()
    assert torch.equal(lin.weight, w1) and float(scaler.get_scale()) == 8.0


def test_model_ema_attached_to_the_optimizer_equals_full_state_ema():
    """ModelEMA.attach_to: trainable parameters in the optimizer's pass, buffers in ModelEMA.update, frozen parameters skipped --
    against ModelEMA over the whole state."""
    from tscd_amd.optim import FusedSGD
    from yolox.utils import ModelEMA
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 1)).cuda()
    for p in net[0].parameters():
        p.requires_grad_(False)                        # "backbone"
    net_r = copy.deepcopy(net)
    tr = lambda m: [p for p in m.parameters() if p.requires_grad]
    opt, opt_r = FusedSGD(tr(net), lr=0.05, momentum=0.9, nesterov=True), torch.optim.SGD(tr(net_r), lr=0.05, momentum=0.9, nesterov=True)
    ema, ema_r = ModelEMA(net, 0.9998), ModelEMA(net_r, 0.9998)
    ema.attach_to(opt, net)
    for it in range(4):
        x = torch.randn(2, 3, 12, 12, device="cuda")
        for m, o, e in ((net, opt, ema), (net_r, opt_r, ema_r)):
            m.train()
            o.zero_grad(set_to_none=True)
            m(x).square().mean().backward()
            o.step()
            e.update(m)
    torch.cuda.synchronize()
    assert ema.updates == ema_r.updates == 4
    a, b = ema.ema.state_dict(), ema_r.ema.state_dict()
    for k in a:
        assert torch.allclose(a[k].float(), b[k].float(), rtol=1e-5, atol=1e-7), k


def test_packed_weight_copies_are_refreshed_in_place_by_the_step():
    """After FusedSGD.step every cached packed copy (conv forward, conv data-gradient, linear forward / data-gradient) of a stepped
    weight equals a fresh packing of the new values, is the same buffer as before, and a lookup launches nothing new."""
    from tscd_amd import ops
    from tscd_amd.optim import FusedSGD
    torch.manual_seed(0)
    conv_w = torch.nn.Parameter(torch.randn(40, 24, 3, 3, device="cuda"))
    lin_w = torch.nn.Parameter(torch.randn(25, 64, device="cuda"))
    frozen = torch.nn.Parameter(torch.randn(16, 16, 1, 1, device="cuda"), requires_grad=False)
    conv5_w = torch.nn.Parameter(torch.randn(33, 72, 5, 5, device="cuda"))      # 25 taps: three staging passes per (cout, cin) tile, 3 cin tiles
    copies5 = {"f": ops.WEIGHTS.get(conv5_w), "d": ops.WEIGHTS.get(conv5_w, None, 1)}
    copies = {"conv_f": ops.WEIGHTS.get(conv_w), "conv_d": ops.WEIGHTS.get(conv_w, None, 1), "lin_f": ops.LINEAR_WEIGHTS.get(lin_w),
              "lin_d": ops.LINEAR_WEIGHTS.get(lin_w, mode=1), "frozen": ops.WEIGHTS.get(frozen)}
    before = {k: v.clone() for k, v in copies.items()}
    opt = FusedSGD([conv_w, lin_w, conv5_w], lr=0.5, momentum=0.9, nesterov=True)
    conv_w.grad, lin_w.grad, conv5_w.grad = torch.randn_like(conv_w), torch.randn_like(lin_w), torch.randn_like(conv5_w)
    opt.step()
    torch.cuda.synchronize()
    assert ops.WEIGHTS.get(conv5_w) is copies5["f"] and torch.equal(copies5["f"], ops.pack_conv_weight(conv5_w))
    assert ops.WEIGHTS.get(conv5_w, None, 1) is copies5["d"] and torch.equal(copies5["d"], ops.pack_conv_weight(conv5_w, None, 1))
    fresh = {"conv_f": ops.pack_conv_weight(conv_w), "conv_d": ops.pack_conv_weight(conv_w, None, 1), "lin_f": ops.pack_linear_weight(lin_w),
             "lin_d": ops.pack_conv_weight(lin_w.detach().reshape(25, 64, 1, 1), mode=1)}
    again = {"conv_f": ops.WEIGHTS.get(conv_w), "conv_d": ops.WEIGHTS.get(conv_w, None, 1), "lin_f": ops.LINEAR_WEIGHTS.get(lin_w),
             "lin_d": ops.LINEAR_WEIGHTS.get(lin_w, mode=1), "frozen": ops.WEIGHTS.get(frozen)}
    for k in fresh:
        assert again[k] is copies[k], k                       # no new buffer: the lookup hit the cache
        assert torch.equal(again[k], fresh[k]), k             # and the buffer holds the new weights
        assert not torch.equal(again[k], before[k]), k
    assert again["frozen"] is copies["frozen"] and torch.equal(again["frozen"], before["frozen"])
    # second step: table reused
    conv_w.grad, lin_w.grad, conv5_w.grad = torch.randn_like(conv_w), torch.randn_like(lin_w), torch.randn_like(conv5_w)
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(ops.WEIGHTS.get(conv_w), ops.pack_conv_weight(conv_w)) and ops.WEIGHTS.get(conv_w) is copies["conv_f"]
