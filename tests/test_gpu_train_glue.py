"""Step-glue behaviour on the GPU: the EMA model's packed-weight caches follow the raw-pointer EMA
update, and the head's eager class branch follows the GradScaler's loss scale (back-off included)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_state_dict_compat import build_model


def _tiny(golden, train):
    cfg, sd, vec, meta = golden
    m = build_model(cfg)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    if train:
        m.train()
        for p in m.backbone.parameters():
            p.requires_grad = False
        for mod in (m.head.stems, m.head.reg_convs, m.head.cls_convs, m.head.reg_preds):
            for p in mod.parameters():
                p.requires_grad = False
    imgs = torch.from_numpy(vec["evalA_imgs"]).float().cuda().half()
    labels = torch.from_numpy(vec["train_labels"]).cuda().half()
    temb = torch.from_numpy(vec["time_embedding"]).cuda().half()
    return m, imgs, labels, temb, dict(lframe=int(meta["lframe"]), gframe=int(meta["gframe"]), resume=False)


def test_ema_forward_sees_updated_weights(golden):
    """ModelEMA.update writes the EMA parameters through raw device pointers (one launch); the packed conv / linear weights
    cached per parameter version must be re-packed afterwards: eval forward of the EMA model after an update == eval
    forward of a freshly built model holding the same (updated) state_dict."""
    from yolox.utils.ema import ModelEMA
    m, imgs, labels, temb, kw = _tiny(golden, train=False)
    ema = ModelEMA(m, decay=0.5)
    ema.decay = lambda n: 0.5                                    # a large step so that stale packs would be far off
    ema.ema.head.debug = {}
    with torch.no_grad():
        ema.ema(imgs, time_embedding=temb, **kw)                  # packs every weight of the EMA model; `score` = obj * max cls per anchor [B,A]
        before = ema.ema.head.debug["score"].float().clone()
        for k, p in m.named_parameters():                         # move the source model, then one EMA update
            if "_preds." in k or "head.stems" in k or "lateral_conv0.conv" in k:
                p.mul_(1.25)
        ema.update(m)
        ema.ema(imgs, time_embedding=temb, **kw)
        after = ema.ema.head.debug["score"].float().clone()
        fresh = build_model(golden[0]).cuda().eval()
        fresh.load_state_dict(ema.ema.state_dict(), strict=True)
        fresh.head.debug = {}
        fresh(imgs, time_embedding=temb, **kw)
        want = fresh.head.debug["score"].float()
    torch.cuda.synchronize()
    assert torch.isfinite(want).all() and not torch.allclose(before, want, rtol=1e-3, atol=1e-3)     # the update did move the outputs
    assert after.shape == want.shape and torch.equal(after, want)            # same weights, same kernels: identical


def _grads(m):
    return {k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}


def test_eager_class_branch_follows_the_loss_scale(golden):
    """The first MCA -> cls_pred -> refined class loss branch runs its backward inside forward, seeded with the loss scale the
    training loop hands to the head: gradients at a scale other than 65536 equal those of the ordinary (non-eager)
    backward, and an overflowing scale is recovered from by the scaler's back-off."""
    import yolox.models.tscd_head as H
    m, imgs, labels, temb, kw = _tiny(golden, train=True)
    one = torch.ones((), device="cuda")
    ref = copy.deepcopy(m)
    scale = torch.full((), 512.0, device="cuda")
    # eager, scale 512
    m.head.loss_scale = scale
    (m(imgs, labels, temb, **kw)["total_loss"] * scale).backward()
    ge = _grads(m)
    # ordinary backward of the same step
    was, H.EAGER_CLASS_BRANCH = H.EAGER_CLASS_BRANCH, False
    try:
        (ref(imgs, labels, temb, **kw)["total_loss"] * scale).backward()
    finally:
        H.EAGER_CLASS_BRANCH = was
    gr = _grads(ref)
    torch.cuda.synchronize()
    assert ge.keys() == gr.keys()
    for k in ("head.agg.mca.kv_cls.weight", "head.agg.linear.weight", "head.cls_pred.weight", "head.cls_pred.bias",
              "head.cls_convs2.0.1.conv.weight"):
        a, b = ge[k].flatten(), gr[k].flatten()
        cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
        assert cos > 0.999 and abs(a.norm().item() / b.norm().item() - 1) < 2e-2, (k, cos, a.norm().item(), b.norm().item())

    # overflow and recovery: a scale that overflows fp16 gradients, default back-off, a handful of steps
    m2, imgs, labels, temb, kw = _tiny(golden, train=True)
    opt = torch.optim.SGD([p for p in m2.parameters() if p.requires_grad], lr=1e-6)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 30, growth_interval=1000)
    w0 = m2.head.cls_pred.weight.detach().clone()
    applied, scales = 0, []
    for it in range(24):
        m2.head.loss_scale = scaler.scale(one)
        loss = m2(imgs, labels, temb, **kw)["total_loss"]
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        before = m2.head.cls_pred.weight.detach().clone()
        scaler.step(opt)
        scaler.update()
        applied += int(not torch.equal(before, m2.head.cls_pred.weight.detach()))
        scales.append(scaler.get_scale())
    assert scales[0] < 2.0 ** 30                      # the first step overflowed and was skipped
    assert applied >= 4, (applied, scales)             # the scaler backed off until the eager branch stopped overflowing
    assert scales[-1] == scales[-2] == scales[-3]      # and settled
    assert torch.isfinite(m2.head.cls_pred.weight).all() and not torch.equal(w0, m2.head.cls_pred.weight.detach())


def test_fused_towers_honour_the_batchnorm_mode(golden):
    """Advisor r03: with the exps' fix_bn option (BatchNorms put in eval() inside a training model, ovis_tscd_large.py:153-160) the
    trained towers must normalise with RUNNING statistics and leave them untouched.  The fused no-grad path follows the modules' own
    mode (== the unfused BaseConv calls, bit for bit); mixed modes are not fused; gradients through an eval-mode BatchNorm are refused
    (the running-statistics backward is not built) instead of silently taking batch statistics."""
    from yolox.models import tower_fusion as TW
    m, imgs, labels, temb, kw = _tiny(golden, train=True)
    head = m.head
    with torch.no_grad():
        x = head.stems[0](m.backbone(imgs)[0].permute(0, 2, 3, 1).half())
    t0, t1 = head.cls_convs2[0], head.reg_convs2[0]
    convs = [t0[0], t1[0], t0[1], t1[1]]
    for c in convs:
        c.bn.eval()
    assert TW.fusable(convs) and not TW.batch_stats(t0[0])
    stats = [c.bn.running_mean.clone() for c in convs]
    with torch.no_grad():
        fused = TW.towers_nograd(x, [t0[0], t1[0]], [t0[1], t1[1]])
        W = head.width
        want_c, want_r = t0[1](t0[0](x)), t1[1](t1[0](x))
    torch.cuda.synchronize()
    assert torch.equal(fused[..., :W], want_c) and torch.equal(fused[..., W:], want_r)
    assert all(torch.equal(s, c.bn.running_mean) for s, c in zip(stats, convs))          # running statistics untouched
    t0[0].bn.train()                                                                       # mixed modes: one launch cannot serve both
    assert not TW.fusable(convs)
    t0[0].bn.eval()
    with pytest.raises(NotImplementedError):
        m(imgs, labels, temb, **kw)                                                        # training forward with gradients: refused, loudly
    for c in convs:
        c.bn.train()
    out = m(imgs, labels, temb, **kw)
    assert torch.isfinite(out["total_loss"])
