"""Fused loss kernels (tscd_amd/csrc/loss.hip) vs the oracle's loss expressions (oracle/tscd_oracle.py
iou_loss / encode_reg_preds and the BCE / SmoothL1 calls of tscd_head, i.e. the reference's tscd_head.py:1008-1203,
losses.py:9-53) evaluated in fp32 on the same fp16 inputs: sums within 1e-3 relative, logit gradients within 2e-3 of the
largest gradient (they are written as fp16)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import tscd_oracle as O


def _scene(B, A, NC, Lmax, seed):
    g = torch.Generator().manual_seed(seed)
    raw = (torch.randn(B, A, 5 + NC, generator=g) * 2).half()
    outputs = raw.clone()
    outputs[..., 0:2] = (torch.rand(B, A, 2, generator=g) * 500).half()
    outputs[..., 2:4] = (torch.rand(B, A, 2, generator=g) * 300 + 4).half()          # up to 304 px: areas above the fp16 range
    labels = torch.zeros(B, Lmax, 5)
    ngt = 7
    labels[:, :ngt, 0] = torch.randint(0, NC, (B, ngt), generator=g).float()
    labels[:, :ngt, 1:3] = torch.rand(B, ngt, 2, generator=g) * 500
    labels[:, :ngt, 3:5] = torch.rand(B, ngt, 2, generator=g) * 250 + 6
    labels = labels.half().float()
    fg = torch.rand(B, A, generator=g) < 0.02
    fg[1] = False                                                                     # a frame without foreground
    mg = torch.where(fg, torch.randint(0, ngt, (B, A), generator=g), torch.full((B, A), -1)).int()
    pi = torch.where(fg, torch.rand(B, A, generator=g), torch.zeros(B, A)).half().float()
    # make some predictions overlap their gt well
    for b in range(B):
        idx = torch.nonzero(fg[b]).flatten()[::2]
        outputs[b, idx, :4] = (labels[b, mg[b, idx].long(), 1:5] * (1 + 0.1 * torch.randn(len(idx), 4, generator=g))).half()
    return raw, outputs, labels, fg, mg, pi


@pytest.mark.parametrize("B,A,NC", [(3, 336, 25), (4, 6804, 30)])
def test_dense_loss_matches_oracle_expressions(B, A, NC):
    from tscd_amd import functional as TF
    raw, outputs, labels, fg, mg, pi = _scene(B, A, NC, 20, B + NC)
    r32 = raw.float().requires_grad_()
    gt = torch.gather(labels[..., 1:5], 1, mg.clamp(min=0).long().unsqueeze(-1).expand(-1, -1, 4))
    gcls = torch.gather(labels[..., 0], 1, mg.clamp(min=0).long()).long()
    iou_l = O.iou_loss(outputs[..., :4].float().reshape(-1, 4)[fg.reshape(-1)], gt.reshape(-1, 4)[fg.reshape(-1)]).sum()
    obj_l = F.binary_cross_entropy_with_logits(r32[..., 4], fg.float(), reduction="sum")
    cls_t = F.one_hot(gcls, NC).float() * pi.unsqueeze(-1)
    cls_l = F.binary_cross_entropy_with_logits(r32[..., 5:][fg], cls_t[fg], reduction="sum")
    (2.0 * obj_l + 3.0 * cls_l).backward()
    rd = raw.cuda().requires_grad_()
    sums = TF.dense_loss(rd, outputs.cuda().contiguous(), fg.to(torch.uint8).cuda(), mg.cuda(), pi.cuda(), labels.cuda().contiguous(), NC)
    (2.0 * sums[1] + 3.0 * sums[2]).backward()
    torch.cuda.synchronize()
    got = sums.detach().cpu()
    for v, ref in zip(got.tolist(), (iou_l.item(), obj_l.item(), cls_l.item())):
        assert abs(v - ref) <= 1e-3 * abs(ref) + 1e-3, (v, ref)
    gd, gr = rd.grad.float().cpu(), r32.grad
    assert torch.all(gd[..., :4] == 0)
    assert (gd - gr).abs().max().item() <= 2e-3 * gr.abs().max().item()
    # deterministic: a second evaluation gives the same bits
    sums2 = TF.dense_loss(raw.cuda(), outputs.cuda().contiguous(), fg.to(torch.uint8).cuda(), mg.cuda(), pi.cuda(), labels.cuda().contiguous(), NC)
    assert torch.equal(sums2.cpu(), got)


@pytest.mark.parametrize("B,A,NC,N", [(3, 336, 25, 150), (4, 6804, 30, 2000)])
def test_row_losses_match_oracle_expressions(B, A, NC, N):
    from tscd_amd import functional as TF
    raw, outputs, labels, fg, mg, pi = _scene(B, A, NC, 20, B + N)
    g = torch.Generator().manual_seed(N)
    # rows: half of them sit on foreground anchors
    fgi = torch.nonzero(fg)
    pick = fgi[torch.randint(0, len(fgi), (N // 2,), generator=g)]
    rb = torch.cat((pick[:, 0], torch.randint(0, B, (N - N // 2,), generator=g)))
    ra = torch.cat((pick[:, 1], torch.randint(0, A, (N - N // 2,), generator=g)))
    cls = (torch.randn(N, NC, generator=g) * 2).half()
    obj = (torch.randn(N, 1, generator=g) * 2).half()
    reg = (torch.randn(N, 4, generator=g) * 1.5).half()
    base = torch.zeros(N, 7 + NC)
    base[:, 0:2] = torch.rand(N, 2, generator=g) * 400
    base[:, 2:4] = base[:, 0:2] + torch.rand(N, 2, generator=g) * 200 + 5
    base = base.half()
    rmask = fg[rb, ra]
    gt = labels[rb, mg[rb, ra].clamp(min=0).long()]
    c32, o32, g32 = cls.float().requires_grad_(), obj.float().requires_grad_(), reg.float().requires_grad_()
    t_cls = F.one_hot(gt[:, 0].long(), NC).float() * pi[rb, ra].unsqueeze(-1)
    l_cls = F.binary_cross_entropy_with_logits(c32[rmask], t_cls[rmask], reduction="sum")
    l_obj = F.binary_cross_entropy_with_logits(o32.view(-1), rmask.float(), reduction="sum")
    tgt = O.encode_reg_preds(gt[rmask, 1:5], base[rmask, :4].float())
    l_reg = F.smooth_l1_loss(g32[rmask], tgt, reduction="sum")
    (l_cls + 2.0 * l_obj + 3.0 * l_reg).backward()
    dev = lambda t: t.cuda()
    args = (dev(rb.int()), dev(ra.int()), A, NC, dev(fg.to(torch.uint8)), dev(mg), dev(pi), dev(labels).contiguous())
    cd, od, gd = dev(cls).requires_grad_(), dev(obj).requires_grad_(), dev(reg).requires_grad_()
    s1 = TF.row_loss(*args, cls_logit=cd)
    s2 = TF.row_loss(*args, obj_logit=od.reshape(-1), reg_delta=gd, base=dev(base)[:N])
    (s1[0] + 2.0 * s2[1] + 3.0 * s2[2]).backward()
    torch.cuda.synchronize()
    for v, ref in ((s1[0], l_cls), (s2[1], l_obj), (s2[2], l_reg)):
        assert abs(v.item() - ref.item()) <= 1e-3 * abs(ref.item()) + 1e-3, (v.item(), ref.item())
    assert s1[1].item() == 0 and s1[2].item() == 0 and s2[0].item() == 0
    for got, ref in ((cd.grad, c32.grad), (od.grad, o32.grad), (gd.grad, g32.grad)):
        assert (got.float().cpu() - ref).abs().max().item() <= 2e-3 * max(ref.abs().max().item(), 1e-6)


@pytest.mark.parametrize("clipped", [False, True])
def test_loss_tail_matches_the_reference_expression(clipped):
    """tscd_loss_combine_f32 vs the expression of get_matcher_refined_losses' tail (tscd_head.py:1157-1203), values bit for bit (same
    fp32 operations in the same order) and gradients to fp32 rounding, with the matched-objectness clip at 15 off and on."""
    from tscd_amd import functional as TF
    g = torch.Generator().manual_seed(7 + int(clipped))
    B, lframe = 16, 4
    dense = (torch.rand(3, generator=g) * 900 + 20).cuda().requires_grad_()
    rows = (torch.rand(3, generator=g) * (4000 if clipped else 60) + (2000 if clipped else 5)).cuda().requires_grad_()
    r_cls = (torch.rand((), generator=g) * 9 + 0.5).cuda().requires_grad_()
    num_fg_b = torch.randint(3, 12, (B,), generator=g).int().cuda()
    ngt_b = torch.randint(2, 7, (B,), generator=g).int().cuda()
    # reference expression (what the head computed with ~25 zero-dimensional torch ops)
    d2, r2, c2 = dense.detach().clone().requires_grad_(), rows.detach().clone().requires_grad_(), r_cls.detach().clone().requires_grad_()
    num_fg = num_fg_b.sum().float().clamp(min=1)
    num_gts = ngt_b.sum().float()
    num_fg_local = num_fg_b[:lframe].sum().float()
    li, lo, lc = d2[0] / num_fg, d2[1] / num_fg, d2[2] / num_fg
    mo, mi = r2[1] / num_fg_local, r2[2] / num_fg_local
    mo = torch.where(mo > 15, mo / mo.detach() * 15, mo)
    ref = 3.0 * li + lo + lc + c2 + mo + 6.0 * mi
    assert bool((r2[1] / num_fg_local > 15)) == clipped
    (ref * 1024.0).backward()
    loss, st = TF.combine_losses(dense, rows, r_cls, num_fg_b, ngt_b, lframe)
    (loss * 1024.0).backward()
    torch.cuda.synchronize()
    assert loss.item() == ref.item()
    want = [ref, 3.0 * li, lo, lc, num_fg / num_gts.clamp(min=1), c2, mo, 6.0 * mi]
    for i, w in enumerate(want):
        assert st[i].item() == w.item(), (i, st[i].item(), w.item())
    for got, exp in ((dense.grad, d2.grad), (rows.grad, r2.grad), (r_cls.grad, c2.grad)):
        assert torch.allclose(got, exp, rtol=2e-6, atol=0), (got, exp)
    assert rows.grad[0].item() == 0.0


def test_loss_backward_forms_the_product_in_fp32_at_the_initial_loss_scale():
    """the upstream scalar of the fused losses is loss_scale * weight / num_fg.  At GradScaler's initial scale
    65536 with num_fg_local = 4 it is 6 * 65536 / 4 = 98304 > fp16 max: cast on its own it is inf, while the product with the stored
    gradient (|d| <= 1 for BCE, <= 1 for SmoothL1) is representable.  The backward multiplies in fp32 and rounds once
    (tscd_scale_grad_f16), like the reference's autograd (tscd_head.py:1157-1203): finite, equal to the fp32 expression."""
    from tscd_amd import functional as TF
    B, A, NC, N = 3, 336, 25, 96
    raw, outputs, labels, fg, mg, pi = _scene(B, A, NC, 20, 11)
    g = torch.Generator().manual_seed(5)
    fgi = torch.nonzero(fg)
    pick = fgi[torch.randint(0, len(fgi), (N // 2,), generator=g)]
    rb = torch.cat((pick[:, 0], torch.randint(0, B, (N - N // 2,), generator=g)))
    ra = torch.cat((pick[:, 1], torch.randint(0, A, (N - N // 2,), generator=g)))
    cls = (torch.randn(N, NC, generator=g) * 2).half()
    obj = (torch.randn(N, generator=g) * 2).half()
    reg = (torch.randn(N, 4, generator=g) * 0.05).half()          # small deltas: SmoothL1 gradient = delta / beta, well inside fp16
    base = torch.zeros(N, 7 + NC)
    base[:, 0:2] = torch.rand(N, 2, generator=g) * 400
    base[:, 2:4] = base[:, 0:2] + torch.rand(N, 2, generator=g) * 200 + 5
    dev = lambda t: t.cuda()
    args = (dev(rb.int()), dev(ra.int()), A, NC, dev(fg.to(torch.uint8)), dev(mg), dev(pi), dev(labels).contiguous())
    cd, od, gd = dev(cls).requires_grad_(), dev(obj).requires_grad_(), dev(reg).requires_grad_()
    rd = raw.cuda().requires_grad_()
    scale, num_fg_local = 65536.0, 4.0
    w = torch.tensor([scale * 1.0 / num_fg_local, scale * 1.0 / num_fg_local, scale * 6.0 / num_fg_local], device="cuda")
    assert not torch.isfinite(w.half()).all()                      # the scalar alone does overflow fp16
    s1 = TF.row_loss(*args, cls_logit=cd)
    s2 = TF.row_loss(*args, obj_logit=od, reg_delta=gd, base=dev(base.half())[:N])
    dense = TF.dense_loss(rd, outputs.cuda().contiguous(), fg.to(torch.uint8).cuda(), mg.cuda(), pi.cuda(), labels.cuda().contiguous(), NC)
    (w[0] * s1[0] + w[1] * s2[1] + w[2] * s2[2] + w[1] * dense[1] + w[0] * dense[2]).backward()
    torch.cuda.synchronize()
    # the stored un-normalised gradients (fp16) times the fp32 scalar, rounded once
    _, d_cls, d_obj, d_reg = __import__("tscd_amd.ops", fromlist=["x"]).loss_rows(*args, cls_logit=dev(cls), obj_logit=dev(obj), reg_delta=dev(reg),
                                                                                  base=dev(base.half())[:N])
    for got, d, s in ((cd.grad, d_cls, w[0]), (od.grad, d_obj, w[1]), (gd.grad, d_reg, w[2])):
        want = (d.float() * s).half().reshape(got.shape)
        assert torch.equal(got, want)
        # finite wherever the fp32 product fits fp16 (|d| * 98304 < 65504 <=> |d| < 0.67: every BCE row here, most SmoothL1 rows);
        # with the scalar cast first EVERY element of the box gradient would be inf or nan
        assert torch.isfinite(got[(d.float() * s).abs().reshape(got.shape) < 65504]).all() and got.abs().max() > 0
    assert torch.isfinite(cd.grad).all() and torch.isfinite(od.grad).all() and torch.isfinite(gd.grad).float().mean() > 0.3
    assert torch.isfinite(rd.grad).all() and torch.all(rd.grad[..., :4] == 0) and rd.grad[..., 4:].abs().max() > 0


@pytest.mark.parametrize("B,A,NC,N", [(4, 6804, 25, 1500), (2, 336, 30, 0), (3, 500, 1, 77)])
def test_class_score_gather_and_its_scatter_backward_equal_torch_indexing(B, A, NC, N):
    """TF.cls_score (one gather forward, one scatter backward) against sigmoid(raw[b, a, 5 + class]) with torch's advanced
    indexing and its autograd (the reference's pred_result[:, 5] path, tscd_head.py:1560-1600): same fp16 values forward; backward
    the same non-zero entries of d raw (fp16, rounded once from the fp32 product like sigmoid_backward)."""
    from tscd_amd import functional as TF
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 1000 + A + NC)
    raw = (torch.randn(B, A, 5 + NC, generator=g) * 3).half().to(dev)
    flat = torch.randperm(B * A, generator=g)[:N].sort().values                      # distinct (frame, anchor) pairs, frame-major
    rb, ra = (flat // A).int().to(dev), (flat % A).int().to(dev)
    rows = torch.zeros(N, 7 + NC).half()
    rows[:, 6] = torch.randint(0, NC, (N,), generator=g).half()
    rows = rows.to(dev)
    up = torch.randn(N, generator=g).half().to(dev)
    r1 = raw.clone().requires_grad_()
    s1 = TF.cls_score(r1, rb, ra, rows)
    (s1.float() * up.float()).sum().backward()
    r0 = raw.clone().requires_grad_()
    s0 = torch.sigmoid(r0[rb.long(), ra.long(), 5 + rows[:, 6].long()])
    (s0.float() * up.float()).sum().backward()
    torch.cuda.synchronize()
    assert s1.shape == (N,) and torch.allclose(s1.float(), s0.float(), rtol=0, atol=1e-3)
    assert r1.grad.shape == raw.shape
    assert torch.equal(r1.grad != 0, r0.grad != 0)
    assert torch.allclose(r1.grad.float(), r0.grad.float(), rtol=2e-3, atol=1e-6)


@pytest.mark.parametrize("B,h,w,W,NC", [(4, 18, 18, 256, 25), (2, 9, 7, 64, 30), (3, 12, 12, 128, 1)])
def test_prediction_level_fused_backward_equals_the_per_conv_path(B, h, w, W, NC):
    """TF.pred_level (one split of the level's raw-map gradient into zero-padded dY's + both bias gradients, weight gradients
    beside the data-gradient chain) against the per-conv autograd path (cat -> slice gradients -> pad -> column sums -> GEMM) on
    the same inputs, the level's map being a batch-strided slice of a clip-wide gradient as in the head: forward bit-identical,
    weight gradients bit-identical (same GEMM on the same padded dY), bias gradients equal to fp32 summation order."""
    from tscd_amd import functional as TF
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(h * 100 + NC)
    feat = torch.randn(B, h, w, 2 * W, generator=g).half().to(dev)
    reg_feat, cls_feat = feat[..., :W], feat[..., W:]
    reg_o = torch.randn(B, h, w, 4, generator=g).half().to(dev)
    mk = lambda co: torch.nn.Conv2d(W, co, 1).to(dev)
    obj_a, cls_a = mk(1), mk(NC)
    obj_b, cls_b = mk(1), mk(NC)
    obj_b.load_state_dict(obj_a.state_dict()); cls_b.load_state_dict(cls_a.state_dict())
    extra = 37                                                                 # other levels' anchors in the clip-wide map
    up = torch.randn(B, h * w + extra, 5 + NC, generator=g).half().to(dev)

    def run(obj, cls, fused):
        if fused:
            raw = TF.pred_level(reg_feat, cls_feat, reg_o, obj, cls)
            assert type(raw.grad_fn).__name__.startswith("_PredLevel")
        else:
            raw = torch.cat([reg_o, TF.conv_bias(reg_feat, obj.weight, obj.bias), TF.conv_bias(cls_feat, cls.weight, cls.bias)], -1)
        allmap = torch.cat([raw.reshape(B, -1, 5 + NC), torch.zeros(B, extra, 5 + NC, dtype=torch.float16, device=dev)], 1)
        (allmap.float() * up.float()).sum().backward()
        torch.cuda.synchronize()
        return raw.detach(), [p.grad.clone() for p in (obj.weight, obj.bias, cls.weight, cls.bias)]

    raw1, g1 = run(obj_a, cls_a, True)
    raw0, g0 = run(obj_b, cls_b, False)
    assert torch.equal(raw0, raw1)
    assert torch.equal(g0[0], g1[0]) and torch.equal(g0[2], g1[2])
    for a, b in ((g0[1], g1[1]), (g0[3], g1[3])):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(a.abs().max()) + 1e-6), (a, b)
