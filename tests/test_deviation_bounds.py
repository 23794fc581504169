"""The deliberate deviations from the reference's fp16 arithmetic, each against its reference-faithful
alternative on the CPU oracle: how far apart the two are.  (SimOTA's fp16 IoU chain has its own pinned fixture:
tests/test_simota_parity.py.)  CPU only; the bounds asserted here are the ones DESIGN.md quotes."""
import numpy as np
import torch

from oracle import nms_ref
from oracle import tscd_oracle as O


def test_mca_fp32_logits_vs_the_reference_fp16_rounding(golden):
    """post_trans.py:640-650 under autocast / model.half(): q k^T, * 25 and * cls_score are each rounded to fp16 before the softmax
    (one fp16 ulp at |logit| ~ 25 is 0.016).  The product keeps the logits in fp32.  On the golden model's aggregation weights and
    300 proposals the two differ by < 1 % of the output scale -- below the fp16 rounding of the outputs' own consumers."""
    cfg, sd, vec, meta = golden
    W = sd["head.agg.mca.q_cls_local.weight"].shape[1]
    g = torch.Generator().manual_seed(3)
    n2, n1 = 300, 80
    x_cls, x_reg = torch.randn(n2, W, generator=g), torch.randn(n2, W, generator=g)
    score = torch.rand(n2, generator=g)
    with torch.no_grad():
        c32, r32 = O.attention_mca_g2l(sd, "head.agg.mca", x_cls, x_reg, score, n1, cfg)
        cfg.mca_fp16_logits = True
        try:
            c16, r16 = O.attention_mca_g2l(sd, "head.agg.mca", x_cls, x_reg, score, n1, cfg)
        finally:
            cfg.mca_fp16_logits = False
    for a, b in ((c32, c16), (r32, r16)):
        rel = (a - b).abs().max().item() / a.abs().max().item()
        assert 0 < rel < 1e-2, rel


def test_per_class_fp32_nms_vs_the_coordinate_offset_trick():
    """post_process.py:58-63 calls torchvision.ops.batched_nms, which shifts every box by class_id * (max coordinate + 1) so that
    classes never overlap.  In fp32 that is the same partition as suppressing class by class (identical keep lists); in the
    evaluator's fp16 the shift of class 24 at 576 px is ~13 900, where fp16 resolves 8 px: boxes are quantised before their IoU
    is taken.  The product suppresses class by class in fp32."""
    g = torch.Generator().manual_seed(11)
    n, nc = 600, 25
    ctr = torch.rand(n, 2, generator=g) * 480 + 40
    wh = torch.rand(n, 2, generator=g) * 90 + 12
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    # near-duplicates so that suppression happens at all
    boxes[n // 2:] = boxes[: n - n // 2] + torch.randn(n - n // 2, 4, generator=g) * 3
    scores = torch.rand(n, generator=g)
    cls = torch.randint(0, nc, (n // 2,), generator=g)
    cls = torch.cat([cls, cls[: n - n // 2]])
    trick32 = nms_ref.batched_nms(boxes, scores, cls, 0.5)
    per_class = []
    for c in range(nc):
        idx = torch.nonzero(cls == c).flatten()
        if len(idx):
            per_class.append(idx[nms_ref.nms(boxes[idx], scores[idx], 0.5)])
    per_class = torch.cat(per_class)
    per_class = per_class[torch.argsort(scores[per_class], descending=True, stable=True)]
    assert torch.equal(trick32, per_class)
    assert 0 < len(trick32) < n                                      # something was suppressed
    # the fp16 evaluation of the same trick: boxes of the high class ids are quantised to 4-8 px
    trick16 = nms_ref.batched_nms(boxes.half(), scores.half(), cls, 0.5)
    a, b = set(trick32.tolist()), set(trick16.tolist())
    jac = len(a & b) / len(a | b)
    assert 0.85 < jac < 1.0, jac                                     # close, and NOT identical: the quantisation is real
    low = cls < 3                                                    # shifts below 2048: fp16 still resolves 1 px
    a_low = {i for i in a if low[i]}
    b_low = {i for i in b if low[i]}
    assert len(a_low ^ b_low) <= max(2, len(a_low) // 20)


def test_fp32_iou_loss_vs_fp16_evaluation():
    """losses.py:9-53 under autocast evaluates prod(br - tl) in fp16: for boxes of 20-120 px the areas (up to ~1.4e4 px^2, fp16 ulp 8)
    move the per-box loss 1 - IoU^2 by up to 0.03 (0.3 % on average), and an area beyond 65504 px^2 (a 256 x 256 box at 576^2) makes
    it inf / NaN.  The product evaluates the loss in fp32 from the fp16 tensors."""
    g = torch.Generator().manual_seed(5)
    n = 400
    t = torch.cat([torch.rand(n, 2, generator=g) * 400 + 60, torch.rand(n, 2, generator=g) * 100 + 20], 1).half().float()   # cx cy w h
    p = (t + torch.randn(n, 4, generator=g) * 6).half().float()
    l32 = O.iou_loss(p, t)
    l16 = O.iou_loss(p.half(), t.half()).float()
    d = (l32 - l16).abs()
    assert torch.isfinite(l16).all() and d.max().item() < 4e-2 and d.mean().item() < 5e-3, (d.max().item(), d.mean().item())
    big_t = torch.tensor([[288.0, 288.0, 300.0, 300.0]])
    big_p = torch.tensor([[280.0, 290.0, 310.0, 290.0]])
    assert not torch.isfinite(O.iou_loss(big_p.half(), big_t.half()).float()).all()       # the reference's fp16 area overflows
    assert torch.isfinite(O.iou_loss(big_p, big_t)).all()
