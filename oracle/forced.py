"""TEST INFRASTRUCTURE -- the CPU oracle evaluated with the PRODUCT's integer decisions handed in.

At the benchmark size (TSCD-L, 16 frames @576^2, 500 proposals per frame, random init) the fp32 oracle and the fp16 product take
different near-tied decisions -- SimOTA foreground anchors, the 500-of-6804 proposal sets, Hungarian columns on fp16 costs -- so
their loss terms are sums over different samples.  Each of those decisions is verified on its own, bit for bit, on identical inputs
(tests/test_gpu_fullsize_c3.py: SimOTA, selection, assignment chain).  What is left to pin is the ARITHMETIC between the decisions:
this module feeds the product's decisions (captured by ``TSCDHead.debug``) into ``oracle.tscd_oracle.tscd_forward`` so that both
sides sum over the same anchors, rows and permutations (reference lines: yolox/models/tscd_head.py:1008-1203 for the losses,
:1712-1851 SimOTA, :1546-1693 selection; tscd_matching.py:912-937 assignments).

Only tests/ and bench.py's cpu_baseline leg import this (the oracle is the checker, never the product path)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import tscd_oracle as O


def forced_aux(dbg, num_classes):
    """``aux`` dict for tscd_forward / tscd_head carrying the product's SimOTA result and proposal sets.

    dbg: TSCDHead.debug of the product's step -- fg_mask u8 [B,A], matched_gt i32 [B,A], pred_iou f32 [B,A] (the matched pair's IoU
    in the fp16 arithmetic of autocast training), agg_idx (per frame ascending anchor indices)."""
    fg_all = dbg["fg_mask"].bool().cpu()
    mg_all = dbg["matched_gt"].cpu().long()
    pi_all = dbg["pred_iou"].float().cpu()
    agg_idx = [a.cpu().long() for a in dbg["agg_idx"]]

    def simota(outputs, labels):
        B, A, _ = outputs.shape
        nlabel = (labels.sum(dim=2) > 0).sum(dim=1)
        fg_ids, cls_t, reg_t, obj_t, fg_masks = [], [], [], [], []
        num_fg, num_gts = 0.0, 0.0
        for b in range(B):
            G = int(nlabel[b])
            num_gts += G
            fg = fg_all[b]
            idx = torch.where(fg)[0]
            g = mg_all[b, idx]
            assert G > 0 or len(idx) == 0
            num_fg += float(len(idx))
            fg_ids.append(idx)
            cls_t.append(F.one_hot(labels[b, g, 0].to(torch.int64), num_classes).float() * pi_all[b, idx].unsqueeze(-1))
            reg_t.append(labels[b, g, 1:5].float())
            obj_t.append(fg.unsqueeze(-1).float())
            fg_masks.append(fg)
        return fg_ids, cls_t, reg_t, obj_t, fg_masks, max(num_fg, 1), num_gts

    return {"force_simota": simota, "force_agg_idx": agg_idx}


class replay_assignments:
    """Context manager: ``O.double_match`` returns the product's (row, column) assignments frame by frame instead of solving."""

    def __init__(self, dbg):
        self.pairs = [(np.asarray(r), np.asarray(c)) for r, c in zip(dbg["hungarian_rows"], dbg["hungarian"])]

    def __enter__(self):
        self._orig = O.double_match
        it = iter(self.pairs)
        O.double_match = lambda *a: next(it)
        return self

    def __exit__(self, *exc):
        O.double_match = self._orig
        return False


def oracle_step(sd, cfg, imgs, targets, temb, lframe, gframe, dbg, grad_names=(), backward=True, retain=(), tower_outputs=False):
    """fp32 oracle train forward (+ backward into ``grad_names``) on the product's decisions.  sd: CPU fp32 state dict (cloned here).
    ``tower_outputs``: additionally hand in the product's trained-tower outputs -- the gathered rows f_cls / f_reg, the class scores and
    the edge block's input maps (dbg["reg_maps"], NHWC) -- so that the proposal stage and the edge block run on IDENTICAL inputs on both
    sides (the towers themselves then receive no gradient in this run).
    -> (loss dict of floats, {name: gradient}, aux)."""
    sd = {k: (v.detach().float().clone() if v.is_floating_point() else v.detach().clone()) for k, v in sd.items()}
    for k in grad_names:
        sd[k].requires_grad_(True)
    aux = forced_aux(dbg, cfg.num_classes)
    if tower_outputs:
        aux["force_rows"] = tuple(dbg[k].detach().float().cpu() for k in ("f_cls", "f_reg", "cls_scores"))
        aux["force_reg_maps"] = [m.detach().permute(0, 3, 1, 2).float().cpu().contiguous() for m in dbg["reg_maps"]]
    with replay_assignments(dbg), torch.set_grad_enabled(bool(grad_names) and backward):
        out = O.tscd_forward(sd, cfg, imgs.float().cpu(), targets.float().cpu(), temb.float().cpu(), 0.5, lframe, gframe, False, True,
                             O.MatcherState(), aux)
        if grad_names and backward:
            for k in retain:                    # activation gradients for diagnostics (e.g. the size of d f_edge next to fp16's range)
                if torch.is_tensor(aux.get(k)) and aux[k].requires_grad:
                    aux[k].retain_grad()
            out["total_loss"].backward()
    losses = {k: (float(v.detach()) if torch.is_tensor(v) else float(v)) for k, v in out.items()}
    grads = {k: sd[k].grad for k in grad_names} if backward else {}
    return losses, grads, aux
