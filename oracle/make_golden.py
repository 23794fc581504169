"""TEST INFRASTRUCTURE (build container only) -- generate tests/golden/*.npz from the reference itself.

Run:  cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

Imports /root/reference through oracle/ref_loader.py (third-party stubs, see there), builds the
reference's own ``Exp`` from exps/TSCD_OVIS/ovis_tscd_large.py with a *tiny* width/depth (0.125/0.33,
so the weights fit a fixture of a few MB; every code path is width-independent), seeds it, and records
for two synthetic clips:

* the reference's state_dict (fp32) and the inputs,
* eval outputs (result / result_ori per local frame), incl. a second clip evaluated with
  ``resume=True`` so the CAFM cross-call state is exercised,
* train outputs: the 9 loss terms and gradients of representative parameters,
* intermediates captured with hooks: FPN maps, wavelet edge maps, MCA outputs (agg / agg_iou), CAFM
  output, TaskAligned output, prediction logits, and the INTEGER outputs: SimOTA fg indices,
  proposal indices (agg_idx), Hungarian permutations.

Only data is written (inputs and expected outputs); no reference source or bytecode is copied.
"""
import os
import sys

sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def synth_clip(B, S, seed, n_gt=3):
    """Blocky low-frequency clip (uint8-valued) + a few bright boxes; differs frame to frame so the
    per-frame proposal counts come out ragged."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.zeros(B, 3, S, S)
    labels = torch.zeros(B, 20, 5)
    for b in range(B):
        cell = (4, 8, 16, 32, 8)[b % 5]
        low = torch.randint(0, 256, (1, 3, S // cell, S // cell), generator=g).float()
        amp = (1.0, 0.35, 0.7, 0.15, 0.5)[b % 5]
        img = 127.0 + amp * (torch.nn.functional.interpolate(low, size=(S, S), mode="nearest")[0] - 127.0)
        for j in range(n_gt):
            cx, cy = 24.0 + 30 * j + 3 * b, 30.0 + 25 * j + 2 * b
            w, h = 28.0 + 6 * j, 36.0 + 4 * j
            labels[b, j] = torch.tensor([float((j + b) % 25), cx, cy, w, h])
            x0, x1 = int(max(cx - w / 2, 0)), int(min(cx + w / 2, S))
            y0, y1 = int(max(cy - h / 2, 0)), int(min(cy + h / 2, S))
            img[:, y0:y1, x0:x1] = torch.tensor([250.0 - 40 * j, 30.0 + 60 * j, 128.0])[:, None, None]
        imgs[b] = img.round().clamp(0, 255)
    return imgs, labels


def main():
    sys.path.insert(0, HERE)
    import ref_loader
    ref_loader.import_reference()
    from yolox.exp import get_exp
    from yolox.data.datasets.vid import get_timing_signal_1d
    from yolox.models.tscd_head import TSCDHead
    from yolox.models.tscd_matching import AwarePositionRegMatcher

    torch.set_num_threads(8)
    exp = get_exp("/root/reference/exps/TSCD_OVIS/ovis_tscd_large.py")
    exp.depth, exp.width = 0.33, 0.125
    exp.minimal_limit, exp.maximal_limit = 140, 165
    torch.manual_seed(0)
    model = exp.get_model()
    B, S, lframe, gframe = 5, 128, 3, 2
    temb = get_timing_signal_1d(torch.arange(0, B), 256)
    # Condition the random init: default init + eval-mode BN at (0,1) running stats makes activations
    # vanish with depth (every output ~ its bias).  Randomise the BN affine and calibrate the running
    # stats with one momentum-1 training-mode pass so eval-mode activations are O(1) and depend on
    # the frame content.
    g = torch.Generator().manual_seed(1)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = 0.8 + 0.4 * torch.rand(m.weight.shape, generator=g)
            m.bias.data = 0.2 * torch.randn(m.bias.shape, generator=g)
            m.momentum = 1.0
    model.train()
    cal_imgs, cal_labels = synth_clip(B, S, 7)
    with torch.no_grad():
        model(cal_imgs, cal_labels, temb, lframe=lframe, gframe=gframe)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 0.03
            m.num_batches_tracked.zero_()
    model.head.local_reg_matcher.last_outputs = None
    # proposal-count control: scores straddle conf_thresh so per-frame counts are ragged
    torch.manual_seed(5)
    for conv in model.head.obj_preds:
        conv.bias.data.fill_(-3.0)
        torch.nn.init.normal_(conv.weight, std=0.3)
    for conv in model.head.cls_preds:
        conv.bias.data.fill_(-4.5)
        torch.nn.init.normal_(conv.weight, std=0.3)

    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rec = {}

    # ---- capture integer intermediates by wrapping reference methods ----
    orig_pw = TSCDHead.postprocess_widx
    orig_fg = TSCDHead.get_fg_idx
    orig_dm = AwarePositionRegMatcher.double_match_embds
    cap = {}

    def pw(self, *a, **k):
        r = orig_pw(self, *a, **k)
        cap["agg_idx"] = [x.clone() for x in r[1]]
        cap["pred_result"] = [x.detach().clone() for x in r[0]]
        return r

    def fg(self, *a, **k):
        r = orig_fg(self, *a, **k)
        cap["fg_idx"] = [torch.as_tensor(x).clone() for x in r[0]]
        cap["num_fg"] = float(r[5])
        return r

    def dm(self, *a, **k):
        r = orig_dm(self, *a, **k)
        cap.setdefault("hungarian", []).append((np.asarray(r[0]).copy(), np.asarray(r[1]).copy()))
        return r

    TSCDHead.postprocess_widx = pw
    TSCDHead.get_fg_idx = fg
    AwarePositionRegMatcher.double_match_embds = dm

    hooks = {}

    def hook(name):
        def f(mod, inp, out):
            hooks[name] = out
        return f

    h = model.head
    model.backbone.register_forward_hook(hook("fpn"))
    h.agg.register_forward_hook(hook("agg"))
    h.agg_iou.register_forward_hook(hook("agg_iou"))
    h.local_reg_matcher.register_forward_hook(hook("cafm"))
    h.task_aligned.register_forward_hook(hook("task_aligned"))
    h.cls_pred.register_forward_hook(hook("cls_preds"))
    h.matcher_obj_pred.register_forward_hook(hook("matcher_obj_preds"))
    h.matcher_reg_pred.register_forward_hook(hook("matcher_reg_preds"))
    for k in range(3):
        h.edge_enhance_reg[k].register_forward_hook(hook(f"edge{k}"))
        h.reg_convs2[k].register_forward_hook(hook(f"reg2_{k}"))

    def dump(prefix):
        rec[prefix + "n_agg"] = np.array([len(x) for x in cap["agg_idx"]])
        for i, x in enumerate(cap["agg_idx"]):
            rec[f"{prefix}agg_idx{i}"] = x.numpy()
        for i, x in enumerate(cap["pred_result"]):
            rec[f"{prefix}pred_result{i}"] = x.numpy()
        for i, (r, c) in enumerate(cap.get("hungarian", [])):
            rec[f"{prefix}hung_row{i}"] = r
            rec[f"{prefix}hung_col{i}"] = c
        rec[prefix + "n_hung"] = np.array(len(cap.get("hungarian", [])))
        for i, t in enumerate(hooks["fpn"]):
            rec[f"{prefix}fpn{i}"] = t.detach().numpy()
        for k in range(3):
            rec[f"{prefix}edge{k}"] = hooks[f"edge{k}"].detach().numpy()
            rec[f"{prefix}reg2_{k}"] = hooks[f"reg2_{k}"].detach().numpy()
        rec[prefix + "agg_cls"] = hooks["agg"][0].detach().numpy()
        rec[prefix + "agg_iou_cls"] = hooks["agg_iou"][0].detach().numpy()
        rec[prefix + "agg_iou_obj"] = hooks["agg_iou"][1].detach().numpy()
        rec[prefix + "cafm"] = hooks["cafm"][-1].detach().numpy()
        rec[prefix + "task_aligned"] = hooks["task_aligned"].detach().numpy()
        rec[prefix + "cls_preds"] = hooks["cls_preds"].detach().numpy()
        rec[prefix + "matcher_obj_preds"] = hooks["matcher_obj_preds"].detach().numpy()
        rec[prefix + "matcher_reg_preds"] = hooks["matcher_reg_preds"].detach().numpy()

    rec["time_embedding"] = temb.numpy()

    # ---- eval: clip A (resume=False) then clip B (resume=True) ----
    model.eval()
    for name, seed, resume in (("evalA_", 10, False), ("evalB_", 11, True)):
        imgs, labels = synth_clip(B, S, seed)
        rec[name + "imgs"] = imgs.numpy().astype(np.uint8)
        cap.clear()
        hooks.clear()
        with torch.no_grad():
            res, res_ori = model(imgs, time_embedding=temb, lframe=lframe, gframe=gframe, resume=resume)
        dump(name)
        for i in range(lframe):
            rec[f"{name}result{i}"] = res[i].numpy() if res[i] is not None else np.zeros((0, 7), np.float32)
            rec[f"{name}result_ori{i}"] = (res_ori[i].numpy() if res_ori[i] is not None
                                            else np.zeros((0, 7), np.float32))
        print(name, "proposals/frame", rec[name + "n_agg"], "dets", [len(rec[f"{name}result{i}"]) for i in range(lframe)])

    # ---- train: clip A, fwd + bwd ----
    model.load_state_dict(sd0)
    model.train()
    imgs, labels = synth_clip(B, S, 10)
    rec["train_labels"] = labels.numpy()
    cap.clear()
    hooks.clear()
    out = model(imgs, labels, temb, lframe=lframe, gframe=gframe, resume=False)
    dump("train_")
    for i, x in enumerate(cap["fg_idx"]):
        rec[f"train_fg_idx{i}"] = x.numpy().astype(np.int64)
    rec["train_num_fg"] = np.array(cap["num_fg"])
    for k, v in out.items():
        rec["train_loss_" + k] = np.array(float(v))
    print({k: float(v) for k, v in out.items()})
    out["total_loss"].backward()
    named = dict(model.named_parameters())
    grads_with = [k for k, p in named.items() if p.grad is not None]
    rec["train_grad_names"] = np.array(grads_with)
    keep = ["head.cls_convs2.0.0.conv.weight", "head.cls_convs2.0.1.conv.weight", "head.reg_convs2.1.0.conv.weight",
            "head.reg_convs2.2.1.bn.weight", "head.edge_enhance_reg.0.0.filter1.0.weight",
            "head.edge_enhance_reg.1.0.filter2.0.weight", "head.agg.mca.q_cls_local.weight",
            "head.agg.mca.kv_reg.weight", "head.agg.mca.linear.weight", "head.agg.linear.weight",
            "head.agg_iou.mca.kv_cls.weight", "head.agg_iou.mca.linear_reg.weight", "head.agg_iou.linear_obj.weight",
            "head.local_reg_matcher.transformer_aware_cross_attention_layers.0.multihead_attn.q_reg.weight",
            "head.local_reg_matcher.transformer_aware_cross_attention_layers.0.CA.fc.0.weight",
            "head.local_reg_matcher.absolute_position_embedding.weight", "head.fc_reg_matcher.weight",
            "head.task_aligned.transformer_cross_attention_layers.0.multihead_attn.k_reg.weight",
            "head.cls_pred.weight", "head.matcher_obj_pred.bias", "head.matcher_reg_pred.weight",
            "head.cls_preds.0.weight", "head.cls_preds.1.bias", "head.obj_preds.2.weight"]
    for k in keep:
        rec["train_grad::" + k] = named[k].grad.numpy()
    # BN running stats after one train step (F4)
    sd1 = model.state_dict()
    for k in ("backbone.backbone.stem.conv.bn.running_mean", "backbone.backbone.dark5.0.bn.running_var",
              "head.cls_convs2.0.0.bn.running_var", "head.stems.1.bn.running_mean"):
        rec["train_bn::" + k] = sd1[k].numpy()

    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "tscd_tiny_weights.npz"), **{k: v.numpy() for k, v in sd0.items()})
    np.savez_compressed(os.path.join(OUT, "tscd_tiny_vectors.npz"), **rec)
    meta = dict(depth=0.33, width=0.125, num_classes=25, minimal_limit=140, maximal_limit=165, B=B, S=S,
                lframe=lframe, gframe=gframe)
    np.savez(os.path.join(OUT, "tscd_tiny_meta.npz"), **{k: np.array(v) for k, v in meta.items()})
    for f in os.listdir(OUT):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
