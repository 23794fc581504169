"""BASELINE config C3 at its full workload: TSCD-L, 16 frames (4 local + 12 global) @576^2, 500 proposals per frame,
train step -- the benchmark's own model, clip and targets (bench.build_model / synth_clip / targets_on_top_proposals).

Every proposal-stage tensor of the product's step is checked against the CPU oracle evaluated on the PRODUCT's own fp16
intermediates (so each stage is compared on identical inputs at N1 = 500, N2 = 6500 per local frame; the MCA aggregate /
backward kernels run with their full split factor, gridDim.y = 16):

  SimOTA          oracle in autocast arithmetic on the product's fp16 outputs   -> foreground set, matched gt, IoU: exact
  selection       reference rule on the product's fp16 scores                   -> anchor indices: exact
  MCA x2 forward  oracle fp32 on the gathered fp16 rows                         -> 2e-2 of max|ref|, 1e-2 of mean|ref|
  MCA backward    oracle autograd, random upstream gradients                    -> cosine > 0.999, norm within 2 %
  Hungarian       scipy chain on the device-produced fp16 cost matrices         -> columns: exact
  CAFM            oracle attention with the product's assignments replayed      -> 6e-2 / 1.5e-2 (LayerNorm'd rows, rms 1)
  TaskAligned, prediction linears                                               -> 6e-2 / 1.5e-2, 3e-2 / 1e-2
and the step itself must be non-degenerate: 500 proposals in every frame, foreground rows among the selected local
proposals, non-zero refined-head losses and gradients.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tscd_oracle as O

L, G, S, NC = 4, 12, 576, 25
STRIDES = (8, 16, 32)


def close(got, ref, rel_max, rel_mean):
    got, ref = torch.as_tensor(got).float().cpu(), torch.as_tensor(ref).float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = (got - ref).abs()
    assert err.max().item() <= rel_max * ref.abs().max().item() + 1e-4, (err.max().item(), ref.abs().max().item())
    assert err.mean().item() <= rel_mean * ref.abs().mean().item() + 1e-5, (err.mean().item(), ref.abs().mean().item())


@pytest.fixture(scope="module")
def c3():
    import bench
    model = bench.build_model(1.0, 1.0, "500").cuda().train()
    imgs, t0, temb = bench.synth_clip(L + G, S, "cuda")
    targets = bench.targets_on_top_proposals(model, imgs, t0, temb, L, G)
    sd = {k: v.detach().float().cpu().clone() if v.is_floating_point() else v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.head.debug = {}
    scale = torch.full((), float(__import__("os").environ.get("TSCD_C3_SCALE", "1024")), device="cuda")
    model.head.loss_scale = scale
    out = model(imgs, targets, temb, lframe=L, gframe=G, resume=False)
    (out["total_loss"] * scale).backward()
    torch.cuda.synchronize()
    dbg = dict(model.head.debug)
    model.head.debug = None
    # the step's parameter gradients, before any later test runs a backward of its own through these modules
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    cfg = O.TSCDConfig(num_classes=NC, minimal_limit=50, maximal_limit=500)
    return dict(model=model, imgs=imgs, targets=targets, temb=temb, out=out, dbg=dbg, sd=sd, cfg=cfg, scale=float(scale), grads=grads)


def test_c3_step_is_the_workload(c3):
    dbg, out, m = c3["dbg"], c3["out"], c3["model"]
    assert dbg["preds_per_frame"] == [500] * (L + G)
    assert int(dbg["refined_fg_rows"]) >= L                               # SimOTA foreground among the selected local proposals
    for k in ("total_loss", "iou_loss", "conf_loss", "cls_loss", "loss_matched_cls", "loss_matched_obj", "loss_matched_iou"):
        assert torch.isfinite(torch.as_tensor(out[k])).all(), k
    assert float(out["loss_matched_cls"]) > 0 and float(out["loss_matched_iou"]) > 0
    for name in ("head.agg.mca.kv_cls.weight", "head.cls_pred.weight", "head.matcher_reg_pred.weight", "head.fc_reg_matcher.weight",
                 "head.local_reg_matcher.transformer_aware_cross_attention_layers.0.multihead_attn.q_reg.weight",
                 "head.task_aligned.transformer_cross_attention_layers.0.multihead_attn.k_reg.weight", "head.cls_convs2.0.0.conv.weight",
                 "head.reg_convs2.2.1.conv.weight", "head.edge_enhance_reg.0.0.filter1.0.weight"):
        g = dict(m.named_parameters())[name].grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0, name


def test_c3_simota_exact(c3):
    dbg, cfg = c3["dbg"], c3["cfg"]
    hw = [(S // s, S // s) for s in STRIDES]
    xs, ys, ss = O.make_grids(hw, STRIDES, torch.float16)
    ref = O.simota_assign(dbg["outputs"].cpu(), c3["targets"].cpu().half(), xs, ys, ss, cfg)
    fg = dbg["fg_mask"].bool().cpu()
    want = torch.stack(list(ref[4]))
    assert torch.equal(fg, want), (int(fg.sum()), int(want.sum()), int((fg != want).sum()))
    lab = c3["targets"].cpu().float()
    mg, pi = dbg["matched_gt"].cpu().long(), dbg["pred_iou"].cpu()
    for b in range(L + G):
        idx = torch.nonzero(fg[b]).flatten()
        assert torch.equal(lab[b, mg[b, idx], 1:5], ref[2][b].float())            # matched boxes
        want_iou = ref[1][b].float().sum(-1).numpy()                                 # NaN where the reference's sum is poisoned
        num = ~np.isnan(want_iou)                                                    # (0 * NaN from a far-away gt: simota.hip header)
        assert np.array_equal(pi[b, idx].numpy()[num], want_iou[num])                # matched IoU (fp16 values): exact
        assert np.isfinite(pi[b, idx].numpy()[~num]).all()
    assert dbg["num_fg"].cpu().tolist() == want.sum(1).tolist()


def test_c3_selection_exact(c3):
    dbg = c3["dbg"]
    score, sel = dbg["score"].float().cpu(), dbg["sel_idx"].cpu()
    for b in range(L + G):
        s = score[b].numpy()
        assert (s >= np.float32(np.float16(0.001))).sum() > 500                   # the 500 clamp binds
        order = np.lexsort((np.arange(len(s)), -s))                                # value desc, then lower anchor index
        want = np.sort(order[:500])
        assert np.array_equal(sel[b, :500].numpy(), want), b


def _mca_oracle(c3, prefix, need_grad=False):
    dbg, cfg, sd = c3["dbg"], c3["cfg"], c3["sd"]
    xc, xr, sc = dbg["f_cls"].detach().float().cpu(), dbg["f_reg"].detach().float().cpu(), dbg["cls_scores"].detach().float().cpu()
    if need_grad:
        xc, xr, sc = xc.requires_grad_(), xr.requires_grad_(), sc.requires_grad_()
        sd = {k: (v.clone().requires_grad_(True) if k.startswith(prefix) and v.is_floating_point() else v) for k, v in sd.items()}
    return O.mca_tscd_g2l_reg(sd, prefix, xc, xr, sc, dbg["preds_per_frame"], L, cfg), (xc, xr, sc, sd)


def test_c3_mca_forward_both_modules(c3):
    dbg = c3["dbg"]
    with torch.no_grad():
        (a_cls, _), _ = _mca_oracle(c3, "head.agg")
        (i_cls, i_obj), _ = _mca_oracle(c3, "head.agg_iou")
    close(dbg["agg_cls"], a_cls, 2e-2, 1e-2)
    close(dbg["agg_iou_cls"], i_cls, 2e-2, 1e-2)
    close(dbg["agg_iou_obj"], i_obj, 2e-2, 1e-2)


def test_c3_mca_backward_full_size(c3):
    dbg, m = c3["dbg"], c3["model"]
    g = torch.Generator().manual_seed(5)
    nl = 500 * L
    gc, go = torch.randn(nl, 1024, generator=g), torch.randn(nl, 1024, generator=g)
    (rc, ro), (xc, xr, sc, sd) = _mca_oracle(c3, "head.agg_iou", need_grad=True)
    ((rc * gc).sum() + (ro * go).sum()).backward()
    mod = m.head.agg_iou
    for p in mod.parameters():
        p.grad = None
    req = {n: p.requires_grad for n, p in mod.named_parameters()}
    for p in mod.parameters():
        p.requires_grad_(True)
    xcg, xrg = dbg["f_cls"].detach().clone().requires_grad_(), dbg["f_reg"].detach().clone().requires_grad_()
    scg = dbg["cls_scores"].detach().clone().requires_grad_()
    c, o = mod(xcg, xrg, scg, None, sim_thresh=0.75, preds_per_frame=dbg["preds_per_frame"], lframe=L, conf_sim_thresh=0.99)
    ((c.float() * gc.cuda()).sum() + (o.float() * go.cuda()).sum()).backward()
    torch.cuda.synchronize()
    checks = [(xcg.grad, xc.grad, "x_cls"), (xrg.grad, xr.grad, "x_reg"), (scg.grad, sc.grad, "score")]
    checks += [(p.grad, sd["head.agg_iou." + k].grad, k) for k, p in mod.named_parameters()]
    for n, p in mod.named_parameters():
        p.requires_grad_(req[n])
    for got, ref, name in checks:
        cos = torch.nn.functional.cosine_similarity(got.float().cpu().flatten(), ref.flatten(), dim=0).item()
        ratio = (got.float().cpu().norm() / ref.norm()).item()
        assert cos > 0.999 and abs(ratio - 1) < 0.02, (name, cos, ratio)


def _scipy_chain(costs_u16, kinds):
    from scipy.optimize import linear_sum_assignment
    order, out = None, []
    for c16, kind in zip(costs_u16, kinds):
        C = c16.view(np.float16).astype(np.float64)
        C = np.where(np.isnan(C), 0.0, C)
        if kind == 0:
            r, c = linear_sum_assignment(C)
            order = np.arange(C.shape[1])
        else:
            r, c = linear_sum_assignment(C[order])
            rest = np.setdiff1d(np.arange(C.shape[1]), c)
            order = np.concatenate((c, rest)) if C.shape[0] < C.shape[1] else c
        out.append((r, c))
    return out


def test_c3_hungarian_chain_exact_and_cafm_task_aligned(c3, monkeypatch):
    from tscd_amd import ops
    dbg, cfg, sd, m = c3["dbg"], c3["cfg"], c3["sd"], c3["model"]
    reg, cls = dbg["agg_iou_obj"].detach(), dbg["agg_iou_cls"].detach()
    spans = [(500 * i, 500 * (i + 1)) for i in range(L)]
    costs = []
    for (pi, ci) in [(0, 0)] + [(i - 1, i) for i in range(1, L)]:
        (p0, p1), (c0, c1) = spans[pi], spans[ci]
        dc = ops.match_cost(reg[p0:p1].contiguous(), reg[c0:c1].contiguous(), cls[p0:p1].contiguous(), cls[c0:c1].contiguous())
        costs.append(np.ascontiguousarray(dc.cpu().numpy().view(np.uint16)))
    want = _scipy_chain(costs, [0] + [2] * (L - 1))
    got = [np.asarray(x) for x in dbg["hungarian"]]
    assert len(got) == L
    for fi in range(L):
        assert np.array_equal(got[fi], want[fi][1]), fi                       # INT, 500 x 500, tie-heavy random-init costs: exact
    # ---- CAFM: the oracle's attention with the product's (exactly verified) assignments replayed ----
    replay = iter(zip(dbg["hungarian_rows"], dbg["hungarian"]))
    monkeypatch.setattr(O, "double_match", lambda *a: tuple(np.asarray(x) for x in next(replay)))
    nl = 500 * L
    with torch.no_grad():
        ref = O.aware_position_reg_matcher(sd, "head.local_reg_matcher", dbg["f_reg"].float().cpu(), reg.float().cpu(), cls.float().cpu(),
                                           dbg["f_edge"].float().cpu(), dbg["preds_per_frame"][:L], c3["temb"][:L].float().cpu(), False,
                                           O.MatcherState(), cfg)
        close(dbg["cafm"], ref, 6e-2, 1.5e-2)
        # fc_reg_matcher + TaskAligned + prediction linears on the product's own inputs
        F = torch.nn.functional
        cafm = dbg["cafm"].float().cpu()
        m4 = F.linear(cafm, sd["head.fc_reg_matcher.weight"], sd["head.fc_reg_matcher.bias"])
        close(dbg["matched4"], m4, 3e-2, 1e-2)
        m4p = dbg["matched4"].float().cpu()
        ta = O.task_aligned(sd, "head.task_aligned", m4p, reg.float().cpu(), dbg["preds_per_frame"][:L], cfg)
        close(dbg["obj_refined"], ta, 6e-2, 1.5e-2)
        close(dbg["cls_preds"], F.linear(dbg["agg_cls"].float().cpu(), sd["head.cls_pred.weight"], sd["head.cls_pred.bias"]), 3e-2, 1e-2)
        close(dbg["matcher_obj_preds"], F.linear(dbg["obj_refined"].float().cpu(), sd["head.matcher_obj_pred.weight"],
                                                 sd["head.matcher_obj_pred.bias"]), 3e-2, 1e-2)
        close(dbg["matcher_reg_preds"], F.linear(m4p, sd["head.matcher_reg_pred.weight"], sd["head.matcher_reg_pred.bias"]), 3e-2, 1e-2)


# ---------------------------------------------------------------------------------------------------------------------------
# The whole step against the oracle with the product's integer decisions forced
# ---------------------------------------------------------------------------------------------------------------------------
GRAD_NAMES = None


def _grad_names():
    """The 24 representative parameters of the tiny golden (towers, wavelet block, both MCAs, CAFM, TaskAligned, prediction heads)."""
    import os
    global GRAD_NAMES
    if GRAD_NAMES is None:
        vec = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tscd_tiny_vectors.npz"))
        GRAD_NAMES = sorted(k[len("train_grad::"):] for k in vec.keys() if k.startswith("train_grad::"))
    return GRAD_NAMES


@pytest.fixture(scope="module")
def c3_forced(c3):
    """fp32 CPU oracle, forward + backward at the benchmark size, on the product's SimOTA result (foreground set, matched boxes, matched
    IoUs), proposal sets and assignment permutations -- each of which is verified exactly by the tests above."""
    from oracle import forced
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    names = [n for n in _grad_names() if n in c3["sd"]]
    import dataclasses
    # the reference's training arithmetic for the box decode: fp16 under autocast (oracle TSCDConfig.decode_autocast)
    cfg = dataclasses.replace(c3["cfg"], decode_autocast=True)
    losses, grads, aux = forced.oracle_step(c3["sd"], cfg, c3["imgs"], c3["targets"], c3["temb"], L, G, c3["dbg"], names,
                                            retain=("f_edge", "f_reg", "f_cls"))
    return dict(losses=losses, grads=grads, aux=aux, names=names)


def test_c3_losses_with_product_decisions_forced(c3, c3_forced):
    """All 8 loss terms of the benchmark's step 0 within 2 % of the fp32 oracle (loss_matched_iou -- 6 x SmoothL1 at the end of the
    longest fp16 chain -- within 7 %), i.e. the bars of the tiny-clip test (tests/test_gpu_e2e.py::test_train_with_the_reference_
    assignment_and_proposal_sets_is_tight) at TSCD-L / 16 x 576^2 / 500 proposals.  Reference: tscd_head.py:1008-1203."""
    out, ref = c3["out"], c3_forced["losses"]
    aux, dbg = c3_forced["aux"], c3["dbg"]
    assert aux["preds_per_frame"] == dbg["preds_per_frame"] == [500] * (L + G)
    for b in range(L + G):
        assert torch.equal(aux["agg_idx"][b], dbg["agg_idx"][b].cpu().long())
    rel = {}
    for k in ("total_loss", "iou_loss", "conf_loss", "cls_loss", "num_fg", "loss_matched_cls", "loss_matched_obj", "loss_matched_iou"):
        rel[k] = abs(float(out[k]) - ref[k]) / max(abs(ref[k]), 1e-6)
    print("C3 loss terms (product / oracle on the product's decisions):",
          {k: (round(float(out[k]), 4), round(ref[k], 4), round(rel[k], 5)) for k in rel})
    # iou_loss: the benchmark's targets are the product's own boxes of 6 proposals per frame, so every other evaluation of the network
    # scores ~5 % worse against them -- the fp32 oracle AND the reference's fp16 arithmetic alike; test_c3_conv_stack_against_the_oracle
    # shows the boxes agree box for box within fp16 noise and that the bias flips with the source of the targets.  Bar: 7 %, as for the
    # other regression term
    assert all(v <= 0.02 for k, v in rel.items() if k not in ("loss_matched_iou", "iou_loss")), rel
    assert rel["loss_matched_iou"] <= 0.07 and rel["iou_loss"] <= 0.07, rel


def _grad_cosines(c3, names, ref_grads):
    cosines, ratios = {}, {}
    for k in names:
        r, g = ref_grads.get(k), c3["grads"].get(k)
        if r is None or g is None:
            continue
        g = g.float().cpu().flatten() / c3["scale"]            # the fixture backpropagates total_loss x the loss scale
        r = r.flatten()
        cosines[k] = torch.nn.functional.cosine_similarity(g, r, dim=0).item()
        ratios[k] = (g.norm() / r.norm()).item()
    return cosines, ratios


def test_c3_gradients_with_product_decisions_forced(c3, c3_forced):
    """The 24 representative parameter gradients (names of the tiny golden) against the fp32 oracle's autograd at full size, the whole
    network in one oracle run (decisions forced, everything else the oracle's own fp32 arithmetic from the images on): >= 10 at cosine
    >= 0.99, all but the two wavelet-block convs >= 0.94, norms within 5 % where the cosine is >= 0.995 (10 % from 0.99).  The wavelet block's filters
    (conv + ReLU on the towers' noisy outputs: ~10 % of max|activation| of fp16 noise at random init flips ReLU masks on the ~2000
    rows that carry gradient) sit at 0.69-0.74 / 0.89 in this run -- the value is a property of the noise realisation: it moved from 0.74 to
    0.69 when the frozen front's batch statistics changed their summation order (CSP pair fusion, 224-row conv tiles) with no change to
    the block or its inputs' producers -- and at the tight bar on identical inputs (test_c3_proposal_stage_gradients_on_identical_tower_outputs),
    which is the test that pins this block."""
    cosines, ratios = _grad_cosines(c3, c3_forced["names"], c3_forced["grads"])
    for k in ("f_edge", "f_reg", "f_cls"):
        ga = c3_forced["aux"][k].grad
        if ga is not None:
            a = ga.abs() * c3["scale"]
            print(f"oracle d {k} x loss scale: max {float(a.max()):.3g}, median {float(a.median()):.3g}, "
                  f"share below fp16's smallest normal (6.1e-5) {float((a < 6.1e-5).float().mean()):.3f}")
    print("C3 gradient cosines:", {k.replace("head.", ""): (round(v, 4), round(ratios[k], 3)) for k, v in sorted(cosines.items(), key=lambda kv: kv[1])})
    assert len(cosines) >= 20, len(cosines)
    assert sum(v >= 0.99 for v in cosines.values()) >= 10, cosines
    for k, v in cosines.items():
        assert v >= (0.60 if "edge_enhance_reg" in k else 0.94), (k, v)
        # norms: within 5 % where the direction is pinned (>= 0.995), 10 % in the band just above 0.99 (the SE gate's 64-element first
        # layer sits AT 0.99 with a norm ratio of 0.92: ReLU masks on fp16-noisy inputs -- it crosses the 0.99 line between builds)
        if v >= 0.995:
            assert abs(ratios[k] - 1) <= 0.05, (k, ratios[k])
        elif v >= 0.99:
            assert abs(ratios[k] - 1) <= 0.10, (k, ratios[k])


@pytest.fixture(scope="module")
def c3_forced_rows(c3):
    """A second oracle run with, additionally, the product's trained-tower OUTPUTS handed in (gathered rows, class scores, the edge
    block's input maps): proposal stage and edge block then see identical inputs on both sides."""
    import dataclasses
    from oracle import forced
    names = [n for n in _grad_names() if n in c3["sd"] and "convs2" not in n]
    cfg = dataclasses.replace(c3["cfg"], decode_autocast=True)
    losses, grads, aux = forced.oracle_step(c3["sd"], cfg, c3["imgs"], c3["targets"], c3["temb"], L, G, c3["dbg"], names, tower_outputs=True)
    return dict(losses=losses, grads=grads, aux=aux, names=names)


def test_c3_proposal_stage_gradients_on_identical_tower_outputs(c3, c3_forced_rows):
    """Edge block, both MCAs, CAFM, TaskAligned and the prediction linears at full size on IDENTICAL inputs (the product's tower outputs
    and decisions handed to the oracle): the refined-head loss terms within 2 % and every gradient at cosine >= 0.99, norms within 3 %
    (measured: all 20 at >= 0.998, the two wavelet-block filters at 1.000 -- the 0.74 / 0.89 of the whole-network run are the towers'
    fp16 noise reaching their ReLUs, not these kernels)."""
    out, ref = c3["out"], c3_forced_rows["losses"]
    rel = {k: abs(float(out[k]) - ref[k]) / max(abs(ref[k]), 1e-6) for k in ("loss_matched_cls", "loss_matched_obj", "loss_matched_iou")}
    print("C3 refined-head losses on identical tower outputs:", {k: (round(float(out[k]), 4), round(ref[k], 4), round(v, 5)) for k, v in rel.items()})
    cosines, ratios = _grad_cosines(c3, c3_forced_rows["names"], c3_forced_rows["grads"])
    print("C3 gradient cosines on identical tower outputs:",
          {k.replace("head.", ""): (round(v, 4), round(ratios[k], 3)) for k, v in sorted(cosines.items(), key=lambda kv: kv[1])})
    assert all(v <= 0.02 for v in rel.values()), rel
    assert len(cosines) >= 14, len(cosines)
    for k, v in cosines.items():
        assert v >= 0.99 and abs(ratios[k] - 1) <= 0.03, (k, v, ratios[k])


def test_c3_conv_stack_against_the_oracle(c3, c3_forced):
    """TSCD-L (width 1.0 / depth 1.0) backbone + PAFPN + stems + towers + Haar edge block at 16 x 576^2 against the fp32 oracle: the FPN
    maps and, on the proposal rows, the tower outputs (cls_convs2 -> f_cls, reg_convs2 -> f_reg, edge block -> f_edge) and the raw
    prediction maps -- each within 1.5x the fp16 noise of the oracle's own PyTorch ops run in fp16 on the device (SURVEY 8d rule)."""
    m, aux, dbg, cfg = c3["model"], c3_forced["aux"], c3["dbg"], c3["cfg"]
    imgs = c3["imgs"]
    sdh = {k: (v.cuda().half() if v.is_floating_point() else v.cuda()) for k, v in c3["sd"].items()}
    with torch.no_grad():
        # ---- fp16 noise floor: the oracle's ops in fp16 on the device, same weights, batch statistics
        nz_fpn = O.yolo_pafpn(sdh, imgs.half(), cfg, True)
        nz = {"fpn": [t.float().cpu() for t in nz_fpn]}
        idx = [a.cuda().long() for a in dbg["agg_idx"]]
        rows = lambda flat: torch.cat([flat[i][idx[i]] for i in range(L + G)]).float().cpu()
        cls_f, reg_f, edge_f, raw = [], [], [], []
        for k, x in enumerate(nz_fpn):
            x = O.base_conv(sdh, f"head.stems.{k}", x, 1, True)
            rf = O.tower(sdh, f"head.reg_convs.{k}", x, True)
            cf = O.tower(sdh, f"head.cls_convs.{k}", x, True)
            v = O.tower(sdh, f"head.cls_convs2.{k}", x, True)
            vr = O.tower(sdh, f"head.reg_convs2.{k}", x, True)
            raw.append(torch.cat([O.conv_pred(sdh, f"head.reg_preds.{k}", rf), O.conv_pred(sdh, f"head.obj_preds.{k}", rf),
                                  O.conv_pred(sdh, f"head.cls_preds.{k}", cf)], 1).flatten(2))
            cls_f.append(v.flatten(2))
            reg_f.append(vr.flatten(2))
            edge_f.append(O.wavelets_hf_block(sdh, f"head.edge_enhance_reg.{k}.0", vr).flatten(2))
        nz["f_cls"] = rows(torch.cat(cls_f, 2).permute(0, 2, 1))
        nz["f_reg"] = rows(torch.cat(reg_f, 2).permute(0, 2, 1))
        nz["f_edge"] = rows(torch.cat(edge_f, 2).permute(0, 2, 1))
        nz["raw"] = torch.cat(raw, 2).permute(0, 2, 1).float().cpu()
        del sdh, nz_fpn
        # ---- the product's pyramid (training-mode BatchNorm: batch statistics, as in the step)
        was = m.training
        m.train()
        fpn = m.backbone(imgs)
        m.train(was)
        torch.cuda.synchronize()

    def check(name, got, ref, noise):
        got, ref, noise = got.float().cpu(), ref.float().cpu(), noise.float().cpu()
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        scale = ref.abs().max().item()
        err, nerr = (got - ref).abs(), (noise - ref).abs()
        print(f"C3 conv stack {name}: max err {err.max().item():.4g} (fp16 noise {nerr.max().item():.4g}), mean {err.mean().item():.4g} "
              f"(noise {nerr.mean().item():.4g}), scale {scale:.4g}")
        assert err.max().item() <= max(1.5 * nerr.max().item(), 2e-3 * scale), (name, err.max().item(), nerr.max().item(), scale)
        assert err.mean().item() <= max(1.5 * nerr.mean().item(), 5e-4 * scale), (name, err.mean().item(), nerr.mean().item())

    for i in range(3):
        check(f"fpn{i}", fpn[i], aux["fpn"][i], nz["fpn"][i])         # the pyramid maps are NCHW-logical (channels-last memory)
    check("f_cls", dbg["f_cls"].detach(), aux["f_cls"].detach(), nz["f_cls"])
    check("f_reg", dbg["f_reg"].detach(), aux["f_reg"].detach(), nz["f_reg"])
    check("f_edge", dbg["f_edge"].detach(), aux["f_edge"].detach(), nz["f_edge"])
    # raw prediction maps [B,A,5+NC]: ``outputs`` are decoded (xy in pixels, wh = exp(logit) * stride, which overflows fp16 for the
    # largest random-init logits): compare in logit space, on the entries that are finite everywhere
    hw = [(S // s, S // s) for s in STRIDES]
    xs, ys, ss = O.make_grids(hw, STRIDES)
    grid, sst = torch.stack((xs, ys), 1)[None], ss[None, :, None]
    dec = lambda t: torch.cat(((t[..., :2] + grid) * sst, torch.exp(t[..., 2:4]) * sst, t[..., 4:]), -1)
    und = lambda o: torch.cat((o[..., :2] / sst - grid, torch.log(o[..., 2:4] / sst), o[..., 4:]), -1)
    got, ref, noise = und(dbg["outputs"].detach().float().cpu()), und(aux["outputs"].detach().float()), nz["raw"]
    ok = torch.isfinite(got) & torch.isfinite(ref) & torch.isfinite(noise)
    assert float(ok.float().mean()) > 0.99
    z = lambda t: torch.where(ok, t, torch.zeros_like(t))
    check("raw prediction logits", z(got), z(ref), z(noise))
    # ---- the IoU-loss term (3 (1 - IoU^2) summed over the foreground anchors, tscd_head.py:1157-1160) on the three sets of boxes, same
    # anchors, same targets.  The product's sum is ~5 % below BOTH oracles' (which agree to 0.2 %): the benchmark's targets are the
    # product's own decoded boxes of 6 proposals per frame (bench.targets_on_top_proposals), and the foreground anchors are those
    # proposals' neighbours, whose random-init predictions are correlated with them -- any OTHER evaluation of the network (fp32, or the
    # reference's fp16 arithmetic) adds independent noise to that correlation and scores a lower IoU against targets taken from the
    # product.  Checked here, independent of the targets: (1) box for box, the product is as close to the fp32 oracle as the fp16 oracle
    # is (1.5x); (2) the bias follows the evaluation the targets are taken from: with the SAME six proposals' boxes read from the fp32
    # oracle instead, the fp32 oracle's sum is the lower one, by a comparable margin.
    fg = dbg["fg_mask"].bool().cpu()
    lab = c3["targets"].float().cpu()
    mg = dbg["matched_gt"].cpu().long()
    o_prod, o_ref, o_noise = dbg["outputs"].detach().float().cpu(), aux["outputs"].detach().float(), dec(nz["raw"])
    boxes = lambda o: o[..., :4].reshape(-1, 4)[fg.reshape(-1)]
    b_prod, b_ref, b_noise = boxes(o_prod), boxes(o_ref), boxes(o_noise)
    tgt = torch.cat([lab[b, mg[b, fg[b]], 1:5] for b in range(L + G)])
    s_prod, s_ref, s_noise = (float(O.iou_loss(b, tgt).sum()) for b in (b_prod, b_ref, b_noise))
    d_prod, d_noise = float(O.iou_loss(b_prod, b_ref).mean()), float(O.iou_loss(b_noise, b_ref).mean())
    print(f"C3 IoU-loss sums over {int(fg.sum())} foreground anchors: product {s_prod:.4f}, fp32 oracle {s_ref:.4f}, fp16 oracle {s_noise:.4f}; "
          f"mean 1 - IoU^2 box for box against the fp32 oracle: product {d_prod:.4f}, fp16 oracle {d_noise:.4f}")
    assert d_prod <= max(1.5 * d_noise, 0.01), (d_prod, d_noise)
    # the same proposals' boxes as the fp32 oracle sees them: per ground-truth box, the anchor whose PRODUCT box it was taken from
    lab2 = lab.clone()
    for b in range(L + G):
        for j in range(lab.shape[1]):
            if float(lab[b, j, 3]) <= 0:
                continue
            a = int((o_prod[b, :, :4] - lab[b, j, 1:5]).abs().sum(1).argmin())
            lab2[b, j, 1:5] = torch.cat((o_ref[b, a, :2], o_ref[b, a, 2:4].clamp(8.0, 400.0)))
    tgt2 = torch.cat([lab2[b, mg[b, fg[b]], 1:5] for b in range(L + G)])
    s2_prod, s2_ref = float(O.iou_loss(b_prod, tgt2).sum()), float(O.iou_loss(b_ref, tgt2).sum())
    print(f"C3 IoU-loss sums with the targets read from the fp32 oracle instead: product {s2_prod:.4f}, fp32 oracle {s2_ref:.4f}")
    assert s_ref > s_prod and s2_prod > s2_ref
    assert 0.4 * (s_ref - s_prod) <= (s2_prod - s2_ref) <= 2.5 * (s_ref - s_prod), (s_prod, s_ref, s2_prod, s2_ref)
