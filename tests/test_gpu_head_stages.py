"""GPU parity of each proposal-stage kernel against the CPU oracle ON THE SAME INPUTS.

The oracle (fp32, pinned to the reference by tests/test_oracle_golden.py) is run on the golden clip; its
intermediates are fed -- rounded to fp16 -- to the product's HIP stages and the outputs are compared with the
oracle's next intermediates.  Integer outputs (proposal indices, Hungarian permutations, NMS keep lists, SimOTA
assignments) must be bit-exact on these inputs; float outputs within the stated fp16 tolerances.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tscd_oracle as O
from test_state_dict_compat import build_model


def close(got, ref, rel_max, rel_mean=None):
    got, ref = torch.as_tensor(got).float().cpu(), torch.as_tensor(ref).float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = (got - ref).abs()
    assert err.max().item() <= rel_max * ref.abs().max().item() + 1e-4, (err.max().item(), ref.abs().max().item())
    if rel_mean is not None:
        assert err.mean().item() <= rel_mean * ref.abs().mean().item() + 1e-5, (err.mean().item(), ref.abs().mean().item())


@pytest.fixture(scope="module")
def ctx(golden):
    cfg, sd, vec, meta = golden
    sd = {k: v.clone() for k, v in sd.items()}
    aux = {}
    imgs = torch.from_numpy(vec["evalA_imgs"]).float()
    temb = torch.from_numpy(vec["time_embedding"])
    with torch.no_grad():
        res = O.tscd_forward(sd, cfg, imgs, None, temb, 0.5, int(meta["lframe"]), int(meta["gframe"]), False, False,
                             O.MatcherState(), aux)
    m = build_model(cfg)
    m.load_state_dict(golden[1], strict=True)
    return cfg, meta, aux, res, m.cuda().eval(), temb


def test_select_proposals_bit_exact(ctx):
    from tscd_amd import ops
    cfg, meta, aux, res, m, temb = ctx
    # scores as the reference forms them: obj * max cls, here from the oracle's decode_res rounded to fp16
    dr = aux["decode_res"].half().float()
    score = (dr[..., 4].half() * dr[..., 5:].max(-1).values.half())          # fp16 product
    idx, cnt = ops.select_proposals(score.cuda().contiguous(), 0.001, cfg.minimal_limit, cfg.maximal_limit)
    cnt = cnt.cpu().tolist()
    for b in range(score.shape[0]):
        s = score[b].float()
        mask = s >= torch.tensor(0.001).half().float()
        if mask.sum() < cfg.minimal_limit:
            top = torch.topk(s, cfg.minimal_limit).indices
            mask[top] = True
        if mask.sum() > cfg.maximal_limit:
            # torch.topk leaves tie order open; the kernel's documented rule: value desc, then lower index
            order = np.lexsort((np.arange(len(s)), -s.numpy()))
            mask = torch.zeros_like(mask)
            mask[torch.from_numpy(order[:cfg.maximal_limit].copy())] = True
        ref = torch.where(mask)[0]
        assert cnt[b] == len(ref)
        assert torch.equal(idx[b, :cnt[b]].cpu().long(), ref)


def test_mca_matches_oracle(ctx):
    cfg, meta, aux, res, m, temb = ctx
    L = int(meta["lframe"])
    h = m.head
    f_cls, f_reg = aux["f_cls"].half().cuda(), aux["f_reg"].half().cuda()
    sc = aux["cls_scores"].half().cuda()
    kw = dict(preds_per_frame=aux["preds_per_frame"], lframe=L, gframe=int(meta["gframe"]), conf_sim_thresh=cfg.conf_sim_thresh)
    with torch.no_grad():
        a, none = h.agg(f_cls, f_reg, sc, None, sim_thresh=cfg.sim_thresh, need_reg=False, **kw)
        c, o = h.agg_iou(f_cls, f_reg, sc, None, sim_thresh=cfg.sim_thresh, **kw)
    torch.cuda.synchronize()
    assert none is None
    # fp16 operands / fp32 accumulation vs the fp32 oracle: max 2e-2, mean 5e-3 (relative to max / mean |ref|)
    close(a, aux["agg_cls"], 2e-2, 1e-2)
    close(c, aux["agg_iou_cls"], 2e-2, 1e-2)
    close(o, aux["agg_iou_obj"], 2e-2, 1e-2)


@pytest.mark.parametrize("W,n_loc,n_glob", [(256, (70, 33), 300), (320, (40,), 100), (128, (64, 64, 1), 0)])
def test_mca_other_widths(W, n_loc, n_glob):
    """Head widths of TSCD-L (d=64), TSCD-X (d=80 -> padded to 96) and TSCD-S (d=32); ragged tiles; no global keys."""
    from yolox.models.post_trans import MCA_tscd_g2l_reg
    g = torch.Generator().manual_seed(W)
    cfg = O.TSCDConfig()
    mod = MCA_tscd_g2l_reg(W, 4 * W, num_heads=4, reconf=True)
    sd = {"m." + k: v.detach().clone() for k, v in mod.state_dict().items()}
    N = sum(n_loc) + n_glob
    # clustered features so that the similarity masks (> .75 / > .99) select more than the diagonal
    centers = torch.randn(12, W, generator=g)
    lab = torch.randint(0, 12, (N,), generator=g)
    x_cls = (centers[lab] + 0.15 * torch.randn(N, W, generator=g)).half()
    x_reg = (centers[lab] + 0.02 * torch.randn(N, W, generator=g)).half()
    sc = torch.rand(N, generator=g).half()
    ppf = list(n_loc) + ([n_glob] if n_glob else [])
    with torch.no_grad():
        rc, ro = O.mca_tscd_g2l_reg(sd, "m", x_cls.float(), x_reg.float(), sc.float(), ppf, len(n_loc), cfg)
        mod = mod.cuda()
        c, o = mod(x_cls.cuda(), x_reg.cuda(), sc.cuda(), None, sim_thresh=0.75, preds_per_frame=ppf, lframe=len(n_loc),
                   conf_sim_thresh=0.99)
    torch.cuda.synchronize()
    close(c, rc, 2e-2, 1e-2)
    close(o, ro, 2e-2, 1e-2)


def test_mca_reference_faithful_fp16_logits_switch():
    """``fp16_logits`` (yolox.models.post_trans.FP16_LOGITS): the forward kernels round q.k^T, x 25 and x class score to fp16 one after
    the other like the reference's autocast / .half() tensors (post_trans.py:655-657) before the fp32 softmax.  On the OUTPUTS the
    effect is ~2e-4 of their scale -- below the fp16 rounding of the outputs themselves -- so the switch is pinned where it acts: the
    softmax row sums (``stats``) of both modes against both arithmetics evaluated with torch on the kernel's own normalised rows.  Each
    mode matches its arithmetic to 1e-4 and is 5-50x further from the other one; the outputs of both modes stay within the fp16
    tolerance of the oracle in the matching mode."""
    import dataclasses
    from tscd_amd import ops
    from yolox.models.post_trans import MCA_tscd_g2l_reg
    W, n_loc, n_glob = 256, 64, 300
    N, DP = n_loc + n_glob, 64
    g = torch.Generator().manual_seed(11)
    x_cls, x_reg = torch.randn(N, W, generator=g).half().cuda(), torch.randn(N, W, generator=g).half().cuda()
    sc = torch.rand(N, generator=g).half().cuda()
    wq, wkv = (torch.randn(W, W, generator=g) / 16).half().cuda(), (torch.randn(2 * W, W, generator=g) / 16).half().cuda()
    q_c, kv_c = ops.linear(x_cls[:n_loc].contiguous(), wq), ops.linear(x_cls, wkv)
    q_r, kv_r = ops.linear(x_reg[:n_loc].contiguous(), wq), ops.linear(x_reg, wkv)
    tiles = ops.mca_tiles([n_loc, n_glob], 1, x_cls.device)
    res = {}
    for mode in (False, True):
        res[mode] = ops.mca_attention(q_c, kv_c, q_r, kv_r, sc, tiles, N, n_loc, W, True, 25.0, 0.75, 0.99, want_stats=True, keep_workspace=True,
                                      fp16_logits=mode)
    torch.cuda.synchronize()
    rows = res[False][5]
    assert torch.equal(rows, res[True][5])                    # the normalised rows do not depend on the switch
    Npad = (N + 31) // 32 * 32 + 32
    h16 = rows.view(torch.float16)
    o = 0
    take = lambda r: (h16[o:o + r * 4 * DP].view(r, 4 * DP), o + r * 4 * DP)
    qn_c, o = take(n_loc); qn_r, o = take(n_loc); kn_c, o = take(Npad); kn_r, o = take(Npad)
    r16 = lambda t: t.half().float()
    scf = sc.float()
    for branch, (qn, kn) in enumerate(((qn_c, kn_c), (qn_r, kn_r))):
        for h in range(4):
            S = qn[:, h * DP:(h + 1) * DP].float() @ kn[:N, h * DP:(h + 1) * DP].float().T            # [n_loc, N], every key visible (one local frame)
            if branch == 0:
                lg32, lg16 = 25.0 * scf[None, :] * S, r16(r16(r16(S) * 25.0) * scf[None, :])
            else:
                lg32, lg16 = 25.0 * S, r16(r16(S) * 25.0)
            l32, l16 = torch.exp(lg32 - 25.0).sum(1), torch.exp(lg16 - 25.0).sum(1)
            rel = lambda a, b: ((a - b).abs() / b.abs()).mean().item()
            for mode, own, other in ((False, l32, l16), (True, l16, l32)):
                got = res[mode][4][:8 * n_loc].view(4, 2, n_loc)[h, branch]
                assert rel(got, own) < 1e-4, (branch, h, mode, rel(got, own))
                assert rel(got, other) > 5 * rel(got, own), (branch, h, mode, rel(got, own), rel(got, other))
    # module level: both modes within the fp16 tolerance of the oracle evaluated in the same arithmetic
    mod = MCA_tscd_g2l_reg(W, 4 * W, num_heads=4, reconf=True)
    sd = {"m." + k: v.detach().clone() for k, v in mod.state_dict().items()}
    ppf = [n_loc, n_glob]
    cfg32 = O.TSCDConfig()
    cfg16 = dataclasses.replace(cfg32)
    cfg16.mca_fp16_logits = True
    mod = mod.cuda()
    kw = dict(sim_thresh=0.75, preds_per_frame=ppf, lframe=1, conf_sim_thresh=0.99)
    with torch.no_grad():
        for cfg, mode in ((cfg32, False), (cfg16, True)):
            ref = O.mca_tscd_g2l_reg(sd, "m", x_cls.float().cpu(), x_reg.float().cpu(), sc.float().cpu(), ppf, 1, cfg)
            got = mod(x_cls, x_reg, sc, None, fp16_logits=mode, **kw)
            torch.cuda.synchronize()
            close(got[0], ref[0], 2e-2, 1e-2)
            close(got[1], ref[1], 2e-2, 1e-2)


def test_cafm_and_task_aligned_match_oracle(ctx):
    cfg, meta, aux, res, m, temb = ctx
    L = int(meta["lframe"])
    h = m.head
    ppf = aux["preds_per_frame"]
    nl = sum(ppf[:L])
    f_reg, f_edge = aux["f_reg"].half().cuda(), aux["f_edge"].half().cuda()
    a_cls, a_obj = aux["agg_iou_cls"].half().cuda(), aux["agg_iou_obj"].half().cuda()
    h.local_reg_matcher._clear_state()
    with torch.no_grad():
        out = h.local_reg_matcher(f_reg, a_obj, a_cls, f_edge, ppf[:L], temb[:L].cuda(), resume=False)[-1]
        hung = list(h.local_reg_matcher.last_hungarian)
        hung_rows = list(h.local_reg_matcher.last_hungarian_rows)
    torch.cuda.synchronize()
    # INT: Hungarian column assignment per local frame.  Exactness on identical inputs (the solver on the device-produced
    # fp16 cost == scipy on that cost, whole chain) is tests/test_gpu_proposal_stages.py::
    # test_device_hungarian_cost_and_assignment_chain.  Here the product's fp16 cost (the reference's own dtype under
    # autocast/.half(): tscd_matching.py:919-930 runs torch.mm on fp16 tensors) meets the ORACLE's fp32 cost, so the
    # criterion is optimality under the oracle's cost: every cost entry is within eps = 3e-3 of the fp32 one, hence an
    # assignment that is optimal for the fp16 matrix costs at most 2*n*eps more than the fp32 optimum.
    from scipy.optimize import linear_sum_assignment
    assert len(hung) == len(aux["hungarian"])
    regs, clss = aux["agg_iou_obj"].half().float(), aux["agg_iou_cls"].half().float()
    nz = lambda t: t / (t.norm(dim=1)[:, None] + 1e-6)
    spans, s0 = [], 0
    for n in ppf[:L]:
        spans.append((s0, s0 + n)); s0 += n
    order, identical = None, 0
    for fi, (a, b) in enumerate(zip(hung, aux["hungarian"])):
        a, b = np.asarray(a), np.asarray(b)
        assert len(a) == len(b)
        assert len(set(a.tolist())) == len(a)           # a valid assignment: no column used twice
        (c0, c1) = spans[fi]
        (p0, p1) = spans[max(fi - 1, 0)]
        C = (1 - ((nz(clss[p0:p1]) @ nz(clss[c0:c1]).t() + nz(regs[p0:p1]) @ nz(regs[c0:c1]).t()) / 2)).double().numpy()
        if fi > 0:
            C = C[order]
        r_opt, c_opt = linear_sum_assignment(C)
        ours = C[np.asarray(hung_rows[fi]), a].sum()
        assert ours <= C[r_opt, c_opt].sum() + 2 * len(a) * 3e-3, (fi, ours, C[r_opt, c_opt].sum())
        identical += int(np.array_equal(a, b))
        rest = np.setdiff1d(np.arange(C.shape[1]), a)
        order = np.arange(C.shape[1]) if fi == 0 else (np.concatenate((a, rest)) if C.shape[0] < C.shape[1] else a)
    print("frames whose assignment equals the fp32 oracle's:", identical, "of", len(hung))
    close(out, aux["cafm"], 6e-2, 1.5e-2)     # LayerNorm'd output, rms 1: reference's own fp16 noise is 5.4e-2 max
    with torch.no_grad():
        matched4 = torch.nn.functional.linear(aux["cafm"], m.head.fc_reg_matcher.weight.cpu(), m.head.fc_reg_matcher.bias.cpu())
        ta = h.task_aligned(matched4.half().cuda(), a_obj, ppf[:L])
        ref = O.task_aligned({k: v.cpu() for k, v in m.state_dict().items()}, "head.task_aligned", matched4, aux["agg_iou_obj"],
                             ppf[:L], cfg)
    torch.cuda.synchronize()
    close(ta, ref, 6e-2, 1.5e-2)


def test_nms_bit_exact_keep_lists(ctx):
    """Final NMS on the oracle's refined predictions: kept candidate indices (INT) must equal the oracle's."""
    from yolox.models.post_process import nms_frames
    cfg, meta, aux, res, m, temb = ctx
    L = int(meta["lframe"])
    ppf = aux["preds_per_frame"]
    nl = sum(ppf[:L])
    det = torch.cat([aux["pred_result"][i] for i in range(L)], 0)
    det[:, :4] = det[:, :4].clamp(-500, 500)     # random-init base boxes reach 1e5 px (inf in fp16); keep them finite
    det = det.half()
    # random-init deltas are huge (exp(dw) overflows fp16 -> inf boxes, which makes the coordinate-offset trick of
    # batched_nms produce NaN IoUs); scale them to a realistic range so the comparison exercises NMS, not inf handling
    cls, obj, reg = aux["cls_preds"].half(), aux["matcher_obj_preds"].half(), (0.05 * aux["matcher_reg_preds"]).half()
    # oracle on the same fp16-rounded inputs (fp16 values, fp32 arithmetic)
    # decode on half tensors: every torch op rounds to fp16, exactly what the reference does in .half() eval
    reg_xyxy = O.decode_reg_preds5(reg, det[:, :4]).float()
    s, cls_pf, obj_pf, reg_pf = 0, [], [], []
    for i in range(L):
        cls_pf.append(cls[s:s + ppf[i]].float()); obj_pf.append(obj[s:s + ppf[i]].float().squeeze(-1)); reg_pf.append(reg_xyxy[s:s + ppf[i]])
        s += ppf[i]
    pr = [det[sum(ppf[:i]):sum(ppf[:i + 1])].float() for i in range(L)]
    # mimic the fp16 rounding of sigmoid / products the kernel (and the reference in .half()) applies
    def q(t):
        return t.half().float()
    ref_res, ref_ori, ref_keep = [], [], []
    for i in range(L):
        sc = q(torch.sigmoid(cls_pf[i])); ob = q(torch.sigmoid(obj_pf[i]))
        loc = torch.where(sc >= q(torch.tensor(0.001)))
        prod = q(ob[loc[0]] * sc[loc[0], loc[1]])
        keepm = prod >= q(torch.tensor(0.001))
        boxes, scores, cids = reg_pf[i][loc[0]][keepm], prod[keepm], loc[1][keepm]
        from oracle.nms_ref import batched_nms
        ref_keep.append(batched_nms(boxes, scores, cids.float(), 0.5))
        ref_res.append((boxes, scores, cids))
    out, ori, keep = nms_frames(det.cuda(), ppf[:L], cfg.num_classes, cls.cuda(), obj.cuda(), reg.cuda(), 0.001, 0.5, return_idx=True)
    torch.cuda.synchronize()
    for i in range(L):
        k = keep[i].cpu().long()
        # INT: kept candidate indices, in order, bit-exact
        assert torch.equal(k, ref_keep[i]), (i, len(k), len(ref_keep[i]), len(set(k.tolist()) & set(ref_keep[i].tolist())))
        assert out[i].shape[0] == len(k)
        # refined boxes: device expf vs libm exp may differ in the last fp16 bit
        assert torch.allclose(out[i][:, :4].float().cpu(), ref_res[i][0][ref_keep[i]], rtol=2e-3, atol=0.07)
        # rows come out in decreasing score order
        sc = (out[i][:, 4].float() * out[i][:, 5].float()).half().float().cpu()
        assert torch.all(sc[:-1] >= sc[1:])


def test_simota_bit_exact(golden):
    """SimOTA on the oracle's decoded outputs (fp16-rounded), fp32 evaluation mode (emu16=0): fg anchors, matched gt
    and matched IoU must reproduce the oracle's assignment exactly."""
    from tscd_amd import ops
    cfg, sd, vec, meta = golden
    sd = {k: v.clone() for k, v in sd.items()}
    aux = {}
    imgs = torch.from_numpy(vec["evalA_imgs"]).float()
    labels = torch.from_numpy(vec["train_labels"])
    temb = torch.from_numpy(vec["time_embedding"])
    with torch.no_grad():
        O.tscd_forward(sd, cfg, imgs, labels, temb, 0.5, int(meta["lframe"]), int(meta["gframe"]), False, True, O.MatcherState(), aux)
    outputs = aux["outputs"].half()
    S = int(meta["S"])
    hw = [(S // s, S // s) for s in cfg.strides]
    xs, ys, ss = O.make_grids(hw, cfg.strides)
    ref = O.simota_assign(outputs.float(), labels, xs, ys, ss, cfg)
    geo = ops.LevelGeometry(hw, cfg.strides)
    fg, mg, pi, nfg, ngt = ops.simota_assign(outputs.cuda().contiguous(), labels.cuda(), geo, cfg.num_classes, 64, emu16=False)
    torch.cuda.synchronize()
    for b in range(outputs.shape[0]):
        ours = torch.nonzero(fg[b]).flatten().cpu()
        assert torch.equal(ours, ref[0][b]), b
        assert int(nfg[b]) == len(ref[0][b])
        np.testing.assert_allclose(pi[b][fg[b].bool()].cpu().numpy(), ref[1][b].sum(-1).numpy(), rtol=1e-5, atol=1e-6)
    assert ngt.cpu().tolist() == [3] * outputs.shape[0]


def test_wavelet_block_vs_reference_golden(golden):
    """WaveletsHFBlock (a6) on the reference's own reg_convs2 maps -> the reference's edge maps (golden, fp32).
    fp16 tolerance: 1.5e-2 of max|ref|, 5e-3 of mean|ref|."""
    cfg, sd, vec, meta = golden
    m = build_model(cfg)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    for k in range(3):
        x = torch.from_numpy(vec[f"evalA_reg2_{k}"]).cuda().half().permute(0, 2, 3, 1).contiguous()
        with torch.no_grad():
            e = m.head.edge_enhance_reg[k][0](x)
        torch.cuda.synchronize()
        close(e.permute(0, 3, 1, 2), vec[f"evalA_edge{k}"], 1.5e-2, 1e-2)


def test_trainable_tower_backward_matches_oracle_autograd():
    """conv -> train-mode BN -> SiLU (x2) + wavelet block: HIP forward and HIP backward (conv wgrad/dgrad, BN+SiLU, both Haar steps)
    vs the oracle's autograd."""
    from yolox.models.network_blocks import BaseConv
    from yolox.models.surrounding_extraction import WaveletsHFBlock
    g = torch.Generator().manual_seed(0)
    torch.manual_seed(0)        # the conv weights come from torch's global generator: do not depend on what ran before this test
    C, B, H, W = 32, 3, 12, 12
    tower = torch.nn.Sequential(BaseConv(C, C, 3, 1), BaseConv(C, C, 3, 1))
    wav = WaveletsHFBlock(C)
    for mod in tower.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
            mod.weight.data = 0.5 + torch.rand(C, generator=g)
            mod.bias.data = 0.2 * torch.randn(C, generator=g)
    x = torch.randn(B, C, H, W, generator=g).half()
    gy = torch.randn(B, C, H, W, generator=g).half()
    # oracle side (fp32 autograd)
    sd = {"t." + k: v.detach().clone().float().requires_grad_(v.is_floating_point() and "running" not in k)
          for k, v in tower.state_dict().items()}
    sd.update({"w." + k: v.detach().clone().float().requires_grad_("filter" in k) for k, v in wav.state_dict().items()})
    y = O.tower(sd, "t", x.float(), True)
    e = O.wavelets_hf_block(sd, "w", y)
    (e * gy.float()).sum().backward()
    tower, wav = tower.cuda().train(), wav.cuda().train()
    ye = wav(tower[1](tower[0](x.cuda().permute(0, 2, 3, 1).contiguous())))
    close(ye.permute(0, 3, 1, 2), e.detach(), 2e-2, 1e-2)
    (ye.float() * gy.cuda().permute(0, 2, 3, 1).float()).sum().backward()
    torch.cuda.synchronize()
    pairs = [("t.0.conv.weight", tower[0].conv.weight), ("t.1.conv.weight", tower[1].conv.weight), ("t.0.bn.weight", tower[0].bn.weight),
             ("t.1.bn.bias", tower[1].bn.bias), ("w.filter1.0.weight", wav.filter1[0].weight), ("w.filter2.0.weight", wav.filter2[0].weight),
             ("w.filter2.0.bias", wav.filter2[0].bias), ("w.filter1.0.bias", wav.filter1[0].bias)]
    for name, p in pairs:
        ref, got = sd[name].grad, p.grad.float().cpu()
        cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item()
        assert cos > 0.995, (name, cos)
        assert abs(got.norm().item() / ref.norm().item() - 1) < 0.03, (name, got.norm().item(), ref.norm().item())


@pytest.mark.parametrize("W,ppf,L", [(64, [40, 25, 90], 2), (256, [70, 33, 150], 2), (128, [64, 64, 1], 3), (128, [200, 180, 150, 170], 3)])
def test_mca_backward_matches_oracle_autograd(W, ppf, L):
    """HIP backward kernels (mca_bwd.hip + GEMM grads) vs the oracle's fp32 autograd."""
    from yolox.models.post_trans import MCA_tscd_g2l_reg
    g = torch.Generator().manual_seed(3)
    cfg = O.TSCDConfig()
    mod = MCA_tscd_g2l_reg(W, 4 * W, num_heads=4, reconf=True)
    N = sum(ppf)
    centers = torch.randn(8, W, generator=g)
    lab = torch.randint(0, 8, (N,), generator=g)
    xc = (centers[lab] + 0.2 * torch.randn(N, W, generator=g)).half()
    xr = (centers[lab] + 0.05 * torch.randn(N, W, generator=g)).half()
    sc = torch.rand(N, generator=g).half()
    gc, go = torch.randn(sum(ppf[:L]), 4 * W, generator=g), torch.randn(sum(ppf[:L]), 4 * W, generator=g)
    sd = {"m." + k: v.detach().clone().requires_grad_(True) for k, v in mod.state_dict().items()}
    xc32, xr32, sc32 = xc.float().requires_grad_(True), xr.float().requires_grad_(True), sc.float().requires_grad_(True)
    rc, ro = O.mca_tscd_g2l_reg(sd, "m", xc32, xr32, sc32, ppf, L, cfg)
    ((rc * gc).sum() + (ro * go).sum()).backward()
    mod = mod.cuda()
    xcg, xrg, scg = xc.cuda().requires_grad_(True), xr.cuda().requires_grad_(True), sc.cuda().requires_grad_(True)
    c, o = mod(xcg, xrg, scg, None, sim_thresh=0.75, preds_per_frame=ppf, lframe=L, conf_sim_thresh=0.99)
    ((c.float() * gc.cuda()).sum() + (o.float() * go.cuda()).sum()).backward()
    torch.cuda.synchronize()
    checks = [(xcg.grad, xc32.grad, "x_cls"), (xrg.grad, xr32.grad, "x_reg"), (scg.grad, sc32.grad, "score")]
    checks += [(p.grad, sd["m." + k].grad, k) for k, p in mod.named_parameters()]
    for got, ref, name in checks:
        cos = torch.nn.functional.cosine_similarity(got.float().cpu().flatten(), ref.flatten(), dim=0).item()
        ratio = (got.float().cpu().norm() / ref.norm()).item()
        assert cos > 0.999 and abs(ratio - 1) < 0.01, (name, cos, ratio)     # measured: 1.0000 / 1.000 on every tensor


@pytest.mark.parametrize("W,ppf,L", [(256, [70, 33, 150, 97], 2), (128, [200, 180, 150, 170, 31], 3), (64, [40, 25, 90], 1)])
def test_mca_backward_is_deterministic(W, ppf, L):
    """No pass of the MCA backward uses atomics (partial sums per split, added in a fixed order): two runs are bit-identical."""
    from yolox.models.post_trans import MCA_tscd_g2l_reg
    g = torch.Generator().manual_seed(11)
    mod = MCA_tscd_g2l_reg(W, 4 * W, num_heads=4, reconf=True).cuda()
    N = sum(ppf)
    centers = torch.randn(8, W, generator=g)
    lab = torch.randint(0, 8, (N,), generator=g)
    xc = (centers[lab] + 0.2 * torch.randn(N, W, generator=g)).half().cuda()
    xr = (centers[lab] + 0.05 * torch.randn(N, W, generator=g)).half().cuda()
    sc = torch.rand(N, generator=g).half().cuda()
    gc, go = torch.randn(sum(ppf[:L]), 4 * W, generator=g).cuda(), torch.randn(sum(ppf[:L]), 4 * W, generator=g).cuda()
    res = []
    for _ in range(2):
        mod.zero_grad(set_to_none=True)
        xcg, xrg, scg = xc.clone().requires_grad_(True), xr.clone().requires_grad_(True), sc.clone().requires_grad_(True)
        c, o = mod(xcg, xrg, scg, None, sim_thresh=0.75, preds_per_frame=ppf, lframe=L, conf_sim_thresh=0.99)
        ((c.float() * gc).sum() + (o.float() * go).sum()).backward()
        torch.cuda.synchronize()
        res.append([xcg.grad.clone(), xrg.grad.clone(), scg.grad.clone()])
    for a, b in zip(*res):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_mca_backward_reads_the_forward_workspace_or_rederives_it_bit_for_bit():
    """ops.mca_backward(fwd_ws=<the forward's workspace>) skips the prep launch and reads the normalised q / k / v rows the forward
    left there; without it they are re-derived from the same inputs: identical gradients either way."""
    from tscd_amd import ops
    d = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    W, ppf, L = 256, [70, 33, 150, 97], 2
    N, N_L = sum(ppf), sum(ppf[:L])
    tiles = ops.mca_tiles(ppf, L, d)
    mk = lambda n, c: torch.randn(n, c, generator=g).half().to(d).contiguous()
    q_c, q_r, kv_c, kv_r = mk(N_L, W), mk(N_L, W), mk(N, 2 * W), mk(N, 2 * W)
    score = torch.rand(N, generator=g).half().to(d)
    for need_reg in (True, False):
        xcat_c, cat3_c, xcat_r, cat3_r, stats, ws = ops.mca_attention(q_c, kv_c, q_r, kv_r, score, tiles, N, N_L, W, need_reg, 25.0, 0.75, 0.99,
                                                                      want_stats=True, keep_workspace=True)
        a = (mk(N_L, 2 * W), mk(N_L, 2 * W) if need_reg else None, mk(N_L, 3 * W), mk(N_L, 3 * W) if need_reg else None)
        args = (q_c, kv_c, q_r, kv_r, score, tiles, N, N_L, W, need_reg, 25.0, 0.75, 0.99, stats, a[0], a[1], a[2], a[3], cat3_c, cat3_r)
        r0 = ops.mca_backward(*args)
        r1 = ops.mca_backward(*args, fwd_ws=ws)
        torch.cuda.synchronize()
        for x, y in zip(r0, r1):
            assert torch.isfinite(x.float()).all() and torch.equal(x, y)


@pytest.mark.parametrize("W,ppf,L", [(256, [70, 33, 150, 97], 2), (128, [200, 180, 150, 170, 31], 3), (320, [40, 25, 90], 1)])
def test_mca_backward_without_class_branch_gradient_equals_zero_tensors_bit_for_bit(W, ppf, L):
    """agg_iou's class output only feeds the assignment cost, detached: no gradient reaches the class branch's attention
    outputs.  ops.mca_backward(dxcat_c=None, dcat3_c=None) runs the passes that leave the zeros' arithmetic out (CLS_G = false in
    mca_bwd.hip): every result equals the one zero tensors give, bit for bit."""
    from tscd_amd import ops
    d = torch.device("cuda")
    g = torch.Generator().manual_seed(17)
    N, N_L = sum(ppf), sum(ppf[:L])
    tiles = ops.mca_tiles(ppf, L, d)
    centers = torch.randn(8, W, generator=g)
    lab = torch.randint(0, 8, (N,), generator=g)
    rows = lambda n, c, s: (centers[lab[:n]].repeat(1, c // W) + s * torch.randn(n, c, generator=g)).half().to(d).contiguous()
    q_c, q_r, kv_c, kv_r = rows(N_L, W, 0.2), rows(N_L, W, 0.05), rows(N, 2 * W, 0.2), rows(N, 2 * W, 0.05)
    score = torch.rand(N, generator=g).half().to(d)
    mk = lambda n, c: torch.randn(n, c, generator=g).half().to(d).contiguous()
    xcat_c, cat3_c, xcat_r, cat3_r, stats, ws = ops.mca_attention(q_c, kv_c, q_r, kv_r, score, tiles, N, N_L, W, True, 25.0, 0.75, 0.99,
                                                                  want_stats=True, keep_workspace=True)
    dx_r, dc3_r = mk(N_L, 2 * W), mk(N_L, 3 * W)
    z = lambda c: torch.zeros(N_L, c, dtype=torch.float16, device=d)
    head = (q_c, kv_c, q_r, kv_r, score, tiles, N, N_L, W, True, 25.0, 0.75, 0.99, stats)
    r0 = ops.mca_backward(*head, z(2 * W), dx_r, z(3 * W), dc3_r, cat3_c, cat3_r, fwd_ws=ws)
    r1 = ops.mca_backward(*head, None, dx_r, None, dc3_r, cat3_c, cat3_r, fwd_ws=ws)
    r2 = ops.mca_backward(*head, None, dx_r, None, dc3_r, cat3_c, cat3_r)
    torch.cuda.synchronize()
    for name, x, y, y2 in zip(("dq_c", "dkv_c", "dq_r", "dkv_r", "dscore"), r0, r1, r2):
        assert torch.isfinite(x.float()).all() and x.float().abs().max() > 0, name
        assert torch.equal(x, y) and torch.equal(x, y2), (name, (x.float() - y.float()).abs().max().item())
    assert torch.count_nonzero(r1[1][:, W:]) == 0        # d(value rows) of the class branch: exactly zero
    with pytest.raises(ValueError):
        ops.mca_backward(*head, None, dx_r, z(3 * W), dc3_r, cat3_c, cat3_r)


def test_mca_module_with_only_the_objectness_output_used_matches_oracle_autograd():
    """The module as agg_iou is used in the train step: the class output feeds nothing with a gradient.  Gradients of the inputs and of
    the parameters that get one vs the oracle's fp32 autograd; the class head's linears get none."""
    from yolox.models.post_trans import MCA_tscd_g2l_reg
    W, ppf, L = 128, [200, 180, 150, 170], 3
    g = torch.Generator().manual_seed(4)
    cfg = O.TSCDConfig()
    mod = MCA_tscd_g2l_reg(W, 4 * W, num_heads=4, reconf=True)
    N = sum(ppf)
    centers = torch.randn(8, W, generator=g)
    lab = torch.randint(0, 8, (N,), generator=g)
    xc = (centers[lab] + 0.2 * torch.randn(N, W, generator=g)).half()
    xr = (centers[lab] + 0.05 * torch.randn(N, W, generator=g)).half()
    sc = torch.rand(N, generator=g).half()
    go = torch.randn(sum(ppf[:L]), 4 * W, generator=g)
    sd = {"m." + k: v.detach().clone().requires_grad_(True) for k, v in mod.state_dict().items()}
    xc32, xr32, sc32 = xc.float().requires_grad_(True), xr.float().requires_grad_(True), sc.float().requires_grad_(True)
    rc, ro = O.mca_tscd_g2l_reg(sd, "m", xc32, xr32, sc32, ppf, L, cfg)
    (ro * go).sum().backward()
    mod = mod.cuda()
    xcg, xrg, scg = xc.cuda().requires_grad_(True), xr.cuda().requires_grad_(True), sc.cuda().requires_grad_(True)
    c, o = mod(xcg, xrg, scg, None, sim_thresh=0.75, preds_per_frame=ppf, lframe=L, conf_sim_thresh=0.99)
    (o.float() * go.cuda()).sum().backward()
    torch.cuda.synchronize()
    checks = [(xcg.grad, xc32.grad, "x_cls"), (xrg.grad, xr32.grad, "x_reg"), (scg.grad, sc32.grad, "score")]
    for k, p in mod.named_parameters():
        ref = sd["m." + k].grad
        if ref is None or float(ref.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            checks.append((p.grad, ref, k))
    assert len(checks) >= 9
    for got, ref, name in checks:
        cos = torch.nn.functional.cosine_similarity(got.float().cpu().flatten(), ref.flatten(), dim=0).item()
        ratio = (got.float().cpu().norm() / ref.norm()).item()
        assert cos > 0.999 and abs(ratio - 1) < 0.01, (name, cos, ratio)


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


@pytest.mark.parametrize("C,heads,nq,nk", [(256, 8, 150, 150), (1024, 8, 77, 77), (64, 8, 40, 130), (320, 8, 33, 90),
                                           (128, 8, 45, 70), (384, 8, 40, 33), (1280, 8, 37, 50)])
def test_cafm_attention_pieces_backward_match_oracle_autograd(C, heads, nq, nk):
    """cos_attn / se_fuse / add_layernorm / linear backward kernels vs. autograd of the oracle's fp32 expressions
    (oracle cos_mh_attention, se_module, referring_cross_attention: the reference's tscd_matching.py:140-181, 264-283, 535-636)."""
    from yolox.models import tscd_matching as M
    import torch.nn.functional as F
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C + nq)
    rnd = lambda *s: torch.randn(*s, generator=g)
    # --- cosine attention, two segments over different key ranges
    q, k, v, go = rnd(nq, C), rnd(nk, C), rnd(nk, C), rnd(nq, C)
    h1, k1 = nq // 3, nk // 2
    segs = [(0, h1, 0, k1), (h1, nq - h1, k1, nk)]
    qd, kd, vd = (t.half().to(dev).requires_grad_() for t in (q, k, v))
    out = M.cos_attention(qd, kd, vd, heads, segs)
    out.backward(go.half().to(dev))
    qr, kr, vr = (t.half().float().requires_grad_() for t in (q, k, v))
    outs = []
    eye = {"a.%s_reg.weight" % n: torch.eye(C) for n in "qkv"}        # the projections are tested with `linear` below
    for (r0, n, a, b) in segs:
        outs.append(O.cos_mh_attention(eye, "a", qr[r0:r0 + n], kr[a:b], vr[a:b], heads))
    ref = torch.cat(outs, 0)
    ref.backward(go.half().float())
    close(out, ref, 2e-2)
    for name, got, want in (("dq", qd.grad, qr.grad), ("dk", kd.grad, kr.grad), ("dv", vd.grad, vr.grad)):
        c = _cos(got.cpu(), want)
        assert c > 0.9995, (name, c)
        assert abs(float(got.float().norm().cpu() / want.norm()) - 1) < 2e-2, name
    # --- SE gate
    se = M.SEModule(2).to(dev)
    reg, edge, pos = rnd(nq, C), rnd(nq, C), rnd(1, C)
    rd, ed = (t.half().to(dev).requires_grad_() for t in (reg, edge))
    pd = pos.to(dev).requires_grad_()
    o = se.fuse(rd, ed, pd)
    o.backward(go.half().to(dev))
    rr, er, pr = reg.half().float().requires_grad_(), edge.half().float().requires_grad_(), pos.clone().requires_grad_()
    w0, w2 = se.fc[0].weight.detach().cpu().clone().requires_grad_(), se.fc[2].weight.detach().cpu().clone().requires_grad_()
    oref = O.se_module({"se.fc.0.weight": w0, "se.fc.2.weight": w2}, "se", rr, er) + pr
    oref.backward(go.half().float())
    close(o, oref, 2e-2)
    for name, got, want in (("dreg", rd.grad, rr.grad), ("dedge", ed.grad, er.grad), ("dpos", pd.grad, pr.grad),
                            ("dw0", se.fc[0].weight.grad, w0.grad), ("dw2", se.fc[2].weight.grad, w2.grad)):
        c = _cos(got.cpu(), want)
        assert c > 0.999, (name, c)
    # --- residual + LayerNorm
    norm = torch.nn.LayerNorm(C).to(dev)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.2 * rnd(C)); norm.bias.copy_(0.1 * rnd(C))
    for with_b in (True, False):
        norm.zero_grad()
        ad, bd = reg.half().to(dev).requires_grad_(), edge.half().to(dev).requires_grad_()
        y = M.add_layernorm(ad, bd if with_b else None, norm)
        y.backward(go.half().to(dev))
        ar, br = reg.half().float().requires_grad_(), edge.half().float().requires_grad_()
        gw, gb = norm.weight.detach().cpu().clone().requires_grad_(), norm.bias.detach().cpu().clone().requires_grad_()
        x = ar + br if with_b else ar
        yr = F.layer_norm(x, (C,), gw, gb, norm.eps)
        yr.backward(go.half().float())
        close(y, yr, 2e-2)
        pairs = [("da", ad.grad, ar.grad), ("dgamma", norm.weight.grad, gw.grad), ("dbeta", norm.bias.grad, gb.grad)]
        if with_b:
            pairs.append(("db", bd.grad, br.grad))
        for name, got, want in pairs:
            c = _cos(got.cpu(), want)
            assert c > 0.9995, (name, with_b, c)
    # --- linear
    lin = torch.nn.Linear(C, 2 * C + 8).to(dev)
    xd = reg.half().to(dev).requires_grad_()
    gl = rnd(nq, 2 * C + 8)
    yl = M.linear(xd, lin)
    yl.backward(gl.half().to(dev))
    xr = reg.half().float().requires_grad_()
    wl, bl = lin.weight.detach().cpu().clone().requires_grad_(), lin.bias.detach().cpu().clone().requires_grad_()
    F.linear(xr, wl, bl).backward(gl.half().float())
    for name, got, want in (("dx", xd.grad, xr.grad), ("dw", lin.weight.grad, wl.grad), ("db", lin.bias.grad, bl.grad)):
        c = _cos(got.cpu(), want)
        assert c > 0.9995, (name, c)


@pytest.mark.parametrize("W,counts", [(256, [[60, 0, 95, 40, 40], [33, 70, 70, 0, 51]]), (64, [[25, 40, 18], [18, 18, 31]])])
def test_cafm_ragged_counts_and_carried_state_match_oracle(W, counts):
    """The matcher on ragged clips: growing / shrinking / equal proposal counts, an empty frame, and a second clip that
    continues from the first one's state (resume=True: first frame matched against carried-over rows).  Assignments (the
    host chain solver's output) must equal the oracle's scipy calls; attention outputs within fp16 tolerance."""
    from yolox.models.tscd_matching import AwarePositionRegMatcher
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(W + len(counts[0]))
    mod = AwarePositionRegMatcher(hidden_channel=W, feedforward_channel=W, num_head=8, decoder_layer_num=1)
    with torch.no_grad():
        for prm in mod.parameters():
            if prm.dim() == 1:
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.1 + (1.0 if "norm.weight" in "" else 0.0))
        for n, prm in mod.named_parameters():
            if n.endswith("norm.weight"):
                prm.fill_(1.0)
    sd = {"m." + k: v.detach().clone().float() for k, v in mod.state_dict().items()}
    mod = mod.to(dev).half().eval()
    cfg = O.TSCDConfig()
    state = O.MatcherState()
    mod._clear_state()
    # cost features: every frame shows the first n objects of one pool (shuffled, slightly perturbed), so any two frames are
    # nested sets and each assignment is unambiguous (matched pairs cost ~0, everything else ~1)
    pool_reg = torch.nn.functional.normalize(torch.randn(128, 4 * W, generator=g), dim=1)
    pool_cls = torch.nn.functional.normalize(torch.randn(128, 4 * W, generator=g), dim=1)

    def cost_feats(n):
        order = torch.randperm(n, generator=g)
        return ((pool_reg[order] + 0.01 * torch.randn(n, 4 * W, generator=g)).half(),
                (pool_cls[order] + 0.01 * torch.randn(n, 4 * W, generator=g)).half())

    for ci, ppf in enumerate(counts):
        n_all = sum(ppf)
        feats = torch.randn(n_all, W, generator=g).half()
        edge = torch.randn(n_all, W, generator=g).half()
        pairs = [cost_feats(n) for n in ppf]
        reg = torch.cat([a for a, _ in pairs], 0)
        cls = torch.cat([b for _, b in pairs], 0)
        temb = torch.randn(len(ppf), 256, generator=g).half()
        resume = ci > 0
        with torch.no_grad():
            out = mod(feats.to(dev), reg.to(dev), cls.to(dev), edge.to(dev), ppf, temb.to(dev), resume=resume)
        torch.cuda.synchronize()
        aux = {}
        ref = O.aware_position_reg_matcher(sd, "m", feats.float(), reg.float(), cls.float(), edge.float(), ppf, temb.float(),
                                           resume, state, cfg, aux)
        hung = [np.asarray(h) for h in mod.last_hungarian]
        assert len(hung) == len(aux["hungarian"])
        for fi, (a, b) in enumerate(zip(hung, aux["hungarian"])):
            assert np.array_equal(a, np.asarray(b)), (ci, fi)
        close(out[-1], ref, 6e-2, 1.5e-2)


@pytest.mark.parametrize("W,n,L", [(256, 96, 4), (64, 40, 3), (256, 500, 2)])
def test_cafm_fused_chain_equals_the_per_op_path_and_the_oracle_autograd(W, n, L):
    """_CafmChain (the clamped training regime: every local frame with the same number of proposals, the whole frame-to-frame chain as
    ONE autograd node with batched k / v projections) against the per-op path on the same inputs: forward bit-identical (same kernels
    on the same rows), every gradient -- features, edge features, and all 11 parameters incl. the time-embedding linear -- equal to
    the per-op autograd within fp16 accumulation noise, and both within tolerance of the oracle's fp32 autograd."""
    import yolox.models.tscd_matching as M
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(W + n + L)
    mod = M.AwarePositionRegMatcher(hidden_channel=W, feedforward_channel=W, num_head=8, decoder_layer_num=1)
    with torch.no_grad():
        for name, prm in mod.named_parameters():
            if prm.dim() == 1:
                prm.copy_(torch.randn(prm.shape, generator=g) * 0.1 + (1.0 if name.endswith("norm.weight") else 0.0))
    sd = {"m." + k: v.detach().clone().float() for k, v in mod.state_dict().items()}
    mod = mod.to(dev).train()
    ppf = [n] * L
    nl = n * L
    pool_reg = torch.nn.functional.normalize(torch.randn(n, 4 * W, generator=g), dim=1)
    pool_cls = torch.nn.functional.normalize(torch.randn(n, 4 * W, generator=g), dim=1)
    regs, clss = [], []
    for _ in range(L):
        order = torch.randperm(n, generator=g)
        regs.append((pool_reg[order] + 0.01 * torch.randn(n, 4 * W, generator=g)).half())
        clss.append((pool_cls[order] + 0.01 * torch.randn(n, 4 * W, generator=g)).half())
    reg, cls = torch.cat(regs).to(dev), torch.cat(clss).to(dev)
    feats0 = torch.randn(nl, W, generator=g).half()
    edge0 = torch.randn(nl, W, generator=g).half()
    temb = torch.randn(L, 256, generator=g).half().to(dev)
    wgt = torch.randn(nl, W, generator=g).half().to(dev)
    used = ("absolute_position_embedding", "transformer_aware_cross_attention_layers.0.CA", "multihead_attn.q_reg", "multihead_attn.k_reg",
            "multihead_attn.v_reg", "transformer_aware_cross_attention_layers.0.norm", "decoder_norm")

    def run(fused):
        M.FUSED_CHAIN = fused
        mod.zero_grad(set_to_none=True)
        f = feats0.to(dev).requires_grad_()
        e = edge0.to(dev).requires_grad_()
        out = mod(f, reg, cls, e, ppf, temb, resume=False)[-1]
        (out.float() * wgt.float()).sum().backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().float().cpu().clone() for k, p in mod.named_parameters() if p.grad is not None}
        return out.detach().float().cpu(), f.grad.float().cpu(), e.grad.float().cpu(), grads, [np.asarray(h) for h in mod.last_hungarian]

    try:
        o1, df1, de1, g1, h1 = run(True)
        o0, df0, de0, g0, h0 = run(False)
    finally:
        M.FUSED_CHAIN = True
    assert all(np.array_equal(a, b) for a, b in zip(h0, h1))
    assert torch.equal(o0, o1)
    cos = lambda a, b: torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()
    assert cos(df0, df1) > 0.9995 and cos(de0, de1) > 0.9995, (cos(df0, df1), cos(de0, de1))
    assert set(g0) == set(g1) and len(g1) == 11 and all(any(u in k for u in used) for k in g1), sorted(g1)
    for k in g0:
        c = cos(g0[k], g1[k])
        r = (g1[k].norm() / g0[k].norm().clamp(min=1e-12)).item()
        assert c > 0.999 and abs(r - 1) < 2e-2, (k, c, r)
    # the oracle's fp32 autograd on the same inputs
    sd_o = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    fo, eo = feats0.float().requires_grad_(), edge0.float().requires_grad_()
    ref = O.aware_position_reg_matcher(sd_o, "m", fo, reg.float().cpu(), cls.float().cpu(), eo, ppf, temb.float().cpu(), False, O.MatcherState(),
                                       O.TSCDConfig(), {})
    (ref * wgt.float().cpu()).sum().backward()
    close(o1, ref.detach(), 6e-2, 1.5e-2)
    assert cos(df1, fo.grad) > 0.995 and cos(de1, eo.grad) > 0.995, (cos(df1, fo.grad), cos(de1, eo.grad))
    for k, v in g1.items():
        gr = sd_o["m." + k].grad
        assert gr is not None and cos(v, gr) > 0.99, (k, cos(v, gr))


@pytest.mark.parametrize("C,heads,segs,nq,nk", [(1024, 8, [(0, 500, 0, 500), (500, 500, 500, 1000), (1000, 500, 1000, 1500)], 1500, 1500),
                                                (256, 8, [(0, 70, 10, 131), (70, 45, 140, 173)], 120, 180),
                                                (320, 8, [(0, 33, 0, 90)], 33, 90)])
def test_cos_attention_key_stationary_backward_equals_the_atomic_one(C, heads, segs, nq, nk):
    """The two-launch backward of the cosine attention (query-stationary dq + delta, key-stationary dk / dv in registers) against
    the one-launch kernel that sums dk / dv over query tiles with fp32 atomics: same forward output bit for bit, dq / dk / dv equal up
    to the order of fp32 partial sums (the two-launch kernels deal a task's key / query tiles to 2-4 waves and add the waves' partial
    sums in wave order; the one-launch kernel walks them in one wave and adds dk / dv with atomics) and the fp16 rounding of the final
    store; key rows no segment attends to get zero gradients; segments that share key rows fall back to the atomic kernel."""
    from yolox.models import tscd_matching as M
    from tscd_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C + nq)
    q, k, v, go = (torch.randn(n_, C, generator=g).half().to(dev) for n_ in (nq, nk, nk, nq))

    def run(flag):
        M.KEY_STATIONARY_BWD = flag
        qd, kd, vd = (t.clone().requires_grad_() for t in (q, k, v))
        out = M.cos_attention(qd, kd, vd, heads, segs)
        out.backward(go)
        torch.cuda.synchronize()
        return out.detach(), qd.grad, kd.grad, vd.grad

    try:
        o1, dq1, dk1, dv1 = run(True)
        o0, dq0, dk0, dv0 = run(False)
    finally:
        M.KEY_STATIONARY_BWD = True
    assert ops.attn_key_tiles(segs, dev) is not None
    assert torch.equal(o0, o1)
    for a, b in ((dq0, dq1), (dk0, dk1), (dv0, dv1)):
        assert torch.isfinite(b.float()).all()
        assert torch.allclose(a.float(), b.float(), rtol=4e-3, atol=4e-3 * float(a.float().abs().max())), float((a.float() - b.float()).abs().max())
        assert _cos(a, b) > 0.99999
    covered = torch.zeros(nk, dtype=torch.bool)
    for (_, n_, k0, k1) in segs:
        covered[k0:k1] = True
    if (~covered).any():
        assert float(dk1[~covered.to(dev)].abs().max()) == 0 and float(dv1[~covered.to(dev)].abs().max()) == 0
    # two segments attending to the same keys: no key-stationary tiling
    assert ops.attn_key_tiles([(0, 10, 0, 50), (10, 10, 25, 60)], dev) is None


def test_clip_wide_cost_matrices_equal_the_per_pair_ones():
    """ops.match_costs_clip (all proposals' [reg | cls] rows normalised once, one GEMM per matrix over two row ranges, results back to
    back in the buffer the single D2H copy takes) against ops.match_cost per pair (tscd_matching.py:919-930): bit-identical fp16
    costs, for equal and ragged frame sizes, the self-match of the first frame included."""
    from tscd_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    for sizes in ([500, 500, 500, 500], [37, 120, 64], [130]):
        n, C = sum(sizes), 1024
        reg = torch.randn(n, C, generator=g).half().to(dev)
        cls = torch.randn(n, C, generator=g).half().to(dev)
        off = [0]
        for m in sizes:
            off.append(off[-1] + m)
        ranges = [(off[0], off[1], off[0], off[1])] + [(off[i - 1], off[i], off[i], off[i + 1]) for i in range(1, len(sizes))]
        flat, shapes = ops.match_costs_clip(reg, cls, ranges)
        o = 0
        for (r0, r1, c0, c1), (nr, nc) in zip(ranges, shapes):
            want = ops.match_cost(reg[r0:r1], reg[c0:c1], cls[r0:r1], cls[c0:c1])
            assert (nr, nc) == tuple(want.shape)
            assert torch.equal(flat[o:o + nr * nc].view(nr, nc), want)
            o += nr * nc
        assert o == flat.numel()


def test_cafm_deferred_join_of_the_assignment_chain_is_the_joined_result():
    """The clamped training regime parks the stream on the worker's latch (hipLaunchHostFunc) and queues CAFM and its backward before
    the assignments exist on the host (TSCD_DEFER_ASSIGNMENTS, tscd_lap_chain_f16_table).  Same clip with the chain joined on the host
    first: identical assignments, bit-identical forward, gradients equal up to the order of the fp32 atomics in the SE gate's backward;
    repeated steps reuse the pinned table and the latch (one generation per frame: the stream is released frame by frame while the
    later matrices are still being solved), and nothing times out."""
    import yolox.models.tscd_matching as M
    dev = torch.device("cuda:0")
    W, n, L = 256, 200, 4
    g = torch.Generator().manual_seed(5)
    mod = M.AwarePositionRegMatcher(hidden_channel=W, feedforward_channel=W, num_head=8, decoder_layer_num=1).to(dev).train()
    nl = n * L
    wgt = torch.randn(nl, W, generator=g).half().to(dev)
    temb = torch.randn(L, 256, generator=g).half().to(dev)

    def clip(seed):
        gg = torch.Generator().manual_seed(seed)
        pool = torch.nn.functional.normalize(torch.randn(n, 4 * W, generator=gg), dim=1)
        reg = torch.cat([(pool[torch.randperm(n, generator=gg)] + 0.01 * torch.randn(n, 4 * W, generator=gg)).half() for _ in range(L)]).to(dev)
        cls = torch.cat([(pool[torch.randperm(n, generator=gg)] + 0.01 * torch.randn(n, 4 * W, generator=gg)).half() for _ in range(L)]).to(dev)
        return reg, cls, torch.randn(nl, W, generator=gg).half(), torch.randn(nl, W, generator=gg).half()

    def run(defer, seed):
        M.DEFER_CHAIN = defer
        reg, cls, f0, e0 = clip(seed)
        mod.zero_grad(set_to_none=True)
        f, e = f0.to(dev).requires_grad_(), e0.to(dev).requires_grad_()
        out = mod(f, reg, cls, e, [n] * L, temb, resume=False)[-1]
        pending = mod._pending is not None
        (out.float() * wgt.float()).sum().backward()
        hung = [np.asarray(h) for h in mod.last_hungarian]          # joins the worker in the deferred form
        torch.cuda.synchronize()
        return out.detach().cpu(), f.grad.float().cpu(), e.grad.float().cpu(), hung, pending

    try:
        gen0 = M._latch().generation
        for seed in (1, 2, 3):
            o1, df1, de1, h1, p1 = run(True, seed)
            o0, df0, de0, h0, p0 = run(False, seed)
            assert p1 and not p0
            assert len(h1) == L and all(np.array_equal(a, b) for a, b in zip(h0, h1))
            assert torch.equal(o0, o1)
            assert torch.allclose(df0, df1, rtol=1e-2, atol=1e-3) and torch.allclose(de0, de1, rtol=1e-2, atol=1e-3)
        assert M._latch().generation == gen0 + 3 * L and M._latch().timeouts == 0      # one generation per frame and clip
    finally:
        M.DEFER_CHAIN = True


def test_cafm_deferred_join_survives_a_failing_chain(monkeypatch):
    """A chain that fails on the worker (here: a kind the clamped regime does not allow, status 5) must not leave the stream parked nor
    feed out-of-range rows to the queued gathers: the foreign call writes identity permutations and raises every generation of the
    clip, the queued forward / backward run to completion with finite values, and the failure surfaces when the worker is joined
    (reading the test hook, or the next prepare())."""
    import yolox.models.tscd_matching as M
    dev = torch.device("cuda:0")
    W, n, L = 256, 64, 3
    g = torch.Generator().manual_seed(11)
    mod = M.AwarePositionRegMatcher(hidden_channel=W, feedforward_channel=W, num_head=8, decoder_layer_num=1).to(dev).train()
    reg = torch.nn.functional.normalize(torch.randn(n * L, 4 * W, generator=g), dim=1).half().to(dev)
    cls = torch.nn.functional.normalize(torch.randn(n * L, 4 * W, generator=g), dim=1).half().to(dev)
    f = torch.randn(n * L, W, generator=g).half().to(dev).requires_grad_()
    e = torch.randn(n * L, W, generator=g).half().to(dev).requires_grad_()
    temb = torch.randn(L, 256, generator=g).half().to(dev)
    real = M.lap_chain_f16

    def broken(base_ptr, shapes, kinds, before=None, **kw):
        return real(base_ptr, shapes, [kinds[0], M.KIND_STATE] + list(kinds[2:]), before, **kw)

    monkeypatch.setattr(M, "lap_chain_f16", broken)
    t0 = M._latch().timeouts
    out = mod(f, reg, cls, e, [n] * L, temb, resume=False)[-1]
    assert mod._pending is not None
    out.float().sum().backward()
    torch.cuda.synchronize()                                  # nothing hangs: the latch was raised on the failure path
    assert torch.isfinite(out.float()).all() and torch.isfinite(f.grad.float()).all()
    assert M._latch().timeouts == t0
    with pytest.raises(ValueError, match="assignment chain failed"):
        mod.last_hungarian
    monkeypatch.setattr(M, "lap_chain_f16", real)
    out2 = mod(f, reg, cls, e, [n] * L, temb, resume=False)[-1]      # the matcher is usable again
    torch.cuda.synchronize()
    assert len(mod.last_hungarian) == L and torch.isfinite(out2.float()).all()


def test_nms_more_candidates_than_one_lds_sort_chunk():
    """The ImageNet-VID regime (no upper proposal clamp, 30 classes): a frame with 1500 proposals has up to 45000
    (proposal, class) candidates -- more than the 16384 keys one LDS sort holds -- next to a small frame.  Kept candidate
    indices must equal the oracle's batched NMS on the same fp16 values, in the same order."""
    from oracle.nms_ref import batched_nms
    from yolox.models.post_process import nms_frames
    g = torch.Generator().manual_seed(17)
    NC, ppf = 30, [1500, 300]
    n = sum(ppf)
    cxy = torch.rand(n, 2, generator=g) * 500 + 30
    wh = torch.rand(n, 2, generator=g) * 120 + 10
    det = torch.zeros(n, 7 + NC)
    det[:, 0:2], det[:, 2:4] = cxy - wh / 2, cxy + wh / 2
    det[:, 4], det[:, 5] = torch.rand(n, generator=g), torch.rand(n, generator=g)
    det[:, 6] = torch.randint(0, NC, (n,), generator=g).float()
    det = det.half()
    cls = (torch.randn(n, NC, generator=g) * 2.0 - 1.0).half()           # ~30 % of the (proposal, class) scores pass 1e-3 * ...
    obj = (torch.randn(n, generator=g) * 1.5 + 1.0).half()
    reg = (0.05 * torch.randn(n, 4, generator=g)).half()
    q = lambda t: t.half().float()
    reg_xyxy = O.decode_reg_preds5(reg, det[:, :4]).float()
    ref_keep, s0 = [], 0
    for c in ppf:
        sc = q(torch.sigmoid(cls[s0:s0 + c].float()))
        ob = q(torch.sigmoid(obj[s0:s0 + c].float()))
        loc = torch.where(sc >= q(torch.tensor(0.001)))
        prod = q(ob[loc[0]] * sc[loc[0], loc[1]])
        km = prod >= q(torch.tensor(0.001))
        boxes, scores, cids = reg_xyxy[s0:s0 + c][loc[0]][km], prod[km], loc[1][km]
        cand_idx = (loc[0] * NC + loc[1])[km]
        ref_keep.append(cand_idx[batched_nms(boxes, scores, cids.float(), 0.5)])
        s0 += c
    assert int((q(torch.sigmoid(cls[:1500].float())) >= 0.001).sum()) > 16384          # the large frame really needs the chunked sort
    out, ori, keep = nms_frames(det.cuda(), ppf, NC, cls.cuda(), obj.cuda(), reg.cuda(), 0.001, 0.5, return_idx=True)
    torch.cuda.synchronize()
    for i in range(len(ppf)):
        k = keep[i].cpu().long()
        # the kernel numbers candidates in expansion order (surviving (row, class) pairs, row-major); map to row*NC + class
        assert out[i].shape[0] == len(k) == len(ref_keep[i]), (i, len(k), len(ref_keep[i]))
        sc = (out[i][:, 4].float() * out[i][:, 5].float()).half().float().cpu()
        assert torch.all(sc[:-1] >= sc[1:])


@pytest.mark.parametrize("n,L,W", [(96, 4, 256), (40, 3, 64)])
def test_row_mapped_se_fuse_and_layernorm_equal_gather_then_plain(n, L, W):
    """The CAFM chain's row maps (ops.row_map into the clip's index table): the row-mapped entry points against materialised gathers /
    scatters around the plain ones -- forward bit-identical (one machine code for both), backward bit-identical in dx / d edge and equal in
    the fp32 accumulators up to the order of the atomic adds."""
    from tscd_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n * L + W)
    nl = n * L
    rnd = lambda *s: torch.randn(*s, generator=g)
    reg, edge, dout, b = (rnd(nl, W).half().to(dev) for _ in range(4))
    w0, w2 = (rnd(32, 2) * 0.5).to(dev).contiguous(), (rnd(2, 32) * 0.5).to(dev).contiguous()
    pos = rnd(L, W).to(dev).contiguous()
    gamma, beta = (1 + 0.1 * rnd(W)).to(dev), (0.1 * rnd(W)).to(dev)
    # index table [inv_0 | c_1 | inv_1 | ...] of random permutations
    perms = [torch.randperm(n, generator=g) for _ in range(2 * L - 1)]
    idx = torch.cat(perms).to(dev)
    c_off = [-1] + [(2 * j - 1) * n for j in range(1, L)]
    inv_off = [2 * j * n for j in range(L)]
    # --- se_fuse: all frames in one launch with a position vector per frame == per-frame plain launches
    out_seg = torch.empty(nl, W, dtype=torch.float16, device=dev)
    ops.se_fuse_ex(reg, edge, w0, w2, pos, out_seg, nl, pos_seg_rows=n, pos_stride=W)
    for j in range(L):
        want = ops.se_fuse(reg[j * n:(j + 1) * n], edge[j * n:(j + 1) * n], w0, w2, pos[j])
        assert torch.equal(out_seg[j * n:(j + 1) * n], want), j
    # --- se_fuse with permuted edge rows (frame j reads frame j-1's edge rows in assignment order)
    j = 2
    c = idx[c_off[j - 1]:c_off[j - 1] + n]
    tgt, e_prev = reg[j * n:(j + 1) * n], edge[(j - 1) * n:j * n]
    got = torch.empty(n, W, dtype=torch.float16, device=dev)
    ops.se_fuse_ex(tgt, e_prev, w0, w2, pos[j - 1], got, n, edge_map=ops.row_map(idx, n, c_off[j - 1]))
    assert torch.equal(got, ops.se_fuse(tgt, e_prev[c].contiguous(), w0, w2, pos[j - 1]))
    # backward of the same: d edge lands at the permuted rows; addends folded in
    add1, add2 = rnd(n, W).half().to(dev), rnd(n, W).half().to(dev)
    d_reg, d_edge = torch.empty(n, W, dtype=torch.float16, device=dev), torch.zeros(n, W, dtype=torch.float16, device=dev)
    acc = torch.zeros(W + 128, device=dev)
    ops.se_fuse_bwd_ex(tgt, e_prev, w0, w2, dout[:n].contiguous(), d_reg, d_edge, n, acc, acc[W:], acc[W + 64:],
                       edge_map=ops.row_map(idx, n, c_off[j - 1]), add_reg=[(add1, n), (add2, n // 2)])
    r_reg, r_edge, r_pos, r_w0, r_w2 = ops.se_fuse_bwd(tgt, e_prev[c].contiguous(), w0, w2, dout[:n].contiguous())
    want_reg = r_reg.float() + add1.float()
    want_reg[:n // 2] += add2[:n // 2].float()
    # (the plain entry rounds d reg to fp16 before the adds the caller does; the folded form rounds once: within one fp16 ulp)
    assert (d_reg.float() - want_reg).abs().max().item() <= 2e-3 * want_reg.abs().max().item() + 1e-3
    want_edge = torch.zeros_like(d_edge)
    want_edge[c] = r_edge
    assert torch.equal(d_edge, want_edge)
    assert torch.allclose(acc[:W], r_pos, rtol=1e-4, atol=1e-4) and torch.allclose(acc[W:W + 64].view(32, 2), r_w0, rtol=1e-3, atol=1e-3)
    # --- add_layernorm through a row map: residual rows gathered by c_j; the decoder norm's un-permute by the segmented inverse map
    a_perm = reg[j * n:(j + 1) * n][idx[c_off[j]:c_off[j] + n]].contiguous()
    out_m = torch.empty(n, W, dtype=torch.float16, device=dev)
    ops.add_layernorm_ex(reg[j * n:(j + 1) * n], b[:n].contiguous(), gamma, beta, 1e-5, out_m, n, ops.row_map(idx, n, c_off[j]))
    assert torch.equal(out_m, ops.add_layernorm(a_perm, b[:n].contiguous(), gamma, beta, 1e-5))
    inv_all = torch.cat([j2 * n + idx[inv_off[j2]:inv_off[j2] + n] for j2 in range(L)])
    out_d = torch.empty(nl, W, dtype=torch.float16, device=dev)
    ops.add_layernorm_ex(reg, None, gamma, beta, 1e-5, out_d, nl, ops.row_map(idx, n, inv_off[0], inv_off[1] - inv_off[0]))
    assert torch.equal(out_d, ops.add_layernorm(reg[inv_all].contiguous(), None, gamma, beta, 1e-5))
    # backward: dx scattered back through the same map (no dx in output order), d gamma / d beta added into the accumulator
    acc2 = torch.zeros(2, W, device=dev)
    dx_m = torch.zeros(nl, W, dtype=torch.float16, device=dev)
    ops.add_layernorm_bwd_ex(reg, None, gamma, dout, 1e-5, None, nl, acc2, a_map=ops.row_map(idx, n, inv_off[0], inv_off[1] - inv_off[0]),
                             dx_mapped=dx_m)
    r_dx, r_dg, r_db = ops.add_layernorm_bwd(reg[inv_all].contiguous(), None, gamma, dout, 1e-5)
    want_dx = torch.zeros_like(dx_m)
    want_dx[inv_all] = r_dx
    assert torch.equal(dx_m, want_dx)
    assert torch.allclose(acc2[0], r_dg, rtol=1e-3, atol=1e-2) and torch.allclose(acc2[1], r_db, rtol=1e-3, atol=1e-2)
    # a second dy contribution is added in fp16 first
    carry = rnd(n, W).half().to(dev)
    acc3, dx3 = torch.zeros(2, W, device=dev), torch.empty(n, W, dtype=torch.float16, device=dev)
    ops.add_layernorm_bwd_ex(reg[:n].contiguous(), b[:n].contiguous(), gamma, dout[:n].contiguous(), 1e-5, dx3, n, acc3, dy_add=carry)
    r3, _, _ = ops.add_layernorm_bwd(reg[:n].contiguous(), b[:n].contiguous(), gamma, (dout[:n] + carry).contiguous(), 1e-5)
    assert torch.equal(dx3, r3)

