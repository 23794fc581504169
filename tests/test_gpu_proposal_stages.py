"""Stage-level GPU parity of the small proposal-stage kernels and of the device Hungarian cost
(row a12) on identical inputs.

* decode_score  vs the reference's ops evaluated on fp16 CPU tensors (tscd_head.py:735-770 decode_outputs, :1546-1575
  cxcywh->xyxy + class max + det rows; each torch op rounds to fp16 like autocast / .half() does on the GPU): boxes and
  scores equal up to 1 fp16 ulp (device expf vs libm), class ids equal wherever the maximum is unique.
* gather_rows / gather_det vs plain indexing of the flattened maps: bit-exact copies (find_feature_score :976-1006).
* scatter_rows = the adjoint of gather_rows: <gather(x), g> == <x, scatter(g)> and scatter == index_put, bit-exact.
* match_cost (EPI_COST epilogue of the MFMA GEMM) vs the oracle's double_match cost (tscd_matching.py:912-937) on the
  golden clip within fp16 tolerance, and the host solver on the DEVICE-produced fp16 costs == scipy on the same costs,
  exactly, for the whole frame-to-frame chain (no ">= 90 %").
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tscd_oracle as O
from test_state_dict_compat import build_model

STRIDES = (8, 16, 32)


def ulp16(x):
    """spacing of fp16 at |x| (>= 2^-24)"""
    x = np.abs(np.asarray(x, np.float32))
    e = np.floor(np.log2(np.maximum(x, 2.0 ** -14)))
    return 2.0 ** (e - 10)


def _levels(B, S, nc, seed):
    g = torch.Generator().manual_seed(seed)
    hw = [(S // s, S // s) for s in STRIDES]
    raw = []
    for (h, w) in hw:
        r = torch.randn(B, h, w, 5 + nc, generator=g)
        r[..., 2:4] *= 1.2                      # exp() of these: boxes from ~0.1 to ~30 strides
        r[..., 4:] = r[..., 4:] * 2.0 - 1.0
        raw.append(r.half())
    return hw, raw


@pytest.mark.parametrize("B,S,nc,training", [(3, 128, 25, True), (2, 192, 30, False), (1, 576, 25, True)])
def test_decode_score_matches_reference_ops_in_fp16(B, S, nc, training):
    from tscd_amd import ops
    hw, raw = _levels(B, S, nc, S + nc)
    geo = ops.LevelGeometry(hw, STRIDES)
    outputs, det, score = ops.decode_score([r.cuda().contiguous() for r in raw], geo, nc, training)
    torch.cuda.synchronize()
    # ---- the reference's op sequence on fp16 CPU tensors ----
    flat = torch.cat([r.reshape(B, -1, 5 + nc) for r in raw], 1)                        # [B,A,5+NC] fp16
    xs, ys, ss = O.make_grids(hw, STRIDES, torch.float16)
    grid = torch.stack((xs, ys), 1)[None]
    st = ss[None, :, None]
    dec = flat.clone()
    dec[..., :2] = (flat[..., :2] + grid) * st                                          # tscd_head.py:768
    dec[..., 2:4] = torch.exp(flat[..., 2:4]) * st                                      # :769
    if training:
        o = outputs.cpu()
        assert torch.equal(o[..., 4:], flat[..., 4:])                                   # raw logits pass through untouched
        assert torch.equal(o[..., :2], dec[..., :2])                                    # add / mul: exactly rounded ops
        d = (o[..., 2:4].float() - dec[..., 2:4].float()).abs().numpy()
        assert (d <= ulp16(dec[..., 2:4].float().numpy())).all()                        # exp: 1 ulp (the stride is a power of two)
    sig = torch.sigmoid(flat[..., 4:])                                                  # eval head: obj/cls .sigmoid() on fp16
    box = torch.empty_like(dec[..., :4])
    box[..., 0] = dec[..., 0] - dec[..., 2] / 2                                         # postprocess_widx :1561-1566
    box[..., 1] = dec[..., 1] - dec[..., 3] / 2
    box[..., 2] = dec[..., 0] + dec[..., 2] / 2
    box[..., 3] = dec[..., 1] + dec[..., 3] / 2
    dt = det.cpu()
    # boxes: built from the kernel's own w/h (1 ulp off at most), so compare against a recomputation from `outputs` exactly
    # when available, else within 1 ulp of the centre coordinate magnitude + the w/h slack
    tol = ulp16(box.float().numpy()) + ulp16(dec[..., 2:4].float().numpy()).max(-1, keepdims=True) * 0.5 + 1e-3
    assert (np.abs(dt[..., :4].float().numpy() - box.float().numpy()) <= 2 * tol).all()
    if training:
        o = outputs.cpu()
        b2 = torch.empty_like(box)
        b2[..., 0] = o[..., 0] - o[..., 2] / 2
        b2[..., 1] = o[..., 1] - o[..., 3] / 2
        b2[..., 2] = o[..., 0] + o[..., 2] / 2
        b2[..., 3] = o[..., 1] + o[..., 3] / 2
        assert torch.equal(dt[..., :4], b2)                                             # same ops on the same fp16 values: exact
    s_ref = sig.float().numpy()
    s_got = torch.cat((dt[..., 4:5], dt[..., 7:]), -1).float().numpy()
    assert (np.abs(s_got - s_ref) <= ulp16(s_ref)).all()                                # sigmoid: 1 ulp
    best, arg = dt[..., 7:].max(-1)
    assert torch.equal(dt[..., 5], best)                                                # class max of the kernel's own sigmoids
    assert torch.equal(dt[..., 6].long(), arg)                                          # first maximum (torch.max rule)
    uniq = (dt[..., 7:] == best.unsqueeze(-1)).sum(-1) == 1
    ref_arg = sig[..., 1:].max(-1).indices
    marg = torch.topk(sig[..., 1:].float(), 2, -1).values
    clear = (marg[..., 0] - marg[..., 1]).numpy() > 2 * ulp16(marg[..., 0].numpy())
    assert torch.equal(arg[torch.from_numpy(clear) & uniq], ref_arg[torch.from_numpy(clear) & uniq])
    assert torch.equal(score.cpu(), dt[..., 4] * dt[..., 5])                            # fp16 product of the two fp16 factors


def _select(B, A, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    cnt = torch.randint(lo, hi + 1, (B,), generator=g).tolist()
    cnt[0] = hi
    if B > 2:
        cnt[2] = 0                                                                      # an empty frame
    cap = max(hi, 1)
    idx = torch.zeros(B, cap, dtype=torch.int32)
    for b in range(B):
        idx[b, :cnt[b]] = torch.randperm(A, generator=g)[:cnt[b]].sort().values.int()
    return idx, cnt


@pytest.mark.parametrize("B,S,W,nf", [(4, 128, 64, 2), (3, 192, 256, 3), (2, 576, 320, 1)])
def test_gather_rows_scatter_adjoint_and_gather_det(B, S, W, nf):
    from tscd_amd import ops
    g = torch.Generator().manual_seed(S + W)
    hw = [(S // s, S // s) for s in STRIDES]
    geo = ops.LevelGeometry(hw, STRIDES)
    A = geo.A
    maps = [[torch.randn(B, h, w, W, generator=g).half() for (h, w) in hw] for _ in range(nf)]
    sel, cnt = _select(B, A, 5, min(A, 300), S)
    off = [0]
    for n in cnt:
        off.append(off[-1] + n)
    N = off[-1]
    dev_maps = [[m.cuda() for m in f] for f in maps]
    sel_d = sel.cuda()
    frame_off = torch.tensor(off, dtype=torch.int32).cuda()
    outs = ops.gather_rows(dev_maps, geo, sel_d, frame_off, N, W)
    torch.cuda.synchronize()
    b_idx = torch.cat([torch.full((cnt[b],), b, dtype=torch.long) for b in range(B)])
    a_idx = torch.cat([sel[b, :cnt[b]].long() for b in range(B)])
    for f in range(nf):
        flat = torch.cat([m.reshape(B, -1, W) for m in maps[f]], 1)                     # [B,A,W] (find_feature_score's view)
        assert torch.equal(outs[f].cpu(), flat[b_idx, a_idx]), f
    # adjoint: scatter the row gradients back into zeroed maps
    grads = [torch.randn(N, W, generator=g).half() for _ in range(nf)]
    zmaps = [[torch.zeros_like(m).cuda() for m in f] for f in maps]
    ops.scatter_rows([x.cuda() for x in grads], zmaps, geo, sel_d, frame_off, N, W)
    torch.cuda.synchronize()
    for f in range(nf):
        want = torch.zeros(B, A, W, dtype=torch.float16)
        want[b_idx, a_idx] = grads[f]                                                   # distinct (frame, anchor) pairs
        got = torch.cat([m.reshape(B, -1, W) for m in zmaps[f]], 1).cpu()
        assert torch.equal(got, want), f
        x = torch.cat([m.reshape(B, -1, W) for m in maps[f]], 1).double()
        lhs = (outs[f].cpu().double() * grads[f].double()).sum()
        rhs = (x * got.double()).sum()
        assert lhs == rhs                                                               # <G x, g> == <x, G^T g>, exactly
    # the adding form (one feature's rows land in maps that already hold another consumer's gradient): == map + scattered rows, rounded once
    base = [[torch.randn(B, h, w, W, generator=g).half() for (h, w) in hw] for _ in range(nf)]
    amaps = [[m.clone().cuda() for m in f] for f in base]
    add = tuple(f % 2 == 0 for f in range(nf))
    ops.scatter_rows([x.cuda() for x in grads], amaps, geo, sel_d, frame_off, N, W, add=add)
    torch.cuda.synchronize()
    for f in range(nf):
        b0 = torch.cat([m.reshape(B, -1, W) for m in base[f]], 1)
        want = b0.clone()
        if add[f]:
            want[b_idx, a_idx] = (b0[b_idx, a_idx].float() + grads[f].float()).half()
        else:
            want[b_idx, a_idx] = grads[f]
        got = torch.cat([m.reshape(B, -1, W) for m in amaps[f]], 1).cpu()
        assert torch.equal(got, want), ("add", f)
    det = torch.randn(B, A, 7 + 25, generator=g).half()
    rows = ops.gather_det(det.cuda().contiguous(), sel_d, frame_off, N)
    torch.cuda.synchronize()
    assert torch.equal(rows.cpu(), det[b_idx, a_idx])


def _scipy_chain(costs_u16, kinds):
    """The reference's frame loop (tscd_matching.py:779-851) on given cost matrices: kind 0 = first frame matched against
    itself, 2 = against the previous frame whose rows the reference keeps in the previous assignment's order."""
    from scipy.optimize import linear_sum_assignment
    order, cols = None, []
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
        cols.append(c)
    return cols


def test_device_hungarian_cost_and_assignment_chain(golden):
    from tscd_amd import ops
    from tscd_amd._host import lap_f16
    from scipy.optimize import linear_sum_assignment
    cfg, sd, vec, meta = golden
    sd = {k: v.clone() for k, v in sd.items()}
    aux = {}
    imgs = torch.from_numpy(vec["evalA_imgs"]).float()
    temb = torch.from_numpy(vec["time_embedding"])
    L = int(meta["lframe"])
    with torch.no_grad():
        O.tscd_forward(sd, cfg, imgs, None, temb, 0.5, L, int(meta["gframe"]), False, False, O.MatcherState(), aux)
    ppf = aux["preds_per_frame"]
    reg16, cls16 = aux["agg_iou_obj"].half(), aux["agg_iou_cls"].half()
    spans, s = [], 0
    for n in ppf[:L]:
        spans.append((s, s + n))
        s += n
    # ---- (1) the cost matrices: device fp16 GEMM vs the oracle's expression on the same fp16-rounded rows (fp32 math) ----
    pairs_idx = [(0, 0)] + [(i - 1, i) for i in range(1, L)]
    nz = lambda t: t / (t.norm(dim=1)[:, None] + 1e-6)
    dev_costs = []
    for (pi, ci) in pairs_idx:
        (p0, p1), (c0, c1) = spans[pi], spans[ci]
        dc = ops.match_cost(reg16[p0:p1].cuda().contiguous(), reg16[c0:c1].cuda().contiguous(), cls16[p0:p1].cuda().contiguous(),
                            cls16[c0:c1].cuda().contiguous())
        torch.cuda.synchronize()
        rr, rc = reg16[p0:p1].float(), reg16[c0:c1].float()
        cr, cc = cls16[p0:p1].float(), cls16[c0:c1].float()
        want = 1 - ((nz(cr) @ nz(cc).t() + nz(rr) @ nz(rc).t()) / 2)
        got = dc.float().cpu()
        # fp16 operands (normalised rows rounded to fp16: 2^-11 relative each) and an fp16 result near 1: 3e-3 absolute
        assert (got - want).abs().max().item() <= 3e-3, (pi, ci, (got - want).abs().max().item())
        dev_costs.append(np.ascontiguousarray(dc.cpu().numpy().view(np.uint16)))
        # ---- (2) the host solver on the device-produced matrix == scipy on the same matrix, exactly ----
        r1, c1_ = lap_f16(dev_costs[-1])
        r2, c2 = linear_sum_assignment(np.where(np.isnan(got.numpy()), 0, got.numpy()).astype(np.float64))
        assert np.array_equal(r1, r2) and np.array_equal(c1_, c2), (pi, ci)
    # ---- (3) the product's matcher (one D2H, chain solver on a worker thread) == the reference's loop with scipy on those
    # device costs: column assignment of every local frame, exactly ----
    m = build_model(cfg)
    m.load_state_dict(golden[1], strict=True)
    h = m.cuda().eval().head
    h.local_reg_matcher._clear_state()
    with torch.no_grad():
        h.local_reg_matcher(aux["f_reg"].half().cuda(), reg16.cuda(), cls16.cuda(), aux["f_edge"].half().cuda(), ppf[:L],
                            temb[:L].cuda(), resume=False)
    torch.cuda.synchronize()
    want_cols = _scipy_chain(dev_costs, [0] + [2] * (L - 1))
    got_cols = [np.asarray(x) for x in h.local_reg_matcher.last_hungarian]
    assert len(got_cols) == len(want_cols)
    for fi, (a, b) in enumerate(zip(got_cols, want_cols)):
        assert np.array_equal(a, b), fi


@pytest.mark.parametrize("n_per,L,C", [(500, 4, 1024), (37, 3, 64), (130, 2, 256)])
def test_cost_matrices_in_one_launch_are_the_per_matrix_launches_bits(n_per, L, C, monkeypatch):
    """ops.match_costs_clip: every cost matrix of a clip from ONE launch (tscd_match_cost_pairs_f16: a wave per 32 x 32 tile) against one
    implicit-GEMM launch per matrix (the EPI_COST epilogue): the same MFMA chain over ascending k, so the fp16 costs -- which decide
    exact ties in the assignment -- are bit-identical; and both within fp16 of 1 - (cos_reg + cos_cls) / 2."""
    from tscd_amd import ops
    g = torch.Generator().manual_seed(n_per + L)
    n = n_per * L
    reg = torch.randn(n, C, generator=g).half().cuda()
    cls = torch.randn(n, C, generator=g).half().cuda()
    pairs = [(0, n_per, 0, n_per)] + [((j - 1) * n_per, j * n_per, j * n_per, (j + 1) * n_per) for j in range(1, L)]
    monkeypatch.setattr(ops, "COST_PAIRS_ONE_LAUNCH", True)
    flat1, shapes1 = ops.match_costs_clip(reg, cls, pairs)
    monkeypatch.setattr(ops, "COST_PAIRS_ONE_LAUNCH", False)
    flat0, shapes0 = ops.match_costs_clip(reg, cls, pairs)
    torch.cuda.synchronize()
    assert shapes0 == shapes1 and torch.equal(flat0, flat1)
    nrm = lambda t: torch.nn.functional.normalize(t.float(), dim=1)
    o = 0
    for (r0, r1, c0, c1) in pairs:
        want = 1 - 0.5 * (nrm(reg[r0:r1]) @ nrm(reg[c0:c1]).T + nrm(cls[r0:r1]) @ nrm(cls[c0:c1]).T)
        got = flat1[o:o + (r1 - r0) * (c1 - c0)].view(r1 - r0, c1 - c0).float()
        assert (got - want).abs().max().item() <= 3e-3
        o += (r1 - r0) * (c1 - c0)

