"""SimOTA in BOTH arithmetic modes, pinned to the reference itself.

tests/golden/simota_fp16.npz holds the reference's own ``TSCDHead.get_fg_idx`` outputs on fp16 tensors (what training
feeds it under autocast, tscd_trainer.py:159-164) for three cases (oracle/make_golden_simota.py): the golden clip, a
tie-heavy 256^2 case with duplicated ground-truth boxes, and a 576^2 bench-like case with coordinates beyond 512 px and
areas that overflow fp16.

CPU: the dtype-faithful oracle must reproduce those outputs bit for bit (fg anchors, matched class, matched IoU as
fp16 values, box targets, counts).  GPU: ``tscd_simota_assign`` with the product default ``emu16=1`` must reproduce the
reference's fp16 outputs, and with ``emu16=0`` the oracle's fp32 evaluation of the same fp16 values -- INT-exact.
"""
import os

import numpy as np
import pytest
import torch

from oracle import tscd_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
STRIDES = (8, 16, 32)


@pytest.fixture(scope="module")
def cases():
    z = np.load(os.path.join(HERE, "golden", "simota_fp16.npz"))
    out = []
    for ci in range(int(z["n_cases"])):
        out.append({k[len(f"c{ci}_"):]: z[k] for k in z.files if k.startswith(f"c{ci}_")})
    return out


def oracle_run(c, dtype):
    S = int(c["S"])
    hw = [(S // s, S // s) for s in STRIDES]
    xs, ys, ss = O.make_grids(hw, STRIDES, dtype)
    outputs = torch.from_numpy(c["outputs"]).to(dtype)
    labels = torch.from_numpy(c["labels"]).to(dtype)
    cfg = O.TSCDConfig(num_classes=outputs.shape[-1] - 5)
    return O.simota_assign(outputs, labels, xs, ys, ss, cfg), hw, cfg


def dense(ref, B, A):
    """per-image lists -> dense fg mask / matched class / matched IoU"""
    fg = torch.stack(list(ref[4])).numpy()
    iou = np.zeros((B, A), np.float32)
    cls = -np.ones((B, A), np.int64)
    for b in range(B):
        idx = np.nonzero(fg[b])[0]
        ct = ref[1][b].float()
        iou[b, idx] = ct.sum(-1).numpy()
        cls[b, idx] = ct.argmax(-1).numpy()
    return fg, cls, iou


def test_oracle_fp16_mode_reproduces_reference(cases):
    """CPU: oracle on fp16 tensors == the reference on fp16 tensors, everything exact (both are CPU torch ops in the
    reference's order; any re-association would show as a flipped near-tie)."""
    for ci, c in enumerate(cases):
        ref, hw, cfg = oracle_run(c, torch.float16)
        B, A = c["fg_mask"].shape
        fg, cls, iou = dense(ref, B, A)
        assert np.array_equal(fg, c["fg_mask"]), ci
        assert np.array_equal(cls, c["matched_cls"]), ci
        assert np.array_equal(iou, c["pred_iou"], equal_nan=True), ci                   # fp16 values, compared exactly
        assert np.array_equal(torch.cat([r.float() for r in ref[2]], 0).numpy(), c["reg_target"]), ci
        assert float(ref[5]) == float(c["num_fg"]) and float(ref[6]) == float(c["num_gts"]), ci


def test_fp16_and_fp32_modes_differ_on_the_bench_like_case(cases):
    """Why the mode matters (and why a test in the wrong mode proves nothing): on the 576^2 case the fp32 evaluation of
    the SAME fp16 values assigns a different foreground set."""
    c = cases[2]
    r16, _, _ = oracle_run(c, torch.float16)
    r32, _, _ = oracle_run(c, torch.float32)
    assert float(r16[5]) != float(r32[5])


@pytest.mark.gpu
@pytest.mark.parametrize("emu16", [1, 0])
def test_hip_simota_bit_exact_both_modes(cases, emu16):
    from tscd_amd import ops
    for ci, c in enumerate(cases):
        B, A = c["fg_mask"].shape
        S = int(c["S"])
        hw = [(S // s, S // s) for s in STRIDES]
        nc = c["outputs"].shape[-1] - 5
        if emu16:
            want_fg, want_cls, want_iou = c["fg_mask"], c["matched_cls"], c["pred_iou"]          # the reference itself
        else:
            ref, _, _ = oracle_run(c, torch.float32)
            want_fg, want_cls, want_iou = dense(ref, B, A)
        outputs = torch.from_numpy(c["outputs"]).cuda().contiguous()
        labels = torch.from_numpy(c["labels"]).cuda()
        geo = ops.LevelGeometry(hw, STRIDES)
        fg, mg, pi, nfg, ngt = ops.simota_assign(outputs, labels, geo, nc, 64, emu16=bool(emu16))
        torch.cuda.synchronize()
        fg, mg, pi = fg.cpu().numpy().astype(bool), mg.cpu().numpy(), pi.cpu().numpy()
        assert np.array_equal(fg, want_fg), (ci, emu16, int(fg.sum()), int(want_fg.sum()), int((fg != want_fg).sum()))
        lab_cls = c["labels"][..., 0].astype(np.int64)
        got_cls = np.where(fg, np.take_along_axis(lab_cls, np.clip(mg, 0, None), 1), -1)
        known = np.isfinite(want_iou) & (want_iou != 0)      # the golden recovers the class from onehot * iou: undefined where that is 0 / NaN
        assert np.array_equal(got_cls[known], want_cls[known]), (ci, emu16)
        if emu16:
            # fp16 values: exact wherever the reference's value is a number.  Where it is NaN -- (matching * ious).sum(0) picks
            # up 0 * NaN from the anchor's pair with ANOTHER, far-away gt (fp16 overflow of prod(br - tl)) and poisons the class
            # target -- the kernel deliberately keeps the matched pair's own IoU (simota.hip header).
            num = ~np.isnan(want_iou)
            assert np.array_equal(pi[num], want_iou[num]), (ci, emu16)
            assert np.isfinite(pi[~num]).all() and (pi[~num] >= 0).all() and (pi[~num] <= 1).all(), (ci, emu16)
        else:
            np.testing.assert_allclose(pi, want_iou, rtol=1e-5, atol=1e-6)
        assert nfg.cpu().tolist() == want_fg.sum(1).tolist()
        assert ngt.cpu().tolist() == (c["labels"].astype(np.float32).sum(-1) > 0).sum(1).tolist()
