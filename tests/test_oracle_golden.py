"""Pins oracle/tscd_oracle.py (the CPU restatement) to vectors produced by the reference itself.

tests/golden/*.npz were written by oracle/make_golden.py, which imports /root/reference in the build
container (TSCD exp ovis_tscd_large with width .125 / depth .33, seeded).  Integer outputs must be
bit-exact; floats within 1e-4 abs / 1e-4 rel of the reference's fp32 results (both sides are fp32 CPU
PyTorch; the slack covers op re-association only).
"""
import numpy as np
import pytest
import torch

from oracle import tscd_oracle as O

ATOL, RTOL = 2e-4, 2e-4


def close(a, b, atol=ATOL, rtol=RTOL):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


def rows_match(a, b, atol=2e-3):
    """Fraction of rows of `a` that have a partner row in `b` (NMS output order can flip on fp near-ties,
    so detections are compared as sets: same count, >= 99% of rows identical within atol)."""
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    if len(a) == 0:
        return 1.0
    hit = 0
    for r in a:
        hit += bool((np.abs(b - r[None]).max(1) <= atol + 1e-3 * np.abs(r).max()).any())
    return hit / len(a)


def clone_sd(sd, grad=False):
    out = {}
    for k, v in sd.items():
        v = v.clone()
        if grad and v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
        out[k] = v
    return out


def test_timing_signal(golden):
    cfg, sd, vec, meta = golden
    close(O.get_timing_signal_1d(torch.arange(0, int(meta["B"])), 256), vec["time_embedding"], 1e-6, 1e-6)


@pytest.fixture(scope="module")
def eval_runs(golden):
    cfg, sd, vec, meta = golden
    sd = clone_sd(sd)
    state = O.MatcherState()
    temb = torch.from_numpy(vec["time_embedding"])
    runs = {}
    for name, resume in (("evalA_", False), ("evalB_", True)):
        imgs = torch.from_numpy(vec[name + "imgs"]).float()
        aux = {}
        with torch.no_grad():
            res, res_ori = O.tscd_forward(sd, cfg, imgs, None, temb, 0.5, int(meta["lframe"]), int(meta["gframe"]),
                                          resume, False, state, aux)
        runs[name] = (res, res_ori, aux)
    return runs


@pytest.mark.parametrize("name", ["evalA_", "evalB_"])
def test_eval_against_reference(golden, eval_runs, name):
    cfg, sd, vec, meta = golden
    res, res_ori, aux = eval_runs[name]
    L = int(meta["lframe"])
    for i in range(3):
        close(aux["fpn"][i], vec[f"{name}fpn{i}"])
    # INT: proposal indices bit-exact
    for i in range(int(meta["B"])):
        assert np.array_equal(aux["agg_idx"][i].numpy(), vec[f"{name}agg_idx{i}"])
        close(aux["pred_result"][i], vec[f"{name}pred_result{i}"])
    # INT: Hungarian permutations bit-exact
    assert len(aux["hungarian"]) == int(vec[name + "n_hung"])
    for i, col in enumerate(aux["hungarian"]):
        assert np.array_equal(np.asarray(col), vec[f"{name}hung_col{i}"]), f"hungarian {i}"
    close(aux["agg_cls"], vec[name + "agg_cls"])
    close(aux["agg_iou_cls"], vec[name + "agg_iou_cls"])
    close(aux["agg_iou_obj"], vec[name + "agg_iou_obj"])
    close(aux["cafm"], vec[name + "cafm"], 5e-4, 5e-4)
    close(aux["cls_preds"], vec[name + "cls_preds"], 5e-4, 5e-4)
    close(aux["matcher_obj_preds"], vec[name + "matcher_obj_preds"], 5e-4, 5e-4)
    close(aux["matcher_reg_preds"], vec[name + "matcher_reg_preds"], 5e-4, 5e-4)
    for i in range(L):
        assert res[i].shape == vec[f"{name}result{i}"].shape
        assert rows_match(res[i], vec[f"{name}result{i}"]) >= 0.99
        assert res_ori[i].shape == vec[f"{name}result_ori{i}"].shape
        assert rows_match(res_ori[i], vec[f"{name}result_ori{i}"]) >= 0.99


def test_edge_maps(golden, eval_runs):
    cfg, sd, vec, meta = golden
    aux = eval_runs["evalA_"][2]
    off = 0
    for k in range(3):
        e = vec[f"evalA_edge{k}"]
        n = e.shape[2] * e.shape[3]
        close(aux["edge_flat"][:, off:off + n].permute(0, 2, 1).reshape(e.shape), e)
        off += n


def test_train_against_reference(golden):
    cfg, sd, vec, meta = golden
    sd = clone_sd(sd, grad=True)
    imgs = torch.from_numpy(vec["evalA_imgs"]).float()
    labels = torch.from_numpy(vec["train_labels"])
    temb = torch.from_numpy(vec["time_embedding"])
    aux = {}
    out = O.tscd_forward(sd, cfg, imgs, labels, temb, 0.5, int(meta["lframe"]), int(meta["gframe"]), False, True,
                         O.MatcherState(), aux)
    # INT: SimOTA foreground indices bit-exact
    for i in range(int(meta["B"])):
        assert np.array_equal(np.asarray(aux["fg_idx"][i]), vec[f"train_fg_idx{i}"]), f"fg_idx {i}"
        assert np.array_equal(aux["agg_idx"][i].numpy(), vec[f"train_agg_idx{i}"])
    for i, col in enumerate(aux["hungarian"]):
        assert np.array_equal(np.asarray(col), vec[f"train_hung_col{i}"])
    for k in O.LOSS_KEYS:
        close(torch.as_tensor(float(out[k])), vec["train_loss_" + k], 1e-4, 1e-4)
    out["total_loss"].backward()
    # every parameter that gets a gradient in the reference gets one here and vice versa
    ref_with = set(vec["train_grad_names"].tolist())
    ours_with = {k for k, v in sd.items() if v.requires_grad and v.grad is not None and k in ref_with}
    assert ours_with == ref_with
    for k in vec:
        if k.startswith("train_grad::"):
            g = sd[k[len("train_grad::"):]].grad
            scale = max(np.abs(vec[k]).max(), 1e-6)
            close(g, vec[k], 2e-4 * scale + 1e-7, 2e-3)
        if k.startswith("train_bn::"):
            close(sd[k[len("train_bn::"):]], vec[k], 1e-5, 1e-4)
