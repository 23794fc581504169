"""Row N4 remainder: the YOLOX base detector's exp / loader / trainer plumbing around the (separately pinned) train step.
CPU: augmentation geometry against hand-derived values of the reference's formulas (yolox/data/data_augment.py:151-231,
datasets/mosaicdetection.py:18-36), optimizer groups and schedule of yolox/exp/yolox_base.py:232-274.  GPU: the trainer loop itself."""
import math
import os
import random

import numpy as np
import pytest
import torch


def test_affine_matrix_is_the_reference_formula():
    from yolox.data import mosaic as MZ
    random.seed(5)
    M, s = MZ.affine_matrix((640, 480), degrees=10.0, translate=0.1, scales=(0.5, 1.5), shear=2.0)
    random.seed(5)                                    # the same draws in the same order: angle, scale, shear x, shear y, tx, ty
    a = math.radians(random.uniform(-10, 10))
    sc = random.uniform(0.5, 1.5)
    shx, shy = math.tan(math.radians(random.uniform(-2, 2))), math.tan(math.radians(random.uniform(-2, 2)))
    tx, ty = random.uniform(-0.1, 0.1) * 640, random.uniform(-0.1, 0.1) * 480
    R = np.array([[sc * math.cos(a), sc * math.sin(a), 0.0], [-sc * math.sin(a), sc * math.cos(a), 0.0]])   # cv2.getRotationMatrix2D(centre 0)
    want = np.array([R[0] + shy * R[1], R[1] + shx * R[0]])
    want[0, 2], want[1, 2] = tx, ty
    assert s == sc and np.allclose(M, want, atol=1e-12)


def test_warp_affine_geometry_and_box_hull():
    from yolox.data import mosaic as MZ
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)
    ident = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(MZ.warp_affine(img, ident, (60, 40)), img)
    shift = np.array([[1.0, 0, 7], [0, 1.0, -3]])                    # dst(x, y) = src(x - 7, y + 3)
    out = MZ.warp_affine(img, shift, (60, 40))
    assert np.array_equal(out[:37, 7:], img[3:, :53]) and (out[:, :7] == 114).all() and (out[37:] == 114).all()
    half = np.array([[0.5, 0, 0], [0, 0.5, 0]])                      # downscale by 2 into a 30 x 20 corner of the canvas
    out = MZ.warp_affine(img, half, (60, 40))
    assert (out[25:, 35:] == 114).all() and abs(float(out[5, 5, 0]) - float(img[10, 10, 0])) <= 1
    # boxes: hull of the four warped corners, clipped
    a = math.radians(30)
    rot = np.array([[math.cos(a), math.sin(a), 10.0], [-math.sin(a), math.cos(a), 20.0]])
    box = np.array([[10.0, 5.0, 30.0, 15.0]])
    got = MZ.boxes_after_affine(box, rot, (100, 100))
    pts = np.array([[10, 5, 1], [30, 15, 1], [10, 15, 1], [30, 5, 1]], dtype=np.float64) @ rot.T
    want = [pts[:, 0].min(), pts[:, 1].min(), pts[:, 0].max(), pts[:, 1].max()]
    assert np.allclose(got[0], np.clip(want, 0, 100))


def test_mosaic_tiles_meet_at_the_centre():
    from yolox.data import mosaic as MZ
    H = W = 64
    for (xc, yc, w, h) in ((70, 50, 64, 48), (40, 90, 30, 64), (96, 96, 64, 64)):
        canvas = np.zeros((2 * H, 2 * W), dtype=np.int32)
        for k in range(4):
            (X1, Y1, X2, Y2), (x1, y1, x2, y2) = MZ.mosaic_tile(k, xc, yc, w, h, 2 * H, 2 * W)
            assert X2 - X1 == x2 - x1 and Y2 - Y1 == y2 - y1 and 0 <= x1 <= x2 <= w and 0 <= y1 <= y2 <= h
            assert 0 <= X1 <= X2 <= 2 * W and 0 <= Y1 <= Y2 <= 2 * H
            canvas[Y1:Y2, X1:X2] += 1
            # the tile corner that touches the centre is kept: tile 0 its bottom-right, tile 3 its top-left, ...
            assert (X2 == xc if k in (0, 2) else X1 == xc) and (Y2 == yc if k in (0, 1) else Y1 == yc)
            assert (x2 == w if k in (0, 2) else x1 == 0) and (y2 == h if k in (0, 1) else y1 == 0)
        assert canvas.max() == 1                                       # the four tiles do not overlap


def test_mosaic_dataset_items_and_closing_the_mosaic():
    from yolox.data import TrainTransform
    from yolox.data import mosaic as MZ
    random.seed(1)
    np.random.seed(1)
    base = MZ.SyntheticDetectionDataset(16, 5, (96, 128), seed=3)
    ds = MZ.MosaicDetection(base, (96, 128), mosaic=True, preproc=TrainTransform(max_labels=120), mosaic_scale=(0.5, 1.5))
    for i in range(4):
        img, lab, info, _ = ds[i]
        assert img.shape == (3, 96, 128) and img.dtype == np.float32 and lab.shape == (120, 5)
        live = lab[lab[:, 3] > 0]
        assert len(live) > 0
        assert (live[:, 1] >= 0).all() and (live[:, 1] <= 128).all() and (live[:, 2] >= 0).all() and (live[:, 2] <= 96).all()
        assert set(np.unique(live[:, 0]).astype(int)) <= set(range(5))
    loader = MZ.YoloBatchLoader(ds, 3, MZ.InfiniteSampler(len(ds), seed=0))
    imgs, labels = loader.next()
    assert imgs.shape == (3, 3, 96, 128) and labels.shape == (3, 120, 5) and len(loader) == 5
    loader.close_mosaic()
    assert ds.enable_mosaic is False
    img, lab, info, iid = ds[2]                                       # plain letterbox of item 2 from now on
    raw, rl, hw, _ = base.pull_item(2)
    assert iid == 2 and info == hw and int((lab[:, 3] > 0).sum()) <= len(rl)


def test_infinite_sampler_shards_one_stream_between_ranks():
    from yolox.data.mosaic import InfiniteSampler
    import itertools
    a = list(itertools.islice(iter(InfiniteSampler(10, seed=4, rank=0, world_size=2)), 10))
    b = list(itertools.islice(iter(InfiniteSampler(10, seed=4, rank=1, world_size=2)), 10))
    whole = list(itertools.islice(iter(InfiniteSampler(10, seed=4)), 20))
    assert a == whole[0::2] and b == whole[1::2] and sorted(whole[:10]) == list(range(10))


def test_base_exp_optimizer_groups_and_schedule():
    """yolox_base.py:232-274: BatchNorm weights without decay, other weights with 5e-4, biases without; lr starts at warmup_lr = 0 and
    follows yoloxwarmcos (quadratic warm-up over 5 epochs, cosine to 5 % of the peak, flat for the last 15)."""
    from yolox.exp import get_exp
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp = get_exp(os.path.join(root, "exps", "default", "yolox_s.py"), None)
    assert (exp.depth, exp.width, exp.num_classes, exp.input_size, exp.no_aug_epochs, exp.max_epoch) == (0.33, 0.5, 80, (640, 640), 15, 300)
    model = exp.get_model()
    opt = exp.get_optimizer(64)
    assert opt is exp.get_optimizer(64) and len(opt.param_groups) == 3
    g_bn, g_w, g_b = opt.param_groups
    n_bn = sum(1 for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d))
    assert len(g_bn["params"]) == n_bn and g_bn["weight_decay"] == 0 and g_w["weight_decay"] == 5e-4 and g_b["weight_decay"] == 0
    assert all(p.dim() == 1 for p in g_bn["params"]) and all(p.dim() == 4 for p in g_w["params"])
    n_all = sum(1 for _ in model.parameters())
    assert len(g_bn["params"]) + len(g_w["params"]) + len(g_b["params"]) == n_all
    assert g_bn["lr"] == 0 and g_bn["nesterov"] and g_bn["momentum"] == 0.9
    peak, ipe = exp.basic_lr_per_img * 64, 100
    sch = exp.get_lr_scheduler(peak, ipe)
    assert sch.update_lr(0) == 0 and abs(sch.update_lr(250) - peak * 0.25) < 1e-12 and abs(sch.update_lr(500) - peak) < 1e-12
    assert abs(sch.update_lr(ipe * 300 - 1) - peak * 0.05) < 1e-12 and abs(sch.update_lr(ipe * 285) - peak * 0.05) < 1e-12
    mid = 500 + (ipe * 285 - 500) // 2
    assert abs(sch.update_lr(mid) - (peak * 0.05 + 0.5 * (peak - peak * 0.05) * (1 + math.cos(math.pi * (mid - 500) / (ipe * 285 - 500))))) < 1e-12
    # preprocess: targets (cls, cx, cy, w, h) follow a multiscale resize
    x, t = torch.zeros(2, 3, 640, 640), torch.tensor([[[1.0, 100, 200, 50, 60]]]).repeat(2, 1, 1)
    x2, t2 = exp.preprocess(x, t.clone(), (480, 512))
    assert x2.shape[-2:] == (480, 512) and torch.allclose(t2[0, 0], torch.tensor([1.0, 100 * 0.8, 200 * 0.75, 50 * 0.8, 60 * 0.75]))


@pytest.mark.gpu
def test_base_trainer_runs_epochs_closes_mosaic_and_checkpoints(tmp_path):
    """tools/train.py's loop on a tiny YOLOX (width .125, 5 classes, 96 x 128 synthetic images): 3 epochs x 4 iterations; mosaic closes
    and the L1 loss starts at the epoch with ``epoch + 1 == max_epoch - no_aug_epochs`` (trainer.py:200-213: with no_aug_epochs = 1 that
    is the SECOND of three epochs -- the reference's rule runs no_aug_epochs + 1 plain epochs), multiscale resize every 10 iterations,
    EMA, checkpoints."""
    import argparse
    from yolox.exp.yolox_base import Exp
    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
    exp = Exp()
    exp.depth, exp.width, exp.num_classes = 0.33, 0.125, 5
    exp.input_size, exp.multiscale_range, exp.synthetic_len = (96, 128), 1, 16
    exp.max_epoch, exp.no_aug_epochs, exp.warmup_epochs, exp.basic_lr_per_img = 3, 1, 1, 1e-4 / 4
    exp.output_dir, exp.seed = str(tmp_path), 0
    args = argparse.Namespace(batch_size=4, fp16=True, experiment_name="tiny", resume=False, ckpt=None, cache=False, async_ckpt=False)
    tr = exp.get_trainer(args)
    tr.train()
    assert tr.max_iter == 4 and len(tr.history) == 12
    rows = tr.history_floats()
    assert all(np.isfinite(r["total_loss"]) for r in rows)
    assert rows[0]["lr"] > 0 and all(r["l1_loss"] == 0 for r in rows[:4]) and all(r["l1_loss"] > 0 for r in rows[4:])
    assert tr.train_loader.dataset.enable_mosaic is False and tr.model.head.use_l1 is True
    assert tr.input_size[0] in (64, 96, 128)                          # a multiscale size was drawn at iteration 10
    for name in ("latest_ckpt.pth", "last_mosaic_epoch_ckpt.pth"):
        assert os.path.isfile(os.path.join(str(tmp_path), "tiny", name)), name
    ck = torch.load(os.path.join(str(tmp_path), "tiny", "latest_ckpt.pth"), weights_only=True)
    assert ck["start_epoch"] == 3 and set(ck["model"].keys()) == set(tr.model.state_dict().keys())
