"""Device-side clip pre-processing: tscd_clip_preproc_u8 + TrainTransformGPU against the host pipeline the reference
runs per frame (vid.py pull_item resize -> data_augment.py TrainTransform: HSV jitter, mirror, letterbox), with the same random decisions."""
import json
import os
import random

import numpy as np
import pytest
import torch


def _host_and_gpu_records(img, annos, input_dim, seed, flip_prob=0.5, hsv_prob=1.0):
    from yolox.data.data_augment import TrainTransform, TrainTransformGPU, resize_bilinear
    h, w = img.shape[:2]
    r = min(input_dim[0] / h, input_dim[1] / w)
    nh, nw = int(h * r), int(w * r)
    scaled = annos.copy()
    scaled[:, :4] *= r
    random.seed(seed); np.random.seed(seed)
    want_img, want_lab = TrainTransform(50, flip_prob, hsv_prob)(resize_bilinear(img, nh, nw), scaled.copy(), input_dim)
    random.seed(seed); np.random.seed(seed)
    meta, lab = TrainTransformGPU(50, flip_prob, hsv_prob)((h, w), (nh, nw), scaled.copy(), input_dim)
    return want_img, want_lab, meta, lab


def test_host_half_draws_the_same_decisions_and_labels():
    rng = np.random.default_rng(0)
    for seed in range(12):
        h, w = int(rng.integers(60, 200)), int(rng.integers(60, 260))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        n = int(rng.integers(0, 4))
        annos = np.zeros((n, 5))
        for i in range(n):
            x1, y1 = rng.uniform(0, w * 0.6), rng.uniform(0, h * 0.6)
            annos[i] = (x1, y1, x1 + rng.uniform(1, w * 0.4), y1 + rng.uniform(1, h * 0.4), rng.integers(0, 5))
        want_img, want_lab, meta, lab = _host_and_gpu_records(img, annos, (96, 128), seed)
        assert np.array_equal(want_lab, lab), seed
        assert tuple(meta[:2]) == (h, w) and meta[4] in (0, 1) and meta[5] in (0, 1)


@pytest.mark.gpu
def test_kernel_matches_the_host_pipeline_per_frame():
    """Every frame of a ragged clip (different source sizes, flips, HSV gains, an unlabeled frame): the kernel's output against
    resize -> HSV -> mirror -> letterbox in numpy, same float32 arithmetic: at most one grey level apart, almost everywhere equal."""
    from yolox.data.data_augment import clip_preproc_gpu
    rng = np.random.default_rng(1)
    sizes = [(72, 128), (90, 64), (128, 128), (45, 200), (64, 97), (200, 90)]
    input_dim = (96, 128)
    frames, metas, wants = [], [], []
    for i, (h, w) in enumerate(sizes):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        img[: h // 3] = img[0, 0]                                   # flat areas: the grey / zero-saturation corner of HSV
        annos = np.array([[5.0, 6.0, w * 0.8, h * 0.7, 2.0]]) if i != 3 else np.zeros((0, 5))
        want_img, _, meta, _ = _host_and_gpu_records(img, annos, input_dim, 100 + i)
        frames.append(img); metas.append(meta); wants.append(want_img)
    hmax, wmax = max(s[0] for s in sizes), max(s[1] for s in sizes)
    batch = torch.zeros((len(sizes), hmax, wmax, 3), dtype=torch.uint8)
    for i, f in enumerate(frames):
        batch[i, :f.shape[0], :f.shape[1]] = torch.from_numpy(f)
    for dt in (torch.float32, torch.float16):
        out = clip_preproc_gpu(batch.cuda(), torch.from_numpy(np.stack(metas)), input_dim, dt).float().cpu().numpy()
        for i, want in enumerate(wants):
            diff = np.abs(out[i] - want)
            assert diff.max() <= 1.0, (i, float(diff.max()))
            assert (diff > 0).mean() < 0.002, (i, float((diff > 0).mean()))
    assert any(m[4] for m in metas) and any(m[5] for m in metas) and any(m[6] or m[7] or m[8] for m in metas)


@pytest.mark.gpu
def test_prefetcher_runs_the_device_transform(tmp_path):
    from PIL import Image
    from yolox.data import DataPrefetcher, TrainTransformGPU
    from yolox.data.datasets import vid
    rng = np.random.default_rng(0)
    images, anns = [], []
    os.makedirs(tmp_path / "train" / "v0")
    for fidx in range(6):
        name = f"v0/img_{fidx}.png"
        Image.fromarray(rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)).save(tmp_path / "train" / name)
        images.append({"id": fidx + 1, "sid": 0, "fid": fidx, "file_name": name, "width": 80, "height": 60})
        anns.append({"id": fidx + 1, "image_id": fidx + 1, "category_id": 1, "bbox": [10, 8, 30, 20], "area": 600, "iscrowd": 0})
    with open(tmp_path / "ann.json", "w") as fh:
        json.dump({"videos": [{"id": 0}], "images": images, "annotations": anns, "categories": [{"id": 1, "name": "a"}]}, fh)
    ds = vid.OVIS(data_dir=str(tmp_path), img_size=(64, 64), preproc=TrainTransformGPU(max_labels=50), mode="random", lframe=2, gframe=2,
                  name="train", COCO_anno=str(tmp_path / "ann.json"), training=True)
    loader = vid.ovis_get_trans_loader(batch_size=4, data_num_workers=0, dataset=ds, rank=0, world_size=1, seed=3)
    pf = DataPrefetcher(loader)
    imgs, targets, temb = pf.next()
    torch.cuda.synchronize()
    assert imgs.shape == (4, 3, 64, 64) and imgs.is_cuda and imgs.dtype == torch.float16 and targets.shape == (4, 100, 5)
    assert float(imgs.max()) <= 255 and torch.all(imgs[:, :, 48:, :] == 114)          # 60x80 -> 48x64, the rest is the fill


# ---------------------------------------------------------------------------------------------------------------------------
# Hand-derived fixtures: tests/golden/clip_preproc_handmade.npz, built by oracle/make_golden_preproc.py from the
# reference's FORMULAS (data_augment.py:21-31, 240-264) on inputs where they give exact values by hand -- letterbox placement / fill /
# plane order, the ratio-1/2 and ratio-1 resizes of block-constant frames, the mirror, and the HSV jitter on greys and the six saturated
# hues.  The bilinear interpolation at general ratios and OpenCV's HSV rounding for general colours stay "parity unpinned" (no cv2 here).
# ---------------------------------------------------------------------------------------------------------------------------
def _handmade():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip_preproc_handmade.npz"))


def test_host_transform_chain_reproduces_the_handmade_fixture():
    from yolox.data.data_augment import _bgr2hsv, _hsv2bgr, preproc, resize_bilinear
    fx = _handmade()
    H, W = (int(v) for v in fx["input_dim"])
    for i in range(len(fx["frames"])):
        h, w, nh, nw, flip, hsv_on = (int(v) for v in fx["meta"][i, :6])
        g = fx["meta"][i, 6:9]
        img = resize_bilinear(fx["frames"][i, :h, :w].copy(), nh, nw)
        if hsv_on:                                   # augment_hsv with the fixture's gains instead of random ones
            hsv = _bgr2hsv(img)
            hsv[..., 0] = (hsv[..., 0] + g[0]) % 180
            hsv[..., 1] = np.clip(hsv[..., 1] + g[1], 0, 255)
            hsv[..., 2] = np.clip(hsv[..., 2] + g[2], 0, 255)
            img = _hsv2bgr(hsv)
        if flip:
            img = img[:, ::-1]
        out, r = preproc(img, (H, W))
        assert r == 1.0 and np.array_equal(out, fx["want"][i]), i


@pytest.mark.gpu
def test_kernel_reproduces_the_handmade_fixture_exactly():
    from yolox.data.data_augment import clip_preproc_gpu
    fx = _handmade()
    H, W = (int(v) for v in fx["input_dim"])
    for dt in (torch.float32, torch.float16):
        out = clip_preproc_gpu(torch.from_numpy(fx["frames"]).cuda(), torch.from_numpy(fx["meta"]), (H, W), dt).float().cpu().numpy()
        for i in range(len(out)):
            assert np.array_equal(out[i], fx["want"][i]), (i, float(np.abs(out[i] - fx["want"][i]).max()))
