"""Mosaic / random-affine / mixup augmentation of the YOLOX base detector's training and its samplers, on numpy.

Contracts mirrored (reference yolox/data/datasets/mosaicdetection.py:18-36 tile placement, :75-160 mosaic + affine + mixup order,
:162-230 mixup; yolox/data/data_augment.py:151-231 affine matrix, box warp and clip; yolox/data/samplers.py InfiniteSampler /
YoloBatchSampler): a wrapped dataset offers ``pull_item(i) -> (img HxWx3 uint8 BGR, labels [n,5] (x1,y1,x2,y2,cls), info, id)``,
``load_anno(i)`` and ``input_dim``; ``MosaicDetection[i]`` returns what the trainer's collate stacks: (img 3xHxW float32, labels
[max_labels,5] (cls,cx,cy,w,h), info, id).  This image has no OpenCV: the affine warp is an own inverse-mapped bilinear resampler
(border 114) -- geometry as cv2.warpAffine's, interpolation parity "unpinned" (cv2 works in 1/32-pixel fixed point) -- and the resize
is data_augment.resize_bilinear.  Randomness comes from ``random`` / ``numpy.random`` like the reference's (seed both for repeatability).
"""
import math
import random

import numpy as np
import torch
from torch.utils.data import Dataset, Sampler

from .data_augment import resize_bilinear


def _draw(value, center=0.0):
    """One augmentation parameter: a float v means uniform(center - v, center + v), a pair (lo, hi) uniform(lo, hi)."""
    if isinstance(value, (int, float)):
        return random.uniform(center - value, center + value)
    lo, hi = value
    return random.uniform(lo, hi)


def affine_matrix(target_wh, degrees=10.0, translate=0.1, scales=0.1, shear=10.0):
    """2x3 matrix of rotation by a (degrees, about the origin) with isotropic scale s, then shear, then translation by a fraction of
    the target size -- M = [R0 + shy R1 ; R1 + shx R0 | t] with R = s [[cos a, sin a], [-sin a, cos a]] -> (M, s)."""
    tw, th = target_wh
    a = math.radians(_draw(degrees))
    s = _draw(scales, center=1.0)
    if s <= 0.0:
        raise ValueError("affine_matrix: the scale range must be positive")
    rot = np.array([[s * math.cos(a), s * math.sin(a), 0.0], [-s * math.sin(a), s * math.cos(a), 0.0]])
    shx, shy = math.tan(math.radians(_draw(shear))), math.tan(math.radians(_draw(shear)))
    M = np.stack((rot[0] + shy * rot[1], rot[1] + shx * rot[0]))
    M[0, 2], M[1, 2] = _draw(translate) * tw, _draw(translate) * th
    return M, s


def warp_affine(img, M, out_wh, border=114):
    """dst(x, y) = src(M^-1 (x, y)) bilinear, ``border`` outside the source (cv2.warpAffine's geometry for a forward matrix M)."""
    ow, oh = out_wh
    h, w = img.shape[:2]
    A = np.vstack((M, [0.0, 0.0, 1.0]))
    inv = np.linalg.inv(A)[:2]
    xs, ys = np.meshgrid(np.arange(ow, dtype=np.float64), np.arange(oh, dtype=np.float64))
    sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
    sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]
    src = np.full((h + 2, w + 2, img.shape[2]), float(border), dtype=np.float32)      # one border pixel all around
    src[1:-1, 1:-1] = img
    inside = (x0 >= -1) & (x0 < w) & (y0 >= -1) & (y0 < h)
    xi, yi = np.clip(x0 + 1, 0, w), np.clip(y0 + 1, 0, h)
    top = src[yi, xi] * (1 - fx) + src[yi, xi + 1] * fx
    bot = src[yi + 1, xi] * (1 - fx) + src[yi + 1, xi + 1] * fx
    out = np.where(inside[..., None], top * (1 - fy) + bot * fy, float(border))
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def boxes_after_affine(boxes, M, out_wh):
    """Axis-aligned hull of each box's four warped corners, clipped to the canvas; boxes [n,4] (x1,y1,x2,y2)."""
    n = len(boxes)
    if n == 0:
        return boxes
    ow, oh = out_wh
    corners = np.ones((n, 4, 3))
    corners[:, :, 0] = boxes[:, [0, 2, 0, 2]]
    corners[:, :, 1] = boxes[:, [1, 3, 3, 1]]
    warped = corners @ M.T                                             # [n,4,2]
    out = np.concatenate((warped.min(1), warped.max(1)), 1)
    out[:, 0::2] = out[:, 0::2].clip(0, ow)
    out[:, 1::2] = out[:, 1::2].clip(0, oh)
    return out


def random_affine(img, targets, target_wh=(640, 640), degrees=10.0, translate=0.1, scales=0.1, shear=10.0):
    M, _ = affine_matrix(target_wh, degrees, translate, scales, shear)
    img = warp_affine(img, M, target_wh)
    if len(targets) > 0:
        targets[:, :4] = boxes_after_affine(targets[:, :4], M, target_wh)
    return img, targets


def mosaic_tile(index, xc, yc, w, h, canvas_h, canvas_w):
    """Where tile ``index`` (0 top-left, 1 top-right, 2 bottom-left, 3 bottom-right of the centre (xc, yc)) of size w x h lands on the
    2H x 2W canvas and which part of the tile that is: -> ((X1,Y1,X2,Y2) on the canvas, (x1,y1,x2,y2) in the tile)."""
    left, top = index in (0, 2), index in (0, 1)
    X1, X2 = (max(xc - w, 0), xc) if left else (xc, min(xc + w, canvas_w))
    Y1, Y2 = (max(yc - h, 0), yc) if top else (yc, min(yc + h, canvas_h))
    cw, ch = X2 - X1, Y2 - Y1
    x1, x2 = (w - cw, w) if left else (0, min(w, cw))          # the tile's edge that touches the centre stays
    y1, y2 = (h - ch, h) if top else (0, min(h, ch))
    return (X1, Y1, X2, Y2), (x1, y1, x2, y2)


class MosaicDetection(Dataset):
    """Four images around a random centre on a 2H x 2W grey canvas -> random affine down to H x W -> optional mixup with a fifth
    (jittered, flipped) image at 0.5 / 0.5 -> ``preproc``.  ``enable_mosaic = False`` (the last ``no_aug_epochs``) leaves the plain
    ``preproc`` of the wrapped dataset's item."""

    def __init__(self, dataset, img_size, mosaic=True, preproc=None, degrees=10.0, translate=0.1, mosaic_scale=(0.5, 1.5),
                 mixup_scale=(0.5, 1.5), shear=2.0, enable_mixup=True, mosaic_prob=1.0, mixup_prob=1.0):
        self._dataset, self.input_dim, self.preproc = dataset, tuple(img_size), preproc
        self.degrees, self.translate, self.scale, self.shear = degrees, translate, mosaic_scale, shear
        self.mixup_scale, self.enable_mosaic, self.enable_mixup = mixup_scale, mosaic, enable_mixup
        self.mosaic_prob, self.mixup_prob = mosaic_prob, mixup_prob

    def __len__(self):
        return len(self._dataset)

    def _fit(self, img, H, W):
        r = min(H / img.shape[0], W / img.shape[1])
        return resize_bilinear(img, int(img.shape[0] * r), int(img.shape[1] * r)), r

    def __getitem__(self, idx):
        H, W = self.input_dim
        if not (self.enable_mosaic and random.random() < self.mosaic_prob):
            img, label, info, img_id = self._dataset.pull_item(idx)
            img, label = self.preproc(img, label, self.input_dim)
            return img, label, info, img_id
        yc, xc = int(random.uniform(0.5 * H, 1.5 * H)), int(random.uniform(0.5 * W, 1.5 * W))
        picks = [idx] + [random.randint(0, len(self._dataset) - 1) for _ in range(3)]
        canvas = np.full((2 * H, 2 * W, 3), 114, dtype=np.uint8)
        labels, img_id = [], None
        for k, i in enumerate(picks):
            img, lab, _, iid = self._dataset.pull_item(i)
            img_id = iid if k == 0 else img_id
            img, r = self._fit(img, H, W)
            h, w = img.shape[:2]
            (X1, Y1, X2, Y2), (x1, y1, x2, y2) = mosaic_tile(k, xc, yc, w, h, 2 * H, 2 * W)
            canvas[Y1:Y2, X1:X2] = img[y1:y2, x1:x2]
            lab = lab.copy()
            if lab.size:
                lab[:, 0:4:2] = r * lab[:, 0:4:2] + (X1 - x1)
                lab[:, 1:4:2] = r * lab[:, 1:4:2] + (Y1 - y1)
            labels.append(lab)
        labels = np.concatenate(labels, 0)
        if len(labels):
            np.clip(labels[:, 0:4:2], 0, 2 * W, out=labels[:, 0:4:2])
            np.clip(labels[:, 1:4:2], 0, 2 * H, out=labels[:, 1:4:2])
        canvas, labels = random_affine(canvas, labels, (W, H), self.degrees, self.translate, self.scale, self.shear)
        if self.enable_mixup and len(labels) and random.random() < self.mixup_prob:
            canvas, labels = self.mixup(canvas, labels)
        img, padded = self.preproc(canvas, labels, self.input_dim)
        return img, padded, (img.shape[1], img.shape[0]), img_id

    def mixup(self, base_img, base_labels):
        H, W = self.input_dim
        jitter = random.uniform(*self.mixup_scale)
        flip = random.uniform(0, 1) > 0.5
        other = []
        while len(other) == 0:                               # an image that has annotations
            j = random.randint(0, len(self) - 1)
            other = self._dataset.load_anno(j)
        img, other, _, _ = self._dataset.pull_item(j)
        fitted, r = self._fit(img, H, W)
        tile = np.full((H, W, 3), 114, dtype=np.uint8)
        tile[:fitted.shape[0], :fitted.shape[1]] = fitted
        tile = resize_bilinear(tile, int(H * jitter), int(W * jitter))
        r *= jitter
        if flip:
            tile = tile[:, ::-1]
        th, tw = tile.shape[:2]
        bh, bw = base_img.shape[:2]
        pad = np.zeros((max(th, bh), max(tw, bw), 3), dtype=np.uint8)
        pad[:th, :tw] = tile
        oy = random.randint(0, pad.shape[0] - bh - 1) if pad.shape[0] > bh else 0
        ox = random.randint(0, pad.shape[1] - bw - 1) if pad.shape[1] > bw else 0
        crop = pad[oy:oy + bh, ox:ox + bw]
        boxes = other[:, :4].astype(np.float64) * r
        boxes[:, 0::2] = boxes[:, 0::2].clip(0, tw)
        boxes[:, 1::2] = boxes[:, 1::2].clip(0, th)
        if flip:
            boxes[:, 0::2] = tw - boxes[:, 2::-2]
        boxes[:, 0::2] = (boxes[:, 0::2] - ox).clip(0, bw)
        boxes[:, 1::2] = (boxes[:, 1::2] - oy).clip(0, bh)
        merged = np.vstack((base_labels, np.hstack((boxes, other[:, 4:5]))))
        mixed = 0.5 * base_img.astype(np.float32) + 0.5 * crop.astype(np.float32)
        return mixed.astype(np.uint8), merged


class InfiniteSampler(Sampler):
    """Endless index stream: a fresh seeded permutation per pass, rank r takes positions r, r + world, ... of the stream."""

    def __init__(self, size, shuffle=True, seed=0, rank=0, world_size=1):
        assert size > 0
        self.size, self.shuffle, self.seed, self.rank, self.world = size, shuffle, int(seed), rank, world_size

    def _stream(self):
        g = torch.Generator()
        g.manual_seed(self.seed)
        while True:
            yield from (torch.randperm(self.size, generator=g) if self.shuffle else torch.arange(self.size)).tolist()

    def __iter__(self):
        import itertools
        yield from itertools.islice(self._stream(), self.rank, None, self.world)

    def __len__(self):
        return self.size // self.world


class YoloBatchLoader:
    """The trainer's loader contract for the base detector: ``len()`` iterations per epoch, ``next() -> (imgs [B,3,H,W] float,
    targets [B,max_labels,5])``, ``close_mosaic()``.  Batches are assembled in-process (``workers`` = 0) or by a torch DataLoader
    over the endless sampler."""

    def __init__(self, dataset, batch_size, sampler, workers=0, device=None):
        self.dataset, self.batch_size, self.device = dataset, batch_size, device
        self._len = max(1, len(sampler) // batch_size)
        if workers > 0:
            from torch.utils.data import BatchSampler, DataLoader
            self._it = iter(DataLoader(dataset, batch_sampler=BatchSampler(sampler, batch_size, drop_last=False), num_workers=workers,
                                       pin_memory=True))
            self._idx = None
        else:
            self._it, self._idx = None, iter(sampler)

    def __len__(self):
        return self._len

    def close_mosaic(self):
        self.dataset.enable_mosaic = False

    def next(self):
        if self._it is not None:
            imgs, labels = next(self._it)[:2]
        else:
            items = [self.dataset[next(self._idx)] for _ in range(self.batch_size)]
            imgs = torch.from_numpy(np.stack([it[0] for it in items]))
            labels = torch.from_numpy(np.stack([it[1] for it in items]))
        if self.device is not None:
            imgs, labels = imgs.to(self.device, non_blocking=True), labels.to(self.device, non_blocking=True)
        return imgs, labels


class SyntheticDetectionDataset:
    """Seeded stand-in for COCODataset (no data set in this image): uint8 noise images of varying size with a few boxes each, the
    ``pull_item`` / ``load_anno`` / ``input_dim`` contract MosaicDetection wraps."""

    def __init__(self, length=64, num_classes=80, img_size=(640, 640), seed=0):
        self.length, self.num_classes, self.input_dim, self.seed = length, num_classes, tuple(img_size), seed

    def __len__(self):
        return self.length

    def _rng(self, i):
        return np.random.default_rng(self.seed * 7919 + i)

    def load_anno(self, i):
        return self._anno(i)[0]

    def _anno(self, i):
        rng = self._rng(i)
        h, w = int(rng.integers(240, 520)), int(rng.integers(320, 680))
        n = int(rng.integers(1, 7))
        x1, y1 = rng.uniform(0, w * 0.6, n), rng.uniform(0, h * 0.6, n)
        bw, bh = rng.uniform(16, w * 0.4, n), rng.uniform(16, h * 0.4, n)
        cls = rng.integers(0, self.num_classes, n)
        return np.stack((x1, y1, np.minimum(x1 + bw, w - 1), np.minimum(y1 + bh, h - 1), cls), 1).astype(np.float32), (h, w)

    def pull_item(self, i):
        labels, (h, w) = self._anno(i)
        img = self._rng(i + 10 ** 6).integers(0, 256, (h, w, 3), dtype=np.uint8)
        return img, labels, (h, w), i

    def __getitem__(self, i):
        return self.pull_item(i)
