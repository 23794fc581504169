#!/usr/bin/env python3
"""Row N2 measurement: the C++ COCO bbox evaluation (tscd_cocoeval_bbox) against the plain-Python restatement of the protocol
(oracle/cocoeval_ref.py; pycocotools is not installed here) on a synthetic OVIS-sized validation set."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def synth(n_img, n_cat, gt_per_img, dt_per_img, seed=0):
    rng = np.random.default_rng(seed)
    n_gt, n_dt = n_img * gt_per_img, n_img * dt_per_img
    def boxes(n):
        xy = rng.uniform(0, 500, (n, 2)); wh = rng.uniform(8, 200, (n, 2))
        return np.concatenate([xy, wh], 1).astype(np.float32)
    gt_box = boxes(n_gt)
    gt_img = np.repeat(np.arange(n_img), gt_per_img).astype(np.int32)
    gt_cat = rng.integers(0, n_cat, n_gt).astype(np.int32)
    gt_crowd = (rng.random(n_gt) < 0.02).astype(np.uint8)
    gt_area = (gt_box[:, 2] * gt_box[:, 3]).astype(np.float32)
    # detections: jittered copies of ground truth + noise boxes
    src = rng.integers(0, n_gt, n_dt)
    dt_box = gt_box[src] + rng.normal(0, 6, (n_dt, 4)).astype(np.float32)
    dt_box[:, 2:] = np.maximum(dt_box[:, 2:], 2)
    dt_img = gt_img[src].copy()
    dt_cat = np.where(rng.random(n_dt) < 0.8, gt_cat[src], rng.integers(0, n_cat, n_dt)).astype(np.int32)
    dt_score = rng.random(n_dt).astype(np.float32)
    return dict(gt_box=gt_box, gt_img=gt_img, gt_cat=gt_cat, gt_crowd=gt_crowd, gt_area=gt_area, dt_box=dt_box, dt_img=dt_img,
                dt_cat=dt_cat, dt_score=dt_score)


def main():
    from tscd_amd._host import cocoeval_bbox
    n_img, n_cat = int(os.environ.get("N_IMG", 4000)), 25
    d = synth(n_img, n_cat, 5, 40)
    t0 = time.time()
    stats = cocoeval_bbox(d["gt_box"], d["gt_img"], d["gt_cat"], d["gt_crowd"], d["gt_area"], d["dt_box"], d["dt_img"], d["dt_cat"],
                          d["dt_score"], n_img, n_cat)
    t_cpp = time.time() - t0
    print(f"C++ tscd_cocoeval_bbox: {n_img} images, {len(d['gt_img'])} boxes, {len(d['dt_img'])} detections, {n_cat} categories: {t_cpp * 1e3:.0f} ms "
          f"(AP {stats[0]:.4f} AP50 {stats[1]:.4f})", flush=True)
    if os.environ.get("WITH_PY", "1") == "1":
        from oracle import cocoeval_ref
        sub = 400
        ds = synth(sub, n_cat, 5, 40)
        gts = [dict(img=int(i), cat=int(c), bbox=b.tolist(), iscrowd=int(cr), area=float(a)) for i, c, b, cr, a in
               zip(ds["gt_img"], ds["gt_cat"], ds["gt_box"], ds["gt_crowd"], ds["gt_area"])]
        dts = [dict(img=int(i), cat=int(c), bbox=b.tolist(), score=float(s)) for i, c, b, s in
               zip(ds["dt_img"], ds["dt_cat"], ds["dt_box"], ds["dt_score"])]
        t0 = time.time()
        ref_stats = cocoeval_ref.evaluate(gts, dts, sub, n_cat)[0]
        t_py = time.time() - t0
        t0 = time.time()
        s2 = cocoeval_bbox(ds["gt_box"], ds["gt_img"], ds["gt_cat"], ds["gt_crowd"], ds["gt_area"], ds["dt_box"], ds["dt_img"], ds["dt_cat"],
                           ds["dt_score"], sub, n_cat)
        t_c2 = time.time() - t0
        print(f"same {sub}-image subset: Python restatement {t_py * 1e3:.0f} ms, C++ {t_c2 * 1e3:.1f} ms; max |stat diff| {np.abs(np.array(ref_stats) - np.array(s2)).max():.2e}")


main()
