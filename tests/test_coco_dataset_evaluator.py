"""SURVEY.md 8f row N4 remainder: the COCO-format data set and the COCO AP evaluator of the base detector (reference:
yolox/data/datasets/coco.py:36-234, yolox/evaluators/coco_evaluator.py:82-330) on a small COCO-format data set written to disk --
no pycocotools / cv2 here: own JSON index, PIL decode, csrc_host/cocoeval.cpp (pinned against the reference's compiled evaluator in
test_cocoeval_ref_pin.py; here additionally against the Python restatement oracle/cocoeval_ref.py on this data)."""
import json
import os

import numpy as np
import pytest
import torch


def _write_coco(root, n_img=6, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "annotations"))
    os.makedirs(os.path.join(root, "val2017"))
    cats = [{"id": 3, "name": "car"}, {"id": 1, "name": "person"}, {"id": 17, "name": "cat"}]
    images, anns, aid = [], [], 1
    for k in range(n_img):
        h, w = int(rng.integers(60, 120)), int(rng.integers(80, 160))
        iid = 100 + 7 * k
        name = f"{iid:012d}.png"
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, "val2017", name))
        images.append({"id": iid, "file_name": name, "width": w, "height": h})
        for _ in range(int(rng.integers(1, 5))):
            x, y = float(rng.uniform(-5, w * 0.6)), float(rng.uniform(0, h * 0.6))
            bw, bh = float(rng.uniform(4, w * 0.5)), float(rng.uniform(4, h * 0.5))
            anns.append({"id": aid, "image_id": iid, "category_id": int(rng.choice([1, 3, 17])), "bbox": [x, y, bw, bh], "area": bw * bh,
                         "iscrowd": int(rng.random() < 0.15), "segmentation": [[0, 0, 1, 1]]})
            aid += 1
    anns.append({"id": aid, "image_id": images[0]["id"], "category_id": 1, "bbox": [5, 5, 10, 10], "area": 0, "iscrowd": 0})    # area 0: dropped
    with open(os.path.join(root, "annotations", "instances_val2017.json"), "w") as fh:
        json.dump({"images": images, "annotations": anns, "categories": cats, "info": {}, "licenses": []}, fh)
    return images, anns, cats


def test_coco_dataset_items_follow_the_reference_contract(tmp_path):
    from yolox.data import COCODataset, ValTransform
    images, anns, cats = _write_coco(str(tmp_path))
    ds = COCODataset(data_dir=str(tmp_path), json_file="instances_val2017.json", name="val2017", img_size=(64, 96), preproc=None)
    assert len(ds) == len(images) and ds.ids == [im["id"] for im in images]
    assert ds.class_ids == [1, 3, 17] and ds._classes == ("person", "car", "cat")
    for k, im in enumerate(images):
        img, labels, (h, w), img_id = ds.pull_item(k)
        assert (h, w) == (im["height"], im["width"]) and int(img_id[0]) == im["id"]
        r = min(64 / h, 96 / w)
        assert img.shape == (int(h * r), int(w * r), 3) and img.dtype == np.uint8
        # labels: non-crowd, positive area, clipped to the image, scaled by r, class = index into the sorted category ids
        want = []
        for a in anns:
            if a["image_id"] != im["id"] or a["iscrowd"] or a["area"] <= 0:
                continue
            x1, y1 = max(0, a["bbox"][0]), max(0, a["bbox"][1])
            x2, y2 = min(w, x1 + max(0, a["bbox"][2])), min(h, y1 + max(0, a["bbox"][3]))
            if x2 >= x1 and y2 >= y1:
                want.append([x1 * r, y1 * r, x2 * r, y2 * r, [1, 3, 17].index(a["category_id"])])
        assert np.allclose(labels, np.array(want, dtype=np.float64).reshape(-1, 5))
        assert np.array_equal(ds.load_anno(k), labels)
    # the BGR decode: channel 0 of the array is the PNG's blue channel
    from PIL import Image
    rgb = np.asarray(Image.open(os.path.join(str(tmp_path), "val2017", images[0]["file_name"])).convert("RGB"))
    assert np.array_equal(ds.load_image(0), rgb[:, :, ::-1])
    # pre-processing hook and the RAM cache give the same items
    ds2 = COCODataset(data_dir=str(tmp_path), json_file="instances_val2017.json", name="val2017", img_size=(64, 96), preproc=ValTransform(), cache=True)
    img, tgt, info, iid = ds2[1]
    assert img.shape == (3, 64, 96) and tgt.shape == (1, 5) and info == (images[1]["height"], images[1]["width"])
    assert np.array_equal(ds2.pull_item(1)[0], ds.pull_item(1)[0])


def test_mosaic_wraps_the_coco_dataset(tmp_path):
    from yolox.data import COCODataset, TrainTransform
    from yolox.data.mosaic import MosaicDetection
    _write_coco(str(tmp_path))
    ds = COCODataset(data_dir=str(tmp_path), json_file="instances_val2017.json", name="val2017", img_size=(64, 96))
    mo = MosaicDetection(ds, mosaic=True, img_size=(64, 96), preproc=TrainTransform(max_labels=20), enable_mixup=True)
    np.random.seed(0)
    import random
    random.seed(0)
    img, lab, info, _ = mo[2]
    assert img.shape == (3, 64, 96) and lab.shape == (20, 5) and np.isfinite(img).all()


class _Loader(list):
    batch_size = 2


def _evaluator(tmp_path, **kw):
    from yolox.data import COCODataset, ValTransform
    from yolox.evaluators import COCOEvaluator
    images, anns, cats = _write_coco(str(tmp_path))
    ds = COCODataset(data_dir=str(tmp_path), json_file="instances_val2017.json", name="val2017", img_size=(64, 96), preproc=ValTransform())
    loader = _Loader()
    loader.dataset = ds
    return COCOEvaluator(loader, (64, 96), 0.01, 0.65, 3, **kw), ds, images, anns


def test_coco_evaluator_on_perfect_and_noisy_detections(tmp_path):
    """Detections equal to the (non-crowd) ground truth, handed over in the model's output format at the network's input scale: AP = AR =
    1 where defined; a noisy set is evaluated to the same numbers as the Python restatement of COCOeval (oracle/cocoeval_ref.py)."""
    ev, ds, images, anns = _evaluator(tmp_path, per_class_AP=True, per_class_AR=True)
    rng = np.random.default_rng(1)

    def outputs_for(noise):
        outs, hs, ws, ids = [], [], [], []
        for im in images:
            h, w = im["height"], im["width"]
            scale = min(64 / h, 96 / w)
            rows = []
            for a in anns:
                if a["image_id"] != im["id"] or a["iscrowd"]:
                    continue                                   # (a copy of a crowd box would claim a regular ground truth it overlaps;
                                                               # the zero-area annotation IS ground truth for the evaluator: it gets its box)
                x, y, bw, bh = a["bbox"]
                jit = noise * rng.standard_normal(4) * np.array([bw, bh, bw, bh])
                box = np.array([x, y, x + bw, y + bh]) + jit
                rows.append(list(box * scale) + [float(rng.uniform(0.5, 1.0)), float(rng.uniform(0.5, 1.0)), [1, 3, 17].index(a["category_id"])])
            outs.append(torch.tensor(rows, dtype=torch.float32) if rows else None)
            hs.append(h); ws.append(w); ids.append(im["id"])
        return outs, (hs, ws), ids

    outs, info, ids = outputs_for(0.0)
    data = ev.convert_to_coco_format(outs, info, ids)
    assert all(d["category_id"] in (1, 3, 17) for d in data)
    d0 = data[0]
    a0 = [a for a in anns if a["image_id"] == d0["image_id"] and not a["iscrowd"]][0]
    assert np.allclose(d0["bbox"], a0["bbox"], atol=1e-3)                       # back in original pixels, xywh
    ap, ap50, summary = ev.evaluate_prediction(data, [0.1, 0.05, 3])
    assert ap == pytest.approx(1.0, abs=1e-6) and ap50 == pytest.approx(1.0, abs=1e-6)
    assert "per class AP" in summary and "person" in summary and "Average forward time" in summary
    # noisy detections: equal to the oracle's COCOeval restatement
    from oracle import cocoeval_ref
    outs, info, ids = outputs_for(0.15)
    data = ev.convert_to_coco_format(outs, info, ids)
    ap, ap50, _ = ev.evaluate_prediction(data, [0.1, 0.05, 3])
    img_of = {i: n for n, i in enumerate(ds.ids)}
    cat_of = {c: n for n, c in enumerate(ds.class_ids)}
    gts = [a for i in ds.ids for a in ds.coco.anns(i)]
    ref = cocoeval_ref.evaluate([dict(img=img_of[a["image_id"]], cat=cat_of[a["category_id"]], bbox=[float(np.float32(v)) for v in a["bbox"]],
                                      iscrowd=int(a.get("iscrowd", 0)), area=float(np.float32(a["area"]))) for a in gts],
                                [dict(img=img_of[d["image_id"]], cat=cat_of[d["category_id"]], bbox=[float(np.float32(v)) for v in d["bbox"]],
                                      score=float(np.float32(d["score"]))) for d in data], len(img_of), len(cat_of))
    stats = ref[0] if isinstance(ref, tuple) else ref
    assert ap == pytest.approx(float(stats[0]), abs=1e-6) and ap50 == pytest.approx(float(stats[1]), abs=1e-6)
    assert 0.0 < ap < 1.0


def test_exp_builds_the_coco_loader_and_evaluator(tmp_path):
    from yolox.exp.yolox_base import Exp
    _write_coco(str(tmp_path))
    os.rename(os.path.join(str(tmp_path), "val2017"), os.path.join(str(tmp_path), "train2017"))
    os.symlink(os.path.join(str(tmp_path), "train2017"), os.path.join(str(tmp_path), "val2017"))
    os.link(os.path.join(str(tmp_path), "annotations", "instances_val2017.json"), os.path.join(str(tmp_path), "annotations", "instances_train2017.json"))
    exp = Exp()
    exp.data_dir, exp.input_size, exp.test_size, exp.num_classes, exp.data_num_workers, exp.seed = str(tmp_path), (64, 96), (64, 96), 3, 0, 0
    loader = exp.get_data_loader(batch_size=2, is_distributed=False)
    imgs, targets = loader.next()[:2]
    assert tuple(imgs.shape) == (2, 3, 64, 96) and tuple(targets.shape) == (2, 120, 5)
    ev = exp.get_evaluator(batch_size=2, is_distributed=False)
    assert type(ev).__name__ == "COCOEvaluator" and len(ev.dataloader.dataset) == 6
    imgs, _, info, ids = next(iter(ev.dataloader))
    assert tuple(imgs.shape) == (2, 3, 64, 96) and len(info) == 2 and len(ids) == 2


@pytest.mark.gpu
def test_base_trainer_trains_and_evaluates_on_a_coco_format_data_set(tmp_path):
    """tools/train.py's loop with ``data_dir`` set: COCODataset -> MosaicDetection -> the HIP train step, and at the end of each epoch
    COCOEvaluator (model.eval() -> postprocess on the HIP NMS -> csrc_host/cocoeval.cpp) -- AP numbers in [0, 1], a ``best`` checkpoint
    contract as the reference's trainer keeps it (trainer.py:317-342)."""
    import argparse
    import random
    from yolox.exp.yolox_base import Exp
    root = str(tmp_path / "coco")
    os.makedirs(root)
    _write_coco(root, n_img=8)
    os.symlink(os.path.join(root, "val2017"), os.path.join(root, "train2017"))
    os.link(os.path.join(root, "annotations", "instances_val2017.json"), os.path.join(root, "annotations", "instances_train2017.json"))
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    exp = Exp()
    exp.depth, exp.width, exp.num_classes = 0.33, 0.125, 3
    exp.data_dir, exp.data_num_workers = root, 0
    exp.input_size, exp.test_size, exp.multiscale_range = (96, 128), (96, 128), 0
    exp.max_epoch, exp.no_aug_epochs, exp.warmup_epochs, exp.basic_lr_per_img, exp.eval_interval = 2, 1, 1, 1e-4 / 4, 1
    exp.output_dir, exp.seed = str(tmp_path / "out"), 0
    args = argparse.Namespace(batch_size=4, fp16=True, experiment_name="tiny_coco", resume=False, ckpt=None, cache=False, async_ckpt=False)
    tr = exp.get_trainer(args)
    tr.train()
    assert type(tr.evaluator).__name__ == "COCOEvaluator"
    assert 0.0 <= tr.best_ap <= 1.0
    assert all(np.isfinite(r["total_loss"]) for r in tr.history_floats())
    assert os.path.isfile(os.path.join(str(tmp_path / "out"), "tiny_coco", "last_epoch_ckpt.pth"))
    # the evaluator on the trained model, directly: a summary with the twelve COCO lines
    ap, ap50, summary = tr.evaluator.evaluate(tr.model, False, half=True)
    assert 0.0 <= ap <= ap50 <= 1.0 and (summary.count("Average Precision") + summary.count("Average Recall") in (0, 12))

