"""The import surface the reference's entry points and exp files rely on, performed against THIS package, and
the clip input pipeline behind it on a tiny OVIS-style dataset written by the test.  CPU only.

Import lists checked (reference file:line):
  tools/tscd_train.py:17-23, tools/tscd_eval.py:18-24       yolox.core.launch, yolox.core.tscd_trainer.Trainer, yolox.exp.get_exp,
                                                             yolox.utils.{configure_nccl,configure_omp,get_num_devices},
                                                             yolox.data.data_augment.{ValTransform,Vid_Val_Transform}, yolox.data.datasets.vid
  exps/TSCD_OVIS/tscd_base.py:10-12,214,321,462,477          yolox.data.datasets.vid, yolox.exp.base_exp.BaseExp, Vid_Val_Transform,
                                                             yolox.models.{YOLOPAFPN}, tscd_head.TSCDHead, tscd.TSCD, yolox.data.TrainTransform,
                                                             yolox.evaluators.ovis_evaluator_v2.OVISEvaluator, yolox.utils.LRScheduler
  exps/TSCD_OVIS/ovis_tscd_large.py:6-9                      exps.TSCD_OVIS.tscd_base.Exp, vid, Vid_Val_Transform
  tools/demo.py:17-22                                        ValTransform, vid_classes.{VID_classes,OVIS_classes}, get_exp,
                                                             yolox.utils.{fuse_model,get_model_info,postprocess,vis}
(loguru and cv2, which those files also import, are third-party packages, not part of this surface.)"""
import json
import os

import numpy as np
import pytest
import torch


def test_reference_entry_point_imports_resolve():
    from yolox.core import launch                                                    # noqa: F401
    from yolox.core.tscd_trainer import Trainer                                      # noqa: F401
    from yolox.exp import get_exp                                                    # noqa: F401
    from yolox.utils import configure_nccl, configure_omp, get_num_devices           # noqa: F401
    from yolox.data.data_augment import ValTransform, Vid_Val_Transform              # noqa: F401
    from yolox.data.datasets import vid
    from yolox.exp.base_exp import BaseExp                                           # noqa: F401
    from yolox.models import YOLOPAFPN, YOLOX, YOLOXHead                             # noqa: F401
    from yolox.models.tscd_head import TSCDHead                                      # noqa: F401
    from yolox.models.tscd import TSCD                                               # noqa: F401
    from yolox.data import TrainTransform                                            # noqa: F401
    from yolox.evaluators.ovis_evaluator_v2 import OVISEvaluator                     # noqa: F401
    from yolox.utils import LRScheduler                                              # noqa: F401
    from yolox.data.datasets.vid_classes import VID_classes, OVIS_classes
    from yolox.utils import fuse_model, get_model_info, postprocess, vis             # noqa: F401
    for name in ("OVIS", "VIDDataset", "ovis_get_trans_loader", "ovis_vid_val_loader", "DataPrefetcher", "get_timing_signal_1d",
                 "collate_fn_ovis", "TrainSampler", "TestSampler", "VIDBatchSampler"):
        assert hasattr(vid, name), name
    assert len(VID_classes) == 30 and len(OVIS_classes) == 25


def test_exp_files_and_tools_load():
    import importlib.util
    from yolox.exp import get_exp
    for f, nc, frames in (("exps/TSCD_OVIS/ovis_tscd_large.py", 25, (4, 12, 8, 24)), ("exps/TSCD_OVIS/ovis_tscd_base.py", 25, None),
                          ("exps/TSCD_VID/vid_tscd_large.py", 30, (4, 12, 1, 31)), ("exps/TSCD_VID/vid_tscd_large_32f.py", 30, (8, 24, 8, 24)),
                          ("exps/default/yolox_s.py", 80, None)):
        exp = get_exp(f)
        assert exp.num_classes == nc, f
        if frames:
            assert (exp.lframe, exp.gframe, exp.lframe_val, exp.gframe_val) == frames, f        # reference: vid_tscd_large.py:27-30
    vid_exp = get_exp("exps/TSCD_VID/vid_tscd_large.py")
    assert vid_exp.maximal_limit == 0 and vid_exp.minimal_limit == 50                           # :42, no upper clamp
    m = get_exp("exps/default/yolox_s.py").get_model()
    assert abs(sum(p.numel() for p in m.parameters()) / 1e6 - 8.97) < 0.05
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for tool in ("tscd_train.py", "tscd_eval.py", "demo.py"):
        spec = importlib.util.spec_from_file_location("tool_" + tool[:-3], os.path.join(root, "tools", tool))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)                                                             # imports resolve, parser builds
        assert hasattr(mod, "make_parser")


@pytest.fixture()
def tiny_ovis(tmp_path):
    """3 videos (7, 5, 2 frames) of 60x80 images, file names <video>/img_<frame>.png, 2 categories."""
    from PIL import Image
    rng = np.random.default_rng(0)
    images, anns, aid = [], [], 1
    for sid, nframes in enumerate((7, 5, 2)):
        os.makedirs(tmp_path / "train" / f"v{sid}", exist_ok=True)
        for fidx in range(nframes):
            name = f"v{sid}/img_{fidx}.png"
            img = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
            Image.fromarray(img).save(tmp_path / "train" / name)
            iid = len(images) + 1
            images.append({"id": iid, "sid": sid, "fid": fidx, "file_name": name, "width": 80, "height": 60})
            if not (sid == 1 and fidx >= 3):                      # a few unlabeled frames
                anns.append({"id": aid, "image_id": iid, "category_id": 1 + (fidx % 2), "bbox": [10 + fidx, 8, 30, 20], "area": 600, "iscrowd": 0})
                aid += 1
    ann = {"videos": [{"id": i} for i in range(3)], "images": images, "annotations": anns,
           "categories": [{"id": 1, "name": "a"}, {"id": 2, "name": "b"}]}
    with open(tmp_path / "ann.json", "w") as fh:
        json.dump(ann, fh)
    return tmp_path


def test_ovis_clip_pipeline_contract(tiny_ovis):
    from yolox.data import TrainTransform
    from yolox.data.data_augment import Vid_Val_Transform
    from yolox.data.datasets import vid
    L, G = 2, 3
    ds = vid.OVIS(data_dir=str(tiny_ovis), img_size=(64, 64), preproc=TrainTransform(max_labels=50), mode="random", lframe=L, gframe=G,
                  name="train", COCO_anno=str(tiny_ovis / "ann.json"), training=True)
    assert len(ds) > 0 and all(len(c) == L + G for c in ds.res)
    for clip in ds.res:                                            # local frames: consecutive frames of one video
        vids = {c.split("/")[0] for c in clip}
        assert len(vids) == 1
        idx = [int(c.rsplit("_", 1)[1].split(".")[0]) for c in clip[:L]]
        assert idx == list(range(idx[0], idx[0] + L)) or len(set(idx)) == 1      # (a 2-frame video is padded by repetition)
    loader = vid.ovis_get_trans_loader(batch_size=L + G, data_num_workers=0, dataset=ds, rank=0, world_size=1, seed=3)
    imgs, targets, infos, raw, paths, temb = next(iter(loader))
    assert imgs.shape == (L + G, 3, 64, 64) and imgs.dtype == torch.float32 and float(imgs.max()) <= 255
    assert targets.shape == (L + G, 100, 5) and temb.shape == (L + G, 256)
    frame_idx = torch.tensor([int(p.rsplit("_", 1)[1].split(".")[0]) for p in paths])
    assert torch.allclose(temb, vid.get_timing_signal_1d(frame_idx, 256))
    lab = targets[0][targets[0].sum(-1) > 0]
    assert lab.shape[0] == 1 and 0 <= lab[0, 0] <= 1                      # (cls, cx, cy, w, h), class index 0/1
    r = min(64 / 60, 64 / 80)
    assert abs(float(lab[0, 3]) - 30 * r) < 1e-3 and abs(float(lab[0, 4]) - 20 * r) < 1e-3
    # letterbox: the image occupies the top-left 48x64 region, the rest is the 114 fill
    assert torch.all(imgs[0][:, int(60 * r):, :] == 114)
    # validation transform keeps corner boxes
    dv = vid.OVIS(data_dir=str(tiny_ovis), img_size=(64, 64), preproc=Vid_Val_Transform(), mode="random", lframe=L, gframe=G, name="train",
                  COCO_anno=str(tiny_ovis / "ann.json"), val=True)
    img, tgt, info, path = dv[dv.res[0][0]]
    assert img.shape == (3, 64, 64) and tgt.shape[1] == 5 and info == (60, 80)


def test_train_sampler_is_rank_aware_and_eval_sharding_is_video_affine(tiny_ovis):
    from yolox.data.data_augment import Vid_Val_Transform
    from yolox.data.datasets import vid
    ds = vid.OVIS(data_dir=str(tiny_ovis), img_size=(64, 64), preproc=Vid_Val_Transform(), mode="random", lframe=1, gframe=1, name="train",
                  COCO_anno=str(tiny_ovis / "ann.json"), val=True)
    n = len(ds.res)
    a, b = vid.TrainSampler(ds, rank=0, world_size=2, seed=1), vid.TrainSampler(ds, rank=1, world_size=2, seed=1)
    ca, cb = [tuple(c) for c in a], [tuple(c) for c in b]
    assert len(ca) == len(cb) == len(a) == len(b) == (n + 1) // 2                # every rank runs the same number of steps per epoch
    assert set(ca) | set(cb) == {tuple(c) for c in ds.res}                      # the epoch's clips are covered ...
    assert len(set(ca) & set(cb)) == 2 * len(ca) - n                            # ... with only the wrap-around padding shared
    ca2 = [tuple(c) for c in a]
    assert ca2 != ca or n < 3                                                   # reshuffled in the next epoch
    # evaluation: clips of one video never leave their rank (CAFM state flows clip -> clip inside a video)
    e0, e1 = vid.TestSampler(ds, 0, 2), vid.TestSampler(ds, 1, 2)
    v0 = {c[0].split("/")[0] for c in e0}
    v1 = {c[0].split("/")[0] for c in e1}
    assert not (v0 & v1) and len(list(e0)) + len(list(e1)) == n
    order = [c[0] for c in e0]
    assert order == [c[0] for c in ds.res if c[0].split("/")[0] in v0]           # dataset order kept inside the shard


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("n", [1, 7, 10, 17])
def test_train_sampler_gives_every_rank_the_same_epoch_length(world, n):
    """ranks with different epoch lengths run different numbers of DDP steps and lr schedules -> mismatched
    collectives.  Like torch's DistributedSampler the permutation is padded by wrapping around to a multiple of the world size."""
    from yolox.data.datasets import vid

    class _DS:
        res = [[f"v{i}/f.jpg"] for i in range(n)]

    samplers = [vid.TrainSampler(_DS(), rank=r, world_size=world, seed=3) for r in range(world)]
    want = -(-n // world)
    got = [[c[0] for c in s] for s in samplers]
    assert all(len(s) == want for s in samplers) and all(len(g) == want for g in got)
    flat = [c for g in got for c in g]
    assert set(flat) == {c[0] for c in _DS.res}                                  # nothing dropped
    assert len(flat) == want * world
    # the same epoch on every rank: interleaving the shards gives ONE permutation (plus its wrapped head)
    inter = [got[i % world][i // world] for i in range(want * world)]
    assert len(set(inter[:n])) == n and inter[n:] == (inter[:n] * world)[:want * world - n]


def test_letterbox_and_transforms():
    from yolox.data.data_augment import TrainTransform, ValTransform, preproc, resize_bilinear
    img = np.zeros((30, 40, 3), np.uint8)
    img[:, 20:] = 200
    out, r = preproc(img, (64, 64))
    assert out.shape == (3, 64, 64) and abs(r - 1.6) < 1e-9
    assert out[0, 0, 0] == 0 and out[0, 0, 63] == 200 and out[0, 47, 5] == 0 and out[0, 48, 5] == 114
    # bilinear resampling: a constant image stays constant, a horizontal ramp stays monotone
    assert np.all(resize_bilinear(np.full((7, 9, 3), 77, np.uint8), 20, 33) == 77)
    ramp = np.tile(np.arange(0, 250, 10, dtype=np.uint8)[None, :, None], (5, 1, 3))
    rr = resize_bilinear(ramp, 5, 60)
    assert np.all(np.diff(rr[0, :, 0].astype(int)) >= 0) and rr[0, 0, 0] == 0 and rr[0, -1, 0] == 240
    v, lab = ValTransform()(img, None, (64, 64))
    assert v.shape == (3, 64, 64) and lab.shape == (1, 5)
    t = TrainTransform(max_labels=7, flip_prob=1.0, hsv_prob=0.0)
    boxes = np.array([[4.0, 5.0, 14.0, 25.0, 1.0]])
    timg, tl = t(img.copy(), boxes, (64, 64))
    assert tl.shape == (7, 5) and tl[0, 0] == 1
    # flipped horizontally: cx' = (W - cx) * r
    assert abs(tl[0, 1] - (40 - 9.0) * 1.6) < 1e-4 and abs(tl[0, 3] - 10 * 1.6) < 1e-4 and abs(tl[0, 4] - 20 * 1.6) < 1e-4


def test_vid_dataset_xml_annotations_and_clip_rules(tmp_path):
    """VIDDataset on an ImageNet-VID-shaped tree written by the test (reference vid.py:48-342): XML boxes clipped, mapped to the class
    index of their WordNet id and scaled by the letterbox ratio; clip rules of photo_to_sequence; the json annotation cache."""
    import random

    from PIL import Image
    from yolox.data import TrainTransform
    from yolox.data.datasets import vid
    from yolox.data.datasets.vid_classes import VID_classes, VID_wnids
    assert len(VID_wnids) == 30 and VID_wnids[VID_classes.index("car")] == "n02958343"
    rng = np.random.default_rng(0)
    videos = []
    for v, nframes in enumerate((9, 5, 3)):
        frames = []
        for f in range(nframes):
            rel = f"Data/VID/train/vid{v}/{f:06d}.JPEG"
            os.makedirs(tmp_path / os.path.dirname(rel), exist_ok=True)
            Image.fromarray(rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)).save(tmp_path / rel, format="JPEG")
            xml = tmp_path / rel.replace("Data", "Annotations").replace("JPEG", "xml")
            os.makedirs(os.path.dirname(xml), exist_ok=True)
            objs = "" if (v == 1 and f < 4) else (
                f"<object><name>{VID_wnids[6]}</name><bndbox><xmax>{90 if f == 0 else 50}</xmax><xmin>10</xmin><ymax>40</ymax><ymin>-3</ymin></bndbox></object>"
                f"<object><name>{VID_wnids[29]}</name><bndbox><xmax>70</xmax><xmin>60</xmin><ymax>55</ymax><ymin>45</ymin></bndbox></object>")
            xml.write_text(f"<annotation><size><width>80</width><height>60</height></size>{objs}</annotation>")
            frames.append(rel)
        videos.append(frames)
    with open(tmp_path / "seq.json", "w") as fh:
        json.dump(videos, fh)
    random.seed(0)
    kw = dict(file_path=str(tmp_path / "seq.json"), img_size=(64, 64), dataset_pth=str(tmp_path), cache_file=str(tmp_path / "cache" / "ann.json"))
    ds = vid.VIDDataset(preproc=TrainTransform(max_labels=50), lframe=2, gframe=2, mode="random", training=True, **kw)
    r = min(64 / 60, 64 / 80)
    a = ds.annotations[videos[0][0]]
    assert a.shape == (2, 5) and np.allclose(a[0], [10 * r, 0, 80 * r, 40 * r, 6]) and a[1, 4] == 29        # clipped to the image
    assert len(ds.annotations[videos[1][0]]) == 0
    assert os.path.isfile(tmp_path / "cache" / "ann.json")
    # video 0: four runs of two consecutive frames; video 1: runs (0,1), (2,3) carry no label and are skipped in training; video 2 too short
    by_video = {}
    for clip in ds.res:
        assert len(clip) == 4 and len({os.path.dirname(c) for c in clip}) == 1
        i0, i1 = (int(os.path.basename(c)[:6]) for c in clip[:2])
        assert i1 == i0 + 1 and i0 % 2 == 0
        assert not {clip[0], clip[1]} & set(clip[2:])                                                  # global frames come from the rest
        by_video.setdefault(os.path.dirname(clip[0]), []).append(i0)
    assert sorted(by_video["Data/VID/train/vid0"]) == [0, 2, 4, 6] and "Data/VID/train/vid1" not in by_video and len(by_video) == 1
    img, target, info, path = ds[ds.res[0][0]]
    assert img.shape == (3, 64, 64) and target.shape == (50, 5) and info == (60, 80)
    imgs, targets, infos, raw, paths, temb = next(iter(vid.ovis_get_trans_loader(batch_size=4, data_num_workers=0, dataset=ds, rank=0, world_size=1)))
    assert imgs.shape == (4, 3, 64, 64) and targets.shape == (4, 100, 5) and temb.shape == (4, 256)
    assert torch.allclose(temb, vid.get_timing_signal_1d(torch.tensor([int(os.path.basename(p)[:6]) for p in paths]), 256))
    # evaluation: every run, the json cache is read back, tnum caps the list
    dv = vid.VIDDataset(preproc=None, lframe=2, gframe=2, mode="random", val=True, tnum=-1, **kw)
    assert len(dv.res) == 4 + 2 and np.array_equal(dv.annotations[videos[0][0]], a)
    assert len(vid.VIDDataset(preproc=None, lframe=2, gframe=2, mode="random", val=True, tnum=3, **kw).res) == 3
    du = vid.VIDDataset(preproc=None, lframe=0, gframe=4, mode="uniform", val=True, tnum=-1, **kw)
    assert du.res[0] == videos[0][0:8:2] and du.res[1] == videos[0][1:8:2]
    # a cache written for another letterbox size is stale: re-parsed (and re-written), never read as it is
    d2 = vid.VIDDataset(preproc=None, lframe=2, gframe=2, mode="random", val=True, tnum=-1, **dict(kw, img_size=(128, 128)))
    assert np.allclose(d2.annotations[videos[0][0]][0, :4], a[0, :4] * 2)
    with open(tmp_path / "cache" / "ann.json") as fh:
        assert json.load(fh)["img_size"] == [128, 128]
    # the reference's object-array .npy list needs unpickling: refused unless asked for; a .json next to it is picked up instead
    arr = np.empty(len(videos), dtype=object)
    for i, v in enumerate(videos):
        arr[i] = v
    os.makedirs(tmp_path / "ref", exist_ok=True)
    np.save(tmp_path / "ref" / "seq.npy", arr, allow_pickle=True)
    kw_npy = dict(kw, file_path=str(tmp_path / "ref" / "seq.npy"), cache_file="")
    with pytest.raises(ValueError, match="allow_pickle"):
        vid.VIDDataset(preproc=None, lframe=2, gframe=2, val=True, tnum=-1, **kw_npy)
    assert len(vid.VIDDataset(preproc=None, lframe=2, gframe=2, val=True, tnum=-1, allow_pickle=True, **kw_npy).res) == 6
    with open(tmp_path / "ref" / "seq.json", "w") as fh:
        json.dump(videos, fh)
    assert len(vid.VIDDataset(preproc=None, lframe=2, gframe=2, val=True, tnum=-1, **kw_npy).res) == 6
