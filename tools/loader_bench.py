#!/usr/bin/env python3
"""Row N1 measurement: clips per second of the OVIS clip pipeline (decode -> letterbox -> HSV / flip -> collate) on the host, per worker
count, on a synthetic 720p video set written to a temp directory; and of the GPU letterbox (uint8 frames uploaded raw)."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def make_set(root, n_vid=4, n_frames=24, hw=(720, 1280)):
    from PIL import Image
    rng = np.random.default_rng(0)
    images, anns, aid = [], [], 1
    for sid in range(n_vid):
        os.makedirs(os.path.join(root, "train", f"v{sid}"), exist_ok=True)
        base = rng.integers(0, 256, (hw[0] // 8, hw[1] // 8, 3), dtype=np.uint8).repeat(8, 0).repeat(8, 1)
        for fidx in range(n_frames):
            name = f"v{sid}/img_{fidx}.jpg"
            Image.fromarray(np.roll(base, fidx * 4, 1)).save(os.path.join(root, "train", name), quality=90)
            iid = len(images) + 1
            images.append({"id": iid, "sid": sid, "fid": fidx, "file_name": name, "width": hw[1], "height": hw[0]})
            for j in range(4):
                anns.append({"id": aid, "image_id": iid, "category_id": 1 + j % 3, "bbox": [50 + 40 * j + fidx, 60 + 30 * j, 200, 150], "area": 30000, "iscrowd": 0})
                aid += 1
    ann = {"videos": [{"id": i} for i in range(n_vid)], "images": images, "annotations": anns, "categories": [{"id": i, "name": str(i)} for i in (1, 2, 3)]}
    with open(os.path.join(root, "ann.json"), "w") as fh:
        json.dump(ann, fh)


def main():
    from yolox.data import TrainTransform
    from yolox.data.datasets import vid
    L, G = 4, 12
    with tempfile.TemporaryDirectory() as root:
        make_set(root)
        from yolox.data import TrainTransformGPU
        for name, tf, workers in [("host transform", TrainTransform, w) for w in (0, 4, 8)] + [("decode only (device transform)", TrainTransformGPU, w) for w in (0, 4, 8)]:
            ds = vid.OVIS(data_dir=root, img_size=(576, 576), preproc=tf(max_labels=50), mode="random", lframe=L, gframe=G, name="train",
                          COCO_anno=os.path.join(root, "ann.json"), training=True)
            loader = vid.ovis_get_trans_loader(batch_size=L + G, data_num_workers=workers, dataset=ds, rank=0, world_size=1, seed=1)
            it = iter(loader)
            next(it)
            n, t0 = 0, time.time()
            while time.time() - t0 < 8.0:
                try:
                    next(it)
                except StopIteration:
                    it = iter(loader)
                    next(it)
                n += 1
            print(f"OVIS pipeline, 16-frame clips of 720p JPEG -> 576^2, {name}, {workers} workers: {n / (time.time() - t0):.2f} clips/s", flush=True)
        if torch.cuda.is_available():
            from yolox.data.data_augment import clip_preproc_gpu, TrainTransformGPU as T
            frames = torch.randint(0, 256, (16, 720, 1280, 3), dtype=torch.uint8).pin_memory()
            tf = T(50)
            metas = torch.from_numpy(np.stack([tf((720, 1280), (324, 576), np.array([[10.0, 10.0, 200.0, 150.0, 1.0]]), (576, 576))[0] for _ in range(16)]))
            for _ in range(3):
                out = clip_preproc_gpu(frames.cuda(non_blocking=True), metas, (576, 576))
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(20):
                out = clip_preproc_gpu(frames.cuda(non_blocking=True), metas, (576, 576))
            torch.cuda.synchronize()
            t_all = (time.time() - t0) / 20
            dev = frames.cuda()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                out = clip_preproc_gpu(dev, metas.cuda(), (576, 576))
            e1.record()
            torch.cuda.synchronize()
            print(f"device transform of a 16-frame 720p clip (resize + HSV + flip + pad): {e0.elapsed_time(e1) / 20 * 1e3:.0f} us kernel, "
                  f"{t_all * 1e3:.2f} ms with the 44 MB upload from pinned memory", flush=True)


main()
