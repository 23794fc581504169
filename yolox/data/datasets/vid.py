"""Clip datasets and loaders of TSCD (interface of the reference's yolox/data/datasets/vid.py: ``OVIS`` :560-717 on top of
``Arg_VID`` :345-557, the samplers :758-830, ``collate_fn_ovis`` :917-937, the loader factories :940-960,
``DataPrefetcher`` :963-1012, ``get_timing_signal_1d`` :1015-1023).

One training iteration = one clip = ``lframe`` consecutive local frames + ``gframe`` frames sampled from the rest of the
video; the batch is the clip (tensor contract: imgs [B,3,H,W] float 0..255 letterboxed, targets [B,100,5]
(cls,cx,cy,w,h; zero rows = padding), img infos, raw targets, paths, time embedding [B,256] of the frame index parsed from
``..._<idx>.jpg``).  On-disk format: the "vid"-style COCO json the reference reads (``videos``, ``images`` with ``sid`` /
``file_name``, ``annotations``, ``categories``), read with a small own index (pycocotools is not in this image) and decoded
with Pillow (OpenCV is not either).

New relative to the reference: the train sampler is RANK-AWARE -- every rank shuffles the clip list with
the same (seed, epoch) generator and takes clips rank, rank + world, ... -- the reference shuffles the full list identically
on every rank, so its multi-GPU runs would train every rank on the same clips."""
import copy
import json
import math
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset as torchDataset
from torch.utils.data.sampler import BatchSampler, Sampler, SequentialSampler

IMAGE_EXT = [".jpg", ".jpeg", ".webp", ".bmp", ".png"]


def get_timing_signal_1d(index_squence, channels, min_timescale=1.0, max_timescale=1.0e4):
    """Sinusoidal embedding of the frame indices: [n] -> [n, channels] (sin | cos halves)."""
    n = channels // 2
    inc = torch.tensor(math.log(max_timescale / min_timescale) / (n - 1))
    inv = min_timescale * torch.exp(torch.arange(0, n) * -inc)
    st = torch.unsqueeze(index_squence, 1) * torch.unsqueeze(inv, 0)
    return torch.cat([torch.sin(st), torch.cos(st)], dim=1)


def imread_bgr(path):
    """Decode an image file to HxWx3 uint8 in BGR channel order (what cv2.imread returns)."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


class CocoIndex:
    """The few lookups the datasets need from a COCO-style annotation file."""

    def __init__(self, path):
        with open(path, "r") as fh:
            self.dataset = json.load(fh)
        self.imgs = {im["id"]: im for im in self.dataset.get("images", [])}
        self.cats = {c["id"]: c for c in self.dataset.get("categories", [])}
        self.anns_of = {}
        for a in self.dataset.get("annotations", []):
            self.anns_of.setdefault(a["image_id"], []).append(a)

    def img_ids(self):
        return sorted(self.imgs)

    def cat_ids(self):
        return sorted(self.cats)

    def anns(self, img_id, iscrowd=False):
        return [a for a in self.anns_of.get(img_id, []) if bool(a.get("iscrowd", 0)) == iscrowd]


class OVIS(torchDataset):
    """OVIS in "vid" COCO format.  ``self.res`` = list of clips (each a list of lframe + gframe file names);
    ``dataset[file_name]`` -> (img, target, (h, w), file_name) after ``preproc``."""

    def __init__(self, data_dir="", img_size=(576, 576), preproc=None, lframe=0, gframe=16, val=False, mode="random", COCO_anno="",
                 name="train", training=False):
        super().__init__()
        if not os.path.isfile(COCO_anno):
            raise FileNotFoundError(f"OVIS annotation file {COCO_anno!r} not found (set exp.data_dir / train_ann / val_ann)")
        self.input_dim = self.img_size = img_size
        self.name, self.val, self.data_dir, self.coco_anno_path = name, val, data_dir, COCO_anno
        self.mode, self.preproc, self.training = mode, preproc, training
        self.coco = CocoIndex(COCO_anno)
        self.name_id_dic = {im["file_name"]: im["id"] for im in self.coco.dataset["images"]}
        self.ids = self.coco.img_ids()
        self.class_ids = self.coco.cat_ids()
        self._classes = tuple(self.coco.cats[c]["name"] for c in self.class_ids)
        self._pos = {i: n for n, i in enumerate(self.ids)}
        self.annotations = [self.load_anno_from_ids(i) for i in self.ids]
        self.res = self.photo_to_sequence(lframe, gframe)

    def __len__(self):
        return len(self.res)

    def load_anno_from_ids(self, id_):
        im = self.coco.imgs[id_]
        width, height = im["width"], im["height"]
        objs = []
        for obj in self.coco.anns(id_):
            x1, y1 = max(0, obj["bbox"][0]), max(0, obj["bbox"][1])
            x2 = min(width, x1 + max(0, obj["bbox"][2]))
            y2 = min(height, y1 + max(0, obj["bbox"][3]))
            if obj.get("area", 1) > 0 and x2 >= x1 and y2 >= y1:
                objs.append(([x1, y1, x2, y2], self.class_ids.index(obj["category_id"])))
        res = np.zeros((len(objs), 5))
        for ix, (box, cls) in enumerate(objs):
            res[ix, :4], res[ix, 4] = box, cls
        r = min(self.img_size[0] / height, self.img_size[1] / width)
        res[:, :4] *= r
        file_name = im["file_name"] if "file_name" in im else "{:012}".format(id_) + ".jpg"
        return res, (height, width), (int(height * r), int(width * r)), file_name

    def _has_labels(self, file_name):
        return len(self.annotations[self._pos[self.name_id_dic[file_name]]][0]) > 0

    def photo_to_sequence(self, lframe, gframe):
        """Clip sampling (vid.py:601-685).  'random' with lframe > 0: every run of `lframe` consecutive frames of a video is
        one clip, completed by `gframe` frames drawn from the rest of that video; training skips clips whose local frames
        carry no label."""
        nvid = len(self.coco.dataset.get("videos", [])) or 1 + max(im["sid"] for im in self.coco.dataset["images"])
        videos = [[] for _ in range(nvid)]
        for im in self.coco.dataset["images"]:
            videos[im["sid"]].append(im["file_name"])
        res = []
        for element in videos:
            if not element:
                continue
            while len(element) < lframe + gframe:              # short videos: repeat the last frame
                element.append(element[-1])
            n = len(element)
            if self.mode == "random":
                if lframe == 0:
                    random.shuffle(element)
                    res += [element[i * gframe:(i + 1) * gframe] for i in range(n // gframe)]
                else:
                    split = n // lframe
                    local_all = element[:split * lframe]
                    for i in np.arange(split) * lframe:
                        l_frame = local_all[i:i + lframe]
                        if self.training and not any(self._has_labels(f) for f in l_frame):
                            continue
                        res.append(l_frame + random.sample(element[:i] + element[i + lframe:], gframe))
            elif self.mode == "uniform":
                split = n // gframe
                uni = element[:split * gframe]
                res += [uni[i::split] for i in range(split)]
            elif self.mode == "gl":
                split = n // lframe
                local_all = element[:split * lframe]
                for i in range(split):
                    g_frame = random.sample(element[:i * lframe] + element[(i + 1) * lframe:], gframe)
                    res.append(local_all[i * lframe:(i + 1) * lframe] + g_frame)
            else:
                raise ValueError(f"unsupported clip sampling mode {self.mode!r}")
        if not self.val:
            random.shuffle(res)
        return res

    def pull_item(self, path):
        annos, _, _, img_path = self.annotations[self._pos[self.name_id_dic[path]]]
        from ..data_augment import resize_bilinear
        img = imread_bgr(os.path.join(self.data_dir, self.name, img_path))
        h, w = img.shape[:2]
        r = min(self.img_size[0] / h, self.img_size[1] / w)
        return resize_bilinear(img, int(h * r), int(w * r)), annos.copy(), (h, w), img_path

    def __getitem__(self, path):
        from ..data_augment import TrainTransformGPU
        if isinstance(self.preproc, TrainTransformGPU):
            return self.pull_raw(path)
        img, target, img_info, path = self.pull_item(path)
        if self.preproc is not None:
            img, target = self.preproc(img, target, self.input_dim)
        return img, target, img_info, path

    def pull_raw(self, path):
        """Device-side pre-processing (row N1): the decoded frame as it is (uint8 HWC BGR), the transformed labels and the kernel's
        per-frame record.  A frame whose letterbox needs the second resample of the host pipeline (pull_item resizes to (nh, nw),
        preproc then maps that onto int(nh * r'), int(nw * r') with r' = min(H / nh, W / nw) -- the identity except when a float
        truncation left nw one short of W) is transformed on the host and handed over as a finished H x W frame."""
        annos, _, _, img_path = self.annotations[self._pos[self.name_id_dic[path]]]
        return _pull_raw(self, os.path.join(self.data_dir, self.name, img_path), annos, img_path)


def _pull_raw(ds, file, annos, ret_path):
    """Shared by the datasets: decoded frame + host half of the device-side transform (see OVIS.pull_raw)."""
    from ..data_augment import TrainTransform, resize_bilinear, val_meta
    img = imread_bgr(file)
    h, w = img.shape[:2]
    H, W = ds.input_dim
    r = min(ds.img_size[0] / h, ds.img_size[1] / w)
    nh, nw = int(h * r), int(w * r)
    r2 = min(H / nh, W / nw)
    if (int(nh * r2), int(nw * r2)) != (nh, nw):
        host = TrainTransform(ds.preproc.max_labels, ds.preproc.flip_prob, ds.preproc.hsv_prob)
        chw, target = host(resize_bilinear(img, nh, nw), annos.copy(), ds.input_dim)
        done = np.ascontiguousarray(chw.transpose(1, 2, 0).astype(np.uint8))
        return done, target, (h, w), ret_path, val_meta((H, W), (H, W))
    meta, target = ds.preproc((h, w), (nh, nw), annos.copy(), ds.input_dim)
    return img, target, (h, w), ret_path, meta


class VIDDataset(torchDataset):
    """ImageNet-VID sequences (reference: vid.py:48-342).  ``file_path``: a list of videos, each a list of frame paths relative to
    ``dataset_pth`` (``Data/VID/.../000000.JPEG``) -- the same list as ``.json`` (the documented format; a ``.json`` next to a ``.npy``
    path is picked up by itself) or the reference's ``.npy`` object array, which can only be read by unpickling and therefore needs
    ``allow_pickle=True`` (or a one-off ``tools/convert_seq_npy.py``).  Annotations are the dataset's XML files
    (path with ``Data`` -> ``Annotations`` and ``JPEG`` -> ``xml``): boxes clipped to the image, class = index of the WordNet id, scaled by
    the letterbox ratio like the reference; parsed once and kept in ``cache_file`` (json).  ``self.res`` = the clips of
    ``photo_to_sequence``; ``dataset[frame path]`` -> (img, target, (h, w), path) after ``preproc``, or the undecorated frame + kernel
    record for a ``TrainTransformGPU``."""

    def __init__(self, file_path="train_seq.npy", img_size=(416, 416), preproc=None, lframe=18, gframe=6, val=False, mode="random",
                 dataset_pth="", tnum=1000, formal=False, traj_linking=False, local_stride=1, training=False, cache_file="",
                 allow_pickle=False):
        super().__init__()
        self.tnum, self.traj_linking, self.input_dim, self.img_size = tnum, traj_linking, img_size, img_size
        self.file_path, self.mode, self.preproc, self.val, self.formal = file_path, mode, preproc, val, formal
        self.local_stride, self.training, self.dataset_pth, self.cache_file = local_stride, training, dataset_pth, cache_file
        if not file_path.endswith(".json") and os.path.isfile(os.path.splitext(file_path)[0] + ".json"):
            file_path = os.path.splitext(file_path)[0] + ".json"          # the converted list next to the reference's .npy wins
        if file_path.endswith(".json"):
            with open(file_path) as fh:
                videos = json.load(fh)
        else:
            # the reference's object-array .npy can only be read by unpickling, which executes what the file says: opt-in only
            # (tools/convert_seq_npy.py turns such a file into .json once)
            if not allow_pickle:
                raise ValueError(f"{file_path}: an object-array .npy sequence list needs unpickling; convert it to .json "
                                 "(tools/convert_seq_npy.py) or pass allow_pickle=True for a file you trust")
            videos = np.load(file_path, allow_pickle=True).tolist()
        videos = [list(v) for v in videos]
        self.annotations = self._load_annotations(videos)
        self.res = self.photo_to_sequence(videos, lframe, gframe)

    def __len__(self):
        return len(self.res)

    # ---- annotations
    def _load_annotations(self, videos):
        """Parsed annotations, cached as json keyed by the letterbox size they were scaled for (a cache written for another
        ``img_size`` is stale and ignored).  Only rank 0 writes the cache, through a temporary file + ``os.replace``."""
        size = [int(self.img_size[0]), int(self.img_size[1])]
        if self.cache_file and os.path.isfile(self.cache_file):
            with open(self.cache_file) as fh:
                blob = json.load(fh)
            if isinstance(blob, dict) and blob.get("img_size") == size and isinstance(blob.get("annotations"), dict):
                return {k: np.asarray(v, dtype=np.float64).reshape(-1, 5) for k, v in blob["annotations"].items()}
        ann = {frame: self.get_annotation(os.path.join(self.dataset_pth, frame), self.img_size) for video in videos for frame in video}
        rank = 0
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                rank = dist.get_rank()
        except Exception:
            pass
        if self.cache_file and rank == 0:
            os.makedirs(os.path.dirname(os.path.abspath(self.cache_file)), exist_ok=True)
            tmp = f"{self.cache_file}.tmp.{os.getpid()}"
            with open(tmp, "w") as fh:
                json.dump({"img_size": size, "annotations": {k: v.tolist() for k, v in ann.items()}}, fh)
            os.replace(tmp, self.cache_file)
        return ann

    @staticmethod
    def get_annotation(image_path, test_size):
        """[n,5] (x1, y1, x2, y2, class) of one frame, scaled by r = min(H / height, W / width) (vid.py:243-285)."""
        import xml.etree.ElementTree as ET
        from .vid_classes import VID_wnids
        root = ET.parse(image_path.replace("Data", "Annotations").replace("JPEG", "xml")).getroot()
        width, height = int(root.find("size/width").text), int(root.find("size/height").text)
        rows = []
        for obj in root.iter("object"):
            bb = obj.find("bndbox")
            x1, y1 = max(0, int(bb.find("xmin").text)), max(0, int(bb.find("ymin").text))
            x2, y2 = min(width, int(bb.find("xmax").text)), min(height, int(bb.find("ymax").text))
            if x2 >= x1 and y2 >= y1:
                rows.append((x1, y1, x2, y2, VID_wnids.index(obj.find("name").text)))
        res = np.zeros((len(rows), 5))
        if rows:
            res[:] = rows
        res[:, :4] *= min(test_size[0] / height, test_size[1] / width)
        return res

    # ---- clips
    def photo_to_sequence(self, videos, lframe, gframe):
        """vid.py:133-241.  Videos shorter than a clip are skipped (``formal``: kept, padded by sampling with replacement); 'random' with
        local frames: consecutive runs of ``lframe`` frames (in training at most 15 per video, evenly spaced, and only runs that carry a
        label) completed by ``gframe`` frames drawn from the rest of the video; ``local_stride`` > 1 interleaves; 'uniform': every
        split-th frame.  Training keeps at most 15000 shuffled clips, evaluation the first ``tnum`` (-1: all)."""
        res = []
        for element in videos:
            n = len(element)
            if n < lframe + gframe:
                if not self.formal:
                    continue
                if lframe == 0:
                    res.append(element)
                else:
                    for i in np.arange(n // lframe) * lframe:
                        res.append(element[i:i + lframe] + random.choices(element[:i] + element[i + lframe:], k=gframe))
                continue
            if self.mode == "uniform":
                split = n // gframe
                uni = element[:split * gframe]
                res += [uni[i::split] for i in range(split)]
                continue
            if self.mode != "random":
                raise ValueError(f"unsupported clip sampling mode {self.mode!r}")
            if lframe == 0:
                split = n // gframe
                random.shuffle(element)
                res += [element[i * gframe:(i + 1) * gframe] for i in range(split)]
                if self.formal and element[split * gframe:]:
                    res.append(element[split * gframe:])
            elif self.local_stride == 1:
                split = n // lframe
                local_all = element[:split * lframe]
                if self.training and split > 15:
                    starts = np.arange(15) * (len(local_all) // 15)
                else:
                    starts = np.arange(split) * lframe
                for i in starts:
                    if self.traj_linking and i != 0:
                        l_frame = local_all[i - 1:i + lframe]
                    else:
                        l_frame = local_all[i:i + lframe]
                        if self.training and not any(len(self.annotations[f]) for f in l_frame):
                            continue
                    res.append(l_frame + random.sample(element[:i] + element[i + lframe:], gframe))
                if self.formal and element[split * lframe:]:
                    res.append(element[split * lframe - 1:] if self.traj_linking else element[split * lframe:])
            else:
                span = lframe * self.local_stride
                for i in range(n // span):
                    res += [element[span * i:span * (i + 1)][j::self.local_stride] for j in range(self.local_stride)]
        if self.val:
            return res if self.tnum == -1 else res[:self.tnum]
        random.shuffle(res)
        return res[:15000]

    # ---- frames
    def pull_item(self, path):
        from ..data_augment import resize_bilinear
        img = imread_bgr(os.path.join(self.dataset_pth, path))
        h, w = img.shape[:2]
        r = min(self.img_size[0] / h, self.img_size[1] / w)
        return resize_bilinear(img, int(h * r), int(w * r)), self.annotations[path].copy(), (h, w), os.path.join(self.dataset_pth, path)

    def __getitem__(self, path):
        from ..data_augment import TrainTransformGPU
        if isinstance(self.preproc, TrainTransformGPU):
            return _pull_raw(self, os.path.join(self.dataset_pth, path), self.annotations[path], os.path.join(self.dataset_pth, path))
        img, target, img_info, path = self.pull_item(path)
        if self.preproc is not None:
            img, target = self.preproc(img, target, self.input_dim)
        return img, target, img_info, path


class TrainSampler(Sampler):
    """Yields clips (lists of file names).  Rank-aware: all ranks draw the same permutation from (seed, epoch), pad it by wrapping
    around to a multiple of the world size (as torch's DistributedSampler does) and keep every world-th clip starting at their rank --
    every rank therefore sees the SAME number of clips per epoch (equal DDP step counts, equal lr schedules, matched collectives);
    call ``set_epoch`` once per epoch."""

    def __init__(self, data_source, rank=None, world_size=None, seed=0):
        self.data_source = data_source
        if rank is None or world_size is None:
            import torch.distributed as dist
            on = dist.is_available() and dist.is_initialized()
            rank, world_size = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
        self.rank, self.world, self.seed, self.epoch = rank, world_size, seed, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _mine(self):
        n = len(self.data_source.res)
        if n == 0:
            return []
        g = torch.Generator().manual_seed(self.seed * 1000003 + self.epoch)
        order = torch.randperm(n, generator=g).tolist()
        total = len(self) * self.world
        while len(order) < total:                       # wrap around (also covers n < world)
            order += order[:total - len(order)]
        return [self.data_source.res[i] for i in order[self.rank:total:self.world]]

    def __iter__(self):
        self.epoch += 1
        return iter(self._mine())

    def __len__(self):
        return (len(self.data_source.res) + self.world - 1) // self.world


class TestSampler(SequentialSampler):
    """Clips in dataset order (the evaluator carries CAFM state from clip to clip of a video).  With ``world_size`` > 1 the
    clips are sharded by VIDEO, never inside one: rank r gets the videos r, r + world, ..."""

    def __init__(self, data_source, rank=0, world_size=1):
        super().__init__(data_source)
        self.data_source, self.rank, self.world = data_source, rank, world_size

    def _mine(self):
        if self.world == 1:
            return list(self.data_source.res)
        vid_of = lambda clip: os.path.dirname(clip[0])
        vids = sorted({vid_of(c) for c in self.data_source.res})
        mine = set(vids[self.rank::self.world])
        return [c for c in self.data_source.res if vid_of(c) in mine]

    def __iter__(self):
        return iter(self._mine())

    def __len__(self):
        return len(self._mine())


class VIDBatchSampler(BatchSampler):
    """Flattens the sampler's clips into batches of `batch_size` file names (batch_size = lframe + gframe: one clip)."""

    def __iter__(self):
        batch = []
        for clip in self.sampler:
            for filename in clip:
                batch.append(filename)
                if len(batch) == self.batch_size:
                    yield batch
                    batch = []
        if batch and not self.drop_last:
            yield batch

    def __len__(self):
        return len(self.sampler)


class VIDBatchSampler_Test(VIDBatchSampler):
    pass


def collate_fn_ovis(batch):
    tar, imgs, ims_info, tar_ori, path, seq = [], [], [], [], [], []
    for img, target, info, p in batch:
        t = torch.zeros([100, 5])
        imgs.append(torch.as_tensor(img))
        tar_ori.append(torch.as_tensor(copy.deepcopy(target)))
        t[:target.shape[0]] = torch.as_tensor(target)
        tar.append(t)
        ims_info.append(info)
        path.append(p)
        seq.append(int(p[p.rfind("/") + 1:p.rfind(".")].split("_")[-1]))
    return torch.stack(imgs), torch.stack(tar), ims_info, tar_ori, path, get_timing_signal_1d(torch.tensor(seq), 256)


def collate_fn_ovis_raw(batch):
    """Clips of undecorated frames for the device-side transform: (frames uint8 [B,hmax,wmax,3], targets, infos, targets as lists, paths,
    time embedding, per-frame kernel records int32 [B,12])."""
    hmax, wmax = max(b[0].shape[0] for b in batch), max(b[0].shape[1] for b in batch)
    frames = torch.zeros((len(batch), hmax, wmax, 3), dtype=torch.uint8)
    tar, ims_info, tar_ori, path, seq, metas = [], [], [], [], [], []
    for i, (img, target, info, p, meta) in enumerate(batch):
        frames[i, :img.shape[0], :img.shape[1]] = torch.from_numpy(img)
        t = torch.zeros([100, 5])
        tar_ori.append(torch.as_tensor(copy.deepcopy(target)))
        t[:target.shape[0]] = torch.as_tensor(target)
        tar.append(t)
        ims_info.append(info)
        path.append(p)
        metas.append(torch.from_numpy(meta))
        seq.append(int(p[p.rfind("/") + 1:p.rfind(".")].split("_")[-1]))
    return frames, torch.stack(tar), ims_info, tar_ori, path, get_timing_signal_1d(torch.tensor(seq), 256), torch.stack(metas)


def _loader(dataset, sampler, data_num_workers):
    from ..data_augment import TrainTransformGPU
    raw = isinstance(getattr(dataset, "preproc", None), TrainTransformGPU)
    return torch.utils.data.DataLoader(dataset, num_workers=data_num_workers, pin_memory=torch.cuda.is_available(),
                                       batch_sampler=sampler, collate_fn=collate_fn_ovis_raw if raw else collate_fn_ovis)


def ovis_get_trans_loader(batch_size, data_num_workers, dataset, rank=None, world_size=None, seed=0):
    return _loader(dataset, VIDBatchSampler(TrainSampler(dataset, rank, world_size, seed), batch_size, drop_last=False), data_num_workers)


def ovis_vid_val_loader(batch_size, data_num_workers, dataset, rank=0, world_size=1):
    return _loader(dataset, VIDBatchSampler_Test(TestSampler(dataset, rank, world_size), batch_size, drop_last=False), data_num_workers)


class DataPrefetcher:
    """Double buffering of the loader's batches: the next clip's H2D copy runs on a side stream from pinned memory while the
    current step computes; ``next()`` -> (imgs, targets, time_embedding) on the device (None at the end of the epoch)."""

    def __init__(self, loader, input_size=None):
        self.loader = iter(loader)
        self.max_iter = len(loader)
        self.input_size = input_size or getattr(getattr(loader, "dataset", None), "input_dim", None)
        self.stream = torch.cuda.Stream()
        self.preload()

    def preload(self):
        try:
            batch = next(self.loader)
        except StopIteration:
            self.next_input = self.next_target = self.time_embedding = None
            return
        self.next_input, self.next_target, self.time_embedding = batch[0], batch[1], batch[5]
        with torch.cuda.stream(self.stream):
            if len(batch) == 7:            # undecorated uint8 frames + kernel records: resize / HSV / flip / pad on the device
                from ..data_augment import clip_preproc_gpu
                frames = self.next_input.cuda(non_blocking=True)
                self.next_input = clip_preproc_gpu(frames, batch[6].cuda(non_blocking=True), self.input_size, torch.float16)
                frames.record_stream(self.stream)
            else:
                self.next_input = self.next_input.cuda(non_blocking=True)
            self.next_target = self.next_target.cuda(non_blocking=True)
            self.time_embedding = self.time_embedding.cuda(non_blocking=True)

    def next(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        out = (self.next_input, self.next_target, self.time_embedding)
        for t in out:
            if t is not None:
                t.record_stream(torch.cuda.current_stream())
        self.preload()
        return out
