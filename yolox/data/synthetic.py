"""Synthetic clip source with the tensor contract of the reference's OVIS/VID loaders + DataPrefetcher
(yolox/data/datasets/vid.py:784-797,963-1012): ``next()`` -> (imgs [B,3,S,S], targets [B,max_labels,5]
(cls,cx,cy,w,h; zero rows = padding), time_embedding [B,256]), one clip = lframe+gframe frames.

The real dataset pipeline (cv2 decode, mosaic/HSV augmentation, COCO json) is row N1 of SURVEY.md section 8f and is not
built; this loader feeds the hot path for smoke runs and benchmarks (SURVEY 8d inputs)."""
import math

import torch


def get_timing_signal_1d(index_sequence, channels, min_timescale=1.0, max_timescale=1.0e4):
    """Sinusoidal frame-index embedding (yolox/data/datasets/vid.py:1015-1023)."""
    n = channels // 2
    inc = math.log(max_timescale / min_timescale) / (n - 1)
    inv = min_timescale * torch.exp(torch.arange(0, n) * -inc)
    st = index_sequence.float().unsqueeze(1) * inv.unsqueeze(0)
    return torch.cat([torch.sin(st), torch.cos(st)], dim=1)


class SyntheticClipLoader:
    def __init__(self, frames, size, num_clips=100, boxes_per_frame=6, max_labels=100, device="cuda", rank=0, world_size=1,
                 seed=0):
        self.frames, self.size, self.num_clips = frames, size, num_clips
        self.boxes, self.max_labels, self.device = boxes_per_frame, max_labels, device
        # rank-aware sharding: every rank walks the same seeded order and takes clips rank, rank+world, ...
        self.ids = list(range(rank, num_clips, world_size))
        self.seed, self.pos = seed, 0

    def __len__(self):
        return len(self.ids)

    def _clip(self, cid):
        g = torch.Generator().manual_seed(self.seed * 100003 + cid)
        S, B = self.size, self.frames
        imgs = torch.rand(B, 3, S, S, generator=g) * 255
        t = torch.zeros(B, self.max_labels, 5)
        sc = S / 576.0
        for j in range(self.boxes):
            t[:, j] = torch.tensor([float(j), (80 + 60 * j) * sc, (100 + 50 * j) * sc, (60 + 10 * j) * sc, (90 + 5 * j) * sc])
        return imgs, t, get_timing_signal_1d(torch.arange(B), 256)

    def next(self):
        cid = self.ids[self.pos % len(self.ids)]
        self.pos += 1
        imgs, t, te = self._clip(cid)
        return imgs.to(self.device, non_blocking=True), t.to(self.device), te.to(self.device)
