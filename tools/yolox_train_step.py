#!/usr/bin/env python3
"""Row N4 measurement: one training step of the YOLOX base detector (backbone + neck + head forward and backward, SimOTA, fused losses,
FusedSGD) on synthetic 640x640 images -- images per second per GPU.  `python tools/yolox_train_step.py [depth width batch]`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tscd_amd.optim import FusedSGD
from yolox.models import YOLOX, YOLOPAFPN, YOLOXHead

depth, width, B = (float(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0.33, 0.5, 16)
torch.manual_seed(0)
model = YOLOX(YOLOPAFPN(depth, width), YOLOXHead(80, width))
for m in model.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.eps, m.momentum = 1e-3, 0.03
model.head.initialize_biases(1e-2)
model = model.cuda().train()
opt = FusedSGD([p for p in model.parameters()], lr=1e-4, momentum=0.9, nesterov=True, weight_decay=5e-4)
from tscd_amd.optim import GradScaler
scaler = GradScaler("cuda", init_scale=2.0 ** 10)
g = torch.Generator().manual_seed(1)
imgs = (torch.rand(B, 3, 640, 640, generator=g) * 255).half().cuda()
labels = torch.zeros(B, 50, 5)
for j in range(8):
    labels[:, j] = torch.tensor([float(j * 7 % 80), 80.0 + 60 * j, 100.0 + 55 * j, 60.0 + 12 * j, 90.0 + 6 * j])
labels = labels.cuda()


def step():
    out = model(imgs, labels)
    opt.zero_grad(set_to_none=True)
    scaler.scale(out["total_loss"]).backward()
    scaler.step(opt)
    scaler.update()
    return out


for _ in range(5):
    out = step()
torch.cuda.synchronize()
n = 20
t0 = time.time()
for _ in range(n):
    out = step()
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print(f"YOLOX d{depth}/w{width}, batch {B} @640^2, train step (fwd + bwd + SimOTA + losses + FusedSGD): {dt * 1e3:.1f} ms = {B / dt:.0f} img/s; "
      f"loss {float(out['total_loss']):.3f}, fg per box {float(out['num_fg']):.2f}", flush=True)
