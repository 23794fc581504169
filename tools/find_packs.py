#!/usr/bin/env python3
"""Which call sites still pack a weight inside a train step (after FusedSGD's batched in-place re-packing)?"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from tscd_amd import ops

dev = torch.device("cuda:0")
model = bench.build_model(1.0, 1.0, "500").to(dev).train()
imgs, t0, temb = bench.synth_clip(16, 576, dev)
targets = bench.targets_on_top_proposals(model, imgs, t0, temb, 4, 12)
opt = bench.make_optimizer(model, lr=1e-6)
scaler = bench.make_scaler()
one = torch.ones((), device=dev)


def step():
    model.head.loss_scale = scaler.scale(one)
    out = model(imgs, targets, temb, lframe=4, gframe=12, resume=False)
    opt.zero_grad(set_to_none=True)
    scaler.scale(out["total_loss"]).backward()
    scaler.step(opt)
    scaler.update()


for _ in range(3):
    step()
cnt = collections.Counter()
orig = ops.pack_conv_weight


def spy(w, cin_pad=None, mode=0):
    fr = [f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in traceback.extract_stack()[-6:-1]]
    cnt[(tuple(w.shape), mode, " <- ".join(reversed(fr[-4:])))] += 1
    return orig(w, cin_pad, mode)


ops.pack_conv_weight = spy
step()
torch.cuda.synchronize()
print("pack calls in one step:", sum(cnt.values()))
for (shape, mode, where), c in cnt.most_common(30):
    print(f"{c:3d}  {shape} mode {mode}  {where}")
