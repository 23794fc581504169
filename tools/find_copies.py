#!/usr/bin/env python3
"""Which Python-level ops issue the small device copies / fills of a train step?  torch.profiler over 2 steps of the bench
workload; prints CPU-op name -> count for every op whose device activity is a memcpy / memset / tiny elementwise kernel."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench

dev = torch.device("cuda:0")
model = bench.build_model(1.0, 1.0, "500").to(dev).train()
imgs, t0, temb = bench.synth_clip(16, 576, dev)
targets = bench.targets_on_top_proposals(model, imgs, t0, temb, 4, 12)
opt = bench.make_optimizer(model, lr=1e-6)
scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10)
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
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
cnt = collections.Counter()
stacks = {}
for e in ev:
    if e.device_type.name == "CUDA" or not e.kernels:
        continue
    for k in e.kernels:
        n = k.name
        if "Memcpy" in n or "Memset" in n or "copyBuffer" in n or "fillBuffer" in n:
            key = (e.name, n[:40])
            cnt[key] += 1
            if key not in stacks and e.stack:
                stacks[key] = [s for s in e.stack if "/repo/" in s][:3]
for (name, kn), c in cnt.most_common(40):
    print(f"{c:5d}  {name:45s} {kn:40s} {stacks.get((name, kn), '')}")
