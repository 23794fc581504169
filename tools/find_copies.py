#!/usr/bin/env python3
"""Which Python-level ops issue the small device copies / fills / dtype conversions of a step?  torch.profiler over one step of the
bench workload (`train`: TSCD-L 16-frame train step, default; `eval`: TSCD-S 8-frame evaluation clip, bench --config c2); prints
CPU-op name -> launches for every op whose device activity is a memcpy / memset / elementwise kernel, with the repo frames."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench

dev = torch.device("cuda:0")
EVAL = len(sys.argv) > 1 and sys.argv[1] == "eval"
if EVAL:
    model = bench.build_model(0.33, 0.5, "500").to(dev).eval().half()
    imgs, t0, temb = bench.synth_clip(8, 576, dev)

    def step():
        with torch.no_grad():
            model(imgs, None, temb, lframe=2, gframe=6, resume=False)
else:
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
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
stacks = {}
total = 0
for e in prof.events():
    if e.device_type.name == "CUDA" or not e.kernels:
        continue
    for k in e.kernels:
        total += 1
        n = k.name
        if any(t in n for t in ("Memcpy", "Memset", "copyBuffer", "fillBuffer", "elementwise", "CatArray", "gather", "index")):
            frames = tuple(s.split("/repo/")[-1] for s in (e.stack or []) if "/repo/" in s and "tools/find_copies" not in s)[:2]
            key = (e.name, n[:36], frames[:1], str(e.input_shapes)[:90])
            cnt[key] += 1
            stacks.setdefault(key, frames)
print("device activities in the step:", total)
for key, c in cnt.most_common(70):
    print(f"{c:5d}  {key[0]:24s} {key[1]:36s} {key[3]} {stacks[key]}")
