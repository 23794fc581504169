#!/usr/bin/env python3
"""Exactly N train steps of the bench workload (TSCD-L 16 frames @576^2, 500-proposal regime, the bench's clip / targets / optimizer) and
nothing else -- the program to put behind `rocprofv3 --kernel-trace --stats`: every kernel total divided by N is a per-step figure
(the first steps carry the one-off captures and packs; N >= 100 keeps them below 1 %).  Usage: python3 tools/steady_steps.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda", 0)
model = bench.build_model(args.depth, args.width, args.props).to(dev).train()
opt = bench.make_optimizer(model, lr=args.lr)
scaler = bench.make_scaler()
ema = bench.FusedEMA(model, optimizer=opt)
imgs, t0_, temb = bench.synth_clip(args.frames, args.size, dev, seed=0)
targets = bench.targets_on_top_proposals(model, imgs, t0_, temb, args.lframe, args.frames - args.lframe)
one = torch.ones((), device=dev)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(N):
    model.head.loss_scale = scaler.scale(one)
    out = model(imgs, targets, temb, lframe=args.lframe, gframe=args.frames - args.lframe, resume=False, next_x=imgs)
    opt.zero_grad(set_to_none=True)
    scaler.scale(out["total_loss"]).backward()
    model.check_assignments()
    scaler.step(opt)
    scaler.update()
    ema.update()
    if i == N - 51:
        torch.cuda.synchronize()
        t = time.perf_counter()
torch.cuda.synchronize()
print(f"{N} steps, last 50: {(time.perf_counter() - t) / 50 * 1e3:.2f} ms per step, loss {float(out['total_loss']):.1f}")
