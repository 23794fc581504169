#!/usr/bin/env python3
"""How the host assignment chain and the step time move as the synthetic run trains (500-proposal regime): per 25 steps, the mean
time of the clip's Hungarian chain on the worker thread, the time the main thread spends inside matcher.finish (waiting for it +
queueing CAFM) and the step time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import yolox.models.tscd_matching as M

MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 1       # tscd_lap_chain_set_parallel: 1 adaptive side-by-side (default), 0 strictly in order, 2 always
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
marks = {}
orig_finish = M.AwarePositionRegMatcher.finish
orig_chain = M.lap_chain_f16


def finish(self, *a, **k):
    t = time.perf_counter()
    r = orig_finish(self, *a, **k)
    marks["finish"] = time.perf_counter() - t
    return r


def chain(base_ptr, shapes, kinds, before=None, **kw):
    if before is not None:
        before()
    t = time.perf_counter()
    r = orig_chain(base_ptr, shapes, kinds, None, **kw)
    marks["chain"] = time.perf_counter() - t
    return r


M.AwarePositionRegMatcher.finish = finish
from tscd_amd import _host
_host._lib.tscd_lap_chain_set_parallel(MODE)
print('chain mode', MODE, flush=True)
_host._lib.tscd_lap_debug_steps(1)
M.lap_chain_f16 = chain


def step():
    model.head.loss_scale = scaler.scale(one)
    out = model(imgs, targets, temb, lframe=args.lframe, gframe=args.frames - args.lframe, resume=False, next_x=imgs)
    opt.zero_grad(set_to_none=True)
    scaler.scale(out["total_loss"]).backward()
    scaler.step(opt)
    scaler.update()
    model.join_prefetch()
    ema.update()
    return out


for _ in range(10):
    step()
for blk in range(10):
    torch.cuda.synchronize()
    t = time.perf_counter()
    ch, fi = [], []
    for _ in range(25):
        out = step()
        ch.append(marks["chain"])
        fi.append(marks["finish"])
    torch.cuda.synchronize()
    print(f"steps {10 + blk * 25:4d}-{10 + blk * 25 + 24:4d}: step {(time.perf_counter() - t) / 25 * 1e3:6.2f} ms | assignment chain on the worker {np.mean(ch) * 1e3:5.2f} ms "
          f"| main thread inside matcher.finish {np.mean(fi) * 1e3:5.2f} ms | loss {float(out['total_loss'].detach()):8.1f} | solves on the int16 / int32 matrix {_host._lib.tscd_lap_debug_width(0, 1)} / {_host._lib.tscd_lap_debug_width(1, 1)} | relaxation passes per step {_host._lib.tscd_lap_debug_steps(1) // 25}", flush=True)
