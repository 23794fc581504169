#!/usr/bin/env python3
"""Per-step wall times of the bench's train step over N steps in the steady state, and what Python's cyclic GC contributes: the
collections that ran (generation, duration) and the steps they landed in.  python tools/step_jitter.py [N=400] [freeze]"""
import gc
import os
import statistics as st
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
FREEZE = len(sys.argv) > 2 and sys.argv[2] == "freeze"
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


def step():
    model.head.loss_scale = scaler.scale(one)
    out = model(imgs, targets, temb, lframe=args.lframe, gframe=args.frames - args.lframe, resume=False, next_x=imgs)
    opt.zero_grad(set_to_none=True)
    scaler.scale(out["total_loss"]).backward()
    model.check_assignments()
    scaler.step(opt)
    scaler.update()
    ema.update()


for _ in range(110):
    step()
torch.cuda.synchronize()
if FREEZE:
    gc.collect()
    gc.freeze()
events = []
t_gc = [0.0]


def cb(phase, info):
    if phase == "start":
        t_gc[0] = time.perf_counter()
    else:
        events.append((time.perf_counter(), info["generation"], (time.perf_counter() - t_gc[0]) * 1e3, info["collected"]))


gc.callbacks.append(cb)
marks = [time.perf_counter()]
for i in range(N):
    step()
    marks.append(time.perf_counter())
torch.cuda.synchronize()
total = (time.perf_counter() - marks[0]) / N * 1e3
gc.callbacks.remove(cb)
dts = [(b - a) * 1e3 for a, b in zip(marks[:-1], marks[1:])]
print(f"{N} steps{' (gc.freeze() after the warm-up)' if FREEZE else ''}: {total:.3f} ms per step; host loop per step: median {st.median(dts):.2f}  p90 {sorted(dts)[int(0.9 * N)]:.2f}  "
      f"p99 {sorted(dts)[int(0.99 * N)]:.2f}  max {max(dts):.2f} ms")
by_gen = {}
for t, g, ms, c in events:
    by_gen.setdefault(g, []).append(ms)
for g in sorted(by_gen):
    v = by_gen[g]
    print(f"  gc generation {g}: {len(v)} collections, {sum(v):.1f} ms in total ({sum(v) / N:.3f} ms per step), longest {max(v):.2f} ms")
w = 20
worst = max(range(0, N - w), key=lambda k: marks[k + w] - marks[k])
best = min(range(0, N - w), key=lambda k: marks[k + w] - marks[k])
print(f"  20-step windows: best {(marks[best + w] - marks[best]) / w * 1e3:.2f}  worst {(marks[worst + w] - marks[worst]) / w * 1e3:.2f} ms per step")
