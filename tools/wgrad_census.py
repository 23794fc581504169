#!/usr/bin/env python3
"""Every weight-gradient launch of a steady-state train step (bench workload), by shape: launches per step, us per launch (device events
around the call on the stream it runs on: GEMM + split-K reduce), TFLOP/s.  Usage: python tools/wgrad_census.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from tscd_amd import ops

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
REC = None
orig_wgrad, orig_lbwd = ops.conv_wgrad, ops.linear_bwd


def wgrad(x, dy, cin, cout, ksize, groups=1, variant=1):
    if REC is None:
        return orig_wgrad(x, dy, cin, cout, ksize, groups, variant)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = orig_wgrad(x, dy, cin, cout, ksize, groups, variant)
    e.record()
    B, H, W, _ = x.shape
    REC.append((("conv", B * H * W, cin, cout, ksize, groups), 2.0 * B * H * W * cin * cout * ksize * ksize * groups, s, e))
    return r


def lbwd(x, dy, weight, need_dx=True, need_dw=True, need_db=False, grad_params=None):
    if REC is None or not need_dw:
        return orig_lbwd(x, dy, weight, need_dx, need_dw, need_db, grad_params)
    # time the parameter half alone
    dx = orig_lbwd(x, dy, weight, need_dx, False, False)[0] if need_dx else None
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    _, dw, db = orig_lbwd(x, dy, weight, False, True, need_db)
    e.record()
    n, in_f = x.shape
    REC.append((("linear", n, in_f, weight.shape[0], 1, 1), 2.0 * n * in_f * weight.shape[0], s, e))
    return dx, dw, db


ops.conv_wgrad, ops.linear_bwd = wgrad, lbwd


def step():
    model.head.loss_scale = scaler.scale(one)
    out = model(imgs, targets, temb, lframe=args.lframe, gframe=args.frames - args.lframe, resume=False, next_x=imgs)
    opt.zero_grad(set_to_none=True)
    scaler.scale(out["total_loss"]).backward()
    scaler.step(opt)
    scaler.update()
    model.join_prefetch()
    ema.update()


for _ in range(20):
    step()
torch.cuda.synchronize()
REC = []
N = 5
for _ in range(N):
    step()
torch.cuda.synchronize()
tab = collections.defaultdict(lambda: [0, 0.0, 0.0])
for key, fl, s, e in REC:
    r = tab[key]
    r[0] += 1
    r[1] += s.elapsed_time(e) * 1e-3
    r[2] += fl
tot = sum(r[1] for r in tab.values())
print(f"weight-gradient calls per step {len(REC) / N:.0f}, {tot / N * 1e3:.3f} ms per step (GEMM + reduce, serialised by the event brackets)")
print("   n/step  us/call  TFLOP/s  ms/step   (kind, M rows, Cin, Cout, k, groups)")
for key, r in sorted(tab.items(), key=lambda kv: -kv[1][1]):
    print(f"  {r[0] / N:6.1f} {r[1] / r[0] * 1e6:8.1f} {r[2] / r[1] / 1e12:8.1f} {r[1] / N * 1e3:8.3f}   {key}")
