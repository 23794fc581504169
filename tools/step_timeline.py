#!/usr/bin/env python3
"""Steady-state timeline of a train step (500-proposal regime, after an 80-step pre-roll): host marks AND device-event marks of the
points that bound the step's critical path  forward up to the cost matrices -> host assignment chain -> rest of forward + backward.
Device marks are events recorded on the main stream where the host passes; all times in ms from the step's start (host clock for host
marks; device marks = elapsed from the step-start event).  Usage: python tools/step_timeline.py [preroll]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import yolox.models.tscd_matching as M
from tscd_amd import ops

PREROLL = int(sys.argv[1]) if len(sys.argv) > 1 else 80
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda", 0)
model = bench.build_model(args.depth, args.width, args.props).to(dev).train()
opt = bench.make_optimizer(model, lr=args.lr)
net = model
if os.environ.get("TSCD_TIMELINE_DDP") == "1":      # the step under DistributedDataParallel over a 1-rank RCCL group (reducer hooks, buckets)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from tscd_amd import ddp as tscd_ddp
    net = tscd_ddp.wrap(model, device_ids=[0])
    net.attach(opt)
scaler = bench.make_scaler()
ema = bench.FusedEMA(model, optimizer=opt)
imgs, t0_, temb = bench.synth_clip(args.frames, args.size, dev, seed=0)
targets = bench.targets_on_top_proposals(model, imgs, t0_, temb, args.lframe, args.frames - args.lframe)
one = torch.ones((), device=dev)
H, D = {}, {}          # host marks (perf_counter), device marks (events)


def mark(name):
    H[name] = time.perf_counter()
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    D[name] = e


def wrap(obj, attr, before=None, after=None):
    f = getattr(obj, attr)

    def g(*a, **k):
        if before:
            mark(before)
        r = f(*a, **k)
        if after:
            mark(after)
        return r
    setattr(obj, attr, g)


wrap(ops, "gather_det", before="counts_on_host")                                   # first call behind the D2H of the proposal counts
wrap(M.AwarePositionRegMatcher, "prepare", before="prepare_in", after="costs_queued")
wrap(M.AwarePositionRegMatcher, "finish", before="finish_in", after="finish_out")
import yolox.models.tscd as tscd_mod
from tscd_amd import functional as TF
wrap(ops, "simota_assign", before="simota_in")
wrap(TF, "dense_loss", before="dense_loss_in")
wrap(tscd_mod.TSCD, "_prefetch", before="prefetch_in", after="prefetch_out")
_orig_prefetch = tscd_mod.TSCD._prefetch


def _prefetch_timed(self, next_x):
    """... plus an event on the PREFETCH stream behind the replayed frozen front: when the next clip's pass really ends on the device
    (device mark prefetch_done; the post segment shares the CUs with whatever of the pass is still running when the chain ends)."""
    r = _orig_prefetch(self, next_x)
    e = torch.cuda.Event(enable_timing=True)
    e.record(tscd_mod._prefetch_stream(next_x.device))
    D["prefetch_done"] = e
    H["prefetch_done"] = time.perf_counter()
    return r


tscd_mod.TSCD._prefetch = _prefetch_timed
orig_chain = M.lap_chain_f16


CH = []                # (start, end) of every chain on the worker's clock; chain k belongs to step k (the deferred join lets the
                       # host leave the step before its chain has ended, so the marks are matched to steps by index afterwards)


def chain(base_ptr, shapes, kinds, before=None, **kw):
    if before is not None:
        before()
    t0 = time.perf_counter()
    r = orig_chain(base_ptr, shapes, kinds, None, **kw)
    CH.append((t0, time.perf_counter()))
    return r


M.lap_chain_f16 = chain


STARTS = []


def step(rec=None):
    mark("start")
    STARTS.append(H["start"])
    model.head.loss_scale = scaler.scale(one)
    out = net(imgs, targets, temb, lframe=args.lframe, gframe=args.frames - args.lframe, resume=False, next_x=imgs)
    mark("forward_out")
    opt.zero_grad(set_to_none=True)
    scaler.scale(out["total_loss"]).backward()
    mark("backward_out")
    if os.environ.get("TSCD_TL_NO_CHECK") != "1":
        model.check_assignments()
    mark("check_out")
    scaler.step(opt)
    scaler.update()
    ema.update()
    mark("step_out")
    if rec is not None:
        rec.append((dict(H), dict(D)))


for _ in range(PREROLL):
    step()
torch.cuda.synchronize()
rec = []
t = time.perf_counter()
N = 30
for _ in range(N):
    step(rec)
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / N * 1e3
_gaps = np.diff([h["start"] for h, _ in rec]) * 1e3
_dev = [rec[i][1]["start"].elapsed_time(rec[i + 1][1]["start"]) for i in range(len(rec) - 1)]
names = ["counts_on_host", "prepare_in", "costs_queued", "chain_start", "simota_in", "dense_loss_in", "prefetch_in", "prefetch_out", "prefetch_done", "finish_in", "chain_end", "finish_out", "forward_out", "backward_out", "step_out"]
print(f"steady state after {PREROLL} steps: {ms:.2f} ms per step (no drains)")
print("start-to-start, ms: host median %.2f max %.2f; device median %.2f max %.2f; the 3 longest device steps %s" %
      (np.median(_gaps), _gaps.max(), np.median(_dev), max(_dev), np.round(sorted(_dev)[-3:], 2).tolist()))
print("host clock, ms from the step's start (median of %d steps):" % N)
M.AwarePositionRegMatcher._join_pending(model.head.local_reg_matcher)
k0 = len(STARTS) - N
chain_rel = {"chain_start": [(CH[k][0] - STARTS[k]) * 1e3 for k in range(k0, len(STARTS))],
             "chain_end": [(CH[k][1] - STARTS[k]) * 1e3 for k in range(k0, len(STARTS))]}
for n in names:
    v = chain_rel[n] if n in chain_rel else [(h[n] - h['start']) * 1e3 for h, _ in rec]
    print(f"  {n:16s} {np.median(v):7.2f}")
print("device clock, ms from the step-start event reaching the main stream (median):")
for n in [x for x in names if x not in ("chain_start", "chain_end")]:
    print(f"  {n:16s} {np.median([d['start'].elapsed_time(d[n]) for _, d in rec]):7.2f}")
lag = [(h["start"] - rec[0][0]["start"]) * 1e3 - rec[0][1]["start"].elapsed_time(d["start"]) for h, d in rec]
print(f"host start minus device start of the same step (how far the host runs ahead), ms: median {np.median(lag):.2f}  min {min(lag):.2f}  max {max(lag):.2f}")
print(f"chain duration {np.median([(b - a) * 1e3 for a, b in CH[k0:]]):.2f} ms; deferred join of the chain: {M.DEFER_CHAIN}")
