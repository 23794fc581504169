#!/usr/bin/env python3
"""cProfile of the host side of a train step (50-proposal regime: the device work is small, the step time is the host's): where the
main thread's time goes per step -- own time by function, top 45."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

PROPS = sys.argv[1] if len(sys.argv) > 1 else "50"
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda", 0)
model = bench.build_model(args.depth, args.width, PROPS).to(dev).train()
opt = bench.make_optimizer(model, lr=args.lr)
scaler = bench.make_scaler()
ema = bench.FusedEMA(model, optimizer=opt)
imgs, t0_, temb = bench.synth_clip(args.frames, args.size, dev, seed=0)
targets = bench.targets_on_top_proposals(model, imgs, t0_, temb, args.lframe, args.frames - args.lframe)
one = torch.ones((), device=dev)


BWD = {"pr": None, "on": False}


def _enable_in_engine_thread(g):
    # runs on autograd's device thread (where the Python backward functions run): profile that thread too
    if BWD["pr"] is not None and not BWD["on"]:
        BWD["on"] = True
        BWD["pr"].enable()
    return g


def step():
    model.head.loss_scale = scaler.scale(one)
    out = model(imgs, targets, temb, lframe=args.lframe, gframe=args.frames - args.lframe, resume=False, next_x=imgs)
    opt.zero_grad(set_to_none=True)
    loss = scaler.scale(out["total_loss"])
    loss.register_hook(_enable_in_engine_thread)
    loss.backward()
    scaler.step(opt)
    scaler.update()
    model.join_prefetch()
    ema.update()


for _ in range(10):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print(f"props {PROPS}: {(time.perf_counter() - t) / 20 * 1e3:.2f} ms per step (unprofiled)")
pr = cProfile.Profile()
BWD["pr"] = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()


def show(prof, title):
    st = pstats.Stats(prof)
    rows = []
    for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
        rows.append((tt / 20 * 1e3, ct / 20 * 1e3, nc / 20, f"{os.path.basename(fn)}:{line}:{name}"))
    rows.sort(reverse=True)
    print(f"---- {title}: own time {sum(r[0] for r in rows):.2f} ms per step")
    print(f"{'own ms':>8} {'cum ms':>8} {'calls':>7}  function (per step)")
    for tt, ct, nc, nm in rows[:40]:
        print(f"{tt:8.3f} {ct:8.3f} {nc:7.1f}  {nm}")


show(pr, "main thread")
show(BWD["pr"], "autograd device thread (Python backward functions)")
