#!/usr/bin/env python3
"""Gradients of one train step with the deferred / batched weight gradients on and off (same weights, same clip): every trainable
parameter's cosine and norm ratio, worst first.  Usage: python tools/wgrad_defer_ab.py [depth width frames lframe size]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from tscd_amd import ops

a = sys.argv[1:]
depth, width, frames, lframe, size = (float(a[0]), float(a[1]), int(a[2]), int(a[3]), int(a[4])) if len(a) >= 5 else (1.0, 1.0, 16, 4, 576)
dev = torch.device("cuda", 0)
bench.NUM_CLASSES, bench.MAX_PROPS = 25, 500
model = bench.build_model(depth, width, "500").to(dev).train()
imgs, t0, temb = bench.synth_clip(frames, size, dev)
targets = bench.targets_on_top_proposals(model, imgs, t0, temb, lframe, frames - lframe)
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
scale = torch.full((), 1024.0, device=dev)


def grads(defer):
    ops.DEFER_LINEAR_WGRAD = defer
    model.load_state_dict(sd)
    for p in model.parameters():
        p.grad = None
    model.head.loss_scale = scale
    out = model(imgs, targets, temb, lframe=lframe, gframe=frames - lframe, resume=False)
    (out["total_loss"] * scale).backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None}


if os.environ.get("DEFER_FIRST") == "1":
    g1 = grads(True)
    g0 = grads(False)
else:
    g0, g1 = grads(False), grads(True)
rows = []
for k in g0:
    a_, b_ = g0[k].flatten(), g1[k].flatten()
    cos = torch.nn.functional.cosine_similarity(a_, b_, dim=0).item() if float(a_.norm()) > 0 else 1.0
    rows.append((cos, float(b_.norm() / a_.norm().clamp(min=1e-30)), k, tuple(g0[k].shape)))
rows.sort()
print(f"{len(rows)} parameters with gradients; worst cosines (deferred vs immediate):")
for r in rows[:12]:
    print(f"  cos {r[0]:.5f}  norm ratio {r[1]:.4f}  {r[2]} {r[3]}")
print("all >= 0.9999:", all(r[0] >= 0.9999 for r in rows))
