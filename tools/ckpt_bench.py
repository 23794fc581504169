#!/usr/bin/env python3
"""Row N3 measurement: how long the training loop is held by a checkpoint -- blocking torch.save (the reference's save_checkpoint) against
the pinned-snapshot + writer-thread path of yolox/utils/checkpoint.py, for the TSCD-L state (model + SGD momentum buffers)."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from yolox.utils import checkpoint

dev = torch.device("cuda:0")
model = bench.build_model(1.0, 1.0, "500").to(dev)
opt = bench.make_optimizer(model)
for g in opt.param_groups:
    for p in g["params"]:
        opt.state[p]["momentum_buffer"] = torch.zeros_like(p)
state = {"start_epoch": 1, "model": model.state_dict(), "optimizer": opt.state_dict()}
nbytes = sum(v.numel() * v.element_size() for v in state["model"].values()) + sum(p.numel() * 4 for g in opt.param_groups for p in g["params"])
with tempfile.TemporaryDirectory() as d:
    torch.cuda.synchronize()
    t0 = time.time()
    checkpoint.save_checkpoint(state, False, d, "blocking", non_blocking=False)
    t_block = time.time() - t0
    for i in range(3):
        torch.cuda.synchronize()
        t0 = time.time()
        checkpoint.save_checkpoint(state, False, d, "async", non_blocking=True)
        t_call = time.time() - t0
        torch.cuda.synchronize()
        t_dev = time.time() - t0
        checkpoint.wait()
        t_all = time.time() - t0
        print(f"state {nbytes / 1e6:.0f} MB: blocking torch.save {t_block * 1e3:.0f} ms | async call returns after {t_call * 1e3:.1f} ms, "
              f"device copies done after {t_dev * 1e3:.1f} ms, file written after {t_all * 1e3:.0f} ms (run {i})", flush=True)
    a = torch.load(os.path.join(d, "async_ckpt.pth"), weights_only=True)
    b = torch.load(os.path.join(d, "blocking_ckpt.pth"), weights_only=True)
    assert all(torch.equal(a["model"][k].cpu(), b["model"][k].cpu()) for k in a["model"])
    print("async file == blocking file")
