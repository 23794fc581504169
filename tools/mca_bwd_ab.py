#!/usr/bin/env python3
"""MCA forward and backward timing at the bench clip's sizes (16 frames, 4 local, 500 / 50 proposals per frame, W 256), per call."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tscd_amd import ops

d = torch.device("cuda")
torch.manual_seed(0)
W, F, L = 256, 16, 4
for P in (500, 50):
    ppf = [P] * F
    N, N_L = F * P, L * P
    tiles = ops.mca_tiles(ppf, L, d)
    centers = torch.randn(40, W, device=d)
    lab = torch.randint(0, 40, (N,), device=d)
    mk = lambda n, c: (centers[lab[:n]].repeat(1, c // W) + 0.3 * torch.randn(n, c, device=d)).half().contiguous()
    q_c, q_r = mk(N_L, W), mk(N_L, W)
    kv_c, kv_r = mk(N, 2 * W), mk(N, 2 * W)
    score = torch.rand(N, device=d).half()
    for need_reg in (True, False):
        xcat_c, cat3_c, xcat_r, cat3_r, stats = ops.mca_attention(q_c, kv_c, q_r, kv_r, score, tiles, N, N_L, W, need_reg, 25.0, 0.75, 0.99,
                                                                  want_stats=True)
        g2 = lambda: torch.randn(N_L, 2 * W, device=d).half()
        g3 = lambda: torch.randn(N_L, 3 * W, device=d).half()
        a = (g2(), g2() if need_reg else None, g3(), g3() if need_reg else None)
        f = lambda: ops.mca_backward(q_c, kv_c, q_r, kv_r, score, tiles, N, N_L, W, need_reg, 25.0, 0.75, 0.99, stats, a[0], a[1], a[2], a[3],
                                     cat3_c, cat3_r)
        fw = lambda: ops.mca_attention(q_c, kv_c, q_r, kv_r, score, tiles, N, N_L, W, need_reg, 25.0, 0.75, 0.99, want_stats=True)
        for _ in range(3):
            fw()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fw()
        e.record()
        torch.cuda.synchronize()
        print(f"P {P} need_reg {need_reg} forward: {s.elapsed_time(e) / 10 * 1e3:8.1f} us per call", flush=True)
        if need_reg:      # agg_iou as the train step calls it: no gradient reaches the class branch's outputs (CLS_G = false passes)
            fn = lambda: ops.mca_backward(q_c, kv_c, q_r, kv_r, score, tiles, N, N_L, W, True, 25.0, 0.75, 0.99, stats, None, a[1], None, a[3],
                                          cat3_c, cat3_r)
            for _ in range(2):
                fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                fn()
            e.record()
            torch.cuda.synchronize()
            print(f"P {P} need_reg True, no class-branch gradient: {s.elapsed_time(e) / 5 * 1e3:8.1f} us per call", flush=True)
        for rnd in range(2):
            for _ in range(2):
                f()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                f()
            e.record()
            torch.cuda.synchronize()
            print(f"P {P} need_reg {need_reg} round {rnd}: {s.elapsed_time(e) / 5 * 1e3:8.1f} us per call", flush=True)
