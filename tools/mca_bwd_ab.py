#!/usr/bin/env python3
"""MCA backward, same box A/B: split passes (mode 2, default) vs the atomic query-stationary passes (mode 1 paired, 0 single) at
the bench clip's sizes (16 frames, 4 local, 500 / 50 proposals per frame, W 256).  Prints the time per call and the largest
difference between the modes' outputs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tscd_amd import ops
from tscd_amd._lib import lib

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
        outs = {}
        for rnd in range(2):
            for mode in (2, 1, 0):
                lib.tscd_mca_backward_set_paired(mode)
                for _ in range(2):
                    outs[mode] = f()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(5):
                    f()
                e.record()
                torch.cuda.synchronize()
                print(f"P {P} need_reg {need_reg} round {rnd} mode {mode}: {s.elapsed_time(e) / 5 * 1e3:8.1f} us per call", flush=True)
        for other in (1, 0):
            diff = [float((x.float() - y.float()).abs().max() / (y.float().abs().max() + 1e-20)) for x, y in zip(outs[2], outs[other]) if x is not None]
            print(f"   mode 2 vs {other}: max |diff| / max |ref| per output = {['%.2e' % v for v in diff]}", flush=True)
lib.tscd_mca_backward_set_paired(2)
