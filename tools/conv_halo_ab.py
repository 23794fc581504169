#!/usr/bin/env python3
"""A/B of the halo-staged conv kernel's tile heights (192 / 224 / 256 rows: variants 6 / 7 / 5, and the automatic choice) on the head
towers' 3x3 shapes of the bench clip.  Interleaved rounds, min and median launch time; outputs compared bit for bit."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tscd_amd import ops

# (B, H, W, Cin, Cout per group, groups)
SHAPES = ((16, 72, 72, 256, 512, 1), (16, 72, 72, 256, 256, 2), (16, 72, 72, 256, 256, 1), (16, 36, 36, 256, 512, 1), (16, 36, 36, 256, 256, 2))


def run(x, wpk, cout, groups, variant, iters):
    kw = dict(variant=variant, groups=groups)
    for _ in range(2):
        y, _ = ops.conv2d(x, wpk, cout, 3, 1, want_stats=True, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        y, _ = ops.conv2d(x, wpk, cout, 3, 1, want_stats=True, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, y


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(0)
    for (B, H, W, Cin, Cout, G) in SHAPES:
        x = torch.randn(B, H, W, Cin * G, device=dev, generator=g).half()
        ws = [(torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) / (Cin * 9) ** 0.5).half() for _ in range(G)]
        wpk = torch.cat([ops.pack_conv_weight(w) for w in ws], 0).contiguous()
        fl = 2.0 * B * H * W * Cout * 9 * Cin * G
        times = {v: [] for v in (6, 7, 5, -1)}
        outs = {}
        for _ in range(4):
            for v in times:
                try:
                    t, y = run(x, wpk, Cout, G, v, 20)
                except Exception as ex:
                    t, y = float("nan"), None
                times[v].append(t)
                outs[v] = y
        same = all(outs[v] is None or torch.equal(outs[v], outs[6]) for v in outs)
        name = {6: "192", 7: "224", 5: "256", -1: "auto"}
        txt = "  ".join(f"{name[v]}: {min(t):6.1f}/{statistics.median(t):6.1f} us ({fl / min(t) / 1e6:5.0f} TF/s)" for v, t in times.items())
        print(f"{(B, H, W, Cin, Cout, G)}: {txt}   bit-identical {same}", flush=True)


main()
