#!/usr/bin/env python3
"""A/B of the tile height (64 / 128 / 192 rows, and the automatic choice) of the 128-channel-wide LDS-DMA conv variant on the
shapes of the bench clip that do not fill the chip with 128-row tiles.  Interleaved rounds, min and median launch time."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tscd_amd import ops
from tscd_amd._lib import lib

SHAPES = ((16, 36, 36, 256, 256, 3), (16, 18, 18, 256, 256, 3), (16, 18, 18, 512, 512, 3), (16, 72, 72, 128, 128, 3),
          (16, 36, 36, 256, 256, 1), (16, 36, 36, 512, 256, 1), (16, 18, 18, 1024, 512, 1), (16, 18, 18, 512, 512, 1),
          (16, 72, 72, 128, 128, 1), (16, 36, 36, 768, 768, 1), (16, 36, 36, 512, 512, 1), (16, 72, 72, 256, 256, 3),
          (8, 36, 36, 256, 256, 3), (8, 72, 72, 128, 128, 3))


def run(x, wpk, cout, k, bm, iters):
    # the 128-wide LDS-DMA form on every shape (variant 1: the halo form would take the 72x72 3x3 otherwise); bm = 0: automatic height
    kw = dict(variant=1, tile_m=bm)
    for _ in range(2):
        ops.conv2d(x, wpk, cout, k, 1, want_stats=True, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv2d(x, wpk, cout, k, 1, want_stats=True, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(0)
    for (B, H, W, Cin, Cout, k) in SHAPES:
        x = torch.randn(B, H, W, Cin, device=dev, generator=g).half()
        w = (torch.randn(Cout, Cin, k, k, device=dev, generator=g) / (Cin * k * k) ** 0.5).half()
        wpk = ops.pack_conv_weight(w)
        fl = 2.0 * B * H * W * Cout * k * k * Cin
        times = {bm: [] for bm in (128, 64, 192, 0)}
        for _ in range(4):
            for bm in times:
                times[bm].append(run(x, wpk, Cout, k, bm, 20))
        auto = lib.tscd_conv_stat_tiles_ex(B * H * W, W, Cin, Cout, k, 1, 1, 1, 0)
        txt = "  ".join(f"{('auto' if bm == 0 else bm)}: {min(t):6.1f}/{statistics.median(t):6.1f} us ({fl / min(t) / 1e6:5.0f} TF/s)" for bm, t in times.items())
        print(f"{(B, H, W, Cin, Cout, k)}: {txt}   [auto tiles {auto}]", flush=True)


main()
