#!/usr/bin/env python3
"""A/B of the implicit-GEMM conv main-loop variants on one device, in one process, interleaved rounds
(cdna_hip_programming.md rule 24): bit-identity against variant 1, then min / median launch time per variant.

    python tools/conv_ab.py [--variants 1,6,-1] [--rounds 5] [--iters 20]

variant 1 = 128-wide LDS-DMA tiles, 5 / 6 = halo-staged 256- / 192-pixel tiles, 0 = register staging, -1 = automatic (a per-call
argument of tscd_conv2d_nhwc_f16_ex)."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tscd_amd import ops
from tscd_amd._lib import lib

SHAPES = ((16, 72, 72, 256, 256, 3), (16, 36, 36, 256, 256, 3), (16, 18, 18, 256, 256, 3), (16, 18, 18, 512, 512, 3),
          (16, 72, 72, 256, 256, 1), (16, 36, 36, 512, 256, 1), (16, 36, 36, 256, 512, 3), (16, 18, 18, 1024, 1024, 1),
          (16, 144, 144, 64, 64, 3), (16, 72, 72, 128, 128, 3), (16, 72, 72, 768, 768, 1))


def run(x, wpk, cout, k, variant, iters):
    for _ in range(2):
        ops.conv2d(x, wpk, cout, k, 1, want_stats=True, variant=variant)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv2d(x, wpk, cout, k, 1, want_stats=True, variant=variant)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="1,6,-1")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    variants = [int(v) for v in args.variants.split(",")]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(0)
    for (B, H, W, Cin, Cout, k) in SHAPES:
        x = torch.randn(B, H, W, Cin, device=dev, generator=g).half()
        w = (torch.randn(Cout, Cin, k, k, device=dev, generator=g) / (Cin * k * k) ** 0.5).half()
        wpk = ops.pack_conv_weight(w)
        fl = 2.0 * B * H * W * Cout * k * k * Cin
        ref = None
        same = {}
        for v in variants:
            y, st = ops.conv2d(x, wpk, Cout, k, 1, want_stats=True, variant=v)
            torch.cuda.synchronize()
            if ref is None:
                ref = (y.clone(), st.clone())
            same[v] = "==" if bool(torch.equal(y, ref[0]) and torch.equal(st, ref[1])) else (
                "~=" if (y.float() - ref[0].float()).abs().max().item() <= 2e-3 * ref[0].float().abs().max().item() else "DIFF")
        ts = {v: [] for v in variants}
        for _ in range(args.rounds):
            for v in variants:
                ts[v].append(run(x, wpk, Cout, k, v, args.iters))
        cells = []
        for v in variants:
            mn, md = min(ts[v]), statistics.median(ts[v])
            cells.append(f"v{v}: {mn:7.1f} us (med {md:7.1f}) {fl / mn / 1e6:6.0f} TF/s {same[v]}")
        print((B, H, W, Cin, Cout, k), " | ".join(cells), flush=True)


if __name__ == "__main__":
    main()
