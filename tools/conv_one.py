#!/usr/bin/env python3
"""One conv shape / one main-loop variant, N launches -- the target of rocprofv3 counter passes.
    python tools/conv_one.py --variant 5 --shape 16,72,72,256,256,3 --iters 20"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tscd_amd import ops
from tscd_amd._lib import lib

ap = argparse.ArgumentParser()
ap.add_argument("--variant", type=int, default=-1)
ap.add_argument("--shape", default="16,72,72,256,256,3")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--debug", type=int, default=0)
a = ap.parse_args()
B, H, W, Cin, Cout, k = [int(v) for v in a.shape.split(",")]
dev = torch.device("cuda:0")
x = torch.randn(B, H, W, Cin, device=dev).half()
w = (torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5).half()
wpk = ops.pack_conv_weight(w)
lib.tscd_conv_set_variant(a.variant)
lib.tscd_conv_set_debug(a.debug)
out = ops.empty_nhwc(B, H, W, Cout, dev)
for _ in range(3):
    ops.conv2d(x, wpk, Cout, k, 1, out=out, want_stats=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    ops.conv2d(x, wpk, Cout, k, 1, out=out, want_stats=True)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / a.iters * 1e3
print(f"variant {a.variant} shape {a.shape}: {t:.1f} us, {2.0 * B * H * W * Cout * k * k * Cin / t / 1e6:.0f} TF/s")
