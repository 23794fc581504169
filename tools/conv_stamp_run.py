#!/usr/bin/env python3
"""Where a k-step of conv_igemm_glds_kernel goes, from in-kernel time stamps (tools/conv_stamp_gen.py builds the stamped library).
Per shape: mean over all waves of the per-step time in the four segments, the main loop as a whole and the epilogue (ns; the stamp
clock ticks at 100 MHz, sums over 8-72 steps per wave and hundreds of waves average it out)."""
import ctypes as C
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
import torch

from tscd_amd import ops
from tscd_amd._lib import LIB_PATH

C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
lib = C.CDLL(os.path.join(here, "_ablate", "libstamp.so"))
lib.tscd_conv2d_nhwc_f16.argtypes = [C.c_void_p] * 7 + [C.c_int] * 10 + [C.c_void_p]
lib.tscd_conv_set_dbg.argtypes = [C.c_void_p]
lib.tscd_conv_set_variant.argtypes = [C.c_int]
lib.tscd_conv_set_halo_auto.argtypes = [C.c_int]
lib.tscd_conv_stat_tiles.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
lib.tscd_conv_set_variant(1)                 # the 128-wide LDS-DMA variant only (4 waves per workgroup)
lib.tscd_conv_set_halo_auto(0)
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
SHAPES = [(16, 36, 36, 256, 256, 3), (16, 18, 18, 512, 512, 3), (16, 72, 72, 128, 128, 3), (16, 72, 72, 256, 256, 3), (16, 72, 72, 256, 128, 1),
          (16, 36, 36, 512, 256, 1), (16, 144, 144, 64, 64, 3)]
for (B, H, W, Cin, Cout, k) in SHAPES:
    x = torch.randn(B, H, W, Cin, device=dev, generator=g).half()
    w = (torch.randn(Cout, Cin, k, k, device=dev, generator=g) / (Cin * k * k) ** 0.5).half()
    wpk = ops.pack_conv_weight(w)
    y = torch.empty(B, H, W, Cout, device=dev, dtype=torch.float16)
    mt = lib.tscd_conv_stat_tiles(B * H * W, W, Cin, Cout, k, 1)
    part = torch.empty((2, wpk.shape[0], mt), dtype=torch.float32, device=dev)
    nwg = mt * ((Cout + 127) // 128)
    dbg = torch.zeros((nwg * 8, 8), dtype=torch.int64, device=dev)
    lib.tscd_conv_set_dbg(dbg.data_ptr())
    for _ in range(3):
        assert lib.tscd_conv2d_nhwc_f16(x.data_ptr(), wpk.data_ptr(), y.data_ptr(), None, None, None, part.data_ptr(), B, H, W, Cin, Cout, k, 1, Cin, Cout, 0, st) == 0
    torch.cuda.synchronize()
    lib.tscd_conv_set_dbg(None)
    d = dbg.cpu().double()
    d = d[d[:, 6] > 0]
    nk = d[:, 6].mean().item()
    steps = nk - 1
    seg = [(d[:, i].mean().item() * 10.0 / steps) for i in range(4)]
    loop = d[:, 4].mean().item() * 10.0
    epi = d[:, 5].mean().item() * 10.0
    print(f"{(B, H, W, Cin, Cout, k)}: {nwg} workgroups, {int(nk)} k-tiles | per step: vmcnt wait {seg[0]:5.0f} ns, barrier {seg[1]:5.0f}, LDS-DMA issue {seg[2]:5.0f}, "
          f"reads + MFMAs + loop {seg[3]:5.0f}  (sum {sum(seg):5.0f}; 8 MFMAs per wave = {8 * 32 / 2.0:.0f} ns at 2.0 GHz) | kernel entry -> loop end {loop / 1e3:6.1f} us, epilogue {epi / 1e3:5.1f} us",
          flush=True)
