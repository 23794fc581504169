#!/usr/bin/env python3
"""Times the ablation switches of tools/conv_ablate_glds_gen.py on the bench clip's shapes that take conv_igemm_glds_kernel."""
import ctypes as C, os, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
import torch
from tscd_amd import ops
from tscd_amd._lib import LIB_PATH
C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
abl = C.CDLL(os.path.join(here, "_ablate", "libabl_glds.so"))
abl.tscd_conv2d_nhwc_f16.argtypes = [C.c_void_p] * 7 + [C.c_int] * 10 + [C.c_void_p]
abl.tscd_conv_set_abl.argtypes = [C.c_int]
abl.tscd_conv_set_halo_auto.argtypes = [C.c_int]
abl.tscd_conv_stat_tiles.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
abl.tscd_conv_set_halo_auto(0)
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
NAMES = {0: "full kernel", 1: "no LDS-DMA after the prologue", 2: "no weight (B) pieces after the prologue", 4: "no fragment reads", 8: "no MFMA", 13: "barriers + waits only", 64: "no main loop",
         128: "no epilogue", 192: "launch + prologue only"}
SHAPES = [(16, 72, 72, 256, 128, 1), (16, 36, 36, 512, 256, 1), (16, 36, 36, 256, 256, 3), (16, 72, 72, 128, 128, 3), (16, 144, 144, 64, 64, 3), (16, 18, 18, 512, 512, 3)]
for (B, H, W, Cin, Cout, k) in SHAPES:
    x = torch.randn(B, H, W, Cin, device=dev, generator=g).half()
    w = (torch.randn(Cout, Cin, k, k, device=dev, generator=g) / (Cin * k * k) ** 0.5).half()
    wpk = ops.pack_conv_weight(w)
    y = torch.empty(B, H, W, Cout, device=dev, dtype=torch.float16)
    mt = abl.tscd_conv_stat_tiles(B * H * W, W, Cin, Cout, k, 1)
    part = torch.empty((2, wpk.shape[0], mt), dtype=torch.float32, device=dev)
    fl = 2.0 * B * H * W * Cout * k * k * Cin
    def launch():
        rc = abl.tscd_conv2d_nhwc_f16(x.data_ptr(), wpk.data_ptr(), y.data_ptr(), None, None, None, part.data_ptr(), B, H, W, Cin, Cout, k, 1, Cin, Cout, 0, st)
        assert rc == 0, rc
    res = {m: [] for m in NAMES}
    for _ in range(3):
        for m in NAMES:
            abl.tscd_conv_set_abl(m)
            for _ in range(2): launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): launch()
            e1.record(); torch.cuda.synchronize()
            res[m].append(e0.elapsed_time(e1) / 20 * 1e3)
    print((B, H, W, Cin, Cout, k), f"{mt} M-tiles, {fl / 1e9:.1f} GFLOP, {(x.numel() + y.numel()) * 2 / 1e6:.0f} MB in + out")
    for m in NAMES:
        print(f"   {NAMES[m]:32s} {min(res[m]):7.1f} us", flush=True)
