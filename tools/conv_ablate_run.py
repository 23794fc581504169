#!/usr/bin/env python3
"""Times the ablation switches of tools/conv_ablate_gen.py on the dominant conv shape (B16 72x72 256->256 3x3)."""
import ctypes as C, os, sys, statistics
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
import torch
from tscd_amd import ops
from tscd_amd._lib import LIB_PATH
C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
abl = C.CDLL(os.path.join(here, "_ablate", "libabl.so"))        # built by tools/conv_ablate_gen.py
abl.tscd_conv2d_nhwc_f16.argtypes = [C.c_void_p] * 7 + [C.c_int] * 10 + [C.c_void_p]
abl.tscd_conv2d_nhwc_f16.restype = C.c_int
abl.tscd_conv_set_abl.argtypes = [C.c_int]
abl.tscd_conv_set_variant.argtypes = [C.c_int]
abl.tscd_conv_stat_tiles.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
B, H, W, Cin, Cout, k = 16, 72, 72, 256, 256, 3
x = torch.randn(B, H, W, Cin, device=dev, generator=g).half()
w = (torch.randn(Cout, Cin, k, k, device=dev, generator=g) / (Cin * k * k) ** 0.5).half()
wpk = ops.pack_conv_weight(w)
y = torch.empty(B, H, W, Cout, device=dev, dtype=torch.float16)
fl = 2.0 * B * H * W * Cout * k * k * Cin
st = torch.cuda.current_stream().cuda_stream
def launch():
    rc = abl.tscd_conv2d_nhwc_f16(x.data_ptr(), wpk.data_ptr(), y.data_ptr(), None, None, None, part.data_ptr(), B, H, W, Cin, Cout, k, 1, Cin, Cout, 0, st)
    assert rc == 0, rc
def run(mode, iters=20):
    abl.tscd_conv_set_abl(mode)
    for _ in range(2): launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): launch()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
NAMES = {0: "full kernel", 1: "no B DMA", 2: "no A halo DMA", 3: "no DMA", 4: "no fragment reads", 7: "MFMA + barriers only", 8: "no MFMA",
         16: "no barriers", 23: "MFMA only", 32: "no setprio", 64: "no main loop (launch + prologue + epilogue)", 192: "launch + prologue only",
         128: "no epilogue", 256: "no output stores", 768: "no stores at all"}
for variant in (6,):
    abl.tscd_conv_set_variant(variant)
    mt = abl.tscd_conv_stat_tiles(B * H * W, W, Cin, Cout, k, 1)
    part = torch.empty((2, 256, mt), dtype=torch.float32, device=dev)
    res = {m: [] for m in NAMES}
    for _ in range(3):
        for m in NAMES:
            res[m].append(run(m))
    print(f"variant {variant} ({'192' if variant == 6 else '256'}-row tiles, {mt} tiles)")
    for m in NAMES:
        t = min(res[m])
        print(f"  abl {m:2d} {NAMES[m]:40s} {t:7.1f} us  ({fl / t / 1e6:6.0f} TF/s-equivalent)", flush=True)
