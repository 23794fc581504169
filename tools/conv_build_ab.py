#!/usr/bin/env python3
"""Same-process A/B of two builds of conv_igemm.hip (shared libraries under tools/_ablate/, e.g. one built from `git show
HEAD:tscd_amd/csrc/conv_igemm.hip` and one from the working tree) on the bench clip's conv shapes, interleaved rounds.  Build them with -Wl,-Bsymbolic: the kernels' host
stubs carry the same names in every build and in the product library (loaded RTLD_GLOBAL here), and an interposed stub launches the
OTHER build's kernel.  The last column is the largest output difference between the first and each other build."""
import ctypes as C, os, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
import torch
from tscd_amd import ops
from tscd_amd._lib import LIB_PATH
C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
names = sys.argv[1:] or ["libold.so", "libnew.so"]
libs = []
for n in names:
    l = C.CDLL(os.path.join(here, "_ablate", n))
    l.tscd_conv2d_nhwc_f16.argtypes = [C.c_void_p] * 7 + [C.c_int] * 10 + [C.c_void_p]
    l.tscd_conv_stat_tiles.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    libs.append(l)
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
SHAPES = [(16, 72, 72, 256, 256, 3), (16, 72, 72, 128, 128, 3), (16, 36, 36, 256, 256, 3), (16, 72, 72, 256, 128, 1), (16, 36, 36, 512, 256, 1), (16, 144, 144, 64, 64, 3), (16, 18, 18, 512, 512, 3)]
for (B, H, W, Cin, Cout, k) in SHAPES:
    x = torch.randn(B, H, W, Cin, device=dev, generator=g).half()
    w = (torch.randn(Cout, Cin, k, k, device=dev, generator=g) / (Cin * k * k) ** 0.5).half()
    wpk = ops.pack_conv_weight(w)
    y = torch.empty(B, H, W, Cout, device=dev, dtype=torch.float16)
    fl = 2.0 * B * H * W * Cout * k * k * Cin
    res = [[] for _ in libs]
    for rnd in range(4):
        for li, l in enumerate(libs):
            mt = l.tscd_conv_stat_tiles(B * H * W, W, Cin, Cout, k, 1)
            part = torch.empty((2, wpk.shape[0], mt), dtype=torch.float32, device=dev)
            def launch():
                rc = l.tscd_conv2d_nhwc_f16(x.data_ptr(), wpk.data_ptr(), y.data_ptr(), None, None, None, part.data_ptr(), B, H, W, Cin, Cout, k, 1, Cin, Cout, 0, st)
                assert rc == 0
            for _ in range(3): launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): launch()
            e1.record(); torch.cuda.synchronize()
            res[li].append(e0.elapsed_time(e1) / 20 * 1e3)
    outs = []
    for l in libs:
        mt = l.tscd_conv_stat_tiles(B * H * W, W, Cin, Cout, k, 1)
        part = torch.empty((2, wpk.shape[0], mt), dtype=torch.float32, device=dev)
        y.zero_()
        assert l.tscd_conv2d_nhwc_f16(x.data_ptr(), wpk.data_ptr(), y.data_ptr(), None, None, None, part.data_ptr(), B, H, W, Cin, Cout, k, 1, Cin, Cout, 0, st) == 0
        torch.cuda.synchronize()
        outs.append(y.float().clone())
    diffs = " ".join(f"{(o - outs[0]).abs().max().item():.2e}/{float((o != outs[0]).float().mean()):.3f}" for o in outs[1:])
    print((B, H, W, Cin, Cout, k), "  ".join(f"{n}: {min(r):6.1f} us ({fl / min(r) / 1e6:5.0f} TF/s)" for n, r in zip(names, res)), " | max diff / fraction differing:", diffs, flush=True)
