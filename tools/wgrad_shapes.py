import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tscd_amd import ops
dev = torch.device("cuda:0")
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
B = 16
for (H, cin, cout, groups) in [(72, 256, 256, 1), (72, 256, 512, 1), (72, 256, 256, 2), (36, 256, 256, 1), (36, 256, 512, 1), (36, 256, 256, 2), (18, 256, 512, 1), (18, 256, 256, 2)]:
    x = torch.randn(B, H, H, cin * groups, device=dev).half()
    dy = torch.randn(B, H, H, cout * groups, device=dev).half()
    t = bench(lambda: ops.conv_wgrad(x, dy, cin, cout, 3, groups=groups))
    fl = 2.0 * B * H * H * 9 * cin * cout * groups
    w = (torch.randn(cout, cin, 3, 3, device=dev) / 48).half()
    if groups == 1:
        wpk = ops.pack_conv_weight(w)
        xx = torch.randn(B, H, H, cin, device=dev).half()
        t2 = bench(lambda: ops.conv2d(xx, wpk, cout, 3, 1))
    else:
        wpk = torch.cat([ops.pack_conv_weight(w) for _ in range(groups)], 0).contiguous()
        xx = torch.randn(B, H, H, cin * groups, device=dev).half()
        t2 = bench(lambda: ops.conv2d(xx, wpk, cout, 3, 1, groups=groups))
    print(f"{H}x{H} {cin}->{cout} x{groups}: wgrad {t:7.1f} us {fl / t / 1e6:6.0f} TFLOP/s | forward conv {t2:7.1f} us {fl / t2 / 1e6:6.0f} TFLOP/s")
