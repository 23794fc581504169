"""GPU parity: MFMA implicit-GEMM conv + BN/SiLU + NHWC helpers vs the CPU oracle (oracle/tscd_oracle.py).

Inputs/weights are rounded to fp16 first so both sides see identical operands; the kernel accumulates in fp32
and stores fp16, so the tolerance is fp16 output rounding: |err| <= 2e-3 * max|ref| (+ 1e-3 abs).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import tscd_oracle as O


def dev():
    return torch.device("cuda:0")


def assert_close(got, ref, rel=2e-3, abs_=1e-3):
    got, ref = got.float().cpu(), ref.float().cpu()
    tol = rel * ref.abs().max().item() + abs_
    err = (got - ref).abs().max().item()
    assert err <= tol, f"max err {err} > tol {tol}"


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride
    (2, 16, 16, 16, 64, 3, 1),      # Focus-like Cin=16 (K=144 -> pad 160)
    (2, 18, 18, 64, 128, 3, 2),     # stride-2 downsample
    (3, 9, 13, 128, 128, 1, 1),     # 1x1, ragged M
    (2, 12, 12, 256, 256, 3, 1),    # head tower shape (W=256)
    (1, 8, 8, 40, 25, 1, 1),        # Cin not multiple of 32, Cout=25 (cls pred)
    (2, 7, 5, 8, 8, 3, 1),          # tiny golden-model width
    (1, 1, 300, 96, 200, 1, 1),     # linear layer as 1x1 conv
    (2, 6, 6, 1024, 512, 1, 1),     # deep K
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_raw_matches_oracle(case):
    from tscd_amd import ops
    B, H, W, Cin, Cout, k, s = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, Cin, H, W, generator=g).half()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).half()
    ref = F.conv2d(x.float(), w.float(), None, s, (k - 1) // 2)
    xg = x.to(dev()).permute(0, 2, 3, 1).contiguous()
    wpk = ops.pack_conv_weight(w.to(dev()))
    y, _ = ops.conv2d(xg, wpk, Cout, k, s)
    torch.cuda.synchronize()
    assert_close(y.permute(0, 3, 1, 2), ref)


def test_conv_channel_slices_and_bias_relu():
    from tscd_amd import ops
    g = torch.Generator().manual_seed(3)
    B, H, W, Cin, Cout = 2, 10, 10, 32, 48
    x = torch.randn(B, Cin, H, W, generator=g).half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 17).half()
    b = torch.randn(Cout, generator=g)
    ref = F.relu(F.conv2d(x.float(), w.float(), b, 1, 1))
    big_in = torch.zeros(B, H, W, 80, dtype=torch.float16, device=dev())
    big_in[..., 16:48] = x.to(dev()).permute(0, 2, 3, 1)
    big_out = torch.full((B, H, W, 96), 7.0, dtype=torch.float16, device=dev())
    ops.conv2d(big_in[..., 16:48], ops.pack_conv_weight(w.to(dev())), Cout, 3, 1, ops.EPI_BIAS_RELU, bias=b.to(dev()),
               out=big_out[..., 8:56])
    torch.cuda.synchronize()
    assert_close(big_out[..., 8:56].permute(0, 3, 1, 2), ref)
    assert (big_out[..., :8] == 7).all() and (big_out[..., 56:] == 7).all()  # neighbours untouched


@pytest.mark.parametrize("training", [True, False])
def test_base_conv_bn_silu(training):
    """BaseConv (network_blocks.py:36-61) in train mode (batch stats + running-stat update) and eval (folded)."""
    from tscd_amd import ops
    g = torch.Generator().manual_seed(11)
    B, H, W, Cin, Cout = 4, 20, 20, 64, 96
    x = torch.randn(B, Cin, H, W, generator=g).half()
    sd = {"c.conv.weight": (torch.randn(Cout, Cin, 3, 3, generator=g) / 24).half().float(),
          "c.bn.weight": 0.5 + torch.rand(Cout, generator=g), "c.bn.bias": 0.3 * torch.randn(Cout, generator=g),
          "c.bn.running_mean": 0.1 * torch.randn(Cout, generator=g), "c.bn.running_var": 0.5 + torch.rand(Cout, generator=g)}
    sd_ref = {k: v.clone() for k, v in sd.items()}
    ref = O.base_conv(sd_ref, "c", x.float(), 1, training)
    d = dev()
    xg = x.to(d).permute(0, 2, 3, 1).contiguous()
    wpk = ops.pack_conv_weight(sd["c.conv.weight"].to(d))
    gam, bet = sd["c.bn.weight"].to(d), sd["c.bn.bias"].to(d)
    rm, rv = sd["c.bn.running_mean"].to(d).clone(), sd["c.bn.running_var"].to(d).clone()
    if training:
        y, part = ops.conv2d(xg, wpk, Cout, 3, 1, ops.EPI_RAW, want_stats=True)
        ss, mean, invstd = ops.bn_finalize(part, B * H * W, gam, bet, rm, rv)
        ops.bn_act(y, ss, ops.ACT_SILU)
        torch.cuda.synchronize()
        np.testing.assert_allclose(rm.cpu().numpy(), sd_ref["c.bn.running_mean"].numpy(), atol=2e-4, rtol=1e-3)
        np.testing.assert_allclose(rv.cpu().numpy(), sd_ref["c.bn.running_var"].numpy(), atol=2e-4, rtol=1e-3)
    else:
        ss = ops.bn_fold(gam, bet, rm, rv)
        y, _ = ops.conv2d(xg, wpk, Cout, 3, 1, ops.EPI_BN_SILU, scale_shift=ss)
        torch.cuda.synchronize()
    assert_close(y.permute(0, 3, 1, 2), ref, rel=3e-3, abs_=2e-3)


def test_focus_pool_upsample_haar():
    from tscd_amd import ops
    g = torch.Generator().manual_seed(5)
    d = dev()
    img = torch.randint(0, 256, (2, 3, 16, 24), generator=g).float()
    f = ops.focus_to_nhwc16(img.to(d))
    tl, tr, bl, br = img[..., ::2, ::2], img[..., ::2, 1::2], img[..., 1::2, ::2], img[..., 1::2, 1::2]
    ref = torch.cat((tl, bl, tr, br), 1)
    torch.cuda.synchronize()
    assert torch.equal(f[..., :12].permute(0, 3, 1, 2).float().cpu(), ref) and (f[..., 12:] == 0).all()

    x = torch.randn(2, 16, 10, 14, generator=g).half()
    xg = x.to(d).permute(0, 2, 3, 1).contiguous()
    buf = torch.zeros(2, 10, 14, 64, dtype=torch.float16, device=d)
    ops.copy_slice(xg, buf[..., :16])
    ops.maxpool(buf[..., :16], buf[..., 16:32])
    ops.maxpool(buf[..., 16:32], buf[..., 32:48])
    ops.maxpool(buf[..., 32:48], buf[..., 48:64])
    torch.cuda.synchronize()
    for i, ks in enumerate((5, 9, 13)):   # SPP 5/9/13 == the 5-pool applied 1x/2x/3x (network_blocks.py:201-223)
        ref = F.max_pool2d(x.float(), ks, 1, ks // 2)
        assert torch.equal(buf[..., 16 * (i + 1):16 * (i + 2)].permute(0, 3, 1, 2).float().cpu(), ref)

    up = torch.zeros(2, 20, 28, 16, dtype=torch.float16, device=d)
    ops.copy_slice(xg, up, upsample=True)
    torch.cuda.synchronize()
    assert torch.equal(up.permute(0, 3, 1, 2).float().cpu(), F.interpolate(x.float(), scale_factor=2, mode="nearest"))

    hf = ops.haar_hf(xg)
    ref_dwt = O.haar_dwt(x.float())
    torch.cuda.synchronize()
    assert_close(hf.permute(0, 3, 1, 2), ref_dwt[:, 16:], rel=1e-3, abs_=1e-3)
    content = torch.randn(2, 16, 10, 14, generator=g).half()
    hf_in = ref_dwt[:, 16:].half()
    out = ops.haar_idwt_mul(hf_in.to(d).permute(0, 2, 3, 1).contiguous(), content.to(d).permute(0, 2, 3, 1).contiguous())
    ref = content.float() * O.haar_idwt(torch.cat((torch.zeros(2, 16, 5, 7), hf_in.float()), 1))
    torch.cuda.synchronize()
    assert_close(out.permute(0, 3, 1, 2), ref, rel=2e-3, abs_=2e-3)


@pytest.mark.parametrize("case", [(5, 16, 16, 32, 32, 1), (5, 4, 4, 32, 8, 1), (3, 12, 12, 32, 32, 3), (2, 20, 20, 256, 256, 3),
                                  (1, 1, 300, 96, 200, 1), (2, 9, 7, 128, 96, 3)])
def test_conv_wgrad_and_dgrad(case):
    """Backward kernels of the trainable head convs vs PyTorch autograd of F.conv2d on the same fp16 operands (fp32
    math): weight gradient (pixel-axis GEMM with ds_read_b64_tr_b16) 1e-3 relative, data gradient 2e-3."""
    from tscd_amd import ops
    B, H, W, Ci, Co, k = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, H, W, Ci, generator=g).half()
    dy = torch.randn(B, H, W, Co, generator=g).half()
    wt = (torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).half()
    xr = x.permute(0, 3, 1, 2).float().requires_grad_(True)
    w = wt.float().requires_grad_(True)
    F.conv2d(xr, w, None, 1, (k - 1) // 2).backward(dy.permute(0, 3, 1, 2).float())
    dw = ops.conv_wgrad(x.to(dev()), dy.to(dev()), Ci, Co, k)
    dx = ops.conv_dgrad(dy.to(dev()), wt.to(dev()))
    torch.cuda.synchronize()
    assert_close(dw, w.grad, rel=1e-3, abs_=1e-3)
    assert_close(dx.permute(0, 3, 1, 2), xr.grad, rel=2e-3, abs_=2e-3)


@pytest.mark.parametrize("case", [(2, 37, 41, 64, 30, False), (3, 20, 20, 32, 4, False), (1, 50, 30, 64, 1, False), (9, 72, 72, 256, 256, True),
                                  (2, 11, 13, 32, 40, True)])
def test_conv_bias_backward_matches_autograd(case):
    """Biased 1x1 conv (prediction convs: 30 / 4 / 1 output channels, padded to 8 for the kernels) and the fused-ReLU form:
    bias, weight and data gradients vs PyTorch autograd in fp32 on the same fp16 operands.  (9,72,72,256) makes the bias /
    mask kernel loop grid-strided (46656 rows > 1024 blocks x 32 rows)."""
    from tscd_amd import functional as TF
    B, H, W, Ci, Co, relu = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn(B, H, W, Ci, generator=g).half()
    wt = (torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5).half()
    bias = torch.randn(Co, generator=g).half()
    gy = torch.randn(B, H, W, Co, generator=g).half()
    xr, wr, br = x.float().requires_grad_(True), wt.float().requires_grad_(True), bias.float().requires_grad_(True)
    xg, wg, bg = x.to(dev()).requires_grad_(True), wt.to(dev()).requires_grad_(True), bias.to(dev()).requires_grad_(True)
    out = TF.conv_bias(xg, wg, bg, relu=relu)
    (out.float() * gy.to(dev()).float()).sum().backward()
    torch.cuda.synchronize()
    ref = F.conv2d(xr.permute(0, 3, 1, 2), wr, br).permute(0, 2, 3, 1)
    if relu:      # the product's own activation pattern (a pre-activation within fp16 rounding of 0 may land on either side)
        ref = ref * (out.detach().float().cpu() > 0)
    (ref * gy.float()).sum().backward()
    assert_close(out, ref.detach(), rel=3e-3, abs_=3e-3)
    scale = max(1.0, float(br.grad.abs().max()))
    assert_close(bg.grad.float() / scale, br.grad / scale, rel=2e-3, abs_=2e-3)
    wscale = max(1.0, float(wr.grad.abs().max()))
    assert_close(wg.grad.float() / wscale, wr.grad / wscale, rel=2e-3, abs_=2e-3)
    assert_close(xg.grad, xr.grad, rel=3e-3, abs_=3e-3)


WGRAD_CASES = [(2, 36, 36, 256, 256, 3), (1, 9, 9, 768, 768, 1), (2, 9, 7, 128, 96, 3), (1, 1, 300, 96, 200, 1), (3, 5, 5, 8, 256, 3),
               (16, 18, 18, 32, 256, 1), (1, 1, 2000, 1024, 32, 1), (2, 3, 2, 64, 64, 3), (1, 40, 40, 128, 128, 3), (5, 4, 4, 32, 8, 1)]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad_lds_dma_matches_register_loop(case):
    """The LDS-DMA weight-gradient loop (swizzled lane-linear tiles, incremental halo test, counted vmcnt ring) against
    the register-staged loop on the same operands: same tiles, splits and k order, so dw must be bit-identical -- images
    smaller than a k-tile, ragged pixel counts, Cin of 8 (a tile spans 16 taps/chunks), splits with a short tail."""
    from tscd_amd import ops
    B, H, W, Ci, Co, k = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, H, W, Ci, generator=g).half().to(dev())
    dy = torch.randn(B, H, W, Co, generator=g).half().to(dev())
    ref = ops.conv_wgrad(x, dy, Ci, Co, k, variant=0).clone()       # per-call plan argument: no process-global switch
    out = ops.conv_wgrad(x, dy, Ci, Co, k).clone()
    torch.cuda.synchronize()
    assert torch.equal(out, ref), float((out - ref).abs().max())


DMA_CASES = [
    # B, H, W, Cin, Cout, k, stride, epilogue-with-stats
    (2, 36, 36, 256, 256, 3, 1),     # head tower level (multi-tile M and N, 72 k-tiles: the ring wraps 18 times)
    (1, 19, 23, 64, 128, 3, 2),      # stride 2, odd extent, ragged M
    (3, 9, 13, 128, 128, 1, 1),      # 1x1, 4 k-tiles
    (2, 5, 7, 32, 64, 1, 1),         # single k-tile (prologue shorter than the ring), BN=64 tile
    (2, 5, 7, 64, 72, 3, 1),         # Cout not a tile multiple
    (1, 6, 6, 1024, 512, 1, 1),      # deep K
    (1, 1, 500, 256, 256, 1, 1),     # linear layer
    (4, 18, 18, 512, 512, 3, 1),     # 18x18 level
    (2, 9, 11, 64, 256, 3, 1),       # one 64-channel chunk per tap, ragged M (128 x 256 variant: 9 k-tiles)
    (1, 5, 5, 128, 384, 1, 1),       # Cout padded to 384: not a multiple of 256 -> stays on the 128-wide tiles
    (2, 8, 8, 192, 256, 3, 2),       # stride 2, 3 chunks per tap
    (2, 72, 72, 256, 256, 3, 1),     # the dominant shape's geometry (72 k-tiles of 32, 81 M-tiles)
    (1, 3, 5, 32, 256, 1, 1),        # a single k-tile on the 256-wide tiles (ring prologue longer than the loop)
    (1, 7, 9, 96, 512, 3, 1),        # 27 k-tiles, two N-tiles of 256, ragged M
]


@pytest.mark.parametrize("case", DMA_CASES)
def test_conv_lds_dma_variant_bit_identical_to_register_staging(case):
    """The LDS-DMA main loop (global_load_lds ring, XOR-swizzled image) must reproduce the register-staged loop bit for bit
    (same MFMA sequence per output), including the BatchNorm partial sums, and both must match the fp32 oracle."""
    from tscd_amd import ops
    from tscd_amd._lib import lib
    B, H, W, Cin, Cout, k, s = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).half()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).half()
    xg = x.to(dev()).permute(0, 2, 3, 1).contiguous()
    wpk = ops.pack_conv_weight(w.to(dev()))
    tall = {}
    # the main-loop plan is a per-call argument (tscd_conv2d_nhwc_f16_ex): nothing process-global is switched
    y0, st0 = ops.conv2d(xg, wpk, Cout, k, s, want_stats=True, variant=0, tile_m=128)
    y1, st1 = ops.conv2d(xg, wpk, Cout, k, s, want_stats=True, variant=1, tile_m=128)       # 128-wide LDS-DMA tiles only
    for bm in (64, 192):                      # the other tile heights of the 128-wide LDS-DMA variant
        tall[bm] = ops.conv2d(xg, wpk, Cout, k, s, want_stats=True, variant=1, tile_m=bm)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    assert torch.equal(st0, st1)
    for bm, (yb, stb) in tall.items():            # same MFMA sequence per output; the statistics are grouped by other row blocks
        assert torch.equal(y0, yb), bm
        M = B * yb.shape[1] * yb.shape[2]
        if lib.tscd_conv_tile_n(Cout) == 128 and Cin % 32 == 0:
            assert stb.shape[-1] == (M + bm - 1) // bm, (bm, stb.shape)
        assert torch.allclose(stb.sum(-1), st0.sum(-1), rtol=2e-5, atol=1e-3), bm
    ref = F.conv2d(x.float(), w.float(), None, s, (k - 1) // 2)
    assert_close(y1.permute(0, 3, 1, 2), ref)


def test_ema_update_matches_foreach_formulation():
    """One-launch EMA (tscd_ema_update_f32) against the reference formulation v *= d; v += (1-d) * src on the same tensors:
    odd sizes, a tensor spanning several chunks, an unaligned view -- bit-identical, src untouched, table cache reused and rebuilt."""
    from tscd_amd import ops
    g = torch.Generator().manual_seed(5)
    sizes = [1, 7, 64, 1000, 16384, 16385, 70001, 3]
    base = torch.randn(sum(sizes) + 1, generator=g).to(dev())
    dst = [torch.randn(n, generator=g).to(dev()) for n in sizes]
    src = [torch.randn(n, generator=g).to(dev()) for n in sizes]
    dst[1] = base[1:8]                      # 4-byte aligned only
    src0 = [t.clone() for t in src]
    for d in (0.37, 0.9998):
        ref = [t.clone() for t in dst]
        torch._foreach_mul_(ref, d)
        torch._foreach_add_(ref, src, alpha=1.0 - d)
        ops.ema_update(dst, src, d)
        torch.cuda.synchronize()
        for a, b in zip(dst, ref):
            assert torch.equal(a, b)
    for a, b in zip(src, src0):
        assert torch.equal(a, b)
    src[2] = src[2].clone()                 # a moved tensor: the cached table must be rebuilt, not reused
    ref = [t.clone() for t in dst]
    torch._foreach_mul_(ref, 0.5)
    torch._foreach_add_(ref, src, alpha=0.5)
    ops.ema_update(dst, src, 0.5)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(dst, ref))


HALO_CASES = [
    (2, 72, 72, 256, 256),       # the dominant shape's geometry: 41 tiles of 256 pixels, tiles span image rows and frames
    (3, 36, 36, 256, 256),       # M = 3888: the last tile is ragged
    (1, 18, 18, 512, 512),       # two N-tiles, 16 channel chunks
    (2, 8, 8, 32, 256),          # one channel chunk (no halo double-buffering), M = 128 < one tile
    (1, 5, 90, 64, 256),         # wide, short image (W = 90: the largest halo class), 2 chunks
    (5, 4, 3, 96, 256),          # tiny images: every pixel is an edge pixel, a tile spans all five frames
]


@pytest.mark.parametrize("case", HALO_CASES)
def test_conv_halo_variant_matches_chunk_major_register_loop_bit_for_bit(case):
    """Variant 5 (input halo staged once per channel chunk, taps read it at row shifts, out-of-image taps redirected to a zero
    block) vs the register-staged loop walking k in the same (channel chunk, tap) order: identical MFMA sequence per output =>
    bit-identical outputs; BatchNorm partial sums agree to fp32 rounding (the 256-row tile sums its rows in another order);
    both match the fp32 oracle.  Also: variant 5 vs the default k order within fp32-accumulation noise."""
    from tscd_amd import ops
    from tscd_amd._lib import lib
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).half()
    xg = x.to(dev()).permute(0, 2, 3, 1).contiguous()
    wpk = ops.pack_conv_weight(w.to(dev()))
    y_def, st_def = ops.conv2d(xg, wpk, Cout, 3, 1, want_stats=True, variant=0)
    y0, st0 = ops.conv2d(xg, wpk, Cout, 3, 1, want_stats=True, variant=0, korder=1)
    y5, st5 = ops.conv2d(xg, wpk, Cout, 3, 1, want_stats=True, variant=5)                # 256-pixel tiles
    yb = ops.conv2d(xg, wpk, Cout, 3, 1, epilogue=ops.EPI_BIAS_RELU, bias=torch.linspace(-1, 1, Cout, device=dev()), variant=5)[0]
    y6, st6 = ops.conv2d(xg, wpk, Cout, 3, 1, want_stats=True, variant=6)                # 192-pixel tiles
    torch.cuda.synchronize()
    assert torch.equal(y0, y5) and torch.equal(y0, y6)
    # per-channel sums of the same fp16 outputs, tiles of another height: compare the totals over the M-tiles
    tot = lambda st: st.double().sum(-1)
    assert torch.allclose(tot(st0), tot(st5), rtol=1e-6, atol=1e-5) and torch.allclose(tot(st0), tot(st6), rtol=1e-6, atol=1e-5)
    assert (y_def.float() - y5.float()).abs().max().item() <= 2e-3 * y_def.float().abs().max().item()
    ref = F.conv2d(x.float(), w.float(), None, 1, 1)
    assert_close(y5.permute(0, 3, 1, 2), ref)
    refb = torch.relu(ref + torch.linspace(-1, 1, Cout).view(1, -1, 1, 1))
    assert_close(yb.permute(0, 3, 1, 2), refb)


GROUP_CASES = [
    # B, H, W, Cin (per group), Cout (per group), k, groups
    (2, 36, 36, 256, 256, 3, 2),     # the trainable towers' second convs on the 36x36 level: halo-staged form, 2 groups
    (2, 18, 18, 256, 256, 3, 4),     # all four towers' second convs, 64-row LDS-DMA tiles
    (1, 9, 11, 64, 128, 3, 2),       # small ragged image, LDS-DMA 128-wide tiles
    (2, 7, 5, 32, 320, 1, 2),        # per-group Cout padded to 384 rows (TSCD-X head width)
    (1, 6, 6, 24, 40, 3, 3),         # Cin % 32 != 0: register-staged loop, BN = 64 tiles
    (3, 72, 72, 128, 128, 3, 2),     # TSCD-S head width on the 72x72 level
]


@pytest.mark.parametrize("case", GROUP_CASES)
def test_grouped_conv_equals_the_separate_launches_bit_for_bit(case):
    """groups convs in ONE launch (tscd_conv2d_nhwc_f16_ex: the towers' second convs, reference tscd_head.py:315-324) == the same convs
    launched one by one on the channel slices: outputs bit-identical (the same MFMA sequence per output as the separate launch of the
    same main-loop form), BatchNorm partial sums equal to fp32 rounding of a different tile height, bias epilogue included."""
    from tscd_amd import ops
    B, H, W, Cin, Cout, k, G = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, H, W, G * Cin, generator=g).half().to(dev())
    ws = [(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).half().to(dev()) for _ in range(G)]
    wpk = torch.cat([ops.pack_conv_weight(w) for w in ws], 0).contiguous()
    yg, stg = ops.conv2d(x, wpk, Cout, k, 1, want_stats=True, groups=G)
    bias = torch.linspace(-1, 1, G * Cout, device=dev())
    yb = ops.conv2d(x, wpk, Cout, k, 1, epilogue=ops.EPI_BIAS_RELU, bias=bias, groups=G)[0]
    torch.cuda.synchronize()
    rows = wpk.shape[0] // G
    for i, w in enumerate(ws):
        xi = x[..., i * Cin:(i + 1) * Cin]
        ref = F.conv2d(xi.permute(0, 3, 1, 2).float().cpu(), w.float().cpu(), None, 1, (k - 1) // 2)
        assert_close(yg[..., i * Cout:(i + 1) * Cout].permute(0, 3, 1, 2), ref)
        # the separate launch, forced onto the same main-loop form where the automatic choice would differ with fewer tiles
        outs = [ops.conv2d(xi, ops.pack_conv_weight(w), Cout, k, 1, want_stats=True, variant=v) for v in (-1, 0, 1, 5, 6)]
        torch.cuda.synchronize()
        same = [torch.equal(o[0], yg[..., i * Cout:(i + 1) * Cout]) for o in outs]
        assert any(same), (i, same)
        y1, st1 = outs[same.index(True)]
        assert torch.allclose(stg[:, i * rows:i * rows + Cout].double().sum(-1), st1[:, :Cout].double().sum(-1), rtol=1e-6, atol=1e-5)
        refb = torch.relu(ref + bias[i * Cout:(i + 1) * Cout].cpu().view(1, -1, 1, 1))
        assert_close(yb[..., i * Cout:(i + 1) * Cout].permute(0, 3, 1, 2), refb)


@pytest.mark.parametrize("groups", [1, 2])
@pytest.mark.parametrize("shape", [(2, 36, 36, 256), (3, 18, 18, 128), (2, 10, 14, 64)])
def test_several_baseconvs_in_one_launch_match_the_single_calls(groups, shape):
    """conv_bn_act_train_multi (tscd_conv_bn_act_train_multi_f16): the towers' first convs as ONE dense conv over the concatenated
    output channels (groups=1: shared input) and their second convs as ONE grouped launch (groups=2), each BatchNorm with its own
    gamma / beta / running statistics, one statistics and one apply launch for all -- against conv_bn_act_train called per conv:
    activations bit-identical, batch statistics and running statistics equal to fp32 rounding (tile heights differ)."""
    from tscd_amd import ops
    B, H, W, C = shape
    nseg = 2
    g = torch.Generator().manual_seed(sum(shape) + groups)
    x = torch.randn(B, H, W, C * (groups if groups > 1 else 1), generator=g).half().to(dev())
    ws = [(torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).half().to(dev()) for _ in range(nseg)]
    mk = lambda: [((torch.rand(C, generator=g) + 0.5).to(dev()), (torch.randn(C, generator=g) * 0.1).to(dev()),
                   torch.zeros(C, device=dev()), torch.ones(C, device=dev())) for _ in range(nseg)]
    bns = mk()
    bns_ref = [tuple(t.clone() for t in b) for b in bns]
    if groups == 1:
        wpk = ops.pack_conv_weight(torch.cat(ws, 0))
    else:
        wpk = torch.cat([ops.pack_conv_weight(w) for w in ws], 0).contiguous()
    out, raw, ss, mean, invstd = ops.conv_bn_act_train_multi(x, wpk, C, bns, 3, groups, keep_raw=True)
    torch.cuda.synchronize()
    for i, w in enumerate(ws):
        xi = x if groups == 1 else x[..., i * C:(i + 1) * C]
        gm, bt, rm, rv = bns_ref[i]
        o1, r1, ss1, m1, iv1 = ops.conv_bn_act_train(xi, ops.pack_conv_weight(w), C, 3, 1, gm, bt, rm, rv, keep_raw=True)
        torch.cuda.synchronize()
        sl = slice(i * C, (i + 1) * C)
        assert torch.equal(raw[..., sl], r1) or (raw[..., sl].float() - r1.float()).abs().max() <= 2e-3 * r1.float().abs().max()
        assert torch.allclose(mean[sl], m1, rtol=1e-5, atol=1e-6) and torch.allclose(invstd[sl], iv1, rtol=1e-5, atol=1e-6)
        assert torch.allclose(ss[:, sl], ss1, rtol=1e-5, atol=1e-6)
        assert torch.allclose(bns[i][2], rm, rtol=1e-5, atol=1e-7) and torch.allclose(bns[i][3], rv, rtol=1e-5, atol=1e-7)
        assert (out[..., sl].float() - o1.float()).abs().max() <= 4e-3 * o1.float().abs().max()


@pytest.mark.parametrize("case", [(2, 36, 36, 256, 256, 3, 2), (1, 9, 7, 64, 96, 3, 2), (2, 18, 18, 128, 128, 1, 3)])
def test_grouped_wgrad_and_two_segment_bn_backward(case):
    """tscd_conv_wgrad_nhwc_f16_ex with groups == the per-group calls on the channel slices (bit-identical when the split count is
    the same, fp32-sum tolerance otherwise); bn_silu_bwd_multi over two gradient tensors and a strided raw slice == the plain
    kernel on the concatenated contiguous tensors, bit for bit."""
    from tscd_amd import ops
    B, H, W, Ci, Co, k, G = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, H, W, G * Ci, generator=g).half().to(dev())
    dy = torch.randn(B, H, W, G * Co, generator=g).half().to(dev())
    dw = ops.conv_wgrad(x, dy, Ci, Co, k, groups=G)
    torch.cuda.synchronize()
    assert dw.shape == (G * Co, Ci, k, k)
    for i in range(G):
        ref = ops.conv_wgrad(x[..., i * Ci:(i + 1) * Ci], dy[..., i * Co:(i + 1) * Co], Ci, Co, k)
        torch.cuda.synchronize()
        got = dw[i * Co:(i + 1) * Co]
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-3 * float(ref.abs().max())), (i, float((got - ref).abs().max()))
    # BatchNorm + SiLU backward over two segments
    C = 2 * Co
    big = torch.randn(B, H, W, C + 16, generator=g).half().to(dev())
    raw = big[..., 8:8 + C]                                      # a channel slice (pixel stride C + 16)
    d0 = torch.randn(B, H, W, Co, generator=g).half().to(dev())
    d1big = torch.randn(B, H, W, Co + 8, generator=g).half().to(dev())
    d1 = d1big[..., :Co]
    gam = [(torch.rand(Co, generator=g) + 0.5).to(dev()) for _ in range(2)]
    ss = torch.stack((torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1)).to(dev())
    mean, invstd = torch.randn(C, generator=g).to(dev()) * 0.1, (torch.rand(C, generator=g) + 0.5).to(dev())
    draw, dg, db = ops.bn_silu_bwd_multi([d0, d1], raw, ss, mean, invstd, gam)
    draw_r, dg_r, db_r = ops.bn_silu_bwd(torch.cat((d0, d1), -1).contiguous(), raw.contiguous(), ss, mean, invstd, torch.cat(gam))
    torch.cuda.synchronize()
    assert torch.equal(draw, draw_r) and torch.equal(dg, dg_r) and torch.equal(db, db_r)
    # one gradient tensor spanning both BatchNorms (the fused first convs' backward): the gamma split is independent of the dY split
    draw2, dg2, db2 = ops.bn_silu_bwd_multi([torch.cat((d0, d1), -1).contiguous()], raw, ss, mean, invstd, gam)
    torch.cuda.synchronize()
    assert torch.equal(draw2, draw_r) and torch.equal(dg2, dg_r) and torch.equal(db2, db_r)


def test_linear_jobs_one_launch_equals_separate_launches():
    """ops.linear_jobs: several independent nn.Linear GEMMs as ONE job-table launch of the 64 x 128 LDS-DMA tile (tscd_conv2d_jobs_nhwc_f16)
    -- the separate launches' bits (an element's K-sum does not depend on which workgroup computes it), with and without bias, into an
    output slice of a wider buffer; a job outside the form (30 outputs) sends the whole call down the one-launch-each path."""
    from tscd_amd import ops
    g = torch.Generator().manual_seed(5)
    dev = torch.device("cuda:0")
    mk = lambda *s: torch.randn(*s, generator=g)
    x1, x2, x3 = mk(2000, 256).half().to(dev), mk(8000, 256).half().to(dev), mk(500, 1024).half().to(dev)
    w1, w2, w3 = (mk(256, 256) * 0.05).to(dev), (mk(512, 256) * 0.05).to(dev), (mk(1024, 1024) * 0.03).to(dev)
    b2, b3 = mk(512).to(dev), mk(1024).half().to(dev)
    wide = torch.zeros(500, 2048, dtype=torch.float16, device=dev)
    jobs = [(x1, w1, None, None), (x2, w2, b2, None), (x3, w3, b3, wide[:, 1024:])]
    got = ops.linear_jobs(jobs)
    want = [ops.linear(x1, w1), ops.linear(x2, w2, b2), ops.linear(x3, w3, b3)]
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert torch.equal(wide[:, 1024:], want[2]) and float(wide[:, :1024].abs().max()) == 0.0
    # a job outside the 128-wide form: every job takes its own launch, same results
    w4 = (mk(30, 256) * 0.05).to(dev)
    got = ops.linear_jobs([(x1, w1, None, None), (x1, w4, None, None)])
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], ops.linear(x1, w4))
