"""GPU parity: CSPDarknet+PAFPN (yolox.models.YOLOPAFPN on HIP kernels) vs the reference's golden FPN maps (eval)
and vs the CPU oracle in train mode (batch-stat BN + running-stat update).

Tolerance: the reference itself runs this path in fp16; its own fp16-vs-fp32 noise on these
fixtures is measured in the test by running the oracle's PyTorch ops in fp16 on the device, and the HIP path
must stay within 1.5x that noise floor (max and mean abs error) and within 10 % of max|ref| absolutely.
Measured on MI355X: HIP path ~0.5x the stock fp16 noise (mean 1.1e-2 vs 2.0e-2 at scale 22).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tscd_oracle as O


def load_backbone(sd, cfg):
    from yolox.models import YOLOPAFPN
    m = YOLOPAFPN(cfg.depth, cfg.width)
    m.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}, strict=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    for p in m.parameters():
        p.requires_grad = False
    return m.cuda()


def fp16_noise_floor(sd, cfg, imgs, training):
    sdh = {k: (v.cuda().half() if v.is_floating_point() else v.cuda()) for k, v in sd.items()}
    with torch.no_grad():
        return O.yolo_pafpn(sdh, imgs.cuda().half(), cfg, training)


def check(got, ref, noise):
    got, ref, noise = got.float().cpu(), torch.as_tensor(ref).float(), noise.float().cpu()
    scale = ref.abs().max().item()
    err, nerr = (got - ref).abs(), (noise - ref).abs()
    assert err.max().item() <= max(1.5 * nerr.max().item(), 2e-3 * scale), (err.max().item(), nerr.max().item(), scale)
    assert err.mean().item() <= max(1.5 * nerr.mean().item(), 5e-4 * scale), (err.mean().item(), nerr.mean().item())
    assert err.max().item() <= 0.1 * scale


def test_backbone_eval_vs_reference_golden(golden):
    cfg, sd, vec, meta = golden
    m = load_backbone(sd, cfg).eval()
    imgs = torch.from_numpy(vec["evalA_imgs"]).float().cuda()
    outs = m(imgs)
    torch.cuda.synchronize()
    noise = fp16_noise_floor(sd, cfg, imgs, False)
    for i in range(3):
        assert outs[i].shape == vec[f"evalA_fpn{i}"].shape
        check(outs[i], vec[f"evalA_fpn{i}"], noise[i])


def test_backbone_half_input_and_half_model(golden):
    cfg, sd, vec, meta = golden
    m = load_backbone(sd, cfg).eval().half()
    imgs = torch.from_numpy(vec["evalA_imgs"]).cuda().half()
    outs = m(imgs)
    torch.cuda.synchronize()
    noise = fp16_noise_floor(sd, cfg, imgs, False)
    for i in range(3):
        check(outs[i], vec[f"evalA_fpn{i}"], noise[i])


def test_backbone_train_mode_vs_oracle(golden):
    cfg, sd, vec, meta = golden
    m = load_backbone(sd, cfg).train()
    imgs = torch.from_numpy(vec["evalA_imgs"]).float()
    sd_ref = {k: v.clone() for k, v in sd.items()}
    ref = O.yolo_pafpn(sd_ref, imgs, cfg, True)
    outs = m(imgs.cuda())
    torch.cuda.synchronize()
    noise = fp16_noise_floor({k: v.clone() for k, v in sd.items()}, cfg, imgs, True)
    for i in range(3):
        check(outs[i], ref[i], noise[i])
    got = m.state_dict()
    for k in ("backbone.stem.conv.bn.running_mean", "backbone.dark5.0.bn.running_var", "C3_n4.conv3.bn.running_var",
              "backbone.dark3.1.m.2.conv2.bn.running_mean"):
        np.testing.assert_allclose(got[k].cpu().numpy(), sd_ref["backbone." + k].numpy(), rtol=2e-2, atol=2e-3)
    assert int(got["backbone.stem.conv.bn.num_batches_tracked"]) == 1


def test_stride2_conv_maxpool_upsample_backward_match_autograd():
    """Row N4 kernels one by one against torch autograd in fp32: the stride-2 conv gradients through the zero-inserted dY, the SPP
    max-pool gradient (ties: first maximum in scan order) and the nearest-upsample gradient."""
    import torch.nn.functional as F
    from tscd_amd import functional as TF
    g = torch.Generator().manual_seed(0)
    # stride-2 BaseConv in training mode (conv -> batch-stat BN -> SiLU)
    x = torch.randn(3, 32, 20, 28, generator=g)
    w = torch.randn(64, 32, 3, 3, generator=g) / 17
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    gy = torch.randn(3, 64, 10, 14, generator=g)
    xr, wr, gr, br = (t.clone().requires_grad_() for t in (x.half().float(), w, gamma, beta))
    y_ref = F.silu(F.batch_norm(F.conv2d(xr, wr, None, 2, 1), None, None, gr, br, True, 0.03, 1e-3))
    y_ref.backward(gy)
    xd = x.half().cuda().permute(0, 2, 3, 1).contiguous().requires_grad_()
    wd, gd, bd = (t.clone().cuda().requires_grad_() for t in (w, gamma, beta))
    rm, rv = torch.zeros(64).cuda(), torch.ones(64).cuda()
    y = TF.conv_bn_silu(xd, wd, gd, bd, rm, rv, 2, True, eps=1e-3, momentum=0.03)
    y.backward(gy.cuda().permute(0, 2, 3, 1).half())
    torch.cuda.synchronize()
    cos = lambda a, b: torch.nn.functional.cosine_similarity(a.float().cpu().flatten(), b.float().flatten(), dim=0).item()
    assert cos(y.permute(0, 3, 1, 2), y_ref.detach()) > 0.9995
    for name, got, want in (("dx", xd.grad.permute(0, 3, 1, 2), xr.grad), ("dw", wd.grad, wr.grad), ("dgamma", gd.grad, gr.grad), ("dbeta", bd.grad, br.grad)):
        c = cos(got, want)
        assert c > 0.999 and abs(float(got.float().norm().cpu() / want.norm()) - 1) < 0.02, (name, c)
    # max pool 5x5 stride 1 with deliberate ties
    xp = torch.randint(0, 6, (2, 16, 9, 11), generator=g).float()
    gp = torch.randn(2, 16, 9, 11, generator=g)
    xpr = xp.clone().requires_grad_()
    F.max_pool2d(xpr, 5, 1, 2).backward(gp)
    xpd = xp.half().cuda().permute(0, 2, 3, 1).contiguous().requires_grad_()
    TF.maxpool5(xpd).backward(gp.cuda().permute(0, 2, 3, 1).half())
    assert torch.allclose(xpd.grad.permute(0, 3, 1, 2).float().cpu(), xpr.grad, atol=2e-2, rtol=2e-3)
    # nearest upsample x2
    xu = torch.randn(2, 8, 5, 7, generator=g)
    gu = torch.randn(2, 8, 10, 14, generator=g)
    xur = xu.clone().requires_grad_()
    F.interpolate(xur, scale_factor=2, mode="nearest").backward(gu)
    xud = xu.half().cuda().permute(0, 2, 3, 1).contiguous().requires_grad_()
    TF.upsample2x(xud).backward(gu.cuda().permute(0, 2, 3, 1).half())
    assert torch.allclose(xud.grad.permute(0, 3, 1, 2).float().cpu(), xur.grad, atol=1e-2, rtol=2e-3)


def test_backbone_and_neck_backward_vs_oracle_autograd(golden):
    """Row N4: the whole CSPDarknet + PAFPN with every parameter trainable, forward + backward on the HIP kernels against the fp32
    oracle's autograd on the same weights and clip (train-mode BatchNorm)."""
    cfg, sd, vec, meta = golden
    from yolox.models import YOLOPAFPN
    m = YOLOPAFPN(cfg.depth, cfg.width)
    m.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}, strict=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    m = m.cuda().train()
    imgs = torch.from_numpy(vec["evalA_imgs"]).float()
    sd_ref = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items() if k.startswith("backbone.")}
    ref = O.yolo_pafpn(sd_ref, imgs, cfg, True)
    g = torch.Generator().manual_seed(5)
    gs = [torch.randn(r.shape, generator=g) for r in ref]
    sum((r * gg).sum() for r, gg in zip(ref, gs)).backward()
    outs = m(imgs.cuda())
    assert all(o.requires_grad for o in outs)
    sum((o.float() * gg.cuda()).sum() for o, gg in zip(outs, gs)).backward()
    torch.cuda.synchronize()
    for o, r in zip(outs, ref):
        c = torch.nn.functional.cosine_similarity(o.detach().float().cpu().flatten(), r.detach().flatten(), dim=0).item()
        assert c > 0.997, c            # fp16 activations through ~60 train-mode BatchNorm layers (the fp16 oracle run: same order)
    # the reference's own fp16 noise on these gradients: the oracle's ops in fp16 on the device, same upstream gradients
    sdh = {k: (v.detach().cuda().half().requires_grad_() if v.is_floating_point() and "running" not in k else v.detach().cuda().clone())
           for k, v in sd.items() if k.startswith("backbone.")}
    sdh = {k: (v.half() if v.is_floating_point() and not v.requires_grad else v) for k, v in sdh.items()}
    refh = O.yolo_pafpn(sdh, imgs.cuda().half(), cfg, True)
    sum((r.float() * gg.cuda()).sum() for r, gg in zip(refh, gs)).backward()
    cosf = lambda a, b: torch.nn.functional.cosine_similarity(a.float().cpu().flatten(), b.float().cpu().flatten(), dim=0).item()
    rows = []
    for k, p in m.named_parameters():
        want = sd_ref["backbone." + k].grad
        assert p.grad is not None and want is not None, k
        rows.append((cosf(p.grad, want), cosf(sdh["backbone." + k].grad, want), float(p.grad.float().norm().cpu() / (want.norm() + 1e-20)), k))
    rows.sort()
    print("worst gradient cosines (HIP, fp16 oracle run, norm ratio) over", len(rows), "parameters:", rows[:6])
    for c, cfloor, ratio, k in rows:
        # fp16 activations through ~60 layers of batch-statistics BatchNorm on 8-channel layers: the rule of SURVEY 8d -- within twice
        # the reference's own fp16-vs-fp32 error -- applied to 1 - cos
        assert 1 - c <= 2 * (1 - cfloor) + 3e-2, (k, c, cfloor)
        assert 0.75 < ratio < 1.33, (k, ratio)
    # and on average the HIP path is no noisier than the fp16 oracle run (measured: 0.0099 against 0.0191)
    mean_hip, mean_floor = sum(1 - r[0] for r in rows) / len(rows), sum(1 - r[1] for r in rows) / len(rows)
    print("mean 1 - cos: HIP", mean_hip, "fp16 oracle run", mean_floor)
    assert mean_hip <= 1.5 * mean_floor + 1e-3


@pytest.mark.parametrize("training", [False, True])
def test_csp_conv1_conv2_as_one_launch_equals_two_launches(golden, training):
    """CSPLayer's conv1 || conv2 as ONE launch over the concatenated output channels (network_blocks.FUSE_CSP, tower_fusion.
    shared_input_pair_nograd) against the two separate launches.  Evaluation mode: bit-identical pyramid (every output channel sums
    its K in the same order).  Training mode: the conv outputs are the same bits, but the batch statistics are summed over the launch's
    M-tiles and the wider launch takes another tile height, so mean / variance differ in their last fp32 bits and fp16 roundings flip
    downstream: the first CSP stage's running statistics agree to 1e-5 relative, the pyramid within the stock fp16 noise bound."""
    from yolox.models import network_blocks as nb
    cfg, sd, vec, meta = golden
    imgs = torch.from_numpy(vec["evalA_imgs"]).float().cuda()
    res = {}
    for fuse in (True, False):
        nb.FUSE_CSP = fuse
        try:
            m = load_backbone(sd, cfg)
            m = m.train() if training else m.eval()
            with torch.no_grad():
                outs = [o.clone() for o in m(imgs)]
            torch.cuda.synchronize()
            res[fuse] = (outs, {k: v.clone() for k, v in m.state_dict().items() if "running" in k})
        finally:
            nb.FUSE_CSP = True
    if not training:
        for a, b in zip(res[True][0], res[False][0]):
            assert torch.equal(a, b)
        return
    for a, b in zip(res[True][0], res[False][0]):
        scale = float(b.float().abs().max())
        assert float((a.float() - b.float()).abs().max()) <= 3e-2 * scale and float((a.float() - b.float()).abs().mean()) <= 2e-3 * scale
    first = [k for k in res[True][1] if k.startswith("backbone.dark2.1.conv1.bn.") or k.startswith("backbone.dark2.1.conv2.bn.")]
    assert len(first) == 4, first
    for k in first:
        a, b = res[True][1][k].double(), res[False][1][k].double()
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-9, k
