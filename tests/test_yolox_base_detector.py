"""YOLOX base detector: the oracle's restatement against the reference's own numbers (tests/golden/yolox_tiny.npz,
oracle/make_golden_yolox.py: YOLOX(YOLOPAFPN(.33, .125), YOLOXHead(5, .125)) on CPU in fp32), and the HIP product -- backbone, neck and head
forward + backward, SimOTA, fused losses -- against both."""
import os

import numpy as np
import pytest
import torch

from oracle import tscd_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "yolox_tiny.npz")
LOSS_KEYS = ("total_loss", "iou_loss", "conf_loss", "cls_loss", "l1_loss", "num_fg")


def _load():
    z = np.load(G)
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    grads = {t: {k.split(":", 1)[1]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{t}_grad:")} for t in "ab"}
    return z, sd, grads


def _cfg(z):
    return O.TSCDConfig(num_classes=int(z["num_classes"]), depth=0.33, width=0.125)


@pytest.mark.parametrize("tag,use_l1", [("a", False), ("b", True)])
def test_oracle_matches_the_reference_losses_and_gradients(tag, use_l1):
    z, sd, grads = _load()
    cfg = _cfg(z)
    sdr = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    out = O.yolox_forward(sdr, cfg, torch.from_numpy(z["imgs"]), torch.from_numpy(z["labels"]), True, use_l1)
    want = z[f"{tag}_losses"]
    for i, k in enumerate(LOSS_KEYS):
        assert abs(float(out[i]) - want[i]) <= 2e-4 * max(1.0, abs(want[i])), (k, float(out[i]), want[i])
    out[0].backward()
    for k, gw in grads[tag].items():
        got = sdr[k].grad
        assert torch.allclose(got, gw, rtol=2e-3, atol=2e-5 * float(gw.abs().max()) + 1e-9), (k, float((got - gw).abs().max()), float(gw.abs().max()))
    if tag == "a":
        assert torch.allclose(sdr["backbone.backbone.stem.conv.bn.running_mean"], torch.from_numpy(z["a_running_mean:backbone.backbone.stem.conv.bn"]), atol=1e-6)


def test_oracle_matches_the_reference_evaluation_output():
    z, sd, _ = _load()
    with torch.no_grad():
        out = O.yolox_forward({k: v.clone() for k, v in sd.items()}, _cfg(z), torch.from_numpy(z["imgs"]), None, False)
    want = torch.from_numpy(z["eval_out"])
    assert out.shape == want.shape and torch.allclose(out, want, rtol=1e-4, atol=1e-4)


def _product(sd, nc):
    from yolox.models import YOLOX, YOLOPAFPN, YOLOXHead
    m = YOLOX(YOLOPAFPN(0.33, 0.125), YOLOXHead(nc, 0.125))
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    m.load_state_dict(sd, strict=True)
    return m.cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,use_l1", [("a", False), ("b", True)])
def test_product_train_step_matches_reference_and_oracle(tag, use_l1):
    """Losses of the HIP product (fp16 activations) against the reference's fp32 numbers, the assignment identical to the oracle's on
    the product's own decoded outputs, and every golden gradient within the fp16 noise measured by running the oracle in fp16."""
    z, sd, grads = _load()
    nc = int(z["num_classes"])
    m = _product(sd, nc).train()
    m.head.use_l1 = use_l1
    m.head.simota_fp16 = False                     # the goldens are the reference in fp32
    imgs, labels = torch.from_numpy(z["imgs"]).cuda(), torch.from_numpy(z["labels"]).cuda()
    out = m(imgs, labels)
    out["total_loss"].backward()
    torch.cuda.synchronize()
    want = z[f"{tag}_losses"]
    got = [float(out[k]) for k in LOSS_KEYS]
    print(tag, "losses product", got, "reference", list(want))
    assert got[5] == want[5]                                               # same number of foreground anchors per box
    for i, k in enumerate(LOSS_KEYS[:5]):
        assert abs(got[i] - want[i]) <= 0.03 * max(1.0, abs(want[i])), (k, got[i], want[i])
    # fp16 noise floor of the gradients: the oracle's ops in fp16 on the device
    cfg = _cfg(z)
    sdh = {k: (v.cuda().half().requires_grad_() if v.is_floating_point() and "running" not in k else (v.cuda().half() if v.is_floating_point() else v.cuda()))
           for k, v in sd.items()}
    oh = O.yolox_forward(sdh, cfg, imgs.half(), labels.half().float().half(), True, use_l1)
    oh[0].float().backward()
    cos = lambda a, b: torch.nn.functional.cosine_similarity(a.float().cpu().flatten(), b.float().cpu().flatten(), dim=0).item()
    pd = dict(m.named_parameters())
    rows = []
    for k, gw in grads[tag].items():
        assert pd[k].grad is not None, k
        if float(gw.abs().max()) == 0.0:             # e.g. the class predictor of a level without foreground anchors
            assert float(pd[k].grad.abs().max()) == 0.0, k
            continue
        rows.append((cos(pd[k].grad, gw), cos(sdh[k].grad, gw), float(pd[k].grad.float().norm().cpu() / (gw.norm() + 1e-20)), k))
    rows.sort()
    print("gradient cosines (HIP, fp16 oracle run, norm ratio):", rows[:5])
    for c, cfloor, ratio, k in rows:
        assert 1 - c <= 2 * (1 - cfloor) + 3e-2, (k, c, cfloor)
        assert 0.7 < ratio < 1.4, (k, ratio)


@pytest.mark.gpu
def test_product_evaluation_output_matches_reference():
    z, sd, _ = _load()
    m = _product(sd, int(z["num_classes"])).eval()
    with torch.no_grad():
        out = m(torch.from_numpy(z["imgs"]).cuda())
    want = torch.from_numpy(z["eval_out"])
    err = (out.float().cpu() - want).abs()
    assert out.shape == want.shape
    assert float(err[..., :4].max()) <= 0.02 * float(want[..., :4].abs().max()) and float(err[..., 4:].max()) <= 2e-2
