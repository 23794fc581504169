"""TEST INFRASTRUCTURE (build container only) -- goldens of the YOLOX base detector from the reference ITSELF.

Run:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_yolox

Builds the reference's ``YOLOX(YOLOPAFPN(depth .33, width .125), YOLOXHead(5, width .125))`` (yolox/models/yolox.py, yolo_head.py,
BatchNorm eps 1e-3 / momentum 0.03 as the exps set them), runs ONE training forward + backward on CPU in fp32 on two synthetic
128x128 images with three boxes each -- once without and once with the L1 term -- and an evaluation forward, and commits
inputs, weights, the loss tuples, the gradients of a dozen parameters and the decoded evaluation output as
``tests/golden/yolox_tiny.npz``.  Only data is written; no reference source or bytecode is copied."""
import os
import sys

sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")

GRAD_KEYS = ["backbone.backbone.stem.conv.conv.weight", "backbone.backbone.dark2.0.conv.weight", "backbone.backbone.dark3.1.m.0.conv2.conv.weight",
             "backbone.backbone.dark5.1.conv2.conv.weight", "backbone.C3_p3.conv3.conv.weight", "backbone.bu_conv2.conv.weight",
             "backbone.C3_n4.conv1.bn.weight", "head.stems.0.conv.weight", "head.cls_convs.1.1.conv.weight", "head.reg_convs.2.0.bn.bias",
             "head.cls_preds.0.weight", "head.reg_preds.1.weight", "head.reg_preds.1.bias", "head.obj_preds.2.bias"]


def main():
    sys.path.insert(0, HERE)
    import ref_loader
    ref_loader.import_reference()
    from yolox.models import YOLOX, YOLOPAFPN, YOLOXHead

    torch.set_num_threads(8)
    torch.manual_seed(7)
    nc = 5
    model = YOLOX(YOLOPAFPN(0.33, 0.125), YOLOXHead(nc, 0.125))
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    model.head.initialize_biases(1e-2)
    g = torch.Generator().manual_seed(3)
    # make the random-init network less degenerate: spread the prediction weights so that SimOTA sees varied boxes / scores
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "_preds" in k and k.endswith("weight"):
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    imgs = torch.rand(2, 3, 128, 128, generator=g) * 255
    labels = torch.zeros(2, 10, 5)
    labels[0, :3] = torch.tensor([[1, 40.0, 44.0, 50.0, 36.0], [3, 90.0, 80.0, 40.0, 60.0], [0, 64.0, 30.0, 100.0, 28.0]])
    labels[1, :3] = torch.tensor([[2, 30.0, 90.0, 36.0, 44.0], [4, 70.0, 70.0, 90.0, 96.0], [1, 100.0, 24.0, 30.0, 30.0]])
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rec = {"imgs": imgs.numpy(), "labels": labels.numpy(), "num_classes": np.array(nc)}
    for k, v in sd0.items():
        rec["w:" + k] = v.numpy()
    for tag, use_l1 in (("a", False), ("b", True)):
        model.load_state_dict(sd0)
        model.train()
        model.head.use_l1 = use_l1
        model.zero_grad(set_to_none=True)
        out = model(imgs, labels)
        out["total_loss"].backward()
        rec[f"{tag}_losses"] = np.array([float(out[k]) for k in ("total_loss", "iou_loss", "conf_loss", "cls_loss", "l1_loss", "num_fg")], np.float64)
        pd = dict(model.named_parameters())
        for k in GRAD_KEYS:
            rec[f"{tag}_grad:{k}"] = pd[k].grad.numpy().copy()
        print(tag, "losses", rec[f"{tag}_losses"])
    rec["a_running_mean:backbone.backbone.stem.conv.bn"] = model.state_dict()["backbone.backbone.stem.conv.bn.running_mean"].numpy().copy()
    model.load_state_dict(sd0)
    model.eval()
    with torch.no_grad():
        rec["eval_out"] = model(imgs).numpy()
    np.savez_compressed(os.path.join(OUT, "yolox_tiny.npz"), **rec)
    print("wrote", os.path.join(OUT, "yolox_tiny.npz"), os.path.getsize(os.path.join(OUT, "yolox_tiny.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
