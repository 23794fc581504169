"""TEST INFRASTRUCTURE (build container only) -- SimOTA goldens from the reference ITSELF on fp16 tensors.

Run:  cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_simota

Training runs the reference's head under autocast on fp16 images and fp16 targets (tscd_trainer.py:159-164):
``outputs`` (decoded boxes + logits), the anchor grids and the labels that reach ``TSCDHead.get_fg_idx``
(tscd_head.py:1712-1851) are all fp16 tensors, so the IoU / in-box chain of ``get_assignments`` (:1330-1426) is
evaluated op by op in fp16 while the class cost runs in fp32 (autocast-off region :1390-1397).  That is the mode the
product's ``tscd_simota_assign(emu16=1)`` reproduces.  This script calls the reference's own ``get_fg_idx`` on fp16
CPU tensors (the same torch ops, rounding each result to fp16) for a few synthetic cases and commits inputs + outputs
as ``tests/golden/simota_fp16.npz``:

  case0  the golden clip's decoded outputs (tests/golden/tscd_tiny_*: S=128, 3 boxes/frame), rounded to fp16
  case1  random-init-like outputs at S=256 with heavy near-ties: boxes from exp(N(0,1.5)), 12 boxes/frame incl. duplicates
  case2  bench-like: S=576 (A=6804), 6 boxes/frame, coordinates above 512 px (0.5 px fp16 grid), areas overflowing fp16

Only data is written; no reference source or bytecode is copied.
"""
import os
import sys

sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")


def grids(S, strides, dtype):
    xs, ys, es = [], [], []
    for s in strides:
        h = w = S // s
        yv, xv = torch.meshgrid([torch.arange(h), torch.arange(w)], indexing="ij")
        xs.append(xv.reshape(1, -1).to(dtype))
        ys.append(yv.reshape(1, -1).to(dtype))
        es.append(torch.full((1, h * w), s).to(dtype))
    return xs, ys, es


def synth_outputs(B, S, nc, seed, box_sigma, n_gt, dup=False):
    g = torch.Generator().manual_seed(seed)
    xs, ys, es = grids(S, (8, 16, 32), torch.float32)
    gx, gy, st = torch.cat(xs, 1)[0], torch.cat(ys, 1)[0], torch.cat(es, 1)[0]
    A = gx.numel()
    out = torch.zeros(B, A, 5 + nc)
    out[..., 0] = (gx + 0.5 + 0.8 * torch.randn(B, A, generator=g)) * st
    out[..., 1] = (gy + 0.5 + 0.8 * torch.randn(B, A, generator=g)) * st
    out[..., 2] = torch.exp(box_sigma * torch.randn(B, A, generator=g)) * st * 3
    out[..., 3] = torch.exp(box_sigma * torch.randn(B, A, generator=g)) * st * 3
    out[..., 4] = 1.5 * torch.randn(B, A, generator=g)
    out[..., 5:] = 1.5 * torch.randn(B, A, nc, generator=g) - 2.0
    # quantise some logits coarsely so that equal costs (exact ties) occur
    out[:, ::3, 4:] = (out[:, ::3, 4:] * 2).round() / 2
    labels = torch.zeros(B, 20, 5)
    for b in range(B):
        for j in range(n_gt):
            cx = S * (0.15 + 0.7 * torch.rand((), generator=g))
            cy = S * (0.15 + 0.7 * torch.rand((), generator=g))
            w = S * (0.05 + 0.5 * torch.rand((), generator=g))
            h = S * (0.05 + 0.6 * torch.rand((), generator=g))
            labels[b, j] = torch.tensor([float((j * 7 + b) % nc), cx, cy, w, h])
        if dup and n_gt >= 4:
            labels[b, 1, 1:] = labels[b, 0, 1:]          # identical boxes, different classes
            labels[b, 3] = labels[b, 2]                  # identical rows
    return out, labels


def main():
    sys.path.insert(0, HERE)
    import ref_loader
    ref_loader.import_reference()
    from yolox.exp import get_exp

    torch.set_num_threads(8)
    exp = get_exp("/root/reference/exps/TSCD_OVIS/ovis_tscd_large.py")
    exp.depth, exp.width = 0.33, 0.125
    torch.manual_seed(0)
    head = exp.get_model().head
    nc = head.num_classes

    cases = []
    # case0: golden clip outputs through OUR oracle forward is not allowed here (the point is the reference) -> use the
    # reference's own decoded outputs from a train-mode forward of the tiny golden model
    gw = np.load(os.path.join(OUT, "tscd_tiny_weights.npz"))
    gv = np.load(os.path.join(OUT, "tscd_tiny_vectors.npz"))
    model = exp.get_model()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in gw.items()}, strict=True)
    model.train()
    cap = {}
    orig = type(model.head).get_fg_idx

    def spy(self, imgs, x_shifts, y_shifts, expanded_strides, labels, outputs, origin_preds, dtype):
        cap["outputs"] = outputs.detach().clone()
        return orig(self, imgs, x_shifts, y_shifts, expanded_strides, labels, outputs, origin_preds, dtype)

    type(model.head).get_fg_idx = spy
    from yolox.data.datasets.vid import get_timing_signal_1d
    imgs = torch.from_numpy(gv["evalA_imgs"]).float()
    labels = torch.from_numpy(gv["train_labels"])
    with torch.no_grad():
        model(imgs, labels, get_timing_signal_1d(torch.arange(0, 5), 256), lframe=3, gframe=2)
    type(model.head).get_fg_idx = orig
    cases.append((128, cap["outputs"], labels))
    cases.append((256,) + synth_outputs(3, 256, nc, 11, 1.5, 12, dup=True))
    o2, l2 = synth_outputs(2, 576, nc, 12, 1.0, 6)
    l2[0, 0, 1:] = torch.tensor([530.3, 541.7, 80.2, 66.6])       # beyond 512 px: fp16 grid is 0.5 px
    l2[1, 1, 1:] = torch.tensor([300.0, 280.0, 420.0, 400.0])     # area 168000 > 65504: inf in fp16
    o2[1, 100:140, 2:4] = 400.0                                    # predicted boxes whose area overflows too
    cases.append((576, o2, l2))

    rec = {"n_cases": np.array(len(cases))}
    for ci, (S, outputs, labels) in enumerate(cases):
        o16, l16 = outputs.half(), labels.half()
        xs, ys, es = grids(S, (8, 16, 32), torch.float16)
        with torch.no_grad():
            fg_ids, cls_t, reg_t, obj_t, fg_masks, num_fg, num_gts = orig(head, None, xs, ys, es, l16, o16, None, torch.float16)[:7]
        B, A = o16.shape[:2]
        fg = torch.stack([m for m in fg_masks]).numpy()
        pred_iou = np.zeros((B, A), np.float32)
        matched_cls = -np.ones((B, A), np.int64)
        for b in range(B):
            idx = np.nonzero(fg[b])[0]
            ct = cls_t[b].float()
            pred_iou[b, idx] = ct.sum(-1).numpy()                      # onehot * iou -> the matched IoU
            matched_cls[b, idx] = ct.argmax(-1).numpy()
            assert list(np.asarray(fg_ids[b])) == list(idx)
        rec[f"c{ci}_S"] = np.array(S)
        rec[f"c{ci}_outputs"] = o16.numpy()
        rec[f"c{ci}_labels"] = l16.numpy()
        rec[f"c{ci}_fg_mask"] = fg
        rec[f"c{ci}_pred_iou"] = pred_iou
        rec[f"c{ci}_matched_cls"] = matched_cls
        rec[f"c{ci}_reg_target"] = torch.cat([r.float() for r in reg_t], 0).numpy()
        rec[f"c{ci}_num_fg"] = np.array(float(num_fg))
        rec[f"c{ci}_num_gts"] = np.array(float(num_gts))
        print(f"case{ci}: S={S} B={B} A={A} num_fg={num_fg} num_gts={num_gts} fg/frame={fg.sum(1).tolist()}")
        # how different is the fp32 evaluation of the same fp16 values? (why the mode matters)
        xs32, ys32, es32 = grids(S, (8, 16, 32), torch.float32)
        with torch.no_grad():
            r32 = orig(head, None, xs32, ys32, es32, l16.float(), o16.float(), None, torch.float32)
        fg32 = torch.stack(list(r32[4])).numpy()
        print(f"        fp32 evaluation of the same inputs: num_fg={r32[5]}, anchors that differ: {(fg32 != fg).sum()}")
    path = os.path.join(OUT, "simota_fp16.npz")
    np.savez_compressed(path, **rec)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
