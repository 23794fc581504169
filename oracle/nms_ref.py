"""TEST INFRASTRUCTURE -- CPU restatement of torchvision.ops.{nms,batched_nms}.

torchvision is a third-party dependency of the reference (requirements.txt:8, ``torchvision>=0.10.0``)
that is absent from /root/reference and from this image, so this follows torchvision's published
algorithm and is anchored on the reference's call sites (yolox/models/post_process.py:58-63,73-78;
yolox/utils/boxes.py:59-70).  NMS parity is therefore "unpinned" by the reference itself; the
restatement is self-checked by tests/test_oracle_nms.py against a brute-force model.

Semantics: candidates are visited in decreasing score order (stable for ties: lower index first); a
candidate is dropped when its IoU with any already-kept box is > ``iou_threshold``; the kept indices
are returned in decreasing score order.  ``batched_nms`` adds ``idx * (max_coordinate + 1)`` to every
box so boxes of different classes never overlap (torchvision/ops/boxes.py, the "coordinate trick").
"""
import numpy as np
import torch


def box_area(boxes):
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def nms(boxes, scores, iou_threshold):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    b = boxes.float()
    order = torch.argsort(scores.float(), descending=True, stable=True)
    b = b[order].numpy()
    n = b.shape[0]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    sup = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if sup[i]:
            continue
        keep.append(i)
        # IoU of box i against every box (float32 arithmetic, as torchvision's CPU kernel)
        iw = np.maximum(np.minimum(b[i, 2], b[:, 2]) - np.maximum(b[i, 0], b[:, 0]), np.float32(0))
        ih = np.maximum(np.minimum(b[i, 3], b[:, 3]) - np.maximum(b[i, 1], b[:, 1]), np.float32(0))
        inter = iw * ih
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (area[i] + area - inter)
        sup |= iou > np.float32(iou_threshold)
    return order[torch.tensor(keep, dtype=torch.int64)]


def batched_nms_vanilla(boxes, scores, idxs, iou_threshold):
    """torchvision's ``_batched_nms_vanilla``: NMS per class without coordinate offsets, survivors in decreasing score order
    (stable here: lower index first among equal scores)."""
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for class_id in torch.unique(idxs):
        cur = torch.where(idxs == class_id)[0]
        keep_mask[cur[nms(boxes[cur], scores[cur], iou_threshold)]] = True
    keep = torch.where(keep_mask)[0]
    return keep[torch.argsort(scores[keep].float(), descending=True, stable=True)]


def batched_nms(boxes, scores, idxs, iou_threshold, vanilla_above=None):
    """``vanilla_above``: torchvision (0.10+) leaves the coordinate trick for per-class NMS when ``boxes.numel()`` exceeds a
    threshold (4000 on the CPU, 20000 on a GPU in current releases); None = always the trick (the historical behaviour and what the
    rest of this oracle uses)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    if vanilla_above is not None and boxes.numel() > vanilla_above:
        return batched_nms_vanilla(boxes, scores, idxs, iou_threshold)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, iou_threshold)
