"""Box helpers of the reference's yolox/utils/boxes.py that the TSCD / YOLOX entry points use: ``postprocess`` (:33-79, the
single-image detector tail of tools/demo.py), ``bboxes_iou`` (:131-154), ``xyxy2xywh`` / ``xyxy2cxcywh`` / ``adjust_box_anns``."""
import numpy as np
import torch

__all__ = ["postprocess", "bboxes_iou", "xyxy2xywh", "xyxy2cxcywh", "adjust_box_anns", "filter_box"]


def filter_box(output, scale_range):
    min_scale, max_scale = scale_range
    w, h = output[:, 2] - output[:, 0], output[:, 3] - output[:, 1]
    return output[(w * h > min_scale * min_scale) & (w * h < max_scale * max_scale)]


def postprocess(prediction, num_classes, conf_thre=0.7, nms_thre=0.45, class_agnostic=False):
    """prediction [B, A, 5+NC] (cx,cy,w,h, obj, class probabilities; YOLOXHead's eval output) -> list of [k,7] rows
    (x1,y1,x2,y2, obj, class_conf, class_id) in decreasing obj*class_conf order, or None -- on the device: one pass builds the
    candidate rows, tscd_amd's NMS kernels (sort, greedy suppression per class or class-agnostic, compaction) do the rest;
    one D2H of the per-image counts at the end.  The input's boxes are converted to corners in place like the reference's."""
    from tscd_amd import ops
    if not prediction.is_cuda:
        raise RuntimeError("postprocess: predictions must be on the GPU (no CPU fallback in the product path)")
    B, A, _ = prediction.shape
    half = prediction[..., 2:4] / 2
    prediction[..., :4] = torch.cat((prediction[..., :2] - half, prediction[..., :2] + half), -1)
    conf, cid = torch.max(prediction[..., 5:5 + num_classes], -1, keepdim=True)
    det = torch.cat((prediction[..., :5], conf, cid.to(prediction.dtype)), -1).to(torch.float16).reshape(B * A, 7).contiguous()
    frame_off = ops.h2d([i * A for i in range(B + 1)], torch.int32, prediction.device)
    rows, idx, cnt = ops.batched_nms(det, frame_off, B, num_classes, conf_thre, nms_thre, A, 2 if class_agnostic else 1, max_rows=A)
    counts = cnt.cpu().tolist()
    return [rows[i, :counts[i]].to(prediction.dtype) if counts[i] > 0 else None for i in range(B)]


def _corners(b, xyxy):
    """[n,4] boxes -> (top-left [n,2], bottom-right [n,2]); centre format is (cx, cy, w, h)."""
    if xyxy:
        return b[:, :2], b[:, 2:4]
    ctr, ext = b[:, :2], b[:, 2:4] * 0.5
    return ctr - ext, ctr + ext


def bboxes_iou(bboxes_a, bboxes_b, xyxy=True):
    """Pairwise IoU [len(a), len(b)] (reference contract: utils/boxes.py:131-154 -- IndexError for anything but [n,4] inputs,
    an empty intersection counts as 0, no epsilon in the denominator)."""
    for b in (bboxes_a, bboxes_b):
        if b.dim() != 2 or b.shape[1] != 4:
            raise IndexError(f"bboxes_iou expects [n,4] boxes, got {tuple(b.shape)}")
    lo_a, hi_a = _corners(bboxes_a, xyxy)
    lo_b, hi_b = _corners(bboxes_b, xyxy)
    lo = torch.maximum(lo_a.unsqueeze(1), lo_b.unsqueeze(0))
    hi = torch.minimum(hi_a.unsqueeze(1), hi_b.unsqueeze(0))
    side = hi - lo
    overlap = side[..., 0] * side[..., 1] * (side > 0).all(dim=2).to(side.dtype)
    ext_a, ext_b = hi_a - lo_a, hi_b - lo_b
    area_a, area_b = ext_a[:, 0] * ext_a[:, 1], ext_b[:, 0] * ext_b[:, 1]
    return overlap / (area_a.unsqueeze(1) + area_b.unsqueeze(0) - overlap)


def adjust_box_anns(bbox, scale_ratio, padw, padh, w_max, h_max):
    """Annotation boxes (x1,y1,x2,y2 numpy rows) after a resize by ``scale_ratio`` and a shift by (padw, padh), clipped to the
    canvas -- in place, like the mosaic code expects (mosaicdetection.py:109-111)."""
    for axis, (shift, limit) in enumerate(((padw, w_max), (padh, h_max))):
        cols = bbox[:, axis::2]
        cols *= scale_ratio
        cols += shift
        np.clip(cols, 0, limit, out=cols)
    return bbox


def xyxy2xywh(bboxes):
    """Corners -> (x1, y1, w, h), in place."""
    bboxes[:, 2:4] -= bboxes[:, 0:2]
    return bboxes


def xyxy2cxcywh(bboxes):
    """Corners -> (cx, cy, w, h), in place."""
    bboxes[:, 2:4] -= bboxes[:, 0:2]
    bboxes[:, 0:2] += bboxes[:, 2:4] * 0.5
    return bboxes
