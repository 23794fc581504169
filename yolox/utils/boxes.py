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


def bboxes_iou(bboxes_a, bboxes_b, xyxy=True):
    if bboxes_a.shape[1] != 4 or bboxes_b.shape[1] != 4:
        raise IndexError
    if xyxy:
        tl = torch.max(bboxes_a[:, None, :2], bboxes_b[:, :2])
        br = torch.min(bboxes_a[:, None, 2:], bboxes_b[:, 2:])
        area_a = torch.prod(bboxes_a[:, 2:] - bboxes_a[:, :2], 1)
        area_b = torch.prod(bboxes_b[:, 2:] - bboxes_b[:, :2], 1)
    else:
        tl = torch.max(bboxes_a[:, None, :2] - bboxes_a[:, None, 2:] / 2, bboxes_b[:, :2] - bboxes_b[:, 2:] / 2)
        br = torch.min(bboxes_a[:, None, :2] + bboxes_a[:, None, 2:] / 2, bboxes_b[:, :2] + bboxes_b[:, 2:] / 2)
        area_a, area_b = torch.prod(bboxes_a[:, 2:], 1), torch.prod(bboxes_b[:, 2:], 1)
    en = (tl < br).type(tl.type()).prod(dim=2)
    area_i = torch.prod(br - tl, 2) * en
    return area_i / (area_a[:, None] + area_b - area_i)


def adjust_box_anns(bbox, scale_ratio, padw, padh, w_max, h_max):
    bbox[:, 0::2] = np.clip(bbox[:, 0::2] * scale_ratio + padw, 0, w_max)
    bbox[:, 1::2] = np.clip(bbox[:, 1::2] * scale_ratio + padh, 0, h_max)
    return bbox


def xyxy2xywh(bboxes):
    bboxes[:, 2] = bboxes[:, 2] - bboxes[:, 0]
    bboxes[:, 3] = bboxes[:, 3] - bboxes[:, 1]
    return bboxes


def xyxy2cxcywh(bboxes):
    bboxes[:, 2] = bboxes[:, 2] - bboxes[:, 0]
    bboxes[:, 3] = bboxes[:, 3] - bboxes[:, 1]
    bboxes[:, 0] = bboxes[:, 0] + bboxes[:, 2] * 0.5
    bboxes[:, 1] = bboxes[:, 1] + bboxes[:, 3] * 0.5
    return bboxes
