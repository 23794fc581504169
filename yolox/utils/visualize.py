"""``vis``: draw detection boxes on an image (reference: yolox/utils/visualize.py:11-43, OpenCV).  numpy only here: boxes are
1-2 px rectangles in a per-class colour, the label text is not rendered (no font rasteriser in this image)."""
import numpy as np

__all__ = ["vis"]

_COLORS = (np.array([[0.000, 0.447, 0.741], [0.850, 0.325, 0.098], [0.929, 0.694, 0.125], [0.494, 0.184, 0.556], [0.466, 0.674, 0.188],
                     [0.301, 0.745, 0.933], [0.635, 0.078, 0.184], [0.300, 0.300, 0.300], [0.600, 0.600, 0.600], [1.000, 0.000, 0.000],
                     [1.000, 0.500, 0.000], [0.749, 0.749, 0.000], [0.000, 1.000, 0.000], [0.000, 0.000, 1.000], [0.667, 0.000, 1.000]]) * 255).astype(np.uint8)


def vis(img, boxes, scores, cls_ids, conf=0.5, class_names=None):
    img = np.ascontiguousarray(img)
    H, W = img.shape[:2]
    for box, score, cid in zip(boxes, scores, cls_ids):
        if score < conf:
            continue
        x0, y0, x1, y1 = [int(np.clip(v, 0, lim - 1)) for v, lim in zip(box, (W, H, W, H))]
        color = _COLORS[int(cid) % len(_COLORS)][::-1]       # BGR image
        img[y0:y0 + 2, x0:x1 + 1] = color
        img[max(y1 - 1, 0):y1 + 1, x0:x1 + 1] = color
        img[y0:y1 + 1, x0:x0 + 2] = color
        img[y0:y1 + 1, max(x1 - 1, 0):x1 + 1] = color
    return img
