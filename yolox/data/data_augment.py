"""Image / label transforms of the clip input pipeline (interface of the reference's yolox/data/data_augment.py:
``preproc`` :248-264, ``TrainTransform`` :267-316, ``ValTransform`` :319-349, ``Vid_Val_Transform`` :351-386).

Same call contracts -- ``transform(img HxWx3 uint8 (BGR, as the decoder delivers it), targets [n,5] (x1,y1,x2,y2,cls),
input_size) -> (img 3xHxW float32 0..255, labels)`` -- on numpy only: this image has no OpenCV, so the letterbox resize is
an own bilinear resampler with OpenCV's INTER_LINEAR sampling geometry (half-pixel centres, edge clamp, no antialias) and
the HSV jitter an own colour-space conversion.  Resize parity with cv2 is therefore "unpinned" (cv2 rounds its fixed-point
interpolation weights to 1/2048; expect +-1 grey level).  ``letterbox_batch_gpu`` is the device-side variant used by the
clip loader when frames are already decoded into one tensor (at ~900 frames/s a per-frame host resize
cannot keep up)."""
import random

import numpy as np


def xyxy2cxcywh(b):
    b = b.copy()
    b[:, 2] = b[:, 2] - b[:, 0]
    b[:, 3] = b[:, 3] - b[:, 1]
    b[:, 0] = b[:, 0] + b[:, 2] * 0.5
    b[:, 1] = b[:, 1] + b[:, 3] * 0.5
    return b


def resize_bilinear(img, out_h, out_w):
    """HxWxC uint8 -> out_h x out_w x C uint8, bilinear, source coordinate (dst + 0.5) * scale - 0.5 clamped to the image."""
    h, w = img.shape[:2]
    if (h, w) == (out_h, out_w):
        return img.copy()
    ys = (np.arange(out_h, dtype=np.float32) + 0.5) * (h / out_h) - 0.5
    xs = (np.arange(out_w, dtype=np.float32) + 0.5) * (w / out_w) - 0.5
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    fy = (ys - y0).astype(np.float32)[:, None, None]
    fx = (xs - x0).astype(np.float32)[None, :, None]
    y0c, y1c = np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
    x0c, x1c = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
    im = img.astype(np.float32)
    if im.ndim == 2:
        im = im[:, :, None]
    top = im[y0c][:, x0c] * (1 - fx) + im[y0c][:, x1c] * fx
    bot = im[y1c][:, x0c] * (1 - fx) + im[y1c][:, x1c] * fx
    out = top * (1 - fy) + bot * fy
    out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out if img.ndim == 3 else out[:, :, 0]


def preproc(img, input_size, swap=(2, 0, 1)):
    """Letterbox: resize by r = min(H_in/h, W_in/w), paste top-left on a 114-grey canvas, HWC -> CHW float32 (0..255)."""
    if img.ndim == 3:
        padded = np.full((input_size[0], input_size[1], 3), 114, dtype=np.uint8)
    else:
        padded = np.full(tuple(input_size), 114, dtype=np.uint8)
    r = min(input_size[0] / img.shape[0], input_size[1] / img.shape[1])
    nh, nw = int(img.shape[0] * r), int(img.shape[1] * r)
    padded[:nh, :nw] = resize_bilinear(img, nh, nw)
    padded = padded.transpose(swap)
    return np.ascontiguousarray(padded, dtype=np.float32), r


def _bgr2hsv(img):
    """uint8 BGR -> (H 0..179, S 0..255, V 0..255) as OpenCV's 8-bit HSV."""
    b, g, r = [img[..., i].astype(np.float32) for i in range(3)]
    v = np.maximum(np.maximum(r, g), b)
    mn = np.minimum(np.minimum(r, g), b)
    d = v - mn
    s = np.where(v > 0, d / np.maximum(v, 1e-12) * 255.0, 0.0)
    dd = np.maximum(d, 1e-12)
    h = np.where(v == r, (g - b) / dd, np.where(v == g, 2.0 + (b - r) / dd, 4.0 + (r - g) / dd)) * 60.0
    h = np.where(d == 0, 0.0, h)
    h = np.where(h < 0, h + 360.0, h) * 0.5
    return np.stack([np.rint(h) % 180, np.rint(s), v], -1).astype(np.int16)


def _hsv2bgr(hsv):
    h = hsv[..., 0].astype(np.float32) * 2.0
    s = hsv[..., 1].astype(np.float32) / 255.0
    v = hsv[..., 2].astype(np.float32)
    c = v * s
    hp = h / 60.0
    x = c * (1 - np.abs(np.mod(hp, 2) - 1))
    z = np.zeros_like(c)
    sel = np.floor(hp).astype(np.int64) % 6
    r = np.choose(sel, [c, x, z, z, x, c])
    g = np.choose(sel, [x, c, c, x, z, z])
    b = np.choose(sel, [z, z, x, c, c, x])
    m = v - c
    return np.clip(np.rint(np.stack([b + m, g + m, r + m], -1)), 0, 255).astype(np.uint8)


def augment_hsv(img, hgain=5, sgain=30, vgain=30):
    """In-place HSV jitter (data_augment.py:21-31): random integer gains on a random subset of the three channels."""
    gains = np.random.uniform(-1, 1, 3) * [hgain, sgain, vgain]
    gains *= np.random.randint(0, 2, 3)
    gains = gains.astype(np.int16)
    hsv = _bgr2hsv(img)
    hsv[..., 0] = (hsv[..., 0] + gains[0]) % 180
    hsv[..., 1] = np.clip(hsv[..., 1] + gains[1], 0, 255)
    hsv[..., 2] = np.clip(hsv[..., 2] + gains[2], 0, 255)
    img[...] = _hsv2bgr(hsv)


def _mirror(image, boxes, prob=0.5):
    _, width, _ = image.shape
    if random.random() < prob:
        image = image[:, ::-1]
        boxes[:, 0::2] = width - boxes[:, 2::-2]
    return image, boxes


class TrainTransform:
    """HSV jitter + horizontal flip + letterbox; targets [n,5] (x1,y1,x2,y2,cls) -> padded [max_labels,5] (cls,cx,cy,w,h)."""

    def __init__(self, max_labels=50, flip_prob=0.5, hsv_prob=1.0):
        self.max_labels, self.flip_prob, self.hsv_prob = max_labels, flip_prob, hsv_prob

    def __call__(self, image, targets, input_dim):
        boxes, labels = targets[:, :4].copy(), targets[:, 4].copy()
        if len(boxes) == 0:
            image, _ = preproc(image, input_dim)
            return image, np.zeros((self.max_labels, 5), dtype=np.float32)
        image_o, boxes_o, labels_o = image.copy(), xyxy2cxcywh(targets[:, :4]), targets[:, 4].copy()
        if random.random() < self.hsv_prob:
            augment_hsv(image)
        image_t, boxes = _mirror(image, boxes, self.flip_prob)
        image_t, r_ = preproc(image_t, input_dim)
        boxes = xyxy2cxcywh(boxes) * r_
        keep = np.minimum(boxes[:, 2], boxes[:, 3]) > 1
        boxes_t, labels_t = boxes[keep], labels[keep]
        if len(boxes_t) == 0:                         # every box collapsed: fall back to the un-augmented frame
            image_t, r_o = preproc(image_o, input_dim)
            boxes_t, labels_t = boxes_o * r_o, labels_o
        t = np.hstack((labels_t[:, None], boxes_t))
        out = np.zeros((self.max_labels, 5))
        out[:min(len(t), self.max_labels)] = t[:self.max_labels]
        return image_t, np.ascontiguousarray(out, dtype=np.float32)


class TrainTransformGPU:
    """Host half of the device-side training transform: draws the SAME random decisions in the same order as
    ``TrainTransform`` (HSV on/off, the three gains and their on/off mask, the flip) and transforms the LABELS, but leaves the image
    alone -- the decoded frame goes to the GPU as uint8 and ``tscd_clip_preproc_u8`` resizes, jitters, flips and pads the whole clip
    there (a per-frame host resize + HSV costs ~90 ms at 720p: 0.66 clips/s per loader worker against ~60 clips/s of the model).

    ``__call__(image_size (h, w) of the decoded frame, resized (nh, nw), targets [n,5] scaled to the resized frame, input_dim)`` ->
    (meta int32[12] for the kernel, labels [max_labels,5] (cls,cx,cy,w,h))."""

    def __init__(self, max_labels=50, flip_prob=0.5, hsv_prob=1.0, hgain=5, sgain=30, vgain=30):
        self.max_labels, self.flip_prob, self.hsv_prob, self.gains = max_labels, flip_prob, hsv_prob, (hgain, sgain, vgain)

    def __call__(self, hw, resized, targets, input_dim):
        (h, w), (nh, nw) = hw, resized
        meta = np.zeros(12, dtype=np.int32)
        meta[:4] = (h, w, nh, nw)
        meta[9:11] = np.array([h / nh, w / nw], dtype=np.float32).view(np.int32)
        # (preproc's own resize is the identity here: its ratio min(H/nh, W/nw) maps (nh, nw) onto itself -- checked by the dataset)
        boxes, labels = targets[:, :4].copy(), targets[:, 4].copy()
        if len(boxes) == 0:
            return meta, np.zeros((self.max_labels, 5), dtype=np.float32)
        boxes_o, labels_o = xyxy2cxcywh(targets[:, :4]), targets[:, 4].copy()
        hsv_on, gains = 0, np.zeros(3, dtype=np.int16)
        if random.random() < self.hsv_prob:
            g = np.random.uniform(-1, 1, 3) * list(self.gains)
            g *= np.random.randint(0, 2, 3)
            hsv_on, gains = 1, g.astype(np.int16)
        flip = 0
        if random.random() < self.flip_prob:
            flip = 1
            boxes[:, 0::2] = nw - boxes[:, 2::-2]
        r2 = min(input_dim[0] / nh, input_dim[1] / nw)      # preproc's own ratio on the resized frame (1.0 unless a truncation left a gap)
        boxes = xyxy2cxcywh(boxes) * r2
        keep = np.minimum(boxes[:, 2], boxes[:, 3]) > 1
        boxes_t, labels_t = boxes[keep], labels[keep]
        if len(boxes_t) == 0:                         # every box collapsed: the un-augmented frame
            hsv_on, flip, gains = 0, 0, np.zeros(3, dtype=np.int16)
            boxes_t, labels_t = boxes_o * r2, labels_o
        meta[4], meta[5] = flip, hsv_on
        meta[6:9] = gains
        t = np.hstack((labels_t[:, None], boxes_t))
        out = np.zeros((self.max_labels, 5))
        out[:min(len(t), self.max_labels)] = t[:self.max_labels]
        return meta, np.ascontiguousarray(out, dtype=np.float32)


def clip_preproc_gpu(frames_u8, meta, input_size, out_dtype=None):
    """frames_u8 [B,hmax,wmax,3] uint8 BGR on the GPU, meta [B,12] int32 (TrainTransformGPU / val_meta) -> [B,3,H,W] letterboxed clip."""
    import torch
    from tscd_amd._lib import call
    from tscd_amd import ops
    B, hmax, wmax, _ = frames_u8.shape
    H, W = input_size
    out_dtype = out_dtype or torch.float16
    assert frames_u8.is_cuda and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous() and out_dtype in (torch.float16, torch.float32)
    meta = meta.to(device=frames_u8.device, dtype=torch.int32).contiguous()
    out = torch.empty((B, 3, H, W), dtype=out_dtype, device=frames_u8.device)
    call("tscd_clip_preproc_u8", frames_u8.data_ptr(), B, hmax, wmax, meta.data_ptr(), out.data_ptr(), H, W, int(out_dtype == torch.float16),
         ops._stream())
    return out


def val_meta(hw, resized):
    """meta of a frame that is only resized and padded (evaluation)."""
    (h, w), (nh, nw) = hw, resized
    meta = np.zeros(12, dtype=np.int32)
    meta[:4] = (h, w, nh, nw)
    meta[9:11] = np.array([h / nh, w / nw], dtype=np.float32).view(np.int32)
    return meta


class ValTransform:
    """Letterbox only; returns (img, zeros[1,5]) like the reference (labels are not used at test time)."""

    def __init__(self, swap=(2, 0, 1), legacy=False):
        self.swap, self.legacy = swap, legacy

    def __call__(self, img, res, input_size):
        img, _ = preproc(img, input_size, self.swap)
        if self.legacy:
            img = _legacy_normalise(img)
        return img, np.zeros((1, 5))


class Vid_Val_Transform:
    """Letterbox + scale the (x1,y1,x2,y2,cls) boxes by the same ratio -> (img, [n,5] (cls,x1,y1,x2,y2))."""

    def __init__(self, swap=(2, 0, 1), legacy=False):
        self.swap, self.legacy = swap, legacy

    def __call__(self, img, res, input_size):
        img, r_ = preproc(img, input_size, self.swap)
        if self.legacy:
            img = _legacy_normalise(img)
        boxes = res[:, :4].copy() * r_
        return img, np.hstack((res[:, 4:5].copy(), boxes))


def _legacy_normalise(img):
    img = img[::-1, :, :].copy() / 255.0
    img -= np.array([0.485, 0.456, 0.406]).reshape(3, 1, 1)
    img /= np.array([0.229, 0.224, 0.225]).reshape(3, 1, 1)
    return img


def letterbox_batch_gpu(frames, input_size, out_dtype=None):
    """Device-side letterbox of a decoded clip: frames [B,3,h,w] (uint8 / float, on the GPU) -> [B,3,H,W] in `out_dtype`
    (default fp16) with the same geometry as ``preproc`` (scale r = min(H/h, W/w), top-left paste, 114 fill), one bilinear
    resample for the whole clip.  Returns (clip, r)."""
    import torch
    import torch.nn.functional as F
    B, C, h, w = frames.shape
    H, W = input_size
    r = min(H / h, W / w)
    nh, nw = int(h * r), int(w * r)
    out = torch.full((B, C, H, W), 114.0, dtype=out_dtype or torch.float16, device=frames.device)
    res = F.interpolate(frames.float(), size=(nh, nw), mode="bilinear", align_corners=False)
    out[:, :, :nh, :nw] = res.round().clamp_(0, 255).to(out.dtype)
    return out, r
