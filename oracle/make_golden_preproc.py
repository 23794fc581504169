"""TEST INFRASTRUCTURE -- writes tests/golden/clip_preproc_handmade.npz: hand-derived input / output pairs for the clip transform chain
of the reference (yolox/data/data_augment.py:21-31 HSV jitter in OpenCV's 8-bit HSV, :240-245 mirror, :248-264 letterbox; the resize of
yolox/data/datasets/vid.py pull_item before it).  cv2 is absent from this image and the reference holds no fixtures for this path, so the
cases are chosen such that the reference's formulas give EXACT values by hand, with no interpolation or rounding in play:

  * letterbox at ratio 1/2 of frames made of constant 2 x 2 blocks (INTER_LINEAR samples at 2 x + 0.5: the mean of a block = its value),
    and at ratio 1 (a copy); placement top-left, fill 114, planes B, G, R, float 0..255;
  * mirror: columns reversed before the letterbox;
  * HSV jitter on greys (S = 0: hue and saturation gains do nothing, V + g clipped to 0..255) and on the six saturated primaries /
    secondaries (H = 0, 30, ... 150 in OpenCV's 0..179 scale: a hue gain of 30 is a rotation to the next one; V or S gains give
    (V', V' (1 - S'/255)) exactly for V' S' a multiple of 255).

What is NOT pinned by this file: the bilinear interpolation itself at non-trivial ratios (cv2 uses 1/2048 fixed-point weights) and the
rounding of OpenCV's HSV conversion for general colours -- "parity unpinned", tests/test_gpu_clip_preproc.py compares those with the
repo's own numpy pipeline only.  Expected planes are BUILT here from the rules above, not computed by the code under test.
Run:  python -m oracle.make_golden_preproc"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# BGR of the six hues at full saturation / value, in hue order (OpenCV H = 0, 30, 60, 90, 120, 150)
HUES = [(0, 0, 255), (0, 255, 255), (0, 255, 0), (255, 255, 0), (255, 0, 0), (255, 0, 255)]
GREYS = [0, 20, 114, 200, 255]


def jitter(px, gains):
    """One pixel of the exact set through (h, s, v) gains by hand."""
    gh, gs, gv = gains
    if px[0] == px[1] == px[2]:                       # grey: S = 0 -> only V moves; a saturation gain lifts S from 0 but C = V S / 255
        v = int(np.clip(px[0] + gv, 0, 255))          # stays 0 only if S stays 0: keep gs <= 0 for frames with greys
        assert gs <= 0
        return (v, v, v)
    k = HUES.index(tuple(px))
    assert gh % 30 == 0
    k2 = (k + gh // 30) % 6
    v, s = int(np.clip(255 + gv, 0, 255)), int(np.clip(255 + gs, 0, 255))
    assert (v * s) % 255 == 0
    c = v * s // 255
    base = HUES[k2]
    return tuple(v if ch == 255 else v - c for ch in base)


def blocks(rows, cols, palette, rng):
    """rows x cols image of 2 x 2 constant blocks drawn from ``palette`` -> (full image, the half-size image of block values)."""
    small = np.array([[palette[rng.integers(len(palette))] for _ in range(cols // 2)] for _ in range(rows // 2)], dtype=np.uint8)
    return np.repeat(np.repeat(small, 2, 0), 2, 1), small


def main():
    rng = np.random.default_rng(7)
    H, W = 8, 8
    palette = HUES + [(g, g, g) for g in GREYS]
    cases = []
    # (frame, resized hw, flip, gains or None, expected resized image BEFORE flip / jitter)
    f0, s0 = blocks(8, 16, palette, rng)              # ratio 1/2 -> 4 x 8
    cases.append((f0, s0, 0, None))
    cases.append((f0, s0, 1, None))
    f1 = np.array([[palette[rng.integers(len(palette))] for _ in range(8)] for _ in range(6)], dtype=np.uint8)   # ratio 1 -> copy
    cases.append((f1, f1, 0, None))
    cases.append((f1, f1, 1, (30, 0, 0)))
    cases.append((f1, f1, 0, (0, 0, -30)))
    cases.append((f0, s0, 1, (60, -30, 0)))
    f2 = np.array([[HUES[(i + j) % 6] for j in range(8)] for i in range(8)], dtype=np.uint8)
    cases.append((f2, f2, 0, (-30, -255, 0)))         # saturation to 0: every pixel becomes white (V = 255)
    cases.append((f2, f2, 0, (0, 0, -255)))           # value to 0: black
    hmax, wmax = max(c[0].shape[0] for c in cases), max(c[0].shape[1] for c in cases)
    frames = np.zeros((len(cases), hmax, wmax, 3), dtype=np.uint8)
    meta = np.zeros((len(cases), 12), dtype=np.int32)
    want = np.full((len(cases), 3, H, W), 114.0, dtype=np.float32)
    for i, (frame, small, flip, gains) in enumerate(cases):
        h, w = frame.shape[:2]
        nh, nw = small.shape[:2]
        frames[i, :h, :w] = frame
        meta[i, :4] = (h, w, nh, nw)
        meta[i, 9:11] = np.array([h / nh, w / nw], dtype=np.float32).view(np.int32)
        meta[i, 4], meta[i, 5] = flip, int(gains is not None)
        img = small.copy()
        if gains is not None:
            meta[i, 6:9] = gains
            img = np.array([[jitter(tuple(int(v) for v in px), gains) for px in row] for row in img], dtype=np.uint8)
        if flip:
            img = img[:, ::-1]
        want[i, :, :nh, :nw] = img.transpose(2, 0, 1).astype(np.float32)
    path = os.path.join(ROOT, "tests", "golden", "clip_preproc_handmade.npz")
    np.savez_compressed(path, frames=frames, meta=meta, want=want, input_dim=np.array([H, W]))
    print(path, frames.shape, want.shape)


if __name__ == "__main__":
    main()
