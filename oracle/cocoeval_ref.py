"""TEST INFRASTRUCTURE -- plain-Python restatement of the COCO bounding-box evaluation (pycocotools.cocoeval.COCOeval:
evaluateImg / accumulate / summarize), the scorer behind the reference's OVISEvaluator.evaluate_prediction
(yolox/evaluators/ovis_evaluator_v2.py:352-424; its own fast path is yolox/layers/cocoeval/cocoeval.cpp).

pycocotools (requirements.txt: ``pycocotools>=2.0.2``) is a third-party dependency that is absent from /root/reference and
from this image, so this follows the published algorithm.  Pinning: everything downstream of the box IoUs -- score ordering with the
100-detection cap, ignore / crowd handling, greedy matching, the interpolated precision / recall arrays -- is checked against the
REFERENCE'S OWN compiled evaluator (oracle/_ref/fast_cocoeval_ref, built by oracle/build_ref.py from yolox/layers/cocoeval/cocoeval.cpp
as it lies in /root/reference; tests/test_cocoeval_ref_pin.py).  Only the IoU formula itself (pycocotools' maskUtils.iou for boxes:
inter / union, inter / detection area for crowd regions) stays a restatement.  The product's C++ implementation
(tscd_amd/csrc_host/cocoeval.cpp) is checked against that module, this restatement and hand-computed cases (tests/test_cocoeval.py).  Deliberately written per (image, category) with Python loops -- small inputs only."""
import numpy as np

IOU_THRS = np.linspace(0.5, 0.95, 10)
REC_THRS = np.linspace(0.0, 1.0, 101)
AREA_RNG = [(0, 1e10), (0, 32 ** 2), (32 ** 2, 96 ** 2), (96 ** 2, 1e10)]
MAX_DETS = [1, 10, 100]


def _iou(d, g, crowd):
    ix = min(d[0] + d[2], g[0] + g[2]) - max(d[0], g[0])
    iy = min(d[1] + d[3], g[1] + g[3]) - max(d[1], g[1])
    if ix <= 0 or iy <= 0:
        return 0.0
    inter = ix * iy
    return inter / (d[2] * d[3] if crowd else d[2] * d[3] + g[2] * g[3] - inter)


def evaluate(gts, dts, n_img, n_cat):
    """gts: list of dict(img, cat, bbox xywh, iscrowd, area); dts: list of dict(img, cat, bbox, score).
    -> (stats [12], precision [T,R,K,A,M], recall [T,K,A,M])"""
    T, R, K, A, M = len(IOU_THRS), len(REC_THRS), n_cat, len(AREA_RNG), len(MAX_DETS)
    precision = -np.ones((T, R, K, A, M))
    recall = -np.ones((T, K, A, M))
    for k in range(n_cat):
        for a, (lo, hi) in enumerate(AREA_RNG):
            per_img = []
            for im in range(n_img):
                g = [x for x in gts if x["img"] == im and x["cat"] == k]
                d = [x for x in dts if x["img"] == im and x["cat"] == k]
                if not g and not d:
                    continue
                gig = [bool(x["iscrowd"]) or x["area"] < lo or x["area"] > hi for x in g]
                order = np.argsort(gig, kind="mergesort")
                g = [g[i] for i in order]
                gig = [gig[i] for i in order]
                dorder = np.argsort([-x["score"] for x in d], kind="mergesort")[:MAX_DETS[-1]]
                d = [d[i] for i in dorder]
                ious = np.array([[_iou(x["bbox"], y["bbox"], y["iscrowd"]) for y in g] for x in d]).reshape(len(d), len(g))
                dtm = np.zeros((T, len(d)), bool)
                dtig = np.zeros((T, len(d)), bool)
                for t, thr in enumerate(IOU_THRS):
                    gtm = -np.ones(len(g), int)
                    for di in range(len(d)):
                        best, m = min(thr, 1 - 1e-10), -1
                        for gi in range(len(g)):
                            if gtm[gi] >= 0 and not g[gi]["iscrowd"]:
                                continue
                            if m > -1 and not gig[m] and gig[gi]:
                                break
                            if ious[di, gi] < best:
                                continue
                            best, m = ious[di, gi], gi
                        if m == -1:
                            continue
                        dtig[t, di] = gig[m]
                        dtm[t, di] = True
                        gtm[m] = di
                darea = np.array([x["bbox"][2] * x["bbox"][3] for x in d])
                out = (darea < lo) | (darea > hi)
                dtig = dtig | (~dtm & out[None, :]) if len(d) else dtig
                per_img.append((np.array([x["score"] for x in d]), dtm, dtig, sum(1 for x in gig if not x)))
            for m, md in enumerate(MAX_DETS):
                if not per_img:
                    continue
                sc = np.concatenate([e[0][:md] for e in per_img])
                inds = np.argsort(-sc, kind="mergesort")
                dtm = np.concatenate([e[1][:, :md] for e in per_img], axis=1)[:, inds]
                dtig = np.concatenate([e[2][:, :md] for e in per_img], axis=1)[:, inds]
                npig = sum(e[3] for e in per_img)
                if npig == 0:
                    continue
                tps = np.cumsum(dtm & ~dtig, axis=1).astype(float)
                fps = np.cumsum(~dtm & ~dtig, axis=1).astype(float)
                for t in range(T):
                    keep = ~dtig[t]
                    tp, fp = tps[t][keep], fps[t][keep]
                    rc = tp / npig
                    pr = tp / (fp + tp + np.spacing(1))
                    recall[t, k, a, m] = rc[-1] if len(rc) else 0
                    pr = pr.tolist()
                    for i in range(len(pr) - 1, 0, -1):
                        if pr[i] > pr[i - 1]:
                            pr[i - 1] = pr[i]
                    idx = np.searchsorted(rc, REC_THRS, side="left")
                    q = np.zeros(R)
                    for ri, pi in enumerate(idx):
                        if pi < len(pr):
                            q[ri] = pr[pi]
                    precision[t, :, k, a, m] = q

    def mp(ts, a, m):
        s = precision[ts, :, :, a, m]
        s = s[s > -1]
        return s.mean() if s.size else -1.0

    def mr(a, m):
        s = recall[:, :, a, m]
        s = s[s > -1]
        return s.mean() if s.size else -1.0

    allt = slice(None)
    stats = [mp(allt, 0, 2), mp(slice(0, 1), 0, 2), mp(slice(5, 6), 0, 2), mp(allt, 1, 2), mp(allt, 2, 2), mp(allt, 3, 2),
             mr(0, 0), mr(0, 1), mr(0, 2), mr(1, 2), mr(2, 2), mr(3, 2)]
    return np.array(stats), precision, recall
