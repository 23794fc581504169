"""Row N2 pinned by the REFERENCE'S OWN native code: oracle/_ref/fast_cocoeval_ref is compiled (oracle/build_ref.py) from the
reference's yolox/layers/cocoeval/cocoeval.cpp as it lies in /root/reference -- COCOevalEvaluateImages (detection <-> ground-truth
matching per image, category, area range and IoU threshold) and COCOevalAccumulate (precision / recall over the data set), the two
calls of its fast_coco_eval_api.py:104-140.  What that module takes as INPUT -- the per-image IoU matrices, which the reference gets
from pycocotools' maskUtils.iou (absent here) -- is computed by the plain box formula of the COCO protocol; everything downstream of
the IoUs (sorting by score with the 100-detection cap, ignore / crowd handling, greedy matching, the 101-point interpolated
precision) is the reference's compiled code.  The product's C++ evaluator (tscd_amd/csrc_host/cocoeval.cpp) and the Python
restatement (oracle/cocoeval_ref.py) must both reproduce its precision and recall arrays.

Those arrays are committed as a fixture (tests/golden/cocoeval_ref_curves.npz, written by oracle/make_golden_cocoeval.py from the
compiled reference): the pin holds everywhere, and the reference-derived shared object is loaded ONLY in the build container (where
/root/reference exists) -- lazily, inside the one test that compares the live build with the fixture -- never on the GPU box."""
import numpy as np
import pytest

import os

from oracle import cocoeval_ref as R

SEEDS = (0, 1, 2, 3)
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cocoeval_ref_curves.npz")


def reference_module():
    """The compiled reference evaluator -- only where the reference tree itself is present (the build container)."""
    if not os.path.isdir("/root/reference"):
        return None
    from oracle import build_ref
    return build_ref.load()


def _scene(seed, n_img=7, n_cat=4):
    rng = np.random.default_rng(seed)
    gts, dts = [], []
    for im in range(n_img):
        for _ in range(int(rng.integers(0, 9))):
            w, h = rng.uniform(8, 200, 2)
            box = [float(rng.uniform(0, 400)), float(rng.uniform(0, 400)), float(w), float(h)]
            gts.append(dict(img=im, cat=int(rng.integers(0, n_cat)), bbox=box, iscrowd=int(rng.random() < 0.15), area=float(w * h)))
        for x in [g for g in gts if g["img"] == im]:
            if rng.random() < 0.8:
                j = rng.normal(0, 6, 4)
                box = [x["bbox"][0] + j[0], x["bbox"][1] + j[1], max(2.0, x["bbox"][2] + j[2]), max(2.0, x["bbox"][3] + j[3])]
                cat = x["cat"] if rng.random() < 0.85 else int(rng.integers(0, n_cat))
                dts.append(dict(img=im, cat=cat, bbox=[float(v) for v in box], score=float(np.round(rng.random(), 2))))       # rounded: score ties
        for _ in range(int(rng.integers(0, 40 if im == 0 else 6))):
            w, h = rng.uniform(4, 150, 2)
            dts.append(dict(img=im, cat=int(rng.integers(0, n_cat)), bbox=[float(rng.uniform(0, 400)), float(rng.uniform(0, 400)), float(w), float(h)],
                            score=float(np.round(rng.random(), 2))))
    dts += [dict(d, score=float(np.round(rng.random(), 2))) for d in dts if d["img"] == 0 and d["cat"] == 0] * 12     # > 100 detections of one (image, category)
    for x in gts + dts:
        x["bbox"] = [float(np.float32(v)) for v in x["bbox"]]
    for x in gts:
        x["area"] = float(np.float32(x["area"]))
    for x in dts:
        x["score"] = float(np.float32(x["score"]))
    return gts, dts, n_img, n_cat


def _reference_curves(REF, gts, dts, n_img, n_cat):
    """precision [T,R,K,A,M], recall [T,K,A,M] from the reference's compiled evaluator."""
    nid = [0]

    def inst(x, det):
        nid[0] += 1
        if det:      # pycocotools loadRes: a detection's area is its box area
            return REF.InstanceAnnotation(nid[0], x["score"], float(np.float32(x["bbox"][2]) * np.float32(x["bbox"][3])), False, False)
        return REF.InstanceAnnotation(nid[0], 0.0, x["area"], bool(x["iscrowd"]), bool(x["iscrowd"]))      # _prepare: ignore = iscrowd

    G = [[[x for x in gts if x["img"] == i and x["cat"] == k] for k in range(n_cat)] for i in range(n_img)]
    D = [[[x for x in dts if x["img"] == i and x["cat"] == k] for k in range(n_cat)] for i in range(n_img)]
    ious = []
    for i in range(n_img):
        row = []
        for k in range(n_cat):
            d = D[i][k]
            order = np.argsort([-x["score"] for x in d], kind="mergesort")[:R.MAX_DETS[-1]]      # pycocotools computeIoU: by score, capped
            row.append([[R._iou(d[j]["bbox"], g["bbox"], bool(g["iscrowd"])) for g in G[i][k]] for j in order])
        ious.append(row)
    evals = REF.COCOevalEvaluateImages([list(a) for a in R.AREA_RNG], R.MAX_DETS[-1], list(R.IOU_THRS), ious,
                                       [[[inst(x, False) for x in c] for c in im] for im in G],
                                       [[[inst(x, True) for x in c] for c in im] for im in D])

    class P:
        recThrs, maxDets, iouThrs, useCats = list(R.REC_THRS), list(R.MAX_DETS), list(R.IOU_THRS), 1
        catIds, areaRng, imgIds = list(range(n_cat)), [list(a) for a in R.AREA_RNG], list(range(n_img))

    out = REF.COCOevalAccumulate(P(), evals)
    T, Rn, K, A, M = len(R.IOU_THRS), len(R.REC_THRS), n_cat, len(R.AREA_RNG), len(R.MAX_DETS)
    return np.array(out["precision"]).reshape(T, Rn, K, A, M), np.array(out["recall"]).reshape(T, K, A, M)


@pytest.mark.parametrize("seed", SEEDS)
def test_product_and_restatement_reproduce_the_reference_evaluator(seed):
    from test_cocoeval import run_cpp
    gts, dts, n_img, n_cat = _scene(seed)
    fx = np.load(FIXTURE)
    p_ref, r_ref = fx[f"precision_{seed}"], fx[f"recall_{seed}"]
    s, p, r = run_cpp(gts, dts, n_img, n_cat)
    np.testing.assert_allclose(p, p_ref, atol=1e-6)
    np.testing.assert_allclose(r, r_ref, atol=1e-6)
    s_o, p_o, r_o = R.evaluate(gts, dts, n_img, n_cat)
    np.testing.assert_allclose(p_o, p_ref, atol=1e-6)
    np.testing.assert_allclose(r_o, r_ref, atol=1e-6)
    # the 12 summary numbers are means over these arrays (COCOeval.summarize): AP = mean of precision > -1 at area 'all', maxDets 100
    pa = p_ref[:, :, :, 0, 2]
    assert abs(s[0] - pa[pa > -1].mean()) < 1e-6
    p50 = p_ref[0, :, :, 0, 2]
    assert abs(s[1] - p50[p50 > -1].mean()) < 1e-6
    ra = r_ref[:, :, 0, 2]
    assert abs(s[8] - ra[ra > -1].mean()) < 1e-6


def test_fixture_is_what_the_compiled_reference_produces():
    """Build container only: the live build of the reference's cocoeval.cpp reproduces the committed fixture exactly."""
    ref = reference_module()
    if ref is None:
        pytest.skip("the compiled reference evaluator is only loaded where /root/reference exists (oracle/build_ref.py)")
    fx = np.load(FIXTURE)
    for seed in SEEDS:
        gts, dts, n_img, n_cat = _scene(seed)
        p, r = _reference_curves(ref, gts, dts, n_img, n_cat)
        np.testing.assert_array_equal(p, fx[f"precision_{seed}"])
        np.testing.assert_array_equal(r, fx[f"recall_{seed}"])
