"""TEST INFRASTRUCTURE -- writes tests/golden/cocoeval_ref_curves.npz: the precision / recall arrays that the reference's OWN compiled
evaluator (oracle/_ref, built by oracle/build_ref.py from /root/reference/yolox/layers/cocoeval/cocoeval.cpp) produces for the four
seeded scenes of tests/test_cocoeval_ref_pin.py.  The fixture is data (numbers), so the pin also holds where the compiled reference
is absent or must not be loaded (the GPU box).  Run in the build container:  python -m oracle.make_golden_cocoeval"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_cocoeval_ref_pin as T
    ref = T.reference_module()
    assert ref is not None, "oracle/_ref is not built: python -m oracle.build_ref (needs /root/reference)"
    out = {}
    for seed in T.SEEDS:
        gts, dts, n_img, n_cat = T._scene(seed)
        p, r = T._reference_curves(ref, gts, dts, n_img, n_cat)
        out[f"precision_{seed}"], out[f"recall_{seed}"] = p.astype(np.float64), r.astype(np.float64)
    path = os.path.join(ROOT, "tests", "golden", "cocoeval_ref_curves.npz")
    np.savez_compressed(path, **out)
    print(path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
