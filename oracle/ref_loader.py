"""TEST INFRASTRUCTURE (build container only) -- imports the read-only reference tree.

Used by oracle/make_golden.py to (a) validate oracle/tscd_oracle.py against the reference itself and
(b) emit the golden vectors committed under tests/golden/.  Nothing here runs on the GPU box:
/root/reference does not exist there and no test / bench / smoke imports this module.

The reference needs third-party modules that are absent from this image.  They are
replaced by minimal stand-ins in ``sys.modules`` *before* the reference is imported:

* loguru            -> no-op logger
* cv2, thop, pycocotools(.coco/.cocoeval/.mask), timm.models.layers -> import-only stubs
* pywt.Wavelet('haar') -> the four analytic Haar taps (dec_lo=[s,s], dec_hi=[-s,s], rec_lo=[s,s],
  rec_hi=[s,-s], s=1/sqrt(2))
* torchvision.ops.{nms,batched_nms,box_area} -> a plain O(n^2) greedy NMS following the published
  torchvision semantics (sort by score desc, stable; suppress IoU > thr; batched = per-class offset
  of boxes by idx*(max_coord+1)).  NMS is therefore "parity unpinned" by the reference.

``torch.Tensor.to`` is wrapped so that the hard-coded ``.to('cuda')`` in
yolox/models/post_trans.py:694-695 lands on the CPU.
"""
import sys
import types
import math

sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference

import torch

REF_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _NoopLogger:
    def __getattr__(self, _):
        def f(*a, **k):
            # logger.catch is used as a decorator
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return self
        return f

    def catch(self, *a, **k):
        if len(a) == 1 and callable(a[0]):
            return a[0]
        return lambda fn: fn


import os as _os
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from oracle.nms_ref import nms, batched_nms, box_area  # noqa: E402  (single restatement, see there)


def install_stubs():
    if "loguru" in sys.modules and getattr(sys.modules["loguru"], "_tscd_stub", False):
        return
    _mod("loguru", logger=_NoopLogger(), _tscd_stub=True)
    _mod("cv2", INTER_LINEAR=1, setNumThreads=lambda *a: None,
         ocl=types.SimpleNamespace(setUseOpenCL=lambda *a: None))
    _mod("thop", profile=lambda *a, **k: (0, 0))
    pc = _mod("pycocotools")
    pc.coco = _mod("pycocotools.coco", COCO=object)
    pc.cocoeval = _mod("pycocotools.cocoeval", COCOeval=object)
    pc.mask = _mod("pycocotools.mask")

    # timm.models.layers: DropPath / to_2tuple / trunc_normal_ (only touched by unused backbones)
    class DropPath(torch.nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    timm = _mod("timm")
    timm.models = _mod("timm.models")
    timm.models.layers = _mod("timm.models.layers", DropPath=DropPath,
                              to_2tuple=lambda x: (x, x) if not isinstance(x, tuple) else x,
                              trunc_normal_=torch.nn.init.trunc_normal_)

    s = 1.0 / math.sqrt(2.0)

    class Wavelet:
        def __init__(self, name):
            assert name == "haar"
            self.dec_lo, self.dec_hi = [s, s], [-s, s]
            self.rec_lo, self.rec_hi = [s, s], [s, -s]

    _mod("pywt", Wavelet=Wavelet)

    tv = _mod("torchvision")
    tv.ops = _mod("torchvision.ops", nms=nms, batched_nms=batched_nms, box_area=box_area)
    tv.ops.boxes = _mod("torchvision.ops.boxes", box_area=box_area, nms=nms, batched_nms=batched_nms)
    tv.transforms = _mod("torchvision.transforms")
    tv.models = _mod("torchvision.models")

    # F7: '.to("cuda")' inside the attention -> CPU
    _orig_to = torch.Tensor.to

    def _to(self, *args, **kwargs):
        args = tuple("cpu" if isinstance(a, str) and a.startswith("cuda") else a for a in args)
        if isinstance(kwargs.get("device"), str) and kwargs["device"].startswith("cuda"):
            kwargs["device"] = "cpu"
        return _orig_to(self, *args, **kwargs)

    torch.Tensor.to = _to
    torch.Tensor.cuda = lambda self, *a, **k: self


def import_reference():
    """Returns the reference's ``yolox`` package (imported from /root/reference)."""
    install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import yolox  # noqa: F401  (the reference's package, NOT this repo's mirror)
    assert yolox.__file__.startswith(REF_ROOT), yolox.__file__
    return yolox
