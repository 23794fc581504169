"""tscd_amd -- MI355X (gfx950) native kernels for the TSCD per-clip hot path.

Layout:  csrc/  hand-written HIP kernels + the C-ABI (declared in include/tscd_hip.h)
         _lib.py ctypes binding of libtscd_hip.so (fails loudly when the library is missing)
         ops.py  thin torch-tensor wrappers (borrow data_ptr()s on the current HIP stream)
The reference-facing module mirror (yolox.models / yolox.exp ...) lives in the top-level ``yolox`` package.
"""
__version__ = "0.1.0"
