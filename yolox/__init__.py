"""Drop-in ``yolox`` import surface for the TSCD hot path, backed by tscd_amd's MI355X kernels.

Mirrors what tools/tscd_train.py, tools/tscd_eval.py and the TSCD exp files import from the reference's
``yolox`` package: same class names, constructor signatures, state_dict keys and forward
contracts; the arithmetic runs in hand-written HIP kernels (tscd_amd/csrc) instead of stock PyTorch ops.
"""
__version__ = "0.3.0"
