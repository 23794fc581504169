"""Torch-tensor wrappers over the C-ABI (include/tscd_hip.h).

Activations are fp16 tensors with LOGICAL shape [B, H, W, C] (NHWC), either contiguous or a channel slice of a
contiguous NHWC buffer (``stride(-1) == 1``, pixel stride ``ld = stride(-2)``) -- concat-free CSP / PAFPN
buffers are plain torch views.  Kernels run on ``torch.cuda.current_stream()`` and only borrow ``data_ptr()``s.
PyTorch is plumbing here (allocation, streams, autograd bookkeeping); all arithmetic is in the HIP kernels.
"""
import torch

from ._lib import call, lib

EPI_RAW, EPI_BIAS, EPI_BIAS_RELU, EPI_BN_SILU, EPI_BN = 0, 1, 2, 3, 4
ACT_NONE, ACT_SILU, ACT_RELU = 0, 1, 2
BN_EPS, BN_MOMENTUM = 1e-3, 0.03  # reference: exps/TSCD_OVIS/ovis_tscd_large.py:109-113


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def require_gpu(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"tscd_amd: {name} is on {t.device}; the TSCD kernels run on an MI355X only "
                           "(no CPU / PyTorch fallback in the product path)")


def nhwc_ld(t):
    """Pixel stride (elements) of an NHWC tensor / channel-slice view; raises if the layout is not NHWC-dense."""
    assert t.dim() == 4 and t.dtype == torch.float16, (t.shape, t.dtype)
    B, H, W, C = t.shape
    s = t.stride()
    ld = s[2] if W > 1 else (s[1] if H > 1 else (s[0] if B > 1 else C))
    ok = (C == 1 or s[3] == 1) and (W == 1 or s[2] == ld) and (H == 1 or s[1] == W * ld) and (B == 1 or s[0] == H * W * ld)
    if not ok or ld < C:
        raise ValueError(f"not an NHWC(-slice) layout: shape {tuple(t.shape)} strides {s}")
    return ld


def empty_nhwc(B, H, W, C, device):
    return torch.empty((B, H, W, C), dtype=torch.float16, device=device)


# ----------------------------------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------------------------------
def pack_conv_weight(w, cin_pad=None, mode=0):
    """torch [Cout,Cin,KH,KW] (fp32/fp16) -> packed fp16 [rows_pad, Kpad] for tscd_conv2d_nhwc_f16.
    mode=1 packs the matrix that makes the same kernel compute the data gradient of a stride-1 conv."""
    require_gpu(w, "weight")
    Cout, Cin, KH, KW = w.shape
    w = w.detach().contiguous()
    if mode == 0:
        inner = cin_pad or Cin
        rows = Cout
    else:
        inner, rows = Cout, Cin
    bn = lib.tscd_conv_tile_n(rows)
    rows_pad = (rows + bn - 1) // bn * bn
    kpad = (KH * KW * inner + 31) // 32 * 32
    out = torch.empty((rows_pad, kpad), dtype=torch.float16, device=w.device)
    assert w.dtype in (torch.float32, torch.float16)
    call("tscd_pack_conv_weight", w.data_ptr(), int(w.dtype == torch.float32), out.data_ptr(), Cout, Cin, inner, KH, KW,
         rows_pad, kpad, mode, _stream())
    return out


class WeightCache:
    """Packed-weight cache keyed on the parameter's storage and in-place version counter, so frozen weights are
    packed once and trainable ones are re-packed only after an optimizer step / load_state_dict."""

    def __init__(self):
        self._c = {}

    def get(self, w, cin_pad=None, mode=0):
        key = (id(w), mode, cin_pad)
        ent = self._c.get(key)
        tag = (w.data_ptr(), w._version, w.dtype, w.device)
        if ent is None or ent[0] != tag:
            ent = (tag, pack_conv_weight(w, cin_pad, mode))
            self._c[key] = ent
        return ent[1]


WEIGHTS = WeightCache()


# ----------------------------------------------------------------------------------------------------
# conv / BN / elementwise
# ----------------------------------------------------------------------------------------------------
def conv_out_hw(H, W, ksize, stride):
    pad = (ksize - 1) // 2
    return (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1


def conv2d(x, wpk, cout, ksize, stride=1, epilogue=EPI_RAW, bias=None, scale_shift=None, out=None, want_stats=False):
    """Implicit-GEMM conv.  x [B,H,W,Cin] fp16 NHWC(-slice).  Returns (y, bn_partial or None)."""
    require_gpu(x, "conv input")
    B, H, W, Cin = x.shape
    ldx = nhwc_ld(x)
    Ho, Wo = conv_out_hw(H, W, ksize, stride)
    if out is None:
        out = empty_nhwc(B, Ho, Wo, cout, x.device)
    assert tuple(out.shape) == (B, Ho, Wo, cout), (out.shape, (B, Ho, Wo, cout))
    ldy = nhwc_ld(out)
    kpad = (ksize * ksize * Cin + 31) // 32 * 32
    assert wpk.shape[1] == kpad and wpk.shape[0] >= cout, (wpk.shape, kpad, cout)
    partial = None
    if want_stats:
        mt = (B * Ho * Wo + lib.tscd_conv_tile_m() - 1) // lib.tscd_conv_tile_m()
        partial = torch.empty((mt, 2, wpk.shape[0]), dtype=torch.float32, device=x.device)
    scale = shift = None
    if scale_shift is not None:
        scale, shift = scale_shift[0], scale_shift[1]
    call("tscd_conv2d_nhwc_f16", x.data_ptr(), wpk.data_ptr(), out.data_ptr(), _ptr(bias), _ptr(scale), _ptr(shift),
         _ptr(partial), B, H, W, Cin, cout, ksize, stride, ldx, ldy, epilogue, _stream())
    return out, partial


def bn_finalize(partial, count, gamma, beta, running_mean, running_var, eps=BN_EPS, momentum=BN_MOMENTUM):
    """-> (scale_shift [2,C], save_mean [C], save_invstd [C]); updates running stats in place when given."""
    C = gamma.shape[0]
    ss = torch.empty((2, C), dtype=torch.float32, device=partial.device)
    mean = torch.empty(C, dtype=torch.float32, device=partial.device)
    invstd = torch.empty(C, dtype=torch.float32, device=partial.device)
    call("tscd_bn_finalize", partial.data_ptr(), partial.shape[0], partial.shape[2], C, int(count), gamma.data_ptr(),
         beta.data_ptr(), eps, momentum, _ptr(running_mean), _ptr(running_var), ss.data_ptr(), mean.data_ptr(),
         invstd.data_ptr(), _stream())
    return ss, mean, invstd


def bn_fold(gamma, beta, running_mean, running_var, eps=BN_EPS):
    C = gamma.shape[0]
    ss = torch.empty((2, C), dtype=torch.float32, device=gamma.device)
    call("tscd_bn_fold", gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), eps, C,
         ss.data_ptr(), _stream())
    return ss


def bn_act(x, scale_shift, act=ACT_SILU, residual=None, out=None):
    """out = act(x*scale+shift) (+ residual); in place on x when ``out`` is None."""
    B, H, W, C = x.shape
    out = x if out is None else out
    call("tscd_bn_act_nhwc_f16", x.data_ptr(), nhwc_ld(x), out.data_ptr(), nhwc_ld(out), scale_shift.data_ptr(),
         _ptr(residual), nhwc_ld(residual) if residual is not None else 0, B * H * W, C, act, _stream())
    return out


def focus_to_nhwc16(img):
    """NCHW image (fp32/fp16) -> [B,H/2,W/2,16] fp16 (12 Focus channels + 4 zero pad)."""
    require_gpu(img, "image")
    img = img.contiguous()
    B, C, H, W = img.shape
    assert C == 3 and img.dtype in (torch.float32, torch.float16)
    out = empty_nhwc(B, H // 2, W // 2, 16, img.device)
    call("tscd_focus_nchw_to_nhwc16", img.data_ptr(), int(img.dtype == torch.float32), out.data_ptr(), B, H, W, _stream())
    return out


def copy_slice(src, dst, upsample=False):
    B, H, W, C = dst.shape
    call("tscd_copy_slice_nhwc_f16", src.data_ptr(), nhwc_ld(src), dst.data_ptr(), nhwc_ld(dst), B, H, W, C, int(upsample),
         _stream())
    return dst


def maxpool(src, dst, ks=5):
    B, H, W, C = src.shape
    call("tscd_maxpool_nhwc_f16", src.data_ptr(), nhwc_ld(src), dst.data_ptr(), nhwc_ld(dst), B, H, W, C, ks, _stream())
    return dst


def haar_hf(x):
    B, H, W, C = x.shape
    hf = empty_nhwc(B, H // 2, W // 2, 3 * C, x.device)
    call("tscd_haar_hf_nhwc_f16", x.data_ptr(), nhwc_ld(x), hf.data_ptr(), B, H, W, C, _stream())
    return hf


def haar_idwt_mul(hf, content, out=None):
    B, H, W, C = content.shape
    assert hf.is_contiguous() and tuple(hf.shape) == (B, H // 2, W // 2, 3 * C)
    out = empty_nhwc(B, H, W, C, content.device) if out is None else out
    call("tscd_haar_idwt_mul_nhwc_f16", hf.data_ptr(), content.data_ptr(), nhwc_ld(content), out.data_ptr(), nhwc_ld(out),
         B, H, W, C, _stream())
    return out
