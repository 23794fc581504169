"""Conv / CSP building blocks of CSPDarknet + PAFPN on the MI355X kernels.

Interface mirror of the reference's yolox/models/network_blocks.py (BaseConv :36-61, Bottleneck :158-180,
SPPBottleneck :201-223, CSPLayer :226-264, Focus :267-289): same submodule names (=> identical state_dict keys)
and constructor arguments.  Differences are internal: activations flow as NHWC fp16 tensors
([B,H,W,C] logical shape), every conv is tscd_amd's MFMA implicit GEMM with the BatchNorm statistics fused
into its epilogue, and CSP / SPP concatenations are channel slices of one pre-allocated buffer that the producing
kernels write into directly (no torch.cat copies).
"""
import torch
import torch.nn as nn

from tscd_amd import functional as TF
from tscd_amd import ops


def get_activation(name="silu", inplace=True):
    if name in ("silu", "swish"):
        return nn.SiLU(inplace=inplace)
    if name == "relu":
        return nn.ReLU(inplace=inplace)
    if name == "lrelu":
        return nn.LeakyReLU(0.1, inplace=inplace)
    raise AttributeError("Unsupported act type: {}".format(name))


# CSPLayer's conv1 || conv2 as one launch on the gradient-free path (TSCD_FUSE_CSP=0: two launches)
import os as _os
FUSE_CSP = _os.environ.get("TSCD_FUSE_CSP", "1") != "0"
_DEFERRED_TICKS = None     # list of num_batches_tracked buffers while a defer_bn_ticks() scope is open


class defer_bn_ticks:
    """Batches the ``num_batches_tracked += 1`` of every BatchNorm that runs inside the scope into ONE fused add at exit
    (134 one-element kernels per TSCD step otherwise).  Re-entrant: only the outermost scope flushes."""

    def __enter__(self):
        global _DEFERRED_TICKS
        self.outer = _DEFERRED_TICKS
        if self.outer is None:
            _DEFERRED_TICKS = []
        return self

    def __exit__(self, *exc):
        global _DEFERRED_TICKS
        if self.outer is None:
            ticks, _DEFERRED_TICKS = _DEFERRED_TICKS, None
            if ticks:
                # a module may run more than once inside the scope (tscd.py runs the next clip's backbone pass in the
                # assignment window): one entry per buffer and pass count, a fused add must not see a tensor twice
                count = {}
                for t in ticks:
                    count[id(t)] = (t, count[id(t)][1] + 1) if id(t) in count else (t, 1)
                by_n = {}
                for t, n in count.values():
                    by_n.setdefault(n, []).append(t)
                for n, ts in by_n.items():
                    torch._foreach_add_(ts, n)
        return False


def _grad_path(base_conv):
    """True when this block has to run on the autograd path: gradients are enabled and its weights train (row N4; the TSCD exps
    freeze the whole backbone and take the in-place no-grad path)."""
    return torch.is_grad_enabled() and base_conv.conv.weight.requires_grad


class BaseConv(nn.Module):
    """Conv2d(no bias, same pad) -> BatchNorm2d -> SiLU, one fused HIP pipeline."""

    def __init__(self, in_channels, out_channels, ksize, stride, groups=1, bias=False, act="silu"):
        super().__init__()
        if groups != 1 or bias or act not in ("silu", "swish"):
            raise NotImplementedError("TSCD path uses groups=1, bias=False, SiLU only")
        pad = (ksize - 1) // 2
        # nn.Conv2d / nn.BatchNorm2d only hold the parameters (state_dict compatibility); they are never called
        self.conv = nn.Conv2d(in_channels, out_channels, ksize, stride, pad, groups=groups, bias=bias)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = get_activation(act, inplace=True)
        self.stride = stride

    def forward(self, x, out=None, residual=None):
        bn = self.bn
        training = self.training and bn.training
        if training and bn.track_running_stats:
            if _DEFERRED_TICKS is not None:
                _DEFERRED_TICKS.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked += 1
        if self.training and not training and _grad_path(self):
            # BatchNorm in eval() inside a TRAINING model with trainable weights (the exps' fix_bn=True option,
            # ovis_tscd_large.py:153-160; every shipped exp sets fix_bn=False) needs the running-statistics backward, which is not
            # built: refuse instead of silently dropping the gradient
            raise NotImplementedError("BaseConv: gradients through an eval-mode BatchNorm (fix_bn=True) are not implemented")
        if not training:
            # evaluation: scale / shift of the folded BatchNorm, computed once per state of the four BN tensors (a .half() model
            # paid four dtype conversions and a fold kernel per conv and clip otherwise: 5 of its 6 launches)
            ver = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version, bn.weight.data_ptr())
            d = self.__dict__
            if d.get("_folded_ver") != ver or d.get("_folded") is None or d["_folded"].device != x.device:
                d["_folded"] = ops.bn_fold(bn.weight.float(), bn.bias.float(), bn.running_mean.float(), bn.running_var.float(), bn.eps)
                d["_folded_ver"] = ver
            return TF.conv_bn_silu(x, self.conv.weight, None, None, None, None, self.stride, False, out=out, residual=residual,
                                   eps=bn.eps, momentum=bn.momentum, folded=d["_folded"])
        y = TF.conv_bn_silu(x, self.conv.weight, bn.weight.float(), bn.bias.float(),
                            bn.running_mean if bn.running_mean.dtype == torch.float32 else bn.running_mean.float(),
                            bn.running_var if bn.running_var.dtype == torch.float32 else bn.running_var.float(),
                            self.stride, training, out=out, residual=residual, eps=bn.eps, momentum=bn.momentum)
        return y

    fuseforward = forward


class Bottleneck(nn.Module):
    def __init__(self, in_channels, out_channels, shortcut=True, expansion=0.5, depthwise=False, act="silu"):
        super().__init__()
        if depthwise:
            raise NotImplementedError("depthwise convs are not on the TSCD path")
        hidden = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = BaseConv(hidden, out_channels, 3, stride=1, act=act)
        self.use_add = shortcut and in_channels == out_channels

    def forward(self, x, out=None):
        if _grad_path(self.conv1):            # row N4: autograd path (new tensors, the residual add is an autograd op)
            y = self.conv2(self.conv1(x))
            return y + x if self.use_add else y
        return self.conv2(self.conv1(x), out=out, residual=x if self.use_add else None)


class SPPBottleneck(nn.Module):
    """1x1 -> max-pools 5/9/13 (the 5-pool cascaded, written into slices of one buffer) -> 1x1."""

    def __init__(self, in_channels, out_channels, kernel_sizes=(5, 9, 13), activation="silu"):
        super().__init__()
        assert tuple(kernel_sizes) == (5, 9, 13)
        hidden = in_channels // 2
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=activation)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=ks, stride=1, padding=ks // 2) for ks in kernel_sizes])
        self.conv2 = BaseConv(hidden * (len(kernel_sizes) + 1), out_channels, 1, stride=1, act=activation)

    def forward(self, x):
        if _grad_path(self.conv1):
            x = self.conv1(x)
            p1 = TF.maxpool5(x)
            p2 = TF.maxpool5(p1)               # cascaded 5x5 pools = the 9x9 and 13x13 windows of the reference
            p3 = TF.maxpool5(p2)
            return self.conv2(torch.cat([x, p1, p2, p3], dim=-1))
        B, H, W, _ = x.shape
        h = self.conv1.conv.out_channels
        buf = ops.empty_nhwc(B, H, W, 4 * h, x.device)
        self.conv1(x, out=buf[..., :h])
        for i in range(3):
            ops.maxpool(buf[..., i * h:(i + 1) * h], buf[..., (i + 1) * h:(i + 2) * h], 5)
        return self.conv2(buf)


class CSPLayer(nn.Module):
    def __init__(self, in_channels, out_channels, n=1, shortcut=True, expansion=0.5, depthwise=False, act="silu"):
        super().__init__()
        hidden = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv3 = BaseConv(2 * hidden, out_channels, 1, stride=1, act=act)
        self.m = nn.Sequential(*[Bottleneck(hidden, hidden, shortcut, 1.0, depthwise, act=act) for _ in range(n)])

    def forward(self, x, out=None):
        if _grad_path(self.conv1):
            x1 = self.m(self.conv1(x))
            return self.conv3(torch.cat([x1, self.conv2(x)], dim=-1))
        B, H, W, _ = x.shape
        h = self.conv1.conv.out_channels
        n = len(self.m)
        if FUSE_CSP and x.is_cuda:
            from . import tower_fusion as TW
            if TW.fusable([self.conv1, self.conv2]):
                # conv1 || conv2 read the same input: ONE launch over the 2h concatenated output channels, straight into the concat
                # buffer [conv1(x) | conv2(x)]; the bottlenecks then run on its first half and the last one writes back into it
                cat = TW.shared_input_pair_nograd(x, self.conv1, self.conv2)
                x1 = cat[..., :h]
                for i, blk in enumerate(self.m):
                    x1 = blk(x1, out=cat[..., :h] if i == n - 1 else None)
                return self.conv3(cat, out=out)
        cat = ops.empty_nhwc(B, H, W, 2 * h, x.device)      # [m(conv1(x)) | conv2(x)]
        self.conv2(x, out=cat[..., h:])
        x1 = self.conv1(x, out=cat[..., :h] if n == 0 else None)
        for i, blk in enumerate(self.m):
            x1 = blk(x1, out=cat[..., :h] if i == n - 1 else None)
        return self.conv3(cat, out=out)


class Focus(nn.Module):
    def __init__(self, in_channels, out_channels, ksize=1, stride=1, act="silu"):
        super().__init__()
        self.conv = BaseConv(in_channels * 4, out_channels, ksize, stride, act=act)

    def forward(self, img_nchw):
        return self.conv(ops.focus_to_nhwc16(img_nchw))
