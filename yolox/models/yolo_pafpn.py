"""YOLOPAFPN neck (interface of the reference's yolox/models/yolo_pafpn.py:12-116) on the MI355X kernels.

Every torch.cat of the reference is a pre-allocated NHWC buffer whose channel slices are written directly by the
producing conv / upsample kernel: [up(fpn_out0) | dark4], [up(fpn_out1) | dark3], [bu_conv2 | fpn_out1],
[bu_conv1 | fpn_out0].  Returns NCHW-logical views (channels-last memory) to keep the reference's contract.
"""
import torch
import torch.nn as nn

from tscd_amd import ops

from .darknet import CSPDarknet
from .network_blocks import BaseConv, CSPLayer


class YOLOPAFPN(nn.Module):
    def __init__(self, depth=1.0, width=1.0, in_features=("dark3", "dark4", "dark5"), in_channels=[256, 512, 1024],
                 depthwise=False, act="silu"):
        super().__init__()
        self.backbone = CSPDarknet(depth, width, depthwise=depthwise, act=act)
        self.in_features = in_features
        self.in_channels = in_channels
        c0, c1, c2 = (int(c * width) for c in in_channels)
        n = round(3 * depth)
        self.upsample = nn.Upsample(scale_factor=2, mode="nearest")
        self.lateral_conv0 = BaseConv(c2, c1, 1, 1, act=act)
        self.C3_p4 = CSPLayer(2 * c1, c1, n, False, depthwise=depthwise, act=act)
        self.reduce_conv1 = BaseConv(c1, c0, 1, 1, act=act)
        self.C3_p3 = CSPLayer(2 * c0, c0, n, False, depthwise=depthwise, act=act)
        self.bu_conv2 = BaseConv(c0, c0, 3, 2, act=act)
        self.C3_n3 = CSPLayer(2 * c0, c1, n, False, depthwise=depthwise, act=act)
        self.bu_conv1 = BaseConv(c1, c1, 3, 2, act=act)
        self.C3_n4 = CSPLayer(2 * c1, c2, n, False, depthwise=depthwise, act=act)
        self._c = (c0, c1, c2)

    def forward_nhwc(self, img):
        p2, carry = self.top_down_nhwc(img)
        return (p2,) + self.bottom_up_nhwc(p2, carry)

    def top_down_nhwc(self, img):
        """Backbone + the top-down half of the neck: -> (pan_out2, carry) -- the stride-8 output is complete here, before the bottom-up
        half (``bottom_up_nhwc(pan_out2, carry)`` -> pan_out1, pan_out0), so a caller can start that level's head next to it."""
        ops.require_gpu(img, "input image")
        c0, c1, c2 = self._c
        B, _, H, W = img.shape
        if H % 32 or W % 32:
            raise ValueError(f"input {H}x{W} must be a multiple of 32 (64 for the wavelet head)")
        dev = img.device
        cat16 = ops.empty_nhwc(B, H // 16, W // 16, 2 * c1, dev)   # [up(fpn_out0) | x1 = dark4]
        cat8 = ops.empty_nhwc(B, H // 8, W // 8, 2 * c0, dev)      # [up(fpn_out1) | x2 = dark3]
        feats = self.backbone(img, outs={"dark4": cat16[..., c1:], "dark3": cat8[..., c0:]})
        x0 = feats["dark5"]
        n16 = ops.empty_nhwc(B, H // 16, W // 16, 2 * c0, dev)     # [bu_conv2(pan_out2) | fpn_out1]
        n32 = ops.empty_nhwc(B, H // 32, W // 32, 2 * c1, dev)     # [bu_conv1(pan_out1) | fpn_out0]
        fpn_out0 = self.lateral_conv0(x0, out=n32[..., c1:])
        ops.copy_slice(fpn_out0, cat16[..., :c1], upsample=True)
        f_out0 = self.C3_p4(cat16)
        fpn_out1 = self.reduce_conv1(f_out0, out=n16[..., c0:])
        ops.copy_slice(fpn_out1, cat8[..., :c0], upsample=True)
        pan_out2 = self.C3_p3(cat8)
        return pan_out2, (n16, n32)

    def bottom_up_nhwc(self, pan_out2, carry):
        c0, c1, c2 = self._c
        n16, n32 = carry
        self.bu_conv2(pan_out2, out=n16[..., :c0])
        pan_out1 = self.C3_n3(n16)
        self.bu_conv1(pan_out1, out=n32[..., :c1])
        pan_out0 = self.C3_n4(n32)
        return pan_out1, pan_out0

    def forward_nhwc_grad(self, img):
        """The same network on the autograd path (the base detector trains backbone and neck): every block returns
        a new tensor, concatenations are torch.cat, the upsample and the SPP pools are autograd Functions over their HIP kernels, the
        stride-2 convs take their gradients from the stride-1 kernels on the zero-inserted dY (tscd_amd/functional.py)."""
        from tscd_amd import functional as TF
        ops.require_gpu(img, "input image")
        if img.shape[2] % 32 or img.shape[3] % 32:
            raise ValueError(f"input {img.shape[2]}x{img.shape[3]} must be a multiple of 32")
        feats = self.backbone(img)
        x2, x1, x0 = feats["dark3"], feats["dark4"], feats["dark5"]
        fpn_out0 = self.lateral_conv0(x0)
        f_out0 = self.C3_p4(torch.cat([TF.upsample2x(fpn_out0), x1], dim=-1))
        fpn_out1 = self.reduce_conv1(f_out0)
        pan_out2 = self.C3_p3(torch.cat([TF.upsample2x(fpn_out1), x2], dim=-1))
        pan_out1 = self.C3_n3(torch.cat([self.bu_conv2(pan_out2), fpn_out1], dim=-1))
        pan_out0 = self.C3_n4(torch.cat([self.bu_conv1(pan_out1), fpn_out0], dim=-1))
        return pan_out2, pan_out1, pan_out0

    def forward(self, input):
        plist = self.__dict__.get("_plist")
        if plist is None:                       # flat parameter list, built once (Module.parameters() walks the module tree)
            plist = self.__dict__["_plist"] = list(self.parameters())
        if torch.is_grad_enabled() and any(p.requires_grad for p in plist):
            if not all(p.requires_grad for p in plist):
                raise NotImplementedError("partially frozen backbone / neck: freeze all of it (TSCD exps) or none (base detector)")
            outs = self.forward_nhwc_grad(input)
        else:
            with torch.no_grad():
                outs = self.forward_nhwc(input)
        return tuple(o.permute(0, 3, 1, 2) for o in outs)
