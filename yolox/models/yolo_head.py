"""YOLOXHead, evaluation forward (interface of the reference's yolox/models/yolo_head.py:18-265): the decoupled base-detector
head of BASELINE config 1 (tools/demo.py: YOLOX-S on one 640x640 image).  Same constructor, submodule names and state_dict
keys as the reference; the convs are tscd_amd's MFMA kernels, box decode + sigmoid one launch (proposal.hip decode_score).
Training of the base detector (get_losses, :267-660) is row N4 of SURVEY.md 8f and is not built."""
import math

import torch
import torch.nn as nn

from tscd_amd import functional as TF
from tscd_amd import ops

from .network_blocks import BaseConv


class YOLOXHead(nn.Module):
    def __init__(self, num_classes, width=1.0, strides=[8, 16, 32], in_channels=[256, 512, 1024], act="silu", depthwise=False,
                 debug=False):
        super().__init__()
        if depthwise:
            raise NotImplementedError("depthwise convs are not on the TSCD path")
        self.n_anchors = 1
        self.num_classes = num_classes
        self.decode_in_inference = True
        self.debug = debug
        self.cls_convs, self.reg_convs = nn.ModuleList(), nn.ModuleList()
        self.cls_preds, self.reg_preds, self.obj_preds = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.stems = nn.ModuleList()
        W = int(256 * width)
        tower = lambda: nn.Sequential(BaseConv(W, W, 3, 1, act=act), BaseConv(W, W, 3, 1, act=act))
        for i in range(len(in_channels)):
            self.stems.append(BaseConv(int(in_channels[i] * width), W, 1, 1, act=act))
            self.cls_convs.append(tower())
            self.reg_convs.append(tower())
            self.cls_preds.append(nn.Conv2d(W, self.n_anchors * num_classes, 1, 1, 0))
            self.reg_preds.append(nn.Conv2d(W, 4, 1, 1, 0))
            self.obj_preds.append(nn.Conv2d(W, self.n_anchors * 1, 1, 1, 0))
        self.use_l1 = False
        self.strides = strides
        self.grids = [torch.zeros(1)] * len(in_channels)

    def initialize_biases(self, prior_prob):
        for convs in (self.cls_preds, self.obj_preds):
            for conv in convs:
                b = conv.bias.view(self.n_anchors, -1)
                b.data.fill_(-math.log((1 - prior_prob) / prior_prob))
                conv.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    def forward(self, xin, labels=None, imgs=None):
        if self.training:
            raise NotImplementedError("training of the YOLOX base detector (yolo_head.py:267-660) is row N4 of SURVEY.md 8f")
        feats = [x.permute(0, 2, 3, 1) for x in xin]                 # channels-last maps -> NHWC views
        feats = [f if f.dtype == torch.float16 else f.half() for f in feats]
        raw = []
        with torch.no_grad():
            for k, x in enumerate(feats):
                x = self.stems[k](x)
                cls_feat = self.cls_convs[k][1](self.cls_convs[k][0](x))
                reg_feat = self.reg_convs[k][1](self.reg_convs[k][0](x))
                reg_o = TF.conv_bias(reg_feat, self.reg_preds[k].weight, self.reg_preds[k].bias)
                obj_o = TF.conv_bias(reg_feat, self.obj_preds[k].weight, self.obj_preds[k].bias)
                cls_o = TF.conv_bias(cls_feat, self.cls_preds[k].weight, self.cls_preds[k].bias)
                raw.append(torch.cat([reg_o, obj_o, cls_o], -1).contiguous())       # [B,h,w,5+NC]
            self.hw = [tuple(f.shape[1:3]) for f in feats]
            geo = ops.LevelGeometry(self.hw, self.strides)
            if not self.decode_in_inference:
                B = raw[0].shape[0]
                flat = torch.cat([r.reshape(B, -1, r.shape[-1]) for r in raw], 1)
                return torch.cat([flat[..., :4], torch.sigmoid(flat[..., 4:])], -1)
            outputs, det, _ = ops.decode_score(raw, geo, self.num_classes, True)
            # [B, A, 5+NC]: decoded cx,cy,w,h | sigmoid(obj) | sigmoid(cls) -- what decode_outputs returns (yolo_head.py:250-265)
            return torch.cat([outputs[..., :4], det[..., 4:5], det[..., 7:]], -1)
