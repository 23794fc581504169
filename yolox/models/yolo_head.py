"""YOLOXHead, evaluation forward (interface of the reference's yolox/models/yolo_head.py:18-265): the decoupled base-detector
head of BASELINE config 1 (tools/demo.py: YOLOX-S on one 640x640 image).  Same constructor, submodule names and state_dict
keys as the reference; the convs are tscd_amd's MFMA kernels, box decode + sigmoid one launch (proposal.hip decode_score).
Training (get_losses, :267-434) is ``forward_train``: the TSCD head's SimOTA and fused loss kernels on the base
detector's own outputs, with the box-branch gradient the frozen TSCD base head never needs."""
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

    def forward_train(self, feats, labels):
        """Training forward + losses (reference: forward :166-224 training branch, get_losses :267-434): decoupled towers and
        prediction convs on the autograd path, decode + SimOTA assignment without gradient (ops.decode_score / ops.simota_assign, the
        kernels of the TSCD head: same assignment code lineage), IoU / objectness / class losses and their logit gradients -- the
        box branch included -- in the fused loss kernel.  -> (loss, 5 * iou_loss, obj_loss, cls_loss, l1_loss, num_fg / num_gts)."""
        nc = self.num_classes
        raw_levels = []
        for k, x in enumerate(feats):
            x = self.stems[k](x)
            cls_feat = self.cls_convs[k](x)
            reg_feat = self.reg_convs[k](x)
            reg_o = TF.conv_bias(reg_feat, self.reg_preds[k].weight, self.reg_preds[k].bias)
            obj_o = TF.conv_bias(reg_feat, self.obj_preds[k].weight, self.obj_preds[k].bias)
            cls_o = TF.conv_bias(cls_feat, self.cls_preds[k].weight, self.cls_preds[k].bias)
            raw_levels.append(torch.cat([reg_o, obj_o, cls_o], -1))                    # [B,h,w,5+NC], with grad
        B = raw_levels[0].shape[0]
        self.hw = [tuple(f.shape[1:3]) for f in feats]
        geo = ops.LevelGeometry(self.hw, self.strides)
        raw_all = torch.cat([r.reshape(B, -1, 5 + nc) for r in raw_levels], 1).contiguous()
        with torch.no_grad():
            outputs, _, _ = ops.decode_score([r.detach().contiguous() for r in raw_levels], geo, nc, True)
            fg_mask, matched_gt, pred_iou, num_fg_b, ngt_b = ops.simota_assign(outputs, labels, geo, nc, labels.shape[1],
                                                                               emu16=self.simota_fp16, center_radius=2.5)
        lab = labels.float().contiguous()
        num_fg = num_fg_b.sum().float().clamp(min=1)
        num_gts = ngt_b.sum().float().clamp(min=1)
        sums = TF.dense_loss(raw_all, outputs, fg_mask, matched_gt, pred_iou, lab, nc, geo=geo)
        loss_iou, loss_obj, loss_cls = sums[0] / num_fg, sums[1] / num_fg, sums[2] / num_fg
        loss_l1 = loss_iou.new_zeros(())
        if self.use_l1:
            # L1 on the raw box logits of the foreground anchors against get_l1_target (:436-441): a dense masked sum, no gather
            dev = raw_all.device
            gx = torch.cat([torch.arange(w, device=dev).repeat(h) for h, w in self.hw]).float()
            gy = torch.cat([torch.arange(h, device=dev).repeat_interleave(w) for h, w in self.hw]).float()
            st = torch.cat([torch.full((h * w,), float(s), device=dev) for (h, w), s in zip(self.hw, self.strides)])
            gt = torch.gather(lab[..., 1:5], 1, matched_gt.long().clamp(min=0).unsqueeze(-1).expand(-1, -1, 4))       # [B,A,4]
            tgt = torch.stack([gt[..., 0] / st - gx, gt[..., 1] / st - gy, torch.log(gt[..., 2] / st + 1e-8), torch.log(gt[..., 3] / st + 1e-8)], -1)
            fg = fg_mask.bool().unsqueeze(-1)
            loss_l1 = (torch.where(fg, (raw_all[..., :4].float() - tgt).abs(), raw_all.new_zeros((), dtype=torch.float32))).sum() / num_fg
        reg_weight = 5.0
        loss = reg_weight * loss_iou + loss_obj + loss_cls + loss_l1
        return loss, reg_weight * loss_iou, loss_obj, loss_cls, loss_l1, num_fg / num_gts

    # The assignment's IoU chain runs in fp32 here.  Under the reference's autocast the chain is evaluated in fp16, where the area of a
    # box above 65504 px^2 (256 x 256) is inf and the IoU of two such boxes NaN -- at 640^2 with a randomly initialised box branch that
    # poisons the class targets from the first step (measured: cls_loss NaN with the fp16 emulation, finite with fp32).  True selects
    # the op-by-op fp16 emulation (ops.simota_assign emu16) for comparisons with a reference run.
    simota_fp16 = False

    def forward(self, xin, labels=None, imgs=None):
        feats = [x.permute(0, 2, 3, 1) for x in xin]                 # channels-last maps -> NHWC views
        feats = [f if f.dtype == torch.float16 else f.half() for f in feats]
        if self.training:
            return self.forward_train(feats, labels)
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
