"""TSCD model wrapper (interface of the reference's yolox/models/tscd.py:11-50)."""
import torch.nn as nn

from .network_blocks import defer_bn_ticks


class TSCD(nn.Module):
    """backbone (YOLOPAFPN) + TSCDHead.  Training returns the reference's loss dict, eval (result, result_ori)."""

    def __init__(self, backbone=None, head=None):
        super().__init__()
        self.backbone = backbone
        self.head = head

    def forward(self, x, targets=None, time_embedding=None, nms_thresh=0.5, lframe=0, gframe=32, resume=False):
        with defer_bn_ticks():
            return self._forward(x, targets, time_embedding, nms_thresh, lframe, gframe, resume)

    def _forward(self, x, targets, time_embedding, nms_thresh, lframe, gframe, resume):
        fpn_outs = self.backbone(x)
        if self.training:
            assert targets is not None
            (loss, iou_loss, conf_loss, cls_loss, l1_loss, num_fg, loss_matched_cls, loss_matched_obj,
             loss_matched_iou) = self.head(fpn_outs, targets, x, time_embedding, lframe=lframe, gframe=gframe, resume=resume)
            return {"total_loss": loss, "iou_loss": iou_loss, "l1_loss": l1_loss, "conf_loss": conf_loss, "cls_loss": cls_loss,
                    "num_fg": num_fg, "loss_matched_cls": loss_matched_cls, "loss_matched_obj": loss_matched_obj,
                    "loss_matched_iou": loss_matched_iou}
        return self.head(fpn_outs, targets, x, time_embedding, nms_thresh=nms_thresh, lframe=lframe, gframe=gframe,
                         resume=resume)
