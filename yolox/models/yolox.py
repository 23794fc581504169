"""YOLOX model wrapper (interface of the reference's yolox/models/yolox.py:11-48)."""
import torch.nn as nn

from .yolo_head import YOLOXHead
from .yolo_pafpn import YOLOPAFPN


class YOLOX(nn.Module):
    """backbone (YOLOPAFPN) + YOLOXHead; eval returns the decoded predictions [B, A, 5+NC]."""

    def __init__(self, backbone=None, head=None):
        super().__init__()
        self.backbone = YOLOPAFPN() if backbone is None else backbone
        self.head = YOLOXHead(80) if head is None else head

    def forward(self, x, targets=None):
        fpn_outs = self.backbone(x)
        if self.training:
            assert targets is not None
            loss, iou_loss, conf_loss, cls_loss, l1_loss, num_fg = self.head(fpn_outs, targets, x)
            return {"total_loss": loss, "iou_loss": iou_loss, "l1_loss": l1_loss, "conf_loss": conf_loss, "cls_loss": cls_loss,
                    "num_fg": num_fg}
        return self.head(fpn_outs)
