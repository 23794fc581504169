"""YOLOX-S base detector (reference: exps/default/yolox_s.py + yolox/exp/yolox_base.py:15-110): depth 0.33, width 0.50, 80 COCO
classes, test size 640x640 -- BASELINE.json config 1 (tools/demo.py on a single image)."""
import os

import torch.nn as nn

from yolox.exp import BaseExp


class Exp(BaseExp):
    def __init__(self):
        super().__init__()
        self.num_classes, self.depth, self.width, self.act = 80, 0.33, 0.50, "silu"
        self.input_size, self.test_size, self.test_conf, self.nmsthre = (640, 640), (640, 640), 0.01, 0.65
        self.exp_name = os.path.split(os.path.realpath(__file__))[1].split(".")[0]

    def get_model(self):
        from yolox.models import YOLOPAFPN, YOLOX, YOLOXHead
        if getattr(self, "model", None) is None:
            in_channels = [256, 512, 1024]
            backbone = YOLOPAFPN(self.depth, self.width, in_channels=in_channels, act=self.act)
            head = YOLOXHead(self.num_classes, self.width, in_channels=in_channels, act=self.act)
            self.model = YOLOX(backbone, head)
            for m in self.model.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eps, m.momentum = 1e-3, 0.03
            self.model.head.initialize_biases(1e-2)
        return self.model

    def get_data_loader(self, batch_size, is_distributed, no_aug=False, cache_img=False):
        raise NotImplementedError("training of the YOLOX base detector is row N4 of SURVEY.md 8f (not built)")

    def get_optimizer(self, batch_size):
        raise NotImplementedError("training of the YOLOX base detector is row N4 of SURVEY.md 8f (not built)")

    def get_lr_scheduler(self, lr, iters_per_epoch):
        raise NotImplementedError("training of the YOLOX base detector is row N4 of SURVEY.md 8f (not built)")

    def get_evaluator(self, batch_size, is_distributed, testdev=False, legacy=False):
        raise NotImplementedError("COCO evaluation of the base detector is outside the TSCD hot path")

    def eval(self, model, evaluator, is_distributed, half=False):
        return evaluator.evaluate(model, is_distributed, half)
