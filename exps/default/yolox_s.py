"""YOLOX-S base detector (reference: exps/default/yolox_s.py over yolox/exp/yolox_base.py): depth 0.33, width 0.50, 80 COCO classes,
640x640 -- BASELINE.json config 1 (tools/demo.py on a single image) and stage 1 of the training recipe (tools/train.py, row N4)."""
import os

from yolox.exp.yolox_base import Exp as BaseDetectorExp


class Exp(BaseDetectorExp):
    def __init__(self):
        super().__init__()
        self.depth, self.width = 0.33, 0.50
        self.exp_name = os.path.split(os.path.realpath(__file__))[1].split(".")[0]
