"""TSCD-L on ImageNet-VID (reference: exps/TSCD_VID/vid_tscd_large.py:12-190): 30 classes, no upper proposal clamp."""
import os

from exps.TSCD_OVIS.ovis_tscd_large import Exp as LargeExp


class Exp(LargeExp):
    def __init__(self):
        super().__init__()
        self.exp_name = os.path.split(os.path.realpath(__file__))[1].split(".")[0]
        self.num_classes = 30
        self.maximal_limit = 0          # vid_tscd_large.py:42 sets only minimal_limit
        self.output_dir = "./VID_TSCD_outputs/TSCD_Large"
        self.lframe, self.gframe, self.lframe_val, self.gframe_val = 8, 24, 8, 24
