"""TSCD-L on ImageNet-VID: the settings of the reference's exps/TSCD_VID/vid_tscd_large.py:12-52 -- 30 classes, clips of 4 local +
12 global frames in training and 1 + 31 in evaluation, a lower proposal clamp of 50 and NO upper clamp (every anchor with
obj * cls >= 0.001 is a proposal)."""
import os

from exps.TSCD_OVIS.ovis_tscd_large import Exp as LargeExp


class Exp(LargeExp):
    def __init__(self):
        super().__init__()
        self.exp_name = os.path.split(os.path.realpath(__file__))[1].split(".")[0]
        self.num_classes = 30
        self.lframe, self.gframe, self.lframe_val, self.gframe_val = 4, 12, 1, 31        # vid_tscd_large.py:27-30
        self.minimal_limit, self.maximal_limit = 50, 0                                   # :42 sets only the lower clamp
        self.output_dir = "./VID_TSCD_outputs/TSCD_Large"
        self.tracker_refine = True
        self.seed = 2024
