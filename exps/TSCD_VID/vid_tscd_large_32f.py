"""Benchmark variant of the ImageNet-VID exp (BASELINE.json config 4): 32-frame clips (8 local + 24 global) in training and
evaluation, and the synthetic-run proposal cap of SURVEY.md 8d (2000 per frame: 60000 (proposal, class) candidates per frame in
the final NMS).  Not a reference exp -- the reference's vid_tscd_large.py uses 4 + 12 / 1 + 31 frames."""
import os

from exps.TSCD_VID.vid_tscd_large import Exp as VidExp


class Exp(VidExp):
    def __init__(self):
        super().__init__()
        self.exp_name = os.path.split(os.path.realpath(__file__))[1].split(".")[0]
        self.lframe, self.gframe, self.lframe_val, self.gframe_val = 8, 24, 8, 24
        self.maximal_limit = 2000
