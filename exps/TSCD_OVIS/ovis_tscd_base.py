"""TSCD-S on OVIS (reference: exps/TSCD_OVIS/ovis_tscd_base.py): the large exp at depth 0.33 / width 0.5."""
import os

from exps.TSCD_OVIS.ovis_tscd_large import Exp as LargeExp


class Exp(LargeExp):
    def __init__(self):
        super().__init__()
        self.depth, self.width = 0.33, 0.50
        self.exp_name = os.path.split(os.path.realpath(__file__))[1].split(".")[0]
        self.output_dir = "./OVIS_TSCD_outputs/TSCD_Base"
