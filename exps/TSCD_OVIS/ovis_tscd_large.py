"""TSCD-L on OVIS: the settings of the reference's exps/TSCD_OVIS/ovis_tscd_large.py:12-51 on top of the base Exp."""
import os

from exps.TSCD_OVIS.tscd_base import Exp as BaseTSCDExp


class Exp(BaseTSCDExp):
    def __init__(self):
        super().__init__()
        self.depth, self.width = 1.0, 1.0
        self.exp_name = os.path.split(os.path.realpath(__file__))[1].split(".")[0]
        self.backbone_name = "MCSP"
        self.data_dir = "path to your datasets"
        self.train_ann, self.val_ann = "annotations/ovis_train_vid.json", "annotations/ovis_valid_vid.json"
        self.num_classes = 25
        self.warmup_epochs, self.no_aug_epochs, self.pre_no_aug, self.eval_interval = 0, 2, 2, 1
        self.gmode, self.lmode = True, True
        self.lframe, self.lframe_val, self.gframe, self.gframe_val = 4, 8, 12, 24
        self.use_loc_emd, self.iou_base, self.reconf, self.loc_fuse_type = True, False, True, "identity"
        self.output_dir = "./OVIS_TSCD_outputs/TSCD_Large"
        self.stem_lr_ratio = 0.1
        self.ota_mode, self.use_pre_nms, self.cat_ota_fg, self.agg_type = True, False, False, "mca"
        self.minimal_limit, self.maximal_limit, self.conf_sim_thresh = 50, 500, 0.99
        self.decouple_reg, self.seed, self.matcher_refine = True, 2024, True

    def _more_args(self):
        d = super()._more_args()
        d.update({"use_pre_nms": self.use_pre_nms, "cat_ota_fg": self.cat_ota_fg, "agg_type": self.agg_type,
                  "maximal_limit": self.maximal_limit, "decouple_reg": self.decouple_reg})
        return d
