"""Base experiment of TSCD on the MI355X kernels: the configuration surface of the reference's
exps/TSCD_OVIS/tscd_base.py:16-499 (same attribute names and defaults, same factory methods), building this
repository's yolox.models classes.  The loaders read OVIS ("vid" COCO json) when ``data_dir`` holds it and fall back to
yolox.data.SyntheticClipLoader otherwise; the evaluator is yolox.evaluators.ovis_evaluator_v2.OVISEvaluator."""
import os
import random

import torch
import torch.distributed as dist
import torch.nn as nn

from yolox.data.data_augment import Vid_Val_Transform
from yolox.data.datasets import vid
from yolox.exp import BaseExp


class Exp(BaseExp):
    def __init__(self):
        super().__init__()
        self.archi_name = "TSCD"
        self.backbone_name = "MCSP"
        # model
        self.num_classes = 25
        self.depth, self.width, self.act = 1.00, 1.00, "silu"
        # proposal / attention
        self.drop_rate, self.head, self.defualt_p = 0, 4, 30
        self.sim_thresh, self.pre_nms, self.ave, self.defualt_pre = 0.75, 0.75, True, 750
        self.use_score, self.perspective, self.fix_bn, self.use_aug, self.use_mask, self.fix_all = True, 0.0, False, False, False, False
        self.gmode, self.lmode, self.both_mode = True, False, False
        self.lframe, self.lframe_val, self.localBlocks = 0, 0, 1
        self.gframe, self.gframe_val, self.tnum = 16, 32, -1
        self.local_stride, self.iou_window, self.globalBlocks = 1, 0, 1
        self.use_ffn, self.use_time_emd, self.use_loc_emd, self.loc_fuse_type, self.use_qkv = True, False, True, "add", True
        self.local_mask, self.local_mask_branch, self.pure_pos_emb, self.loc_conf, self.iou_base = False, "", False, False, False
        self.reconf, self.ota_mode, self.ota_cls, self.traj_linking = False, False, False, False
        self.minimal_limit, self.vid_cls, self.vid_reg, self.conf_sim_thresh = 0, True, False, 0.99
        # data
        self.data_num_workers = 12
        self.input_size = (576, 576)
        self.multiscale_range = 3
        self.data_dir = "path to your datasets"
        self.mosaic_prob, self.mixup_prob, self.hsv_prob, self.flip_prob = 1.0, 1.0, 1.0, 0.5
        self.degrees, self.translate, self.mosaic_scale, self.enable_mixup, self.mixup_scale, self.shear = 10.0, 0.1, (0.1, 2), True, (0.5, 1.5), 2.0
        # training
        self.warmup_epochs, self.max_epoch, self.warmup_lr, self.min_lr_ratio = 1, 7, 0, 0.1
        self.basic_lr_per_img = 0.002 / 64.0
        self.scheduler, self.no_aug_epochs, self.ema = "yoloxwarmcos", 2, True
        self.weight_decay, self.momentum = 5e-4, 0.9
        self.print_interval, self.eval_interval, self.save_history_ckpt = 10, 1, True
        self.exp_name = os.path.split(os.path.realpath(__file__))[1].split(".")[0]
        # testing
        self.test_size, self.test_conf, self.nmsthre = (576, 576), 0.001, 0.5

    # ------------------------------------------------------------------ model
    def _more_args(self):
        return {"use_ffn": self.use_ffn, "use_time_emd": self.use_time_emd, "use_loc_emd": self.use_loc_emd,
                "loc_fuse_type": self.loc_fuse_type, "use_qkv": self.use_qkv, "local_mask": self.local_mask,
                "local_mask_branch": self.local_mask_branch, "pure_pos_emb": self.pure_pos_emb, "loc_conf": self.loc_conf,
                "iou_base": self.iou_base, "reconf": self.reconf, "ota_mode": self.ota_mode, "ota_cls": self.ota_cls,
                "traj_linking": self.traj_linking, "iou_window": self.iou_window, "globalBlocks": self.globalBlocks,
                "minimal_limit": self.minimal_limit, "conf_sim_thresh": self.conf_sim_thresh}

    def get_model(self):
        if self.backbone_name != "MCSP":
            raise NotImplementedError("only the CSPDarknet ('MCSP') backbone is on the TSCD hot path (SURVEY.md section 2)")
        from yolox.models import TSCD, TSCDHead, YOLOPAFPN
        in_channels = [256, 512, 1024]
        backbone = YOLOPAFPN(self.depth, self.width, in_channels=in_channels)
        for p in backbone.parameters():
            p.requires_grad = False
        head = TSCDHead(self.num_classes, self.width, in_channels=in_channels, heads=self.head, drop=self.drop_rate,
                        use_score=self.use_score, defualt_p=self.defualt_p, sim_thresh=self.sim_thresh, pre_nms=self.pre_nms,
                        ave=self.ave, defulat_pre=self.defualt_pre, test_conf=self.test_conf, use_mask=self.use_mask,
                        gmode=self.gmode, lmode=self.lmode, both_mode=self.both_mode, localBlocks=self.localBlocks,
                        **self._more_args())
        for mod in (head.stems, head.reg_convs, head.cls_convs, head.reg_preds):
            for p in mod.parameters():
                p.requires_grad = False
        if self.fix_all:
            for mod in (head.obj_preds, head.cls_preds):
                for p in mod.parameters():
                    p.requires_grad = False
        self.model = TSCD(backbone, head)
        for m in self.model.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eps, m.momentum = 1e-3, 0.03
                if self.fix_bn:
                    m.eval()
        self.model.head.initialize_biases(1e-2)
        return self.model

    # ------------------------------------------------------------------ optimisation
    def _param_groups(self, split_head):
        pg0, pg1, pg2, pg3 = [], [], [], []
        for k, v in self.model.named_modules():
            if hasattr(v, "bias") and isinstance(v.bias, nn.Parameter):
                pg2.append(v.bias)
            if isinstance(v, nn.BatchNorm2d) or "bn" in k:
                pg0.append(v.weight)
            elif hasattr(v, "weight") and isinstance(v.weight, nn.Parameter):
                if split_head and ("head.stem" in k or "head.reg_convs" in k or "head.cls_convs" in k):
                    pg3.append(v.weight)
                else:
                    pg1.append(v.weight)
        return pg0, pg1, pg2, pg3

    def get_optimizer(self, batch_size):
        if "optimizer" not in self.__dict__:
            lr = self.warmup_lr if self.warmup_epochs > 0 else self.basic_lr_per_img * batch_size
            pg0, pg1, pg2, pg3 = self._param_groups(split_head=hasattr(self, "stem_lr_ratio"))
            # fused multi-tensor SGD on the GPU: same update rule; with amp the GradScaler's inf check then stays on the
            # device instead of draining the GPU once per step (tscd_trainer.py:171-176 semantics unchanged)
            fused = bool(pg0) and pg0[0].is_cuda
            if fused:
                # one launch for all groups, with the EMA of the trained parameters in the same pass (tscd_amd/optim.py, row N3)
                from tscd_amd.optim import FusedSGD
                opt = FusedSGD(pg0, lr=lr, momentum=self.momentum, nesterov=True)
            else:
                opt = torch.optim.SGD(pg0, lr=lr, momentum=self.momentum, nesterov=True)
            opt.add_param_group({"params": pg1, "weight_decay": self.weight_decay})
            opt.add_param_group({"params": pg2})
            if pg3:
                opt.add_param_group({"params": pg3, "lr": lr * self.stem_lr_ratio, "weight_decay": self.weight_decay})
            self.optimizer = opt
        return self.optimizer

    def get_lr_scheduler(self, lr, iters_per_epoch):
        from yolox.utils import LRScheduler
        return LRScheduler(self.scheduler, lr, iters_per_epoch, self.max_epoch, warmup_epochs=self.warmup_epochs,
                           warmup_lr_start=self.warmup_lr, no_aug_epochs=self.no_aug_epochs, min_lr_ratio=self.min_lr_ratio)

    def random_resize(self, data_loader, epoch, rank, is_distributed):
        """Multiscale size in multiples of 64 (tscd_base.py:380-399).  All ranks draw from one shared seeded RNG, so the
        reference's barrier + 16-byte broadcast every 10 iterations is not needed."""
        if not hasattr(self, "random_size"):
            base = int(self.input_size[0] / 64)
            self.random_size = (base - self.multiscale_range, base + self.multiscale_range)
        if not hasattr(self, "_size_rng"):
            self._size_rng = random.Random(self.seed if self.seed is not None else 0)
        size = self._size_rng.randint(*self.random_size)
        factor = self.input_size[1] * 1.0 / self.input_size[0]
        return (int(64 * size), 64 * int(size * factor))

    def preprocess(self, inputs, targets, tsize):
        sy, sx = tsize[0] / self.input_size[0], tsize[1] / self.input_size[1]
        if sx != 1 or sy != 1:
            inputs = nn.functional.interpolate(inputs, size=tsize, mode="bilinear", align_corners=False)
            targets[..., 1::2] = targets[..., 1::2] * sx
            targets[..., 2::2] = targets[..., 2::2] * sy
        return inputs, targets

    # ------------------------------------------------------------------ data / eval
    def _synthetic(self, frames, size, is_distributed=False):
        from yolox.data import SyntheticClipLoader
        rank = dist.get_rank() if is_distributed and dist.is_initialized() else 0
        world = dist.get_world_size() if is_distributed and dist.is_initialized() else 1
        return SyntheticClipLoader(frames, size, rank=rank, world_size=world, seed=self.seed or 0)

    def _ann(self, name):
        return os.path.join(self.data_dir, getattr(self, name, ""))

    def get_data_loader(self, batch_size, is_distributed, no_aug=False, cache_img=False):
        """OVIS clips (tscd_base.py:199-219 / ovis_tscd_large.py:199-217) behind the device prefetcher when ``data_dir`` holds the
        dataset; a synthetic clip source with the same tensor contract otherwise (smoke runs, benchmarks)."""
        assert batch_size == self.lframe + self.gframe, "one clip per iteration: -b must equal lframe + gframe"
        if not os.path.isfile(self._ann("train_ann")):
            return self._synthetic(batch_size, self.input_size[0], is_distributed)
        from yolox.data import DataPrefetcher, TrainTransform
        from yolox.data.data_augment import TrainTransformGPU
        # row N1: decode on the loader workers, resize / HSV / flip / pad of the whole clip in one kernel on the device (a per-frame host
        # resize + HSV sustains 0.66 clips/s per worker; `gpu_preproc = False` keeps the host transform)
        tf = TrainTransformGPU if (getattr(self, "gpu_preproc", True) and torch.cuda.is_available()) else TrainTransform
        dataset = vid.OVIS(img_size=self.input_size, preproc=tf(max_labels=50, flip_prob=self.flip_prob, hsv_prob=self.hsv_prob),
                           mode="random", lframe=self.lframe, gframe=self.gframe, data_dir=self.data_dir, name="train",
                           COCO_anno=self._ann("train_ann"), training=True)
        loader = vid.ovis_get_trans_loader(batch_size=batch_size, data_num_workers=self.data_num_workers, dataset=dataset,
                                           seed=self.seed or 0)
        return _Prefetched(loader)

    def get_eval_loader(self, batch_size, tnum=None, data_num_workers=8, formal=False):
        assert batch_size == self.lframe_val + self.gframe_val
        if not os.path.isfile(self._ann("val_ann")):
            return self._synthetic(batch_size, self.test_size[0])
        dataset = vid.OVIS(data_dir=self.data_dir, img_size=self.test_size, mode="random", COCO_anno=self._ann("val_ann"), name="train",
                           lframe=self.lframe_val, gframe=self.gframe_val, preproc=Vid_Val_Transform(), val=True)
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        return vid.ovis_vid_val_loader(batch_size=batch_size, data_num_workers=data_num_workers, dataset=dataset, rank=rank, world_size=world)

    def get_evaluator(self, val_loader):
        from yolox.evaluators.ovis_evaluator_v2 import OVISEvaluator
        return OVISEvaluator(dataloader=val_loader, img_size=self.test_size, confthre=self.test_conf, nmsthre=self.nmsthre,
                             num_classes=self.num_classes, lframe=self.lframe_val, gframe=self.gframe_val, first_only=False)

    def get_trainer(self, args):
        from yolox.core import Trainer
        return Trainer(self, args)

    def eval(self, model, evaluator, is_distributed, half=False):
        return evaluator.evaluate(model, is_distributed, half)


class _Prefetched:
    """The trainer's clip source contract -- ``next()`` -> (imgs, targets, time_embedding) on the device, ``len()`` = iterations per
    epoch -- over a torch DataLoader: a DataPrefetcher (pinned H2D on a side stream) that is re-armed at the end of every epoch."""

    def __init__(self, loader):
        from yolox.data import DataPrefetcher
        self.loader, self._mk = loader, DataPrefetcher
        self.pf = DataPrefetcher(loader)

    def __len__(self):
        return len(self.loader)

    def next(self):
        out = self.pf.next()
        if out[0] is None:                 # epoch exhausted: the sampler reshuffles (rank-aware, seeded by the epoch) on re-iteration
            self.pf = self._mk(self.loader)
            out = self.pf.next()
        return out
