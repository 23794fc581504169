"""Exp of the YOLOX base detector -- stage 1 of the published recipe (README.md:86-89) -- with the contract of
the reference's yolox/exp/yolox_base.py:15-330: the configuration attributes (model, loader, augmentation, schedule, test), the
factory methods the base trainer calls (get_model, get_data_loader, random_resize, preprocess, get_optimizer, get_lr_scheduler,
get_trainer, eval) and their defaults.

MI355X specifics: the optimizer is tscd_amd.optim.FusedSGD (one launch for the three parameter groups + the EMA of the stepped
parameters); ``get_data_loader`` takes any dataset with the ``pull_item / load_anno / input_dim`` contract through ``self.dataset_fn``
-- with ``data_dir`` set it is yolox.data.COCODataset over the COCO-format JSON there (own index, PIL decode: no pycocotools / cv2 in
this image) and ``get_evaluator`` is COCOEvaluator on csrc_host/cocoeval.cpp; without a data set (none ships here) the default is the
seeded synthetic stand-in of yolox/data/mosaic.py.  Either way the loader is the MosaicDetection -> TrainTransform chain the reference
builds (:138-165)."""
import os
import random

import torch
import torch.distributed as dist
import torch.nn as nn

from .base_exp import BaseExp


class Exp(BaseExp):
    def __init__(self):
        super().__init__()
        self.archi_name = "YOLOX"
        # model (yolox_base.py:19-27)
        self.num_classes, self.depth, self.width, self.act = 80, 1.00, 1.00, "silu"
        # loader (:29-50): multiscale range = input +- 5 * 32
        self.data_num_workers, self.input_size, self.multiscale_range = 8, (640, 640), 5
        self.data_dir, self.train_ann, self.val_ann, self.test_ann = None, "instances_train2017.json", "instances_val2017.json", "instances_test2017.json"
        self.train_name, self.val_name = "train2017", "val2017"
        # augmentation (:52-72)
        self.mosaic_prob, self.mixup_prob, self.hsv_prob, self.flip_prob = 1.0, 1.0, 1.0, 0.5
        self.degrees, self.translate, self.mosaic_scale, self.shear = 10.0, 0.1, (0.1, 2), 2.0
        self.enable_mixup, self.mixup_scale = True, (0.5, 1.5)
        # schedule (:74-104)
        self.warmup_epochs, self.max_epoch, self.warmup_lr, self.min_lr_ratio = 5, 300, 0, 0.05
        self.basic_lr_per_img, self.scheduler, self.no_aug_epochs, self.ema = 0.01 / 64.0, "yoloxwarmcos", 15, True
        self.weight_decay, self.momentum = 5e-4, 0.9
        self.print_interval, self.eval_interval, self.save_history_ckpt = 10, 10, False
        self.exp_name = os.path.split(os.path.realpath(__file__))[1].split(".")[0]
        # test (:106-115)
        self.test_size, self.test_conf, self.nmsthre = (640, 640), 0.01, 0.65
        self.dataset_fn = None          # () -> dataset with pull_item / load_anno / input_dim; None = the synthetic stand-in
        self.synthetic_len = 256

    # ------------------------------------------------------------------ model
    def get_model(self):
        from yolox.models import YOLOPAFPN, YOLOX, YOLOXHead
        if getattr(self, "model", None) is None:
            in_channels = [256, 512, 1024]
            self.model = YOLOX(YOLOPAFPN(self.depth, self.width, in_channels=in_channels, act=self.act),
                               YOLOXHead(self.num_classes, self.width, in_channels=in_channels, act=self.act))
        for m in self.model.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eps, m.momentum = 1e-3, 0.03
        self.model.head.initialize_biases(1e-2)
        self.model.train()
        return self.model

    # ------------------------------------------------------------------ data
    def get_data_loader(self, batch_size, is_distributed, no_aug=False, cache_img=False):
        """MosaicDetection(dataset, preproc = TrainTransform(max_labels 120)) drawn by an endless rank-sharded sampler; the batch is
        the per-rank share of ``batch_size`` (yolox_base.py:138-196)."""
        from yolox.data import TrainTransform
        from yolox.data.mosaic import InfiniteSampler, MosaicDetection, SyntheticDetectionDataset, YoloBatchLoader
        if self.dataset_fn is not None:
            base = self.dataset_fn()
        elif self.data_dir is not None:
            from yolox.data import COCODataset
            base = COCODataset(data_dir=self.data_dir, json_file=self.train_ann, name=self.train_name, img_size=self.input_size,
                               preproc=None, cache=cache_img)
        else:
            base = SyntheticDetectionDataset(self.synthetic_len, self.num_classes, self.input_size, seed=self.seed or 0)
        self.dataset = MosaicDetection(base, mosaic=not no_aug, img_size=self.input_size,
                                       preproc=TrainTransform(max_labels=120, flip_prob=self.flip_prob, hsv_prob=self.hsv_prob),
                                       degrees=self.degrees, translate=self.translate, mosaic_scale=self.mosaic_scale,
                                       mixup_scale=self.mixup_scale, shear=self.shear, enable_mixup=self.enable_mixup,
                                       mosaic_prob=self.mosaic_prob, mixup_prob=self.mixup_prob)
        world = dist.get_world_size() if is_distributed else 1
        rank = dist.get_rank() if is_distributed else 0
        sampler = InfiniteSampler(len(self.dataset), seed=self.seed or 0, rank=rank, world_size=world)
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        return YoloBatchLoader(self.dataset, max(1, batch_size // world), sampler, workers=0 if (self.dataset_fn is None and self.data_dir is None) else self.data_num_workers,
                               device=device)

    def random_resize(self, data_loader, epoch, rank, is_distributed):
        """A new training size every 10 iterations: 32 * uniform{H/32 - r .. H/32 + r}, drawn on rank 0 and broadcast (:198-218)."""
        size = torch.zeros(2, dtype=torch.long, device="cuda" if torch.cuda.is_available() else "cpu")
        if rank == 0:
            if not hasattr(self, "random_size"):
                base = int(self.input_size[0] / 32)
                self.random_size = (base - self.multiscale_range, base + self.multiscale_range)
            k = random.randint(*self.random_size)
            size[0], size[1] = int(32 * k), 32 * int(k * self.input_size[1] * 1.0 / self.input_size[0])
        if is_distributed:
            dist.barrier()
            dist.broadcast(size, 0)
        return (int(size[0]), int(size[1]))

    def preprocess(self, inputs, targets, tsize):
        """Bilinear resize of the batch to ``tsize`` and the matching rescale of the (cx, cy, w, h) targets (:220-230)."""
        sy, sx = tsize[0] / self.input_size[0], tsize[1] / self.input_size[1]
        if sx != 1 or sy != 1:
            inputs = nn.functional.interpolate(inputs, size=tsize, mode="bilinear", align_corners=False)
            targets[..., 1::2] = targets[..., 1::2] * sx
            targets[..., 2::2] = targets[..., 2::2] * sy
        return inputs, targets

    # ------------------------------------------------------------------ solver
    def get_optimizer(self, batch_size):
        """Nesterov SGD over three groups -- BatchNorm weights (no decay), other weights (decay), biases (no decay) -- starting at the
        warm-up learning rate (:232-259)."""
        if "optimizer" not in self.__dict__:
            lr = self.warmup_lr if self.warmup_epochs > 0 else self.basic_lr_per_img * batch_size
            bn_w, w, b = [], [], []
            for name, mod in self.model.named_modules():
                if isinstance(getattr(mod, "bias", None), nn.Parameter):
                    b.append(mod.bias)
                if isinstance(mod, nn.BatchNorm2d) or "bn" in name:
                    bn_w.append(mod.weight)
                elif isinstance(getattr(mod, "weight", None), nn.Parameter):
                    w.append(mod.weight)
            on_gpu = bool(bn_w) and bn_w[0].is_cuda
            if on_gpu:
                from tscd_amd.optim import FusedSGD as SGD
            else:
                SGD = torch.optim.SGD
            opt = SGD(bn_w, lr=lr, momentum=self.momentum, nesterov=True)
            opt.add_param_group({"params": w, "weight_decay": self.weight_decay})
            opt.add_param_group({"params": b})
            self.optimizer = opt
        return self.optimizer

    def get_lr_scheduler(self, lr, iters_per_epoch):
        from yolox.utils import LRScheduler
        return LRScheduler(self.scheduler, lr, iters_per_epoch, self.max_epoch, warmup_epochs=self.warmup_epochs,
                           warmup_lr_start=self.warmup_lr, no_aug_epochs=self.no_aug_epochs, min_lr_ratio=self.min_lr_ratio)

    # ------------------------------------------------------------------ evaluation
    def get_eval_loader(self, batch_size, is_distributed, testdev=False, legacy=False):
        """COCODataset(val / test annotations, ValTransform) read in order, each rank its contiguous share (:261-291)."""
        from yolox.data import COCODataset, ValTransform
        valdataset = COCODataset(data_dir=self.data_dir, json_file=self.val_ann if not testdev else self.test_ann,
                                 name=self.val_name if not testdev else "test2017", img_size=self.test_size, preproc=ValTransform(legacy=legacy))
        world = dist.get_world_size() if is_distributed else 1
        rank = dist.get_rank() if is_distributed else 0
        per = max(1, batch_size // world)
        sampler = (torch.utils.data.distributed.DistributedSampler(valdataset, num_replicas=world, rank=rank, shuffle=False)
                   if is_distributed else torch.utils.data.SequentialSampler(valdataset))
        return torch.utils.data.DataLoader(valdataset, batch_size=per, sampler=sampler, num_workers=0 if self.data_num_workers == 0 else
                                           min(self.data_num_workers, 4), pin_memory=True)

    def get_evaluator(self, batch_size, is_distributed, testdev=False, legacy=False):
        """COCOEvaluator over get_eval_loader (:293-306); None without a data set (``data_dir`` unset: the trainer then skips the
        evaluation epochs)."""
        if self.data_dir is None:
            return None
        from yolox.evaluators import COCOEvaluator
        return COCOEvaluator(dataloader=self.get_eval_loader(batch_size, is_distributed, testdev, legacy), img_size=self.test_size,
                             confthre=self.test_conf, nmsthre=self.nmsthre, num_classes=self.num_classes, testdev=testdev)

    def get_trainer(self, args):
        from yolox.core import BaseTrainer
        return BaseTrainer(self, args)

    def eval(self, model, evaluator, is_distributed, half=False):
        return evaluator.evaluate(model, is_distributed, half)
