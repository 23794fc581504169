"""Training loop of the YOLOX base detector (contract of the reference's yolox/core/trainer.py:36-390): epochs of
``max_iter`` iterations, AMP with a GradScaler, EMA, the yoloxwarmcos schedule, a new multiscale size every 10 iterations, mosaic
switched off and the L1 loss switched on for the last ``no_aug_epochs`` epochs, ``latest`` / ``last_mosaic_epoch`` / ``last_epoch`` /
``best`` checkpoints.  The step itself is the HIP path: YOLOX.forward -> fused losses -> backward kernels -> tscd_amd.optim.FusedSGD."""
import collections
import os
import time

import torch

from tscd_amd import ddp as tscd_ddp
from yolox.utils import ModelEMA, all_reduce_norm, checkpoint, get_local_rank, get_rank, get_world_size, is_parallel


class BaseTrainer:
    def __init__(self, exp, args):
        self.exp, self.args = exp, args
        self.max_epoch = exp.max_epoch
        self.amp_training = getattr(args, "fp16", True)
        from tscd_amd.optim import GradScaler
        self.scaler = GradScaler("cuda", enabled=self.amp_training)
        self.is_distributed = get_world_size() > 1
        self.rank, self.local_rank = get_rank(), get_local_rank()
        self.device = "cuda:{}".format(self.local_rank)
        self.use_model_ema = exp.ema
        self.data_type = torch.float16 if self.amp_training else torch.float32
        self.input_size = exp.input_size
        self.best_ap = 0
        self.file_name = os.path.join(exp.output_dir, getattr(args, "experiment_name", None) or exp.exp_name)
        self.history = collections.deque(maxlen=2048)
        if self.rank == 0:
            os.makedirs(self.file_name, exist_ok=True)

    # ------------------------------------------------------------------ loop (trainer.py:70-131)
    def train(self):
        self.before_train()
        try:
            for self.epoch in range(self.start_epoch, self.max_epoch):
                self.before_epoch()
                for self.iter in range(self.max_iter):
                    self.train_one_iter()
                    self.after_iter()
                self.after_epoch()
        finally:
            checkpoint.wait()

    def train_one_iter(self):
        t0 = time.time()
        inps, targets = self.prefetcher.next()
        inps, targets = inps.to(self.data_type), targets.to(self.data_type)
        targets.requires_grad = False
        inps, targets = self.exp.preprocess(inps, targets, self.input_size)
        outputs = self.model(inps, targets)            # the kernels compute in fp16 themselves: no autocast region needed
        loss = outputs["total_loss"]
        self.optimizer.zero_grad()
        self.scaler.scale(loss).backward()
        self.scaler.step(self.optimizer)
        self.scaler.update()
        if self.use_model_ema:
            self.ema_model.update(self.model)
        lr = self.lr_scheduler.update_lr(self.progress_in_iter + 1)
        for g in self.optimizer.param_groups:
            g["lr"] = lr
        self.history.append({"iter_time": time.time() - t0, "lr": lr,
                             **{k: (v.detach() if torch.is_tensor(v) else v) for k, v in outputs.items()}})
        return outputs

    # ------------------------------------------------------------------ setup (trainer.py:133-198)
    def before_train(self):
        torch.cuda.set_device(self.local_rank)
        model = self.exp.get_model()
        model.to(self.device)
        self.optimizer = self.exp.get_optimizer(self.args.batch_size)
        model = self.resume_train(model)
        self.no_aug = self.start_epoch >= self.max_epoch - self.exp.no_aug_epochs
        self.train_loader = self.exp.get_data_loader(batch_size=self.args.batch_size, is_distributed=self.is_distributed, no_aug=self.no_aug,
                                                     cache_img=getattr(self.args, "cache", False))
        self.prefetcher = self.train_loader            # contract: next() -> (imgs, targets) on the device
        self.max_iter = len(self.train_loader)
        self.lr_scheduler = self.exp.get_lr_scheduler(self.exp.basic_lr_per_img * self.args.batch_size, self.max_iter)
        if self.is_distributed:
            model = tscd_ddp.wrap(model, device_ids=[self.local_rank])      # DDP(..., broadcast_buffers=False), trainer.py:165-166
            model.attach(self.optimizer)
        if self.use_model_ema:
            self.ema_model = ModelEMA(model, 0.9999)
            self.ema_model.updates = self.max_iter * self.start_epoch
            if hasattr(self.optimizer, "attach_ema"):
                self.ema_model.attach_to(self.optimizer, model)
        self.model = model
        self.model.train()
        self.evaluator = self.exp.get_evaluator(batch_size=self.args.batch_size, is_distributed=self.is_distributed)

    def _inner(self):
        return self.model.module if is_parallel(self.model) else self.model

    def before_epoch(self):
        """trainer.py:200-213: from epoch ``max_epoch - no_aug_epochs`` on, no mosaic / mixup and the extra L1 box loss."""
        if self.epoch + 1 == self.max_epoch - self.exp.no_aug_epochs or self.no_aug:
            self.train_loader.close_mosaic()
            self._inner().head.use_l1 = True
            self.exp.eval_interval = 1
            if not self.no_aug:
                self.save_ckpt("last_mosaic_epoch")

    def after_epoch(self):
        self.save_ckpt("latest")
        if self.evaluator is not None and (self.epoch + 1) % self.exp.eval_interval == 0:
            all_reduce_norm(self._inner())
            self.evaluate_and_save_model()

    def after_iter(self):
        if (self.progress_in_iter + 1) % 10 == 0 and getattr(self.exp, "multiscale_range", 0) > 0:
            self.input_size = self.exp.random_resize(self.train_loader, self.epoch, self.rank, self.is_distributed)

    @property
    def progress_in_iter(self):
        return self.epoch * self.max_iter + self.iter

    def resume_train(self, model):
        self.start_epoch = 0
        ckpt_file = getattr(self.args, "ckpt", None)
        if getattr(self.args, "resume", False):
            ckpt_file = ckpt_file or os.path.join(self.file_name, "latest_ckpt.pth")
            checkpoint.wait()
            ckpt = torch.load(ckpt_file, map_location=self.device, weights_only=True)
            model.load_state_dict(ckpt["model"])
            self.optimizer.load_state_dict(ckpt["optimizer"])
            self.best_ap = ckpt.get("best_ap", 0)
            se = getattr(self.args, "start_epoch", None)
            self.start_epoch = se - 1 if se is not None else ckpt.get("start_epoch", 0)
        elif ckpt_file:
            ckpt = torch.load(ckpt_file, map_location=self.device, weights_only=True)["model"]
            checkpoint.load_ckpt(model, ckpt)
        return model

    def evaluate_and_save_model(self):
        model = self.ema_model.ema if self.use_model_ema else self._inner()
        was = model.training
        model.eval()
        try:
            ap50_95, ap50, summary = self.exp.eval(model, self.evaluator, self.is_distributed)
        finally:
            model.train(was)
        self.save_ckpt("last_epoch", ap50_95 > self.best_ap)
        self.best_ap = max(self.best_ap, ap50_95)
        return ap50_95, ap50, summary

    def save_ckpt(self, ckpt_name, update_best_ckpt=False):
        if self.rank != 0:
            return
        model = self.ema_model.ema if self.use_model_ema else self._inner()
        state = {"start_epoch": self.epoch + 1, "model": model.state_dict(), "optimizer": self.optimizer.state_dict(), "best_ap": self.best_ap}
        checkpoint.save_checkpoint(state, update_best_ckpt, self.file_name, ckpt_name, non_blocking=getattr(self.args, "async_ckpt", True))

    def history_floats(self, last=None):
        rows = list(self.history) if last is None else list(self.history)[-last:]
        return [{k: (float(v) if torch.is_tensor(v) else v) for k, v in r.items()} for r in rows]
