"""TSCD training / evaluation driver (contract of the reference's yolox/core/tscd_trainer.py:90-468).

Same per-iteration order as the reference's ``train_one_iter`` (:153-192): clip from the loader, fp16 cast (T1),
multiscale ``exp.preprocess``, forward, ``GradScaler``-scaled backward, SGD step, EMA (decay .9998), LR update.  The
model computes in fp16 inside the HIP kernels, so no autocast context is needed.  DDP: gradients all-reduced over
RCCL, the parameters that never get a gradient are frozen so the bucket set is static, buffers are not
broadcast (BN statistics stay per rank until ``all_reduce_norm`` before evaluation, as in the reference)."""
import collections
import os
import time

import torch
from torch.nn.parallel import DistributedDataParallel as DDP

from tscd_amd import ddp as tscd_ddp

from yolox.utils import checkpoint
from yolox.utils import ModelEMA, all_reduce_norm, get_local_rank, get_rank, get_world_size, is_parallel, synchronize

DEAD_PARAMS = ("transformer_self_attention_layers", "transformer_ffn_layers", "edge_feature_embedding", "attn.position_embedding",
               "agg.linear_obj", "agg.mca.linear_reg", "agg_iou.linear.", "agg_iou.mca.linear.")


class Trainer:
    HISTORY_LEN = 2048

    def __init__(self, exp, args, val_loader=None, val=False):
        self.exp, self.args, self.val_loader, self.val = exp, args, val_loader, val
        self.max_epoch = exp.max_epoch
        self.amp_training = getattr(args, "fp16", True)
        from tscd_amd.optim import GradScaler          # torch.amp.GradScaler with FusedSGD's one-launch inf check (tscd_trainer.py:113)
        self.scaler = GradScaler("cuda", enabled=self.amp_training)
        self.is_distributed = get_world_size() > 1
        self.rank, self.local_rank = get_rank(), get_local_rank()
        self.device = "cuda:{}".format(self.local_rank)
        self.use_model_ema = exp.ema
        self.input_size = exp.input_size
        self.best_ap = 0
        self.file_name = os.path.join(exp.output_dir, getattr(args, "experiment_name", None) or exp.exp_name)
        self.lframe, self.gframe = exp.lframe, exp.gframe
        # the last HISTORY_LEN iterations (print / eval intervals read a window of it): bounded, a run of 10^6 iterations keeps
        # 10^6 dicts of device tensors alive otherwise
        self.history = collections.deque(maxlen=self.HISTORY_LEN)
        if self.rank == 0:
            os.makedirs(self.file_name, exist_ok=True)

    # ------------------------------------------------------------------ setup
    def before_train(self):
        torch.cuda.set_device(self.local_rank)
        model = self.exp.get_model()
        for k, p in model.named_parameters():
            if any(d in k for d in DEAD_PARAMS):
                p.requires_grad = False
        model.to(self.device)
        self.optimizer = self.exp.get_optimizer(self.args.batch_size)
        model = self.resume_train(model)
        self.train_loader = self.exp.get_data_loader(batch_size=self.args.batch_size, is_distributed=self.is_distributed)
        self.prefetcher = self.train_loader          # tensor contract: next() -> (imgs, targets, time_embedding)
        self.max_iter = len(self.train_loader)
        self.lr_scheduler = self.exp.get_lr_scheduler(self.exp.basic_lr_per_img * self.args.batch_size, self.max_iter)
        if self.is_distributed:
            # DDP(model, device_ids=[local_rank], broadcast_buffers=False) of the reference (tscd_trainer.py:230-231) as a flat-arena
            # gradient all-reduce (tscd_amd/ddp.py); the fused optimizer reads the reduced arena
            model = tscd_ddp.wrap(model, device_ids=[self.local_rank])
            model.attach(self.optimizer)
        if self.use_model_ema:
            self.ema_model = ModelEMA(model, 0.9998)
            self.ema_model.updates = self.max_iter * self.start_epoch
            if hasattr(self.optimizer, "attach_ema"):
                self.ema_model.attach_to(self.optimizer, model)      # EMA of the trained parameters inside the optimizer's launch
        self.model = model
        self.model.train()

    def resume_train(self, model):
        self.start_epoch = 0
        ckpt_file = getattr(self.args, "ckpt", None)
        if getattr(self.args, "resume", False):
            ckpt_file = ckpt_file or os.path.join(self.file_name, "latest_ckpt.pth")
            checkpoint.wait()
            ckpt = torch.load(ckpt_file, map_location=self.device, weights_only=True)
            model.load_state_dict(ckpt["model"])
            self.optimizer.load_state_dict(ckpt["optimizer"])
            self.start_epoch = ckpt.get("start_epoch", 0)
        elif ckpt_file:
            ckpt = torch.load(ckpt_file, map_location=self.device, weights_only=True)["model"]
            checkpoint.load_ckpt(model, ckpt)      # shape-tolerant fine-tune load (yolox/utils/checkpoint.py:11-33)
        return model

    # ------------------------------------------------------------------ loop
    def train(self):
        self.before_train()
        try:
            for self.epoch in range(self.start_epoch, self.max_epoch):
                for self.iter in range(self.max_iter):
                    self.train_one_iter()
                    self.after_iter()
                self.after_epoch()
        finally:
            checkpoint.wait()             # the last checkpoint's writer thread

    def after_iter(self):
        """Reference tscd_trainer.py:298-366: a ``latest`` checkpoint every 2000 iterations of an epoch (iteration 0 included),
        multiscale resize every 10."""
        if self.iter % 2000 == 0:
            self.save_ckpt("latest")
        if (self.progress_in_iter + 1) % 10 == 0 and getattr(self.exp, "multiscale_range", 0) > 0:
            self.input_size = self.exp.random_resize(self.train_loader, self.epoch, self.rank, self.is_distributed)

    def after_epoch(self):
        """Reference :287-293, :394-425: ``latest`` after every epoch; every ``eval_interval`` epochs the (EMA) model is evaluated
        and saved as ``last_epoch``, copied to ``best_ckpt.pth`` when its AP50:95 is the best so far."""
        self.save_ckpt("latest")
        if self.val_loader is not None and (self.epoch + 1) % max(1, getattr(self.exp, "eval_interval", 1)) == 0:
            self.evaluate_and_save_model()

    def evaluate_and_save_model(self):
        summary = self.evaluate()
        self.model.train()
        ap50_95 = float(summary[0])
        self.save_ckpt("last_epoch", ap50_95 > self.best_ap)
        self.best_ap = max(self.best_ap, ap50_95)
        return summary

    @property
    def progress_in_iter(self):
        return self.epoch * self.max_iter + self.iter

    def _next_batch(self):
        inps, targets, time_embedding = self.prefetcher.next()
        inps, targets, time_embedding = inps.half(), targets.half(), time_embedding.half()      # tscd_trainer.py:159-161
        targets.requires_grad = False
        inps, targets = self.exp.preprocess(inps, targets, self.input_size)
        return inps, targets, time_embedding

    def _join_prefetch(self):
        m = self.model.module if is_parallel(self.model) else self.model
        if hasattr(m, "join_prefetch"):
            m.join_prefetch()

    def train_one_iter(self):
        t0 = time.time()
        # one clip of look-ahead: the model runs the NEXT clip's frozen backbone pass inside this step's assignment window
        # (yolox/models/tscd.py).  Never across the end of an epoch or a multiscale resize (the next clip's size is not decided yet).
        batch, self._ahead = getattr(self, "_ahead", None), None
        inps, targets, time_embedding = batch if batch is not None else self._next_batch()
        resize_next = (self.progress_in_iter + 1) % 10 == 0 and getattr(self.exp, "multiscale_range", 0) > 0
        if self.iter + 1 < self.max_iter and not resize_next:
            self._ahead = self._next_batch()
        next_x = self._ahead[0] if self._ahead is not None else None
        # the head's eager class branch backpropagates inside forward: hand it this step's loss scale (device tensor, no sync)
        m = self.model.module if is_parallel(self.model) else self.model
        if hasattr(getattr(m, "head", None), "loss_scale"):
            if not hasattr(self, "_one"):
                self._one = torch.ones((), dtype=torch.float32, device=inps.device)
            m.head.loss_scale = self.scaler.scale(self._one)
        outputs = self.model(inps, targets, time_embedding, lframe=self.exp.lframe, gframe=self.exp.gframe, resume=False,
                             next_x=next_x)
        loss = outputs["total_loss"]
        self.optimizer.zero_grad()
        self.scaler.scale(loss).backward()
        if hasattr(m, "check_assignments"):
            m.check_assignments()       # a failed / timed-out assignment chain raises here, BEFORE its step is applied
        self.scaler.step(self.optimizer)
        self.scaler.update()
        if self.use_model_ema:         # (reads the frozen front's running statistics as they were BEFORE a pending prefetched pass)
            self.ema_model.update(self.model)
        lr = self.lr_scheduler.update_lr(self.progress_in_iter + 1)
        for g in self.optimizer.param_groups:
            g["lr"] = lr
        # loss terms stay device tensors (a float() here would drain the GPU every iteration and undo the look-ahead above);
        # history_floats() converts on demand (print / eval intervals)
        self.history.append({"iter_time": time.time() - t0, "lr": lr,
                             **{k: (v.detach() if torch.is_tensor(v) else v) for k, v in outputs.items()}})
        return outputs

    def history_floats(self, last=None):
        """The recorded iterations (or the last ``last`` of them) with every loss term as a Python float (synchronises)."""
        rows = list(self.history) if last is None else list(self.history)[-last:]
        return [{k: (float(v) if torch.is_tensor(v) else v) for k, v in r.items()} for r in rows]

    # ------------------------------------------------------------------ eval / checkpoints
    def evaluate(self):
        model = self.ema_model.ema if self.use_model_ema else self.model
        model = model.module if is_parallel(model) else model
        all_reduce_norm(model)
        evaluator = self.exp.get_evaluator(self.val_loader)
        res = self.exp.eval(model, evaluator, self.is_distributed)
        synchronize()
        return res

    def save_ckpt(self, ckpt_name, update_best_ckpt=False):
        if self.rank != 0:
            return
        model = self.ema_model.ema if self.use_model_ema else self.model
        model = model.module if is_parallel(model) else model
        # a prefetched pass of the next clip may be pending: the live model is saved with the statistics it had before that pass
        # (step_state_dict), so a resumed run applies that clip once; the EMA copy never saw the pass (ModelEMA.update)
        live = self.model.module if is_parallel(self.model) else self.model
        sd = live.step_state_dict() if (model is live and hasattr(live, "step_state_dict")) else model.state_dict()
        state = {"start_epoch": self.epoch + 1, "model": sd, "optimizer": self.optimizer.state_dict()}
        checkpoint.save_checkpoint(state, update_best_ckpt, self.file_name, ckpt_name, non_blocking=getattr(self.args, "async_ckpt", True))
