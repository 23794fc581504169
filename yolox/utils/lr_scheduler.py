"""Learning-rate schedules with the reference's ``LRScheduler(name, lr, iters_per_epoch, total_epochs, **kw).update_lr(it)``
contract (yolox/utils/lr_scheduler.py:9-205).  TSCD uses "yoloxwarmcos" (exps/TSCD_OVIS/tscd_base.py:163)."""
import math


class LRScheduler:
    def __init__(self, name, lr, iters_per_epoch, total_epochs, **kwargs):
        self.name, self.lr = name, lr
        self.iters_per_epoch, self.total_epochs = iters_per_epoch, total_epochs
        self.total_iters = iters_per_epoch * total_epochs
        self.__dict__.update(kwargs)
        if name not in ("cos", "warmcos", "yoloxwarmcos", "multistep"):
            raise ValueError("Scheduler version {} not supported.".format(name))

    def _cos(self, lo, hi, frac):
        return lo + 0.5 * (hi - lo) * (1.0 + math.cos(math.pi * frac))

    def update_lr(self, iters):
        lr, total = self.lr, self.total_iters
        if self.name == "cos":
            return self._cos(0.0, lr, iters / total)
        if self.name == "multistep":
            gamma = getattr(self, "gamma", 0.1)
            for m in self.milestones:
                if iters >= int(total * m / self.total_epochs):
                    lr *= gamma
            return lr
        warm = self.iters_per_epoch * getattr(self, "warmup_epochs", 0)
        if self.name == "warmcos":
            start = getattr(self, "warmup_lr_start", 1e-6)
            if iters <= warm:
                return (lr - start) * iters / float(warm) + start
            return self._cos(0.0, lr, (iters - warm) / (total - warm))
        # yoloxwarmcos: quadratic warm-up, cosine to min_lr, flat min_lr for the last no_aug epochs
        start = getattr(self, "warmup_lr_start", 0)
        min_lr = lr * getattr(self, "min_lr_ratio", 0.2)
        no_aug = self.iters_per_epoch * getattr(self, "no_aug_epochs", 0)
        if iters <= warm:
            return (lr - start) * pow(iters / float(warm), 2) + start
        if iters >= total - no_aug:
            return min_lr
        return self._cos(min_lr, lr, (iters - warm) / (total - warm - no_aug))
