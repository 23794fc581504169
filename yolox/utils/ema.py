"""ModelEMA with the reference's contract (yolox/utils/ema.py:22-60): EMA over every floating tensor of the
state_dict (parameters AND buffers, so the frozen backbone's BN running stats are tracked), decay ramp
d = decay * (1 - exp(-n/2000)).  The update is two multi-tensor launches instead of ~1200 small ones."""
import math
from copy import deepcopy

import torch
import torch.nn as nn


def is_parallel(model):
    """torch's wrappers, or tscd_amd.ddp.ArenaDDP (flat-arena gradient all-reduce): anything that holds the model as ``.module``."""
    if isinstance(model, (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel)):
        return True
    return type(model).__name__ == "ArenaDDP" and hasattr(model, "module")


class ModelEMA:
    def __init__(self, model, decay=0.9999, updates=0):
        self.ema = deepcopy(model.module if is_parallel(model) else model).eval()
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))
        for p in self.ema.parameters():
            p.requires_grad_(False)

    def attach_to(self, optimizer, model):
        """Hands the EMA copies of the parameters ``optimizer`` (tscd_amd.optim.FusedSGD) steps to its fused pass; ``update`` then
        covers the rest of the state that can change -- buffers, and parameters some other mechanism trains.  Frozen parameters
        (requires_grad False, not in the optimizer) equal their copies at construction and stay equal: they are skipped, which is
        what the full-state formula d * w + (1 - d) * w gives up to one rounding."""
        inner = model.module if is_parallel(model) else model
        ema_p = dict(self.ema.named_parameters())
        stepped = {id(p) for g in optimizer.param_groups for p in g["params"]}
        pairs = {p: ema_p[k] for k, p in inner.named_parameters() if id(p) in stepped}
        optimizer.attach_ema(pairs, self._next_decay)
        self._fused_keys = {k for k, p in inner.named_parameters() if id(p) in stepped}
        self._skip_keys = self._fused_keys | {k for k, p in inner.named_parameters() if not p.requires_grad}
        self._objs = None
        self._fused_pending = False

    def _next_decay(self):
        # called by the optimizer's step: the EMA update counter advances there, update() must not advance it again
        self.updates += 1
        self._fused_pending = True
        return self.decay(self.updates)

    @torch.no_grad()
    def update(self, model):
        if getattr(self, "_fused_pending", False):
            self._fused_pending = False           # the optimizer's pass already counted this update
        else:
            self.updates += 1
        d = self.decay(self.updates)
        inner = model.module if is_parallel(model) else model
        # The (ema tensor, model tensor) pairs are collected once: walking two state_dicts of ~900 entries costs ~3.5 ms of host
        # time per step, more than the update itself.  They are re-collected when ANY parameter / buffer object of either model
        # reports a new address (p.data = ..., .to(), .half(): ~0.1 ms per step to check) and, to notice replaced sub-modules,
        # unconditionally every 128th update.  The cached aliases keep their storages alive, so a stale pair can never point
        # at freed memory.
        self._calls = getattr(self, "_calls", 0) + 1
        objs = getattr(self, "_objs", None)
        stale = objs is None or self._objs_of != id(inner) or self._calls % 128 == 0
        if not stale:
            stale = tuple(t.data_ptr() for t in objs) != self._ptrs
        if stale:
            msd = inner.state_dict()
            skip = getattr(self, "_skip_keys", ())
            self._pairs = [(v, msd[k].detach()) for k, v in self.ema.state_dict().items() if v.dtype.is_floating_point and k not in skip]
            self._objs = list(inner.parameters()) + list(inner.buffers()) + list(self.ema.parameters()) + list(self.ema.buffers())
            self._ptrs = tuple(t.data_ptr() for t in self._objs)
            self._objs_of = id(inner)
        dst = [a for a, _ in self._pairs]
        src = [b if b.dtype == a.dtype else b.to(a.dtype) for a, b in self._pairs]
        # A prefetched pass of the NEXT clip (yolox/models/tscd.py) may already have advanced the frozen front's running statistics:
        # the reference's EMA after step t sees them with clip t only (tscd_trainer.py:171-176) -- read the pre-pass snapshot instead
        early = inner.pre_prefetch_buffers() if hasattr(inner, "pre_prefetch_buffers") else {}
        tag = "ema"
        if early:
            src = [early.get(b.data_ptr(), b) for b in src]
            tag = "ema_snapshot"
        if dst and all(a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.is_contiguous()
                       and b.is_contiguous() for a, b in zip(dst, src)):
            from tscd_amd import ops          # one launch, every element read and written once (elementwise.hip)
            ops.ema_update(dst, src, d, tag=tag)
        else:                                 # CPU models (tests), half-precision copies
            torch._foreach_mul_(dst, d)
            torch._foreach_add_(dst, src, alpha=1.0 - d)
