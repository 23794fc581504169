"""Step tail of the TSCD trainer as one launch: Nesterov-momentum SGD over the reference's parameter
groups (ovis_tscd_large.py:164-197) with the GradScaler's unscale / skip-on-overflow (tscd_trainer.py:171-176) and the EMA
update of the same parameters (ema.py:48-60), csrc/elementwise.hip: tscd_sgd_ema_step_f32.

``FusedSGD`` is a ``torch.optim.SGD`` (same constructor arguments, same ``state_dict`` layout -- ``momentum_buffer`` per parameter
-- so checkpoints interchange with the reference's) whose ``step`` makes one kernel launch for all groups.  It declares
``_step_supports_amp_scaling``: ``GradScaler.step`` hands it ``grad_scale`` and ``found_inf`` as device tensors and never
synchronises.  ``attach_ema(ema_of)`` adds the EMA copies of the trainable parameters to the same pass; ``ModelEMA.update`` then
only has to cover what else can change (BatchNorm buffers): frozen parameters equal their EMA copies and stay equal."""
import numpy as np
import torch

from . import ops
from ._lib import call, lib


class FusedSGD(torch.optim.SGD):
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        super().__init__(params, lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov)
        self._ema_of = {}          # id(parameter) -> EMA copy (fp32 tensor of the same shape)
        self._ema_decay = None     # callable: () -> decay of this step (advances the EMA's update counter)
        self._table = None
        self._reducer = None       # tscd_amd.ddp.ArenaReducer: gradients are then read from its flat arena (averaged over the ranks)
        self._gbase = None         # this step's device array of gradient pointers (built by check_finite or step, whichever runs first)

    def attach_reducer(self, reducer):
        """Multi-GPU: ``reducer.finish()`` (all-reduce of the flat gradient arena) runs at the head of check_finite / step, and the
        kernels read the averaged gradients from the arena (static addresses: no per-step pointer upload)."""
        self._reducer = reducer
        self._table = None

    def attach_ema(self, ema_of, decay_fn):
        """ema_of: dict parameter -> its EMA tensor; decay_fn(): the decay d of the coming update (called once per step)."""
        self._ema_of = {id(p): e for p, e in ema_of.items()}
        self._ema_decay = decay_fn
        self._table = None

    def handles_ema_of(self, p):
        return id(p) in self._ema_of

    def _build(self):
        ch = lib.tscd_ema_chunk_elems()
        rec, sig, params = [], [], []
        for gi, group in enumerate(self.param_groups):
            if group["dampening"] != 0 or group.get("maximize", False):
                raise NotImplementedError("FusedSGD: dampening / maximize are not on the TSCD path")
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if st.get("momentum_buffer") is None:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                buf, g, e = st["momentum_buffer"], p.grad, self._ema_of.get(id(p))
                for t in (p, g, buf) + ((e,) if e is not None else ()):
                    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()):
                        raise TypeError("FusedSGD: contiguous fp32 CUDA parameters, gradients and EMA copies only")
                pp, pb, pe, n = p.data_ptr(), buf.data_ptr(), (e.data_ptr() if e is not None else 0), p.numel()
                pi = len(params)
                # gradient: byte offset + parameter index (tscd_sgd_ema_step_indirect_f32) -- the table does not depend on where
                # this step's gradients were allocated
                for o in range(0, n, ch):
                    rec.append((pp + 4 * o, 4 * o, pb + 4 * o, (pe + 4 * o) if pe else 0, min(ch, n - o), gi | (pi << 8)))
                sig.append((pp, pb, pe))
                params.append(p)
        dev = params[0].device if params else None
        table = torch.from_numpy(np.array(rec, dtype=np.int64).reshape(-1, 6)).to(dev) if rec else None
        # the same chunks as rows of the non-finite check (tscd_grad_gather_check_f32: {dst = NULL, byte offset, n, parameter index})
        chk = torch.from_numpy(np.array([(0, r[1], r[4], r[5] >> 8) for r in rec], dtype=np.int64).reshape(-1, 4)).to(dev) if rec else None
        self._table = (tuple(sig), table, len(rec), params, len(self.param_groups), chk)
        self._static_gbase = None

    def _ensure_table(self):
        if len(self.param_groups) > 8:
            raise NotImplementedError("FusedSGD: at most 8 parameter groups")
        # the chunk table names parameters, momentum buffers and EMA copies (stable addresses); gradients are re-allocated by
        # zero_grad(set_to_none=True) every step and reach the kernel through a per-step array of ~150 pointers
        # (checked once per step: check_finite and step share it -- _gbase is this step's marker; the walk over ~150 parameters is 0.3 ms
        # of host time)
        if self._table is not None and self._gbase is not None and self._table[4] == len(self.param_groups):
            return self._table
        if self._table is None or self._table[0] != self._signature() or self._table[4] != len(self.param_groups):
            self._build()
        return self._table

    def _grad_pointers(self, params):
        """Device array of this step's gradient pointers, built once per step (check_finite and step share it).  With a reducer the
        gradients are the arena's slices: static, uploaded once."""
        if self._gbase is not None:
            return self._gbase
        if self._reducer is not None:
            if self._static_gbase is None or self._static_gbase[0] != self._reducer.generation:
                ptrs = [self._reducer.view_of(p).data_ptr() for p in params]
                self._static_gbase = (self._reducer.generation, ops.h2d(ptrs, torch.int64, params[0].device))
            self._gbase = self._static_gbase[1]
            return self._gbase
        for p in params:
            g = p.grad
            if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.numel() == p.numel()):
                raise TypeError("FusedSGD: contiguous fp32 CUDA gradients only")
        self._gbase = ops.h2d([p.grad.data_ptr() for p in params], torch.int64, params[0].device)
        return self._gbase

    @torch.no_grad()
    def check_finite(self, found_inf):
        """ORs 1 into the device scalar ``found_inf`` when any gradient of this step holds Inf / NaN: ONE launch over the same chunk
        table as the step (torch's GradScaler needs 6 multi-tensor launches for the ~150 gradient tensors).  With a reducer attached
        the check runs on the all-reduced arena, so every rank takes the same skip decision."""
        if self._reducer is not None:
            self._reducer.finish()
        self._gbase = None               # a new step begins here (a previous one whose step() never ran must not lend its pointers)
        sig, table, n, params, _, chk = self._ensure_table()
        if n == 0:
            return
        if not (found_inf.is_cuda and found_inf.dtype == torch.float32 and found_inf.numel() == 1):
            raise TypeError("FusedSGD.check_finite: found_inf must be an fp32 device scalar")
        call("tscd_grad_gather_check_f32", chk.data_ptr(), n, self._grad_pointers(params).data_ptr(), 1.0, found_inf.data_ptr(), ops._stream())

    def _signature(self):
        sig = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                buf = self.state[p].get("momentum_buffer") if p in self.state else None
                e = self._ema_of.get(id(p))
                sig.append((p.data_ptr(), buf.data_ptr() if buf is not None else -1, e.data_ptr() if e is not None else 0))
        return tuple(sig)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None, found_inf=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._reducer is not None:
            self._reducer.finish()        # (no-op when check_finite already joined this step's all-reduce)
        sig, table, n, params, _, _ = self._ensure_table()
        if n == 0:
            return loss
        lr = (lib_float * len(self.param_groups))(*[float(g["lr"]) for g in self.param_groups])
        wd = (lib_float * len(self.param_groups))(*[float(g["weight_decay"]) for g in self.param_groups])
        g0 = self.param_groups[0]
        if any(g["momentum"] != g0["momentum"] or g["nesterov"] != g0["nesterov"] for g in self.param_groups):
            raise NotImplementedError("FusedSGD: one momentum / nesterov setting for all groups")
        d = float(self._ema_decay()) if (self._ema_decay is not None and self._ema_of) else 1.0
        gs = grad_scale if grad_scale is not None else getattr(self, "grad_scale", None)
        fi = found_inf if found_inf is not None else getattr(self, "found_inf", None)
        for t in (gs, fi):
            if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.numel() == 1):
                raise TypeError("FusedSGD: grad_scale / found_inf must be fp32 device scalars")
        gbase, self._gbase = self._grad_pointers(params), None
        call("tscd_sgd_ema_step_indirect_f32", table.data_ptr(), n, gbase.data_ptr(), lr, wd, len(self.param_groups), float(g0["momentum"]),
             int(bool(g0["nesterov"])), ops._ptr(gs), ops._ptr(fi), d, 1.0 - d, ops._stream())
        # raw-pointer writes: bump the version counters so caches keyed on them (packed weights, folded BN, graphs) notice
        torch.autograd.graph.increment_version(params)
        if self._ema_of:
            torch.autograd.graph.increment_version([self._ema_of[id(p)] for p in params if id(p) in self._ema_of])
        ops.refresh_packed_weights(params)       # every packed fp16 copy of the stepped weights, in place, one launch
        return loss


import ctypes as _C

lib_float = _C.c_float


class GradScaler(torch.amp.GradScaler):
    """torch.amp.GradScaler (the reference's ``torch.cuda.amp.GradScaler``, tscd_trainer.py:113,171-176) whose inf check over a
    FusedSGD's gradients is that optimizer's own one-launch pass (FusedSGD.check_finite) instead of torch's per-dtype / per-device
    multi-tensor kernels; everything else -- scale growth / backoff, state_dict, other optimizers -- is the base class."""

    def __init__(self, device="cuda", **kw):
        super().__init__(device, **kw)

    def _check_inf_per_device(self, optimizer):
        if not isinstance(optimizer, FusedSGD):
            return super()._check_inf_per_device(optimizer)
        _scale, _ = self._check_scale_growth_tracker("_check_inf_per_device")
        found_inf = torch.zeros((), dtype=torch.float32, device=_scale.device)
        optimizer.check_finite(found_inf)
        self._per_optimizer_states[id(optimizer)]["found_inf_per_device"] = {_scale.device: found_inf}
        return self._per_optimizer_states[id(optimizer)]["found_inf_per_device"]
