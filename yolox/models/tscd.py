"""TSCD model wrapper (interface of the reference's yolox/models/tscd.py:11-50)."""
import os

import torch
import torch.nn as nn

from tscd_amd import graphs, ops

from .network_blocks import defer_bn_ticks
from .tscd_head import HeadPrefix

# Cross-step overlap (training, frozen backbone): while the host solves the clip's assignments the device has ~3 ms with
# nothing to do, and the launch-bound stretch after it leaves more holes.  `forward(..., next_x=<next clip>)` runs the NEXT
# clip's frozen FRONT -- the backbone pass and the head's frozen half (stems, base towers, box prediction convs:
# ovis_tscd_large.py:141-149) -- inside that window on a low-priority side stream; the next call (same tensor object) picks the
# result up instead of recomputing it.  Frozen weights do not change at the optimizer step, so the features are exactly
# those of an in-place pass, and this clip's step no longer has them on its critical path (forward up to the cost matrices -> host
# assignment chain -> rest).  TSCD_PREFETCH_BACKBONE=0 disables it.
PREFETCH_BACKBONE = os.environ.get("TSCD_PREFETCH_BACKBONE", "1") != "0"
_PREFETCHED = {}        # id(model) -> (clip tensor, pyramid, ready event, BatchNorm snapshot record); module-level: the model must stay deep-copyable
_PREFETCH_STREAM = {}   # device -> low-priority stream


_REJECTED_STREAMS = []       # candidates found to share a hardware queue with the main stream (kept alive: their slots stay taken)


def _shares_queue(main, cand, dev):
    """True when work queued on `main` behind work on `cand` does not start until that work is over: HIP multiplexes a process's
    streams onto a few in-order hardware queues (4 by default), so two streams can be one queue.  Probe: ~3 ms of fills on the
    candidate, then one tiny fill on the main stream -- if the tiny one is done while the candidate is still busy, they are apart."""
    big = torch.empty((64 << 20,), dtype=torch.float32, device=dev)
    small = torch.empty((64,), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(cand):
        for _ in range(48):
            big.fill_(0.0)
        busy = torch.cuda.Event()
        busy.record(cand)
    with torch.cuda.stream(main):
        small.fill_(1.0)
        done = torch.cuda.Event()
        done.record(main)
    done.synchronize()
    shared = busy.query()            # the candidate's work already over when the main stream's tiny launch got through
    torch.cuda.synchronize(dev)
    return shared


def _prefetch_stream(dev):
    """The stream the next clip's frozen pass replays on: low priority where the runtime has such a level, and on a hardware queue of
    its own -- under DistributedDataParallel (RCCL's streams are created first) a fresh stream was seen to land on the main stream's
    queue, and the whole assignment window then ran BEHIND the 6 ms graph (15.1 vs 13.5 ms per step)."""
    st = _PREFETCH_STREAM.get(dev)
    if st is None:
        main = torch.cuda.current_stream(dev)
        for attempt in range(8):
            try:
                st = torch.cuda.Stream(device=dev, priority=1)      # below the default stream where the runtime has such a level
            except Exception:
                st = torch.cuda.Stream(device=dev)
            if attempt == 7 or not _shares_queue(main, st, dev):
                break
            _REJECTED_STREAMS.append(st)
        _PREFETCH_STREAM[dev] = st
    return st


class _FrozenFront(nn.Module):
    """backbone + the head's frozen half (stems, base towers, box prediction convs) as ONE gradient-free pass: what
    ``graphs.frozen_pass`` captures and replays.  The modules are the model's own (shared, not copied); this wrapper is never registered
    in the model's module tree (state_dict untouched)."""

    def __init__(self, backbone, head):
        super().__init__()
        self.backbone = backbone
        self.prefix = nn.ModuleList(head.prefix_modules())
        object.__setattr__(self, "_head", head)

    def forward(self, x):
        # (One in-order pass.  Forking the stride-8 level's head prefix onto a second stream next to the neck's bottom-up half -- its
        # large launches beside the small levels' underfilled ones -- was built and measured in round 4: 13.48 vs 12.36 ms per step in
        # the same process, because the extra stream lands on a hardware queue that another stream of the step already parks on the
        # assignment latch, and the two-ranks-on-one-GPU rehearsal deadlocked.  yolo_pafpn.top_down_nhwc / bottom_up_nhwc and
        # TSCDHead.frozen_prefix_level keep the split for a runtime that gives the branch a queue of its own.)
        return self._head.frozen_prefix(self.backbone(x)).tensors()


_FRONTS = {}            # id(model) -> (weakref(model), _FrozenFront)


class TSCD(nn.Module):
    """backbone (YOLOPAFPN) + TSCDHead.  Training returns the reference's loss dict, eval (result, result_ori)."""

    def __init__(self, backbone=None, head=None):
        super().__init__()
        self.backbone = backbone
        self.head = head

    def _front(self):
        """The frozen front of this model (backbone + head prefix) when every parameter of it is frozen, else None."""
        import weakref
        if not hasattr(self.head, "prefix_modules"):
            return None
        hit = _FRONTS.get(id(self))
        if hit is None or hit[0]() is not self or hit[1].backbone is not self.backbone or hit[1]._head is not self.head:
            hit = _FRONTS[id(self)] = (weakref.ref(self), _FrozenFront(self.backbone, self.head))
        front = hit[1]
        # frozen as the exps freeze them (ovis_tscd_large.py:116-117,141-149): the backbone, and the box prediction convs (whose output
        # is part of the prefix); stems and base towers run without autograd whatever their flags say, exactly as TSCDHead always has
        if any(p.requires_grad for p in graphs.parameters_of(self.backbone)) or any(p.requires_grad for p in graphs.parameters_of(self.head.reg_preds)):
            return None
        front.training = self.training          # the graph key carries the mode; the children keep their own flags
        return front

    def forward(self, x, targets=None, time_embedding=None, nms_thresh=0.5, lframe=0, gframe=32, resume=False, next_x=None):
        with defer_bn_ticks():
            return self._forward(x, targets, time_embedding, nms_thresh, lframe, gframe, resume, next_x)

    def _pyramid(self, x):
        """What the head consumes: the prefetched / replayed frozen front (a HeadPrefix: stem outputs, base-tower features, box logits)
        when backbone AND head prefix are frozen -- the TSCD exps -- else the pyramid itself."""
        hit = _PREFETCHED.pop(id(self), None)
        if hit is not None and hit[0] is x:
            torch.cuda.current_stream().wait_event(hit[2])
            for t in hit[1]:
                ops.share(t)        # allocated under the prefetch stream, consumed on this one (ops.py: stream-safety rule)
            return HeadPrefix.from_tensors(hit[1])
        front = self._front() if x.is_cuda else None
        if front is not None:
            with torch.no_grad():
                return HeadPrefix.from_tensors(graphs.frozen_pass(front, x))
        if not torch.is_grad_enabled() or (not self.training and not any(p.requires_grad for p in self.backbone.parameters())):
            return graphs.frozen_pass(self.backbone, x)
        return self.backbone(x)

    def join_prefetch(self):
        """Orders the current stream behind a pending prefetched backbone pass (it updates the BatchNorm running statistics):
        call before anything else reads or writes the backbone's buffers -- EMA update, checkpointing, evaluation."""
        hit = _PREFETCHED.get(id(self))
        if hit is not None:
            torch.cuda.current_stream().wait_event(hit[2])

    def check_assignments(self):
        """Joins the assignment chain the device consumed without the host (deferred regime, tscd_matching.py) and raises what its
        worker raised, or a latch timeout: call between backward and the optimizer step, so that a step computed with the fallback
        (identity / stale) permutations is never applied.  By then the chain is over (it ends ~2/3 into the step, the host queues the
        backward until later): the join does not wait in the steady state."""
        m = getattr(self.head, "local_reg_matcher", None)
        if m is not None and hasattr(m, "_join_pending"):
            m._join_pending()

    def _bn_snapshot_lists(self, front):
        """(live, snapshot) lists of the running_mean / running_var buffers the frozen front updates, the snapshot tensors allocated
        once; re-collected when a buffer object was replaced (.half() / .to())."""
        hit = self.__dict__.get("_bn_snap")
        if hit is not None and all(b is m._buffers[n] for (m, n), b in zip(hit[2], hit[0])):
            return hit[0], hit[1]
        owners = [(m, n) for m in front.modules() if isinstance(m, nn.BatchNorm2d) and m.track_running_stats
                  for n in ("running_mean", "running_var")]
        live = [m._buffers[n] for m, n in owners]
        if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in live):
            self.__dict__["_bn_snap"] = None
            return None, None
        snap = [torch.zeros_like(t) for t in live]
        self.__dict__["_bn_snap"] = (live, snap, owners)
        return live, snap

    def pre_prefetch_buffers(self):
        """{data_ptr of a live running-statistics buffer: its value BEFORE the pending prefetched pass} -- empty without a pending pass.
        The prefetched pass of clip t+1 runs inside step t and advances the frozen front's BatchNorm running statistics; the
        reference's EMA / checkpoint after step t see them with clip t only (tscd_trainer.py:171-176, utils/ema.py:48-60): those
        readers take the snapshot the pass left behind.  The current stream is ordered behind the snapshot copy (not behind the pass)."""
        hit = _PREFETCHED.get(id(self))
        if hit is None or len(hit) < 4 or hit[3] is None:
            return {}
        live, snap, ev = hit[3]
        torch.cuda.current_stream().wait_event(ev)
        return {a.data_ptr(): b for a, b in zip(live, snap)}

    def step_state_dict(self):
        """state_dict() as the reference's loop would see it after the current step: with a prefetched pass pending, the frozen
        front's running statistics are the pre-pass snapshot and its num_batches_tracked counters are one lower (a resumed run
        then applies that clip once, not twice)."""
        sd = self.state_dict()
        snap = self.pre_prefetch_buffers()
        if not snap:
            return sd
        hit = self.__dict__.get("_bn_snap")
        fronts = {id(m) for (m, _) in hit[2]}
        for name, mod in self.named_modules():
            if id(mod) in fronts:
                for n in ("running_mean", "running_var"):
                    sd[f"{name}.{n}"] = snap[mod._buffers[n].data_ptr()].clone()
                # (under defer_bn_ticks the counters of a pass tick when the step's forward returns: they already include the pass)
                sd[f"{name}.num_batches_tracked"] = sd[f"{name}.num_batches_tracked"] - 1
        return sd

    def _prefetch(self, next_x):
        side = _prefetch_stream(next_x.device)
        side.wait_stream(torch.cuda.current_stream())            # next_x and every buffer the pass recycles are settled
        front = self._front()
        with torch.cuda.stream(side), torch.no_grad():
            live, snap = self._bn_snapshot_lists(front)
            snap_rec = None
            if live:
                ops.ema_update(snap, live, 0.0, tag="bn_snapshot")   # snap = 0 * snap + 1 * live: one launch, 0.6 MB
                sev = torch.cuda.Event()
                sev.record(side)
                snap_rec = (live, snap, sev)
            outs = graphs.frozen_pass(front, next_x)               # hipGraph replay after the first clip of a shape
            ev = torch.cuda.Event()
            ev.record(side)
        _PREFETCHED[id(self)] = (next_x, outs, ev, snap_rec)

    def _forward(self, x, targets, time_embedding, nms_thresh, lframe, gframe, resume, next_x=None):
        fpn_outs = self._pyramid(x)
        if self.training:
            assert targets is not None
            frozen = self._front() is not None
            if PREFETCH_BACKBONE and next_x is not None and next_x.is_cuda and frozen:
                self.head.window_hook = lambda: self._prefetch(next_x)
            try:
                (loss, iou_loss, conf_loss, cls_loss, l1_loss, num_fg, loss_matched_cls, loss_matched_obj,
                 loss_matched_iou) = self.head(fpn_outs, targets, x, time_embedding, lframe=lframe, gframe=gframe, resume=resume)
            finally:
                self.head.window_hook = None
            return {"total_loss": loss, "iou_loss": iou_loss, "l1_loss": l1_loss, "conf_loss": conf_loss, "cls_loss": cls_loss,
                    "num_fg": num_fg, "loss_matched_cls": loss_matched_cls, "loss_matched_obj": loss_matched_obj,
                    "loss_matched_iou": loss_matched_iou}
        return self.head(fpn_outs, targets, x, time_embedding, nms_thresh=nms_thresh, lframe=lframe, gframe=gframe,
                         resume=resume)
