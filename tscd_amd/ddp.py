"""Data-parallel wiring of the clip-sharded train step: one process per GPU, one clip per rank, gradients averaged over RCCL / xGMI
(reference contract: yolox/core/tscd_trainer.py:230-231 ``DDP(model, device_ids=[local_rank], broadcast_buffers=False)`` over the
process group of yolox/core/launch.py; after ``backward()`` every rank holds the mean gradient).

Not torch's DistributedDataParallel: its reducer copies every parameter's gradient into a bucket with a launch of its own (~150 small
launches on the backward stream of a TSCD-L step) and runs its hooks on autograd's thread -- +0.8-1.1 ms per step before any
communication (profiles/r03_ddp_ab.txt).  Here:

* ONE flat fp32 arena holds the gradients of all trainable parameters, in the order in which the backward pass finishes them
  (recorded on the first step on rank 0 and broadcast, so every rank uses the same layout), cut into ``nchunks`` byte-balanced chunks;
* a post-accumulate hook per parameter only counts; when the last gradient of a chunk has arrived, ONE gather launch
  (tscd_grad_gather_check_f32: per-step source pointers, static destinations) copies the chunk's gradients into the arena on the
  communication stream and ONE all-reduce (ReduceOp.AVG on RCCL; gloo: the gather pre-divides, then SUM) follows it there -- the
  early chunks (proposal stage: MCA, CAFM, TaskAligned, prediction linears) travel while the towers' backward still computes;
* the backward pass's final callback launches what is left, makes the caller's stream wait for the collectives and points every
  ``p.grad`` at its arena slice (the mean gradient, as after torch DDP's backward);
* ``FusedSGD.attach_reducer`` reads the arena directly (static addresses) and runs its inf check on the REDUCED arena, so every rank
  takes the same skip decision; ``broadcast_buffers=False`` semantics: BatchNorm statistics stay per rank (tscd_trainer.py:231).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of the 78.8 MB of TSCD-L gradients moves 2 * 7/8 * 78.8 MB per
GPU over one link, ~0.9 ms at 8 ranks -- fewer, larger collectives (3-4 chunks of ~20-26 MB) keep each at the per-link bandwidth.

The static set of parameters without a gradient must be frozen (``requires_grad=False``), as for torch DDP: a trainable
parameter that receives no gradient raises at the end of the pass.  One backward pass per optimizer step (no gradient accumulation
across passes)."""
import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn


class ArenaReducer:
    def __init__(self, params, process_group=None, nchunks=4):
        self.pg = process_group if process_group is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.pg)
        self.backend = dist.get_backend(self.pg)
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "ArenaReducer: no trainable parameters"
        p0 = self.params[0]
        if not all(p.dtype == torch.float32 and p.device == p0.device for p in self.params):
            raise TypeError("ArenaReducer: fp32 parameters on one device")
        self.device = p0.device
        self.cuda = self.device.type == "cuda"
        self.nchunks_wanted = max(1, int(nchunks))
        self.comm = torch.cuda.Stream(self.device) if self.cuda else None
        self.generation = 0              # bumped when the arena layout changes (FusedSGD re-reads the addresses)
        self.publish_grads = True        # p.grad = arena slice after the pass (the mean gradient, as after torch DDP's backward)
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._layout(list(range(len(self.params))), 1)
        self._order, self._learned = [], False
        self._works, self._active, self._finished = [], False, True
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._ready)
        if self.cuda:
            from . import ops
            ops.FOREIGN_GRAD_HOOKS_OK = True     # this reducer flushes the deferred weight gradients before it reads a gradient

    # ------------------------------------------------------------------ layout
    def _layout(self, order, nchunks):
        """Arena = the parameters in ``order`` (16-byte aligned slices), cut into ``nchunks`` chunks of about equal bytes."""
        offs, off = {}, 0
        for i in order:
            offs[i] = off
            off += (self.params[i].numel() + 3) // 4 * 4
        total = off
        self.arena = torch.zeros((max(total, 4),), dtype=torch.float32, device=self.device)
        self.views = [None] * len(self.params)
        for i in order:
            p = self.params[i]
            self.views[i] = self.arena[offs[i]:offs[i] + p.numel()].view_as(p)
        target, cuts, acc = total / nchunks, [], 0
        chunks, cur = [], []
        for i in order:
            cur.append(i)
            acc += (self.params[i].numel() + 3) // 4 * 4
            if acc >= target * (len(chunks) + 1) and len(chunks) < nchunks - 1:
                chunks.append(cur)
                cur = []
        if cur:
            chunks.append(cur)
        self.chunks = []
        for members in chunks:
            lo = offs[members[0]]
            hi = offs[members[-1]] + (self.params[members[-1]].numel() + 3) // 4 * 4
            self.chunks.append({"members": members, "lo": lo, "hi": hi, "table": None, "n": 0})
        self.chunk_of = {}
        for c, ck in enumerate(self.chunks):
            for i in ck["members"]:
                self.chunk_of[i] = c
        if self.cuda:
            from ._lib import lib
            ch = lib.tscd_ema_chunk_elems()
            for ck in self.chunks:
                rows = []
                for slot, i in enumerate(ck["members"]):
                    n, base = self.params[i].numel(), self.views[i].data_ptr()
                    for o in range(0, n, ch):
                        rows.append((base + 4 * o, 4 * o, min(ch, n - o), slot))
                ck["table"] = torch.from_numpy(np.array(rows, dtype=np.int64).reshape(-1, 4)).to(self.device)
                ck["n"] = len(rows)
        self.generation += 1

    def view_of(self, p):
        return self.views[self._index[id(p)]]

    # ------------------------------------------------------------------ the pass
    def _begin(self):
        self._active, self._finished = True, False
        self._arrived = [0] * len(self.chunks)
        self._launched = [False] * len(self.chunks)
        self._got = [False] * len(self.params)
        self._works = []
        torch.autograd.Variable._execution_engine.queue_callback(self._end_of_pass)

    def _ready(self, p):
        if not self._active:
            self._begin()
        i = self._index[id(p)]
        if self._got[i]:
            raise RuntimeError("ArenaReducer: a parameter's gradient arrived twice in one pass (one backward pass per step)")
        self._got[i] = True
        if not self._learned:
            self._order.append(i)
        c = self.chunk_of[i]
        self._arrived[c] += 1
        if self._arrived[c] == len(self.chunks[c]["members"]) and len(self.chunks) > 1:
            self._launch(c)

    def _launch(self, c):
        ck = self.chunks[c]
        self._launched[c] = True
        buf = self.arena[ck["lo"]:ck["hi"]]
        grads = [self.params[i].grad for i in ck["members"]]
        if any(g is None for g in grads):
            missing = [i for i in ck["members"] if self.params[i].grad is None]
            raise RuntimeError(f"ArenaReducer: {len(missing)} trainable parameter(s) received no gradient (freeze the static unused set, "
                               f"SURVEY F5); first index {missing[0]}")
        avg_in_collective = self.backend == "nccl"
        scale = 1.0 if avg_in_collective else 1.0 / self.world
        if self.cuda:
            from . import ops
            ops.flush_deferred_wgrads()                     # registered-but-unwritten weight gradients are computed first
            main = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)
                for g in grads:
                    if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()):
                        raise TypeError("ArenaReducer: contiguous fp32 CUDA gradients only")
                src = ops.h2d([g.data_ptr() for g in grads], torch.int64, self.device)
                ops.call("tscd_grad_gather_check_f32", ck["table"].data_ptr(), ck["n"], src.data_ptr(), scale, 0, ops._stream())
                work = dist.all_reduce(buf, op=dist.ReduceOp.AVG if avg_in_collective else dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._keep = getattr(self, "_keep", [])
            self._keep.append((grads, src))                 # read by the communication stream: alive until the pass has joined it
        else:                                               # CPU rehearsal of the plumbing (gloo, tests): torch copies
            for i, g in zip(ck["members"], grads):
                self.views[i].copy_(g)
            buf.mul_(scale)
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._works.append(work)

    def _end_of_pass(self):
        self.finish()

    def finish(self):
        """Launches the chunks that are still pending, orders the current stream behind every collective of the pass and publishes the
        mean gradients.  Idempotent; FusedSGD calls it before it reads the arena."""
        if self._finished:
            return
        self._finished, self._active = True, False
        for c in range(len(self.chunks)):
            if not self._launched[c]:
                self._launch(c)
        for w in self._works:
            w.wait()                     # RCCL: the CURRENT STREAM waits (no host block); gloo: the host waits
        self._works = []
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.comm)
            self._keep = []
        if self.publish_grads:
            for p, v in zip(self.params, self.views):
                p.grad = v
        if not self._learned:
            self._learn_order()

    def _learn_order(self):
        """After the first pass: rank 0's gradient-ready order becomes the arena order on every rank."""
        self._learned = True
        order = self._order if len(self._order) == len(self.params) else list(range(len(self.params)))
        box = [order]
        dist.broadcast_object_list(box, src=dist.get_global_rank(self.pg, 0) if self.pg is not dist.group.WORLD else 0, group=self.pg)
        order = list(box[0])
        assert sorted(order) == list(range(len(self.params))), "ArenaReducer: the ranks' parameter lists differ"
        old = [v.clone() for v in self.views] if self.publish_grads else None
        self._layout(order, min(self.nchunks_wanted, len(order)))
        if old is not None:              # this step's (already averaged) gradients move with the layout
            for p, v, o in zip(self.params, self.views, old):
                v.copy_(o)
                p.grad = v


class ArenaDDP(nn.Module):
    """``model`` with an ArenaReducer over its trainable parameters.  ``.module`` is the wrapped model (as for torch's wrapper: the
    trainer unwraps it with ``is_parallel``); forward just calls it."""

    def __init__(self, module, process_group=None, nchunks=4):
        super().__init__()
        self.module = module
        self.reducer = ArenaReducer(list(module.parameters()), process_group, nchunks)

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def attach(self, optimizer):
        """FusedSGD reads the reduced arena directly (and checks it for Inf / NaN after the reduction)."""
        if hasattr(optimizer, "attach_reducer"):
            optimizer.attach_reducer(self.reducer)
        return optimizer


def wrap(model, device_ids=None, process_group=None, nchunks=4):
    """model -> ArenaDDP(model); ``broadcast_buffers=False`` semantics as the reference's trainer (tscd_trainer.py:231): each rank keeps
    the BatchNorm running statistics of its own clips.  ``device_ids`` is accepted for the reference call's shape (one device per
    process: the model's own)."""
    return ArenaDDP(model, process_group, nchunks)
