"""hipGraph replay of a frozen sub-network pass (the TSCD backbone + PAFPN: ~110 convs, 330 launches per clip in training
mode where its BatchNorms still take batch statistics, ~150 in evaluation).

Nothing in such a pass depends on the host: frozen weights, one input tensor, a fixed launch sequence.  The first call for an
(input shape, dtype, mode) signature runs eagerly -- its result is the one returned -- and records the same launch sequence on a
side stream into a graph; later calls copy the input into the graph's static input, replay (one host call instead of ~330)
and return COPIES of the pyramid (the consumer keeps them across the next replay: the trainer runs the next clip's pass while
this clip's backward still reads them).  What replay cannot do in Python is done by hand: the BatchNorm
``num_batches_tracked`` ticks.  A graph is dropped when a parameter or buffer of the module was written by anything but the
pass itself (load_state_dict, EMA copy, optimizer on an unfrozen backbone): the packed-weight cache would hand the eager path a
new buffer while the graph kept the old one.

TSCD_GRAPH_BACKBONE=0 turns it off (the eager pass is the cross-check: tests/test_gpu_graphs.py)."""
import os

import torch

ENABLED = os.environ.get("TSCD_GRAPH_BACKBONE", "1") != "0"
_GRAPHS = {}          # (id(module), shape, dtype, training) -> _Entry
MAX_GRAPHS = 8


class _Entry:
    __slots__ = ("graph", "static_x", "outs", "ticks", "stats", "versions", "module_ref")


def _state_versions(module, training):
    """What a recorded pass depends on besides its input.  Training-mode passes read-modify-write the BatchNorm buffers inside the
    graph (always current), so only parameter writes invalidate them; an evaluation pass bakes the folded BatchNorm scale / shift in,
    so buffer writes (training passes, EMA updates, load_state_dict) invalidate it too."""
    ps, bs = _tensors_of(module)
    v = sum(p._version for p in ps)
    if not training:
        v += sum(b._version for b in bs)
    return v


_TENSORS = {}     # id(module) -> (weakref, parameters, buffers, sentinel): module.parameters() walks the module tree (2 ms per call on CSPDarknet + PAFPN)


def _sentinel(module):
    """(owner, kind, name, tensor) of one floating-point buffer (or parameter): .half() / .to() / .cuda() REPLACE every such buffer
    (Module._apply), so the cached lists are stale exactly when this one is no longer the tensor its owner holds."""
    for owner in module.modules():
        for name, b in owner._buffers.items():
            if b is not None and b.is_floating_point():
                return owner, "_buffers", name, b
    for owner in module.modules():
        for name, q in owner._parameters.items():
            if q is not None:
                return owner, "_parameters", name, q
    return None


def _tensors_of(module):
    import weakref
    hit = _TENSORS.get(id(module))
    if hit is not None and hit[0]() is module:
        sen = hit[3]
        if sen is None or getattr(sen[0], sen[1]).get(sen[2]) is sen[3]:
            return hit[1], hit[2]
    hit = _TENSORS[id(module)] = (weakref.ref(module), tuple(module.parameters()), tuple(module.buffers()), _sentinel(module))
    return hit[1], hit[2]


def parameters_of(module):
    """tuple(module.parameters()), cached (see _tensors_of)."""
    return _tensors_of(module)[0]


def _running_stats(module):
    from yolox.models.network_blocks import BaseConv
    out = []
    for m in module.modules():
        if isinstance(m, BaseConv) and m.training and m.bn.training and m.bn.track_running_stats:
            out += [m.bn.running_mean, m.bn.running_var]
    return out


def _tick_buffers(module):
    from yolox.models.network_blocks import BaseConv
    return [m.bn.num_batches_tracked for m in module.modules()
            if isinstance(m, BaseConv) and m.training and m.bn.training and m.bn.track_running_stats]


def _apply_ticks(ticks):
    from yolox.models import network_blocks as nb
    if not ticks:
        return
    if nb._DEFERRED_TICKS is not None:
        nb._DEFERRED_TICKS.extend(ticks)
    else:
        torch._foreach_add_(ticks, 1)


def frozen_pass(module, x):
    """module(x) for a module whose parameters do not require grad, under no_grad, on the current stream."""
    if not ENABLED or not x.is_cuda or torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        return module(x)
    key = (id(module), tuple(x.shape), x.dtype, x.stride(), module.training)
    ent = _GRAPHS.get(key)
    if ent is not None and (ent.module_ref() is not module or ent.versions != _state_versions(module, module.training)):
        del _GRAPHS[key]
        ent = None
    if ent is None:
        with torch.no_grad():
            outs = module(x)                         # this call's result; also warms every cache the capture must hit
        if len(_GRAPHS) < MAX_GRAPHS:
            _capture(module, x, key)
        return outs
    cur = torch.cuda.current_stream()
    ent.static_x.copy_(x, non_blocking=True)
    ent.graph.replay()
    _apply_ticks(ent.ticks)
    if ent.stats:                         # the replayed kernels wrote the running statistics: same version bump as the eager pass
        torch.autograd.graph.increment_version(ent.stats)
    outs = tuple(o.clone() for o in ent.outs)
    for o in outs:
        o.record_stream(cur)
    return outs


def _capture(module, x, key):
    import weakref
    from yolox.models import network_blocks as nb
    ent = _Entry()
    ent.static_x = x.clone()
    ent.ticks = _tick_buffers(module)
    ent.stats = _running_stats(module)
    ent.module_ref = weakref.ref(module)
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream(device=x.device)
    side.wait_stream(cur)
    saved, nb._DEFERRED_TICKS = nb._DEFERRED_TICKS, []           # the recorded pass executes nothing: its ticks are not real
    try:
        g = torch.cuda.CUDAGraph()
        # thread_local: other threads keep making HIP calls while this one records (the assignment worker waits on its copy event, the
        # process group's watchdog polls its events) -- only this thread's calls are checked against the capture
        with torch.no_grad(), torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            outs = module(ent.static_x)
    finally:
        nb._DEFERRED_TICKS = saved
    cur.wait_stream(side)
    ent.graph = g
    ent.outs = tuple(outs) if isinstance(outs, (tuple, list)) else (outs,)
    ent.versions = _state_versions(module, module.training)
    _GRAPHS[key] = ent


def clear():
    _GRAPHS.clear()
    _TENSORS.clear()
