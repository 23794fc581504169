"""Torch-tensor wrappers over the C-ABI (include/tscd_hip.h).

Activations are fp16 tensors with LOGICAL shape [B, H, W, C] (NHWC), either contiguous or a channel slice of a
contiguous NHWC buffer (``stride(-1) == 1``, pixel stride ``ld = stride(-2)``) -- concat-free CSP / PAFPN
buffers are plain torch views.  Kernels run on ``torch.cuda.current_stream()`` and only borrow ``data_ptr()``s.
PyTorch is plumbing here (allocation, streams, autograd bookkeeping); all arithmetic is in the HIP kernels.
"""
import weakref

import torch

from ._lib import call, lib

EPI_RAW, EPI_BIAS, EPI_BIAS_RELU, EPI_BN_SILU, EPI_BN = 0, 1, 2, 3, 4
ACT_NONE, ACT_SILU, ACT_RELU = 0, 1, 2
BN_EPS, BN_MOMENTUM = 1e-3, 0.03  # reference: exps/TSCD_OVIS/ovis_tscd_large.py:109-113


_raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice


def _stream():
    """hipStream_t of torch's current stream on the current device (the raw C accessors: torch.cuda.current_stream() costs
    ~8 us of Python per call, which at ~500 kernel launches per step made the forward host-bound)."""
    return _raw_stream(_cur_device())


def _ptr(t):
    return 0 if t is None else t.data_ptr()


# ----------------------------------------------------------------------------------------------------
# stream safety of per-call temporaries
# ----------------------------------------------------------------------------------------------------
# RULE.  A wrapper's per-call temporaries (workspaces, BatchNorm partial sums, scale / shift rows, outputs) are allocated with
# ``temp`` / ``ztemp`` while the stream the kernel will be launched on is torch's CURRENT stream, and the launch passes ``_stream()``:
# the caching allocator hands a freed block only to later work of the SAME stream, so a temporary can die on the host while its
# kernel is still queued.  A launch on any other stream than the one that was current at allocation reads memory the allocating
# stream's pool may recycle underneath it (round 3: a "launch-stream override" variant faulted exactly so).  A tensor that does cross
# streams (an output consumed behind an event wait) is handed over with ``share(t, stream)`` (= Tensor.record_stream).
# TSCD_DEBUG_STREAMS=1 checks the rule at every C-ABI call: the launch stream must be torch's current stream, and (outside backward
# passes, where the autograd engine does the hand-over) each pointer argument that lies in a live ``temp`` allocation must have been
# allocated under -- or shared with -- the launching stream.
import bisect as _bisect
import os as _os0

DEBUG_STREAMS = _os0.environ.get("TSCD_DEBUG_STREAMS", "0") == "1"
_TEMP_STARTS, _TEMP_ENTRIES = [], {}       # sorted block starts; start -> [end, allocating raw stream, set(streams shared with)]
_STREAM_FIRST = {"tscd_stream_host_func"}  # entry points whose stream is not the last argument


def _drop_temp(a, token):
    e = _TEMP_ENTRIES.get(a)
    if e is not None and e[3] is token:
        del _TEMP_ENTRIES[a]
        i = _bisect.bisect_left(_TEMP_STARTS, a)
        if i < len(_TEMP_STARTS) and _TEMP_STARTS[i] == a:
            del _TEMP_STARTS[i]


def _register_temp(t):
    n = t.numel() * t.element_size()
    if n == 0 or not t.is_cuda:
        return t
    a = t.data_ptr()
    i = _bisect.bisect_left(_TEMP_STARTS, a)
    # a new allocation replaces whatever the registry still knows about the bytes it covers
    while i < len(_TEMP_STARTS) and _TEMP_STARTS[i] < a + n:
        del _TEMP_ENTRIES[_TEMP_STARTS[i]]
        del _TEMP_STARTS[i]
    if i > 0 and _TEMP_ENTRIES[_TEMP_STARTS[i - 1]][0] > a:
        _TEMP_ENTRIES[_TEMP_STARTS[i - 1]][0] = a
    _TEMP_STARTS.insert(i, a)
    token = object()
    # the entry lives as long as the tensor OBJECT it was made for: once that is gone the block may belong to anything (a parameter
    # created later, a torch-internal buffer), which this registry knows nothing about.  (A view that outlives its base is then no
    # longer checked: the check errs on the quiet side.)
    ref = weakref.ref(t, lambda _r, a=a, token=token: _drop_temp(a, token))
    _TEMP_ENTRIES[a] = [a + n, _stream(), set(), token, ref]
    return t


def _temp_entry(ptr):
    i = _bisect.bisect_right(_TEMP_STARTS, ptr) - 1
    if i < 0:
        return None
    e = _TEMP_ENTRIES[_TEMP_STARTS[i]]
    return e if ptr < e[0] else None


if DEBUG_STREAMS:
    def temp(*a, **k):
        """torch.empty for a per-call temporary (see RULE above); registered for the TSCD_DEBUG_STREAMS check."""
        return _register_temp(torch.empty(*a, **k))

    def ztemp(*a, **k):
        return _register_temp(torch.zeros(*a, **k))

    _plain_call = call

    def call(name, *args):      # noqa: F811  (the checking form of _lib.call)
        launch = args[0] if name in _STREAM_FIRST else args[-1]
        launch = 0 if launch is None else int(launch)
        cur = _stream()
        if launch != cur:
            raise AssertionError(f"TSCD_DEBUG_STREAMS: {name} launches on stream {launch:#x} while torch's current stream is {cur:#x}")
        # Inside a backward pass the autograd engine hands tensors across streams itself: it orders the consumer behind the producer
        # and records the consumer's stream on the gradient (torch/csrc/autograd/input_buffer.cpp), which this registry cannot
        # see -- there only the launch stream is checked.
        in_backward = torch._C._current_graph_task_id() >= 0
        for j, v in enumerate(args):
            if in_backward:
                break
            if isinstance(v, int) and v >= (1 << 32):
                e = _temp_entry(v)
                if e is not None and e[1] != launch and launch not in e[2]:
                    raise AssertionError(f"TSCD_DEBUG_STREAMS: argument {j} of {name} is a temporary allocated under stream {e[1]:#x} "
                                         f"but the launch is on stream {launch:#x} (allocate under the launching stream or ops.share it)")
        return _plain_call(name, *args)
else:
    temp, ztemp = torch.empty, torch.zeros


def share(t, stream=None):
    """Hands a tensor allocated under another stream over to ``stream`` (default: the current one): Tensor.record_stream, so its
    block is not recycled before that stream's queued work has passed, plus the debug registry's note."""
    st = stream if stream is not None else torch.cuda.current_stream()
    t.record_stream(st)
    if DEBUG_STREAMS:
        e = _temp_entry(t.data_ptr())
        if e is not None:
            e[2].add(int(st.cuda_stream))
    return t


# ----------------------------------------------------------------------------------------------------
# parameter gradients on their own stream
# ----------------------------------------------------------------------------------------------------
# Nothing in a backward pass waits for a weight gradient: the data gradients form the dependent chain, the weight gradients only
# feed the optimizer.  Inside a backward pass `param_grad_stream` moves the enclosed launches (a weight-gradient GEMM, its split-K
# reduce, a bias column sum) to a second stream that follows the current one by an event; the pass's final callback makes the
# caller's stream wait for it, so whatever reads the gradients next (GradScaler's unscale, the optimizer) is ordered behind them.
# The tensors the launches READ are kept alive until that join (the caching allocator would otherwise hand their blocks to later
# work on the first stream); the gradients themselves are allocated under the second stream and reach `param.grad` through
# autograd's AccumulateGrad, which adopts a fresh contiguous gradient without touching it.  Hence the conditions: the parameter has
# no gradient yet (`grad is None`), this is its first contribution in this pass, and no reducer hooks are waiting for it
# (torch.distributed initialised with more than one rank, or TSCD_ONE_GRAD_STREAM=1: every launch stays on the pass's own stream).
import contextlib
import os as _os

# Off by default: with the whole backward queued ahead of the device (deferred assignment chain) a second queue gained 0.05 ms at
# most on the bench step (profiles/r03_stream_layouts.txt); TSCD_PARAM_GRAD_STREAM=1 turns it on.
PARAM_GRAD_STREAM = _os.environ.get("TSCD_PARAM_GRAD_STREAM", "0") == "1"
_PG = {}


class _ParamGradState:
    def __init__(self, dev):
        self.stream = torch.cuda.Stream(dev)
        self.keep, self.seen, self.task = [], set(), -1

    def join(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        self.keep.clear()
        self.seen.clear()
        self.task = -1


def _param_grad_state(params):
    if not PARAM_GRAD_STREAM or _os.environ.get("TSCD_ONE_GRAD_STREAM") == "1":
        return None
    task = torch._C._current_graph_task_id()
    if task < 0:
        return None
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        return None
    dev = _cur_device()
    st = _PG.get(dev)
    if st is None:
        st = _PG[dev] = _ParamGradState(dev)
    if st.task != task:
        if st.task >= 0:          # a pass that ended without its callback (an exception in backward): order behind its leftovers
            st.join()
        torch.autograd.Variable._execution_engine.queue_callback(st.join)
        st.task = task
    for prm in params:
        if prm.grad is not None or id(prm) in st.seen:
            # accumulation into an existing gradient runs a kernel on the pass's own stream: stay there, behind what is in flight
            torch.cuda.current_stream().wait_stream(st.stream)
            return None
    st.seen.update(id(prm) for prm in params)
    return st


@contextlib.contextmanager
def param_grad_stream(params, *reads):
    """with param_grad_stream([weight, ...], x, dy): dw = conv_wgrad(x, dy, ...)  -- see the note above.  Yields True when the body
    runs on the second stream."""
    st = _param_grad_state(params)
    if st is None:
        yield False
        return
    ev = torch.cuda.Event()
    ev.record()
    st.stream.wait_event(ev)
    st.keep.extend(reads)
    with torch.cuda.stream(st.stream):
        yield True


def require_gpu(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"tscd_amd: {name} is on {t.device}; the TSCD kernels run on an MI355X only "
                           "(no CPU / PyTorch fallback in the product path)")


def nhwc_ld(t):
    """Pixel stride (elements) of an NHWC tensor / channel-slice view; raises if the layout is not NHWC-dense."""
    assert t.dim() == 4 and t.dtype == torch.float16, (t.shape, t.dtype)
    B, H, W, C = t.shape
    s = t.stride()
    ld = s[2] if W > 1 else (s[1] if H > 1 else (s[0] if B > 1 else C))
    ok = (C == 1 or s[3] == 1) and (W == 1 or s[2] == ld) and (H == 1 or s[1] == W * ld) and (B == 1 or s[0] == H * W * ld)
    if not ok or ld < C:
        raise ValueError(f"not an NHWC(-slice) layout: shape {tuple(t.shape)} strides {s}")
    return ld


def empty_nhwc(B, H, W, C, device):
    return temp((B, H, W, C), dtype=torch.float16, device=device)


# ----------------------------------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------------------------------
def _pack_dims(shape, cin_pad, mode):
    Cout, Cin, KH, KW = shape
    inner, rows = ((cin_pad or Cin), Cout) if mode == 0 else (Cout, Cin)
    bn = lib.tscd_conv_tile_n(rows)
    return Cout, Cin, inner, KH, KW, (rows + bn - 1) // bn * bn, (KH * KW * inner + 31) // 32 * 32


def pack_conv_weight(w, cin_pad=None, mode=0):
    """torch [Cout,Cin,KH,KW] (fp32/fp16) -> packed fp16 [rows_pad, Kpad] for tscd_conv2d_nhwc_f16.
    mode=1 packs the matrix that makes the same kernel compute the data gradient of a stride-1 conv."""
    require_gpu(w, "weight")
    Cout, Cin, inner, KH, KW, rows_pad, kpad = _pack_dims(w.shape, cin_pad, mode)
    w = w.detach().contiguous()
    out = torch.empty((rows_pad, kpad), dtype=torch.float16, device=w.device)    # cached by WeightCache: persistent, not a per-call temporary
    assert w.dtype in (torch.float32, torch.float16)
    call("tscd_pack_conv_weight", w.data_ptr(), int(w.dtype == torch.float32), out.data_ptr(), Cout, Cin, inner, KH, KW,
         rows_pad, kpad, mode, _stream())
    return out


class WeightCache:
    """Packed-weight cache keyed on the parameter's storage and in-place version counter, so frozen weights are
    packed once and trainable ones are re-packed only after an optimizer step / load_state_dict."""

    def __init__(self):
        self._c = {}

    @staticmethod
    def _tag(w):
        return (w.data_ptr(), w._version, w.dtype, w.device, tuple(w.shape))

    def _lookup(self, key, w, make, shape4=None, cin_pad=None, mode=0):
        # id() values are recycled once a tensor dies, so an entry is only valid while its weakref still points at `w`
        ent = self._c.get(key)
        if ent is None or ent[2]() is not w or ent[0] != self._tag(w):
            if len(self._c) > 4096:
                self._c = {k: v for k, v in self._c.items() if v[2]() is not None}
            # (tag, packed, weakref, [Cout, Cin, KH, KW] the packing sees, cin_pad, mode): the last three let refresh() repack in place
            ent = (self._tag(w), make(), weakref.ref(w), tuple(shape4 if shape4 is not None else w.shape), cin_pad, mode)
            self._c[key] = ent
            _REFRESH_TABLES.clear()
        return ent[1]

    def get(self, w, cin_pad=None, mode=0):
        return self._lookup((id(w), mode, cin_pad), w, lambda: pack_conv_weight(w, cin_pad, mode), None, cin_pad, mode)


_REFRESH_TABLES = {}      # (ids of the stepped parameters) -> (signature, jobs, blockmap, nblocks, entries)


def refresh_packed_weights(params):
    """Re-packs, IN PLACE and in one launch, every cached packed copy (conv forward / data-gradient order, linear) of the given
    parameters -- called by the optimizer right after its step, so the next forward and backward find every copy current and launch
    no pack kernel (~90 per TSCD-L step otherwise).  Copies whose parameter was replaced or reshaped are left to the lazy path."""
    import numpy as np
    key = tuple(id(p) for p in params)
    hit = _REFRESH_TABLES.get(key)
    if hit is None:
        want = set(key)
        ents = []
        for cache in (WEIGHTS, LINEAR_WEIGHTS):
            for k, e in cache._c.items():
                w = e[2]()
                if w is not None and id(w) in want and w.is_cuda and w.is_contiguous() and w.dtype in (torch.float32, torch.float16):
                    ents.append((cache, k))
        if not ents:
            _REFRESH_TABLES[key] = None
            return 0
        be = lib.tscd_pack_block_elems()
        jobs, bmap, sig = [], [], []
        for ji, (cache, k) in enumerate(ents):
            tag, packed, ref, shape4, cin_pad, mode = cache._c[k]
            w = ref()
            Cout, Cin, inner, KH, KW, rows_pad, kpad = _pack_dims(shape4, cin_pad, mode)
            assert tuple(packed.shape) == (rows_pad, kpad)
            jobs.append((w.data_ptr(), packed.data_ptr(), Cout | (Cin << 32), inner | (KH << 32), KW | (rows_pad << 32), kpad | (mode << 32),
                         int(w.dtype == torch.float32)))
            nb = ((Cout + be - 1) // be) * ((Cin + be - 1) // be)      # (cout, cin) tiles of edge `be`
            bmap.extend((ji, b) for b in range(nb))
            sig.append((w.data_ptr(), packed.data_ptr()))
        dev = params[0].device
        jt = torch.from_numpy(np.array(jobs, dtype=np.int64).reshape(-1, 7)).to(dev)
        bt = torch.from_numpy(np.array(bmap, dtype=np.int32).reshape(-1, 2)).to(dev)
        hit = _REFRESH_TABLES[key] = (tuple(sig), jt, bt, len(bmap), ents)
    if hit is None:
        return 0
    sig, jt, bt, nb, ents = hit
    live = []
    for cache, k in ents:
        e = cache._c.get(k)
        w = e[2]() if e is not None else None
        if w is None:
            _REFRESH_TABLES.clear()
            return 0
        live.append((w.data_ptr(), e[1].data_ptr()))
    if tuple(live) != sig:                       # a parameter or a packed copy moved: rebuild next time, lazy path now
        _REFRESH_TABLES.clear()
        return 0
    call("tscd_pack_conv_weights_batched", jt.data_ptr(), bt.data_ptr(), nb, _stream())
    for cache, k in ents:
        e = cache._c[k]
        cache._c[k] = (cache._tag(e[2]()),) + e[1:]
    return len(ents)


WEIGHTS = WeightCache()


# ----------------------------------------------------------------------------------------------------
# conv / BN / elementwise
# ----------------------------------------------------------------------------------------------------
# bench.py sets this to a list to time every conv launch with device events on the launch stream
CONV_PROFILE = None
# ... and this one to a dict to time the other kernel families of the step (bench.py: `roofline_secondary`): name -> [(e0, e1, flop, bytes)]
KERNEL_PROFILE = None


def _profiled(name, work):
    """Brackets a wrapper's launches with device events on the launch stream while KERNEL_PROFILE is a dict; ``work(result, *args,
    **kwargs) -> (algorithmic FLOP, algorithmic bytes)`` of the call."""
    def deco(f):
        def g(*a, **k):
            prof = KERNEL_PROFILE
            if prof is None:
                return f(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = f(*a, **k)
            e1.record()
            prof.setdefault(name, []).append((e0, e1, tuple(work(r, *a, **k))))
            return r
        g.__name__, g.__doc__ = f.__name__, f.__doc__
        return g
    return deco


def _mca_work(N, N_L, W, need_reg, backward):
    """(FLOP, bytes) of one MCA call; the FLOP figure needs the per-frame proposal counts, which the caller of the profile knows: the
    entry carries the sizes (negative FLOP = "see bytes field"): bench.py turns (N, N_L, W, need_reg, backward) into SURVEY 8d's numbers."""
    return (N, N_L, W, int(need_reg), int(backward))


_F32_CACHE = {}     # id(parameter) -> (weakref, version, data_ptr, fp32 copy)


def f32(t):
    """fp32 view of a small parameter (bias, norm weight) for kernels that take fp32 vectors.  For an nn.Parameter that autograd
    does not have to see, the converted copy is kept until the parameter is written (a .half() evaluation model paid one
    conversion kernel per call and layer otherwise)."""
    if t is None or t.dtype == torch.float32:
        return t
    if not isinstance(t, torch.nn.Parameter) or (t.requires_grad and torch.is_grad_enabled()):
        return t.float()
    ent = _F32_CACHE.get(id(t))
    if ent is not None and ent[0]() is t and ent[1] == t._version and ent[2] == t.data_ptr():
        return ent[3]
    if len(_F32_CACHE) > 8192:
        _F32_CACHE.clear()
    v = t.detach().float()
    _F32_CACHE[id(t)] = (weakref.ref(t), t._version, t.data_ptr(), v)
    return v


def conv_out_hw(H, W, ksize, stride):
    pad = (ksize - 1) // 2
    return (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1


def conv2d(x, wpk, cout, ksize, stride=1, epilogue=EPI_RAW, bias=None, scale_shift=None, out=None, want_stats=False, groups=1,
           variant=-1, korder=0, tile_m=0):
    """Implicit-GEMM conv.  x [B,H,W,groups*Cin] fp16 NHWC(-slice).  Returns (y [B,Ho,Wo,groups*cout], bn_partial or None).
    ``groups`` > 1: that many convs of Cin -> cout channels in one launch, group g on channels [g*Cin, (g+1)*Cin) of x with the g-th
    packed matrix of ``wpk`` ([groups * rows_pad, Kpad]) into channels [g*cout, (g+1)*cout) of y.  ``variant`` / ``korder`` / ``tile_m``:
    per-call main-loop plan of tscd_conv2d_nhwc_f16_ex (tests and A/B timing; the product passes the automatic values)."""
    require_gpu(x, "conv input")
    B, H, W, Cx = x.shape
    assert Cx % groups == 0
    Cin = Cx // groups
    ldx = nhwc_ld(x)
    Ho, Wo = conv_out_hw(H, W, ksize, stride)
    if out is None:
        out = empty_nhwc(B, Ho, Wo, groups * cout, x.device)
    assert tuple(out.shape) == (B, Ho, Wo, groups * cout), (out.shape, (B, Ho, Wo, groups * cout))
    ldy = nhwc_ld(out)
    kpad = (ksize * ksize * Cin + 31) // 32 * 32
    rows_g = wpk.shape[0] // groups
    assert wpk.shape[1] == kpad and rows_g >= cout and (groups == 1 or rows_g % lib.tscd_conv_tile_n(cout) == 0), (wpk.shape, kpad, cout, groups)
    partial = None
    plain = groups == 1 and variant == -1 and korder == 0 and tile_m == 0
    if want_stats:
        # one partial-sum entry per M-tile of the main loop the launch takes
        mt = lib.tscd_conv_stat_tiles(B * Ho * Wo, Wo, Cin, cout, ksize, stride) if plain else \
            lib.tscd_conv_stat_tiles_ex(B * Ho * Wo, Wo, Cin, cout, ksize, stride, groups, variant, tile_m)
        partial = temp((2, wpk.shape[0], mt), dtype=torch.float32, device=x.device)
    scale = shift = None
    if scale_shift is not None:
        scale, shift = scale_shift[0], scale_shift[1]
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if plain:
        call("tscd_conv2d_nhwc_f16", x.data_ptr(), wpk.data_ptr(), out.data_ptr(), _ptr(bias), _ptr(scale), _ptr(shift),
             _ptr(partial), B, H, W, Cin, cout, ksize, stride, ldx, ldy, epilogue, _stream())
    else:
        call("tscd_conv2d_nhwc_f16_ex", x.data_ptr(), wpk.data_ptr(), out.data_ptr(), _ptr(bias), _ptr(scale), _ptr(shift),
             _ptr(partial), B, H, W, Cin, cout, ksize, stride, ldx, ldy, epilogue, groups, variant, korder, tile_m, _stream())
    if prof is not None:
        e1.record()
        prof.append((lib.tscd_conv_tile_n(cout), Cin, 2.0 * B * Ho * Wo * groups * cout * ksize * ksize * Cin, e0, e1,
                     (B, H, W, Cin, cout, ksize, stride) + ((f"x{groups} groups",) if groups > 1 else ())))
    return out, partial


def bn_finalize(partial, count, gamma, beta, running_mean, running_var, eps=BN_EPS, momentum=BN_MOMENTUM):
    """-> (scale_shift [2,C], save_mean [C], save_invstd [C]); updates running stats in place when given."""
    C = gamma.shape[0]
    ss = temp((2, C), dtype=torch.float32, device=partial.device)
    mean = temp(C, dtype=torch.float32, device=partial.device)
    invstd = temp(C, dtype=torch.float32, device=partial.device)
    call("tscd_bn_finalize", partial.data_ptr(), partial.shape[2], partial.shape[1], C, int(count), gamma.data_ptr(),
         beta.data_ptr(), eps, momentum, _ptr(running_mean), _ptr(running_var), ss.data_ptr(), mean.data_ptr(),
         invstd.data_ptr(), _stream())
    if running_mean is not None:
        torch.autograd.graph.increment_version((running_mean, running_var))
    return ss, mean, invstd


def conv_bn_act_train(x, wpk, cout, ksize, stride, gamma, beta, running_mean, running_var, eps=BN_EPS, momentum=BN_MOMENTUM,
                      act=ACT_SILU, residual=None, out=None, keep_raw=False):
    """BaseConv in training mode in one host call (conv + batch statistics + apply): -> (out, raw, scale_shift [2,C], mean, invstd).
    keep_raw: raw conv output in its own buffer (the backward needs it); otherwise the apply pass runs in place on ``out``.
    ``out`` may BE ``residual`` (a bottleneck writing back over its own input): the raw output then takes a buffer of its own and
    the apply pass reads and writes each element of ``out`` in place."""
    if out is not None and residual is not None and out.data_ptr() == residual.data_ptr():
        keep_raw = True
    if CONV_PROFILE is not None:            # roofline instrumentation brackets the conv launch alone: separate calls
        raw, part = conv2d(x, wpk, cout, ksize, stride, EPI_RAW, out=None if keep_raw else out, want_stats=True)
        B, Ho, Wo, _ = raw.shape
        ss, mean, invstd = bn_finalize(part, B * Ho * Wo, gamma, beta, running_mean, running_var, eps, momentum)
        o = bn_act(raw, ss, act, residual=residual, out=(out if out is not None else empty_nhwc(B, Ho, Wo, cout, x.device)) if keep_raw else None)
        return o, raw, ss, mean, invstd
    B, H, W, Cin = x.shape
    Ho, Wo = conv_out_hw(H, W, ksize, stride)
    dev = x.device
    if out is None:
        out = empty_nhwc(B, Ho, Wo, cout, dev)
    raw = empty_nhwc(B, Ho, Wo, cout, dev) if keep_raw else out
    mt = lib.tscd_conv_stat_tiles(B * Ho * Wo, Wo, Cin, cout, ksize, stride)     # one partial-sum entry per M-tile of the variant taken
    partial = temp((2, wpk.shape[0], mt), dtype=torch.float32, device=dev)
    small = temp((4, cout), dtype=torch.float32, device=dev)
    call("tscd_conv_bn_act_train_f16", x.data_ptr(), wpk.data_ptr(), raw.data_ptr(), nhwc_ld(raw), partial.data_ptr(), wpk.shape[0],
         B, H, W, Cin, cout, ksize, stride, nhwc_ld(x), gamma.data_ptr(), beta.data_ptr(), eps, momentum, _ptr(running_mean),
         _ptr(running_var), small.data_ptr(), out.data_ptr(), nhwc_ld(out), _ptr(residual),
         nhwc_ld(residual) if residual is not None else 0, act, _stream())
    if running_mean is not None:
        # the kernel wrote the running statistics through raw pointers: bump their version counters so that what is keyed on them
        # (the folded evaluation scale / shift of BaseConv, evaluation graphs) notices the next time the module runs in eval mode
        torch.autograd.graph.increment_version((running_mean, running_var))
    return out, raw, small[:2], small[2], small[3]


def _ptr_array(tensors):
    import ctypes as C
    return (C.c_void_p * len(tensors))(*[0 if t is None else t.data_ptr() for t in tensors])


def conv_bn_act_train_multi(x, wpk, seg_c, bns, ksize, groups, eps=BN_EPS, momentum=BN_MOMENTUM, act=ACT_SILU, keep_raw=False):
    """Several BaseConvs (3x3 / 1x1, stride 1) in training mode through ONE conv launch + ONE statistics launch + ONE apply launch.
    ``bns``: list of nseg (gamma, beta, running_mean, running_var) fp32 tuples, one BatchNorm of seg_c channels each.
    groups == 1: the convs share their input x [B,H,W,Cin]; wpk = their packed matrices as one [nseg*seg_c (padded), Kpad] matrix.
    groups == nseg: conv g reads channels [g*Cin, ...) of x; wpk = [groups * rows_pad, Kpad].
    -> (out [B,H,W,nseg*seg_c], raw (== out unless keep_raw), scale_shift [2,Ctot], mean [Ctot], invstd [Ctot])."""
    B, H, W, Cx = x.shape
    nseg = len(bns)
    assert groups in (1, nseg) and Cx % groups == 0
    Cin = Cx // groups
    Ctot = nseg * seg_c
    dev = x.device
    out = empty_nhwc(B, H, W, Ctot, dev)
    raw = empty_nhwc(B, H, W, Ctot, dev) if keep_raw else out
    cout_launch = Ctot if groups == 1 else seg_c
    if CONV_PROFILE is not None:            # roofline instrumentation brackets the conv launch alone: separate calls
        raw_, part = conv2d(x, wpk, cout_launch, ksize, 1, EPI_RAW, out=raw, want_stats=True, groups=groups)
        sss = []
        for i, (g, b, rm, rv) in enumerate(bns):      # per-module statistics from the module's rows of the partial table
            rows_g = wpk.shape[0] // groups
            r0 = i * seg_c if groups == 1 else i * rows_g
            sub = torch.stack((part[0, r0:r0 + seg_c], part[1, r0:r0 + seg_c])).contiguous()
            sss.append(bn_finalize(sub, B * H * W, g, b, rm, rv, eps, momentum))
        ss = torch.cat([t[0] for t in sss], 1).contiguous()
        mean, invstd = torch.cat([t[1] for t in sss]), torch.cat([t[2] for t in sss])
        bn_act(raw, ss, act, out=out if keep_raw else None)
        return out, raw, ss, mean, invstd
    mt = lib.tscd_conv_stat_tiles_ex(B * H * W, W, Cin, cout_launch, ksize, 1, groups, -1, 0)
    partial = temp((2, wpk.shape[0], mt), dtype=torch.float32, device=dev)
    small = temp((4, Ctot), dtype=torch.float32, device=dev)
    rms = [t[2] for t in bns]
    call("tscd_conv_bn_act_train_multi_f16", x.data_ptr(), wpk.data_ptr(), raw.data_ptr(), nhwc_ld(raw), partial.data_ptr(), B, H, W, Cin,
         seg_c, nseg, groups, ksize, nhwc_ld(x), _ptr_array([t[0] for t in bns]), _ptr_array([t[1] for t in bns]), eps, momentum,
         _ptr_array(rms) if rms[0] is not None else None, _ptr_array([t[3] for t in bns]) if rms[0] is not None else None,
         small.data_ptr(), out.data_ptr(), nhwc_ld(out), act, _stream())
    if rms[0] is not None:
        torch.autograd.graph.increment_version(tuple(rms) + tuple(t[3] for t in bns))
    return out, raw, small[:2], small[2], small[3]


def bn_fold(gamma, beta, running_mean, running_var, eps=BN_EPS):
    C = gamma.shape[0]
    ss = torch.empty((2, C), dtype=torch.float32, device=gamma.device)      # cached per BatchNorm state by the callers: persistent
    call("tscd_bn_fold", gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), eps, C,
         ss.data_ptr(), _stream())
    return ss


@_profiled("bn_act", lambda r, x, scale_shift, act=ACT_SILU, residual=None, out=None:
           (0.0, 2.0 * x.numel() * (2 + (residual is not None))))
def bn_act(x, scale_shift, act=ACT_SILU, residual=None, out=None):
    """out = act(x*scale+shift) (+ residual); in place on x when ``out`` is None."""
    B, H, W, C = x.shape
    out = x if out is None else out
    call("tscd_bn_act_nhwc_f16", x.data_ptr(), nhwc_ld(x), out.data_ptr(), nhwc_ld(out), scale_shift.data_ptr(),
         _ptr(residual), nhwc_ld(residual) if residual is not None else 0, B * H * W, C, act, _stream())
    return out


def focus_to_nhwc16(img):
    """NCHW image (fp32/fp16) -> [B,H/2,W/2,16] fp16 (12 Focus channels + 4 zero pad)."""
    require_gpu(img, "image")
    img = img.contiguous()
    B, C, H, W = img.shape
    assert C == 3 and img.dtype in (torch.float32, torch.float16)
    out = empty_nhwc(B, H // 2, W // 2, 16, img.device)
    call("tscd_focus_nchw_to_nhwc16", img.data_ptr(), int(img.dtype == torch.float32), out.data_ptr(), B, H, W, _stream())
    return out


def copy_slice(src, dst, upsample=False):
    B, H, W, C = dst.shape
    call("tscd_copy_slice_nhwc_f16", src.data_ptr(), nhwc_ld(src), dst.data_ptr(), nhwc_ld(dst), B, H, W, C, int(upsample),
         _stream())
    return dst


def zero_insert2(dy):
    """dY [B,Ho,Wo,C] -> [B,2Ho,2Wo,C] with dY at the even positions (stride-2 conv backward, elementwise.hip)."""
    B, Ho, Wo, C = dy.shape
    out = empty_nhwc(B, 2 * Ho, 2 * Wo, C, dy.device)
    call("tscd_zero_insert2_nhwc_f16", dy.data_ptr(), nhwc_ld(dy), out.data_ptr(), nhwc_ld(out), B, Ho, Wo, C, _stream())
    return out


def maxpool5_bwd(x, dy):
    B, H, W, C = x.shape
    dx = empty_nhwc(B, H, W, C, x.device)
    call("tscd_maxpool5_bwd_nhwc_f16", x.data_ptr(), nhwc_ld(x), dy.data_ptr(), nhwc_ld(dy), dx.data_ptr(), nhwc_ld(dx), B, H, W, C, _stream())
    return dx


def upsample2_bwd(dy):
    B, H2, W2, C = dy.shape
    dx = empty_nhwc(B, H2 // 2, W2 // 2, C, dy.device)
    call("tscd_upsample2_bwd_nhwc_f16", dy.data_ptr(), nhwc_ld(dy), dx.data_ptr(), nhwc_ld(dx), B, H2 // 2, W2 // 2, C, _stream())
    return dx


def maxpool(src, dst, ks=5):
    B, H, W, C = src.shape
    call("tscd_maxpool_nhwc_f16", src.data_ptr(), nhwc_ld(src), dst.data_ptr(), nhwc_ld(dst), B, H, W, C, ks, _stream())
    return dst


def haar_hf(x):
    B, H, W, C = x.shape
    hf = empty_nhwc(B, H // 2, W // 2, 3 * C, x.device)
    call("tscd_haar_hf_nhwc_f16", x.data_ptr(), nhwc_ld(x), hf.data_ptr(), B, H, W, C, _stream())
    return hf


def haar_idwt_mul(hf, content, out=None):
    B, H, W, C = content.shape
    assert hf.is_contiguous() and tuple(hf.shape) == (B, H // 2, W // 2, 3 * C)
    out = empty_nhwc(B, H, W, C, content.device) if out is None else out
    call("tscd_haar_idwt_mul_nhwc_f16", hf.data_ptr(), content.data_ptr(), nhwc_ld(content), out.data_ptr(), nhwc_ld(out),
         B, H, W, C, _stream())
    return out


def haar_idwt(hf):
    """Inverse Haar transform with LL = 0: [B,h,w,3C] -> [B,2h,2w,C]; also the backward (adjoint) of haar_hf."""
    assert hf.is_contiguous() and hf.dtype == torch.float16
    B, h, w, C3 = hf.shape
    C = C3 // 3
    out = empty_nhwc(B, 2 * h, 2 * w, C, hf.device)
    call("tscd_haar_idwt_mul_nhwc_f16", hf.data_ptr(), 0, C, out.data_ptr(), C, B, 2 * h, 2 * w, C, _stream())
    return out


def haar_idwt_mul_bwd(d, hf, content):
    """-> (dhf [B,h,w,3C], dcontent [B,H,W,C]) for out = content * idwt(hf)."""
    B, H, W, C = content.shape
    assert d.is_contiguous() and hf.is_contiguous() and d.dtype == torch.float16 and tuple(d.shape) == (B, H, W, C)
    dcontent = empty_nhwc(B, H, W, C, d.device)
    dhf = torch.empty_like(hf)
    call("tscd_haar_idwt_mul_bwd_nhwc_f16", d.data_ptr(), hf.data_ptr(), content.data_ptr(), nhwc_ld(content), dcontent.data_ptr(),
         dhf.data_ptr(), B, H, W, C, _stream())
    return dhf, dcontent


# ----------------------------------------------------------------------------------------------------
# GEMM (nn.Linear) on the conv kernel
# ----------------------------------------------------------------------------------------------------
EPI_COST = 5


def pack_linear_weight(w):
    """nn.Linear weight [out, in] -> packed [rows_pad, Kpad] (a 1x1 conv)."""
    return pack_conv_weight(w.reshape(w.shape[0], w.shape[1], 1, 1))


class _LinearCache(WeightCache):
    def get(self, w, mode=0):
        shape4 = (w.shape[0], w.shape[1], 1, 1)
        if mode == 0:
            return self._lookup((id(w), "lin"), w, lambda: pack_linear_weight(w), shape4, None, 0)
        return self._lookup((id(w), "lin_dgrad"), w, lambda: pack_conv_weight(w.detach().reshape(shape4), mode=1), shape4, None, 1)


LINEAR_WEIGHTS = _LinearCache()


def linear(x, weight, bias=None, out=None, wpk=None):
    """x [n, K] fp16 (row stride may exceed K) @ weight[out,K]^T (+ bias) -> [n, out] fp16 via the MFMA conv kernel (called
    directly as a 1 x n image: this wrapper sits on the launch-bound stretch of the step, so it skips conv2d's generic layout
    checks)."""
    n, K = x.shape
    nout = weight.shape[0] if weight is not None else out.shape[1]
    wpk = LINEAR_WEIGHTS.get(weight) if wpk is None else wpk
    if out is None:
        out = temp((n, nout), dtype=torch.float16, device=x.device)
    if n == 0:
        return out
    if not (x.is_cuda and x.dtype == torch.float16 and x.stride(1) == 1 and out.stride(1) == 1 and K % 8 == 0
            and wpk.shape[1] == (K + 31) // 32 * 32 and wpk.shape[0] >= nout) or CONV_PROFILE is not None:
        x4, o4 = x.unsqueeze(0).unsqueeze(0), out.unsqueeze(0).unsqueeze(0)       # generic path (checks, profiling)
        if bias is not None:
            conv2d(x4, wpk, nout, 1, 1, EPI_BIAS, bias=bias.float(), out=o4)
        else:
            conv2d(x4, wpk, nout, 1, 1, EPI_RAW, out=o4)
        return out
    b32 = None
    if bias is not None:
        b32 = bias if bias.dtype == torch.float32 else bias.float()
    call("tscd_conv2d_nhwc_f16", x.data_ptr(), wpk.data_ptr(), out.data_ptr(), 0 if b32 is None else b32.data_ptr(), 0, 0, 0,
         1, 1, n, K, nout, 1, 1, x.stride(0), out.stride(0), EPI_RAW if b32 is None else EPI_BIAS, _stream())
    return out


import ctypes as _C


class _ConvJobC(_C.Structure):
    _fields_ = [("x", _C.c_void_p), ("wpk", _C.c_void_p), ("y", _C.c_void_p), ("bias", _C.c_void_p), ("B", _C.c_int), ("H", _C.c_int),
                ("W", _C.c_int), ("Cin", _C.c_int), ("Cout", _C.c_int), ("ldx", _C.c_int), ("ldy", _C.c_int), ("epilogue", _C.c_int)]


LINEAR_JOBS = _os.environ.get("TSCD_LINEAR_JOBS", "1") != "0"


def linear_jobs(jobs):
    """Several INDEPENDENT ``linear`` calls -- jobs = [(x, weight, bias or None, out or None)] -- as ONE launch where every job is of the
    128-wide LDS-DMA form (K % 32 == 0, more than 64 outputs); otherwise (or with TSCD_LINEAR_JOBS=0) one launch each.  -> list of outputs,
    the separate launches' bits."""
    outs, recs, keep = [], [], []
    ok = LINEAR_JOBS and CONV_PROFILE is None and 2 <= len(jobs) <= lib.tscd_conv2d_jobs_max()
    for (x, weight, bias, out) in jobs:
        n, K = x.shape
        nout = weight.shape[0]
        if out is None:
            out = temp((n, nout), dtype=torch.float16, device=x.device)
        outs.append(out)
        if not ok:
            continue
        wpk = LINEAR_WEIGHTS.get(weight)
        if not (n > 0 and x.is_cuda and x.dtype == torch.float16 and x.stride(1) == 1 and out.stride(1) == 1 and wpk.shape[1] == K
                and wpk.shape[0] >= nout and lib.tscd_conv2d_jobs_supported(K, nout)):
            ok = False
            continue
        b32 = None if bias is None else (bias if bias.dtype == torch.float32 else bias.float())
        keep.append((wpk, b32))
        recs.append(_ConvJobC(x.data_ptr(), wpk.data_ptr(), out.data_ptr(), None if b32 is None else b32.data_ptr(), 1, 1, n, K, nout,
                              x.stride(0), out.stride(0), EPI_RAW if b32 is None else EPI_BIAS))
    if ok:
        arr = (_ConvJobC * len(recs))(*recs)
        call("tscd_conv2d_jobs_nhwc_f16", _C.cast(arr, _C.c_void_p), len(recs), _stream())
        return outs
    for (x, weight, bias, _), out in zip(jobs, outs):
        linear(x, weight, bias, out=out)
    return outs


# ----------------------------------------------------------------------------------------------------
# proposal stage
# ----------------------------------------------------------------------------------------------------
_H2D_CACHE = {}


def h2d(data, dtype, device):
    """Small host table -> device WITHOUT a stream-blocking pageable copy: stage through pinned memory and copy
    asynchronously on the current stream.  Every index / tile / offset table of the proposal stage goes through here.
    Tables up to 4 KB are cached by content (the same tile / offset tables recur clip after clip); callers treat the
    result as read-only."""
    t = torch.as_tensor(data, dtype=dtype)
    if device is None or torch.device(device).type != "cuda":
        return t
    small = t.numel() * t.element_size() <= 4096
    if small:
        key = (dtype, str(device), tuple(t.shape), t.numpy().tobytes())
        hit = _H2D_CACHE.get(key)
        if hit is not None:
            return hit
    out = t.pin_memory().to(device, non_blocking=True)
    if small:
        if len(_H2D_CACHE) > 1024:
            _H2D_CACHE.clear()
        _H2D_CACHE[key] = out
    return out


def _i32(vals, device):
    return h2d(vals, torch.int32, device)


class LevelGeometry:
    """Per-clip FPN level table (hw, w, stride) as host int arrays for the C-ABI (no device copy needed)."""

    def __init__(self, hw_list, strides):
        import ctypes as C
        self.hw = [h * w for h, w in hw_list]
        self.w = [w for _, w in hw_list]
        self.strides = list(strides)
        self.A = sum(self.hw)
        self.c_hw = (C.c_int * 3)(*self.hw)
        self.c_w = (C.c_int * 3)(*self.w)
        self.c_stride = (C.c_int * 3)(*self.strides)


def decode_score(raw_levels, geo, num_classes, want_outputs):
    """raw_levels: 3 contiguous tensors [B, h, w, 5+NC] fp16.  -> outputs [B,A,5+NC] or None, det [B,A,7+NC], score [B,A]."""
    B = raw_levels[0].shape[0]
    dev = raw_levels[0].device
    nch = 5 + num_classes
    for r in raw_levels:
        assert r.is_contiguous() and r.shape[-1] == nch and r.dtype == torch.float16
    outputs = temp((B, geo.A, nch), dtype=torch.float16, device=dev) if want_outputs else None
    det = temp((B, geo.A, 7 + num_classes), dtype=torch.float16, device=dev)
    score = temp((B, geo.A), dtype=torch.float16, device=dev)
    call("tscd_decode_score", raw_levels[0].data_ptr(), raw_levels[1].data_ptr(), raw_levels[2].data_ptr(), geo.c_hw, geo.c_w,
         geo.c_stride, B, num_classes, _ptr(outputs), det.data_ptr(), score.data_ptr(), _stream())
    return outputs, det, score


def select_proposals(score, conf_thresh, min_keep, max_keep):
    """-> (idx [B, cap] int32 ascending anchor ids, count [B] int32)."""
    B, A = score.shape
    cap = max_keep if max_keep > 0 else A
    cap = max(cap, min_keep, 1)
    idx = temp((B, cap), dtype=torch.int32, device=score.device)
    cnt = temp((B,), dtype=torch.int32, device=score.device)
    call("tscd_select_proposals", score.data_ptr(), B, A, float(conf_thresh), int(min_keep), int(max_keep), cap, idx.data_ptr(),
         cnt.data_ptr(), _stream())
    return idx, cnt


def _map_table(maps_per_feature):
    import ctypes as C
    ptrs = []
    for f in range(3):
        for k in range(3):
            ptrs.append(maps_per_feature[f][k].data_ptr() if f < len(maps_per_feature) else 0)
    return (C.c_void_p * 9)(*ptrs)


def _maps_ld(maps_per_feature, W):
    """Common pixel stride of the feature maps ([B,h,w,W] NHWC tensors or W-channel slices of wider buffers)."""
    ld = None
    for f in maps_per_feature:
        for m in f:
            l = nhwc_ld(m)
            assert m.shape[-1] == W and (ld is None or l == ld), "feature maps of one gather must share their pixel stride"
            ld = l
    return ld


def gather_rows(maps_per_feature, geo, sel_idx, frame_off, N, W):
    """maps_per_feature: list (<=3) of per-level lists of [B,h,w,W] maps (NHWC or channel slices of one pixel stride).
    -> list of [N, W] tensors."""
    nm = len(maps_per_feature)
    dev = sel_idx.device
    outs = [temp((N, W), dtype=torch.float16, device=dev) for _ in range(nm)]
    ld = _maps_ld(maps_per_feature, W)
    B = maps_per_feature[0][0].shape[0]
    call("tscd_gather_rows", _map_table(maps_per_feature), geo.c_hw, sel_idx.data_ptr(), sel_idx.shape[1], frame_off.data_ptr(),
         B, N, W, ld, nm, _ptr(outs[0]), _ptr(outs[1] if nm > 1 else None), _ptr(outs[2] if nm > 2 else None), 0, _stream())
    return outs


def scatter_rows(grads, grad_maps_per_feature, geo, sel_idx, frame_off, N, W, add=None):
    """Adjoint of gather_rows: writes grads[f][n] into the (pre-zeroed) grad maps; ``add[f]``: feature f's rows are ADDED to what its
    maps already hold (another consumer's gradient of the same feature maps) -- no zero fill and no tensor add for that feature."""
    nm = len(grads)
    B = grad_maps_per_feature[0][0].shape[0]
    gs = [g.contiguous() for g in grads]
    ld = _maps_ld(grad_maps_per_feature, W)
    mode = 1
    for f in range(nm):
        if add is not None and add[f]:
            mode |= 2 << f
    call("tscd_gather_rows", _map_table(grad_maps_per_feature), geo.c_hw, sel_idx.data_ptr(), sel_idx.shape[1],
         frame_off.data_ptr(), B, N, W, ld, nm, _ptr(gs[0]), _ptr(gs[1] if nm > 1 else None), _ptr(gs[2] if nm > 2 else None), mode,
         _stream())


def gather_det(det, sel_idx, frame_off, N, rows=False):
    """-> pred_result rows [N,nd]; with ``rows`` also the (frame, anchor) of every row as two int32 vectors."""
    B, A, nd = det.shape
    out = temp((N, nd), dtype=torch.float16, device=det.device)
    if rows:
        rb = temp((2, N), dtype=torch.int32, device=det.device)
        call("tscd_gather_det_rows", det.data_ptr(), A, nd, sel_idx.data_ptr(), sel_idx.shape[1], frame_off.data_ptr(), B, N,
             out.data_ptr(), rb[0].data_ptr(), rb[1].data_ptr(), _stream())
        return out, rb[0], rb[1]
    call("tscd_gather_det", det.data_ptr(), A, nd, sel_idx.data_ptr(), sel_idx.shape[1], frame_off.data_ptr(), B, N,
         out.data_ptr(), _stream())
    return out


# ----------------------------------------------------------------------------------------------------
# MCA
# ----------------------------------------------------------------------------------------------------
def mca_tiles(preds_per_frame, lframe, device):
    """q-tile table: 32-row tiles that never straddle a frame.  -> int32 [ntiles,4] (row0,nrows,loc_start,loc_end)."""
    tiles, start = [], 0
    for i in range(lframe):
        n = preds_per_frame[i]
        for r in range(0, n, 32):
            tiles.append((start + r, min(32, n - r), start, start + n))
        start += n
    return h2d(tiles, torch.int32, device).reshape(-1, 4)


def mca_splits(ntiles):
    return max(1, min(16, (1024 + ntiles - 1) // max(ntiles, 1)))


@_profiled("mca_forward", lambda r, q_c, kv_c, q_r, kv_r, score, tiles, N, N_L, W, need_reg, *a, **k: _mca_work(N, N_L, W, need_reg, False))
def mca_attention(q_c, kv_c, q_r, kv_r, score, tiles, N, N_L, W, need_reg, scale, sim_thresh, conf_thresh, splits=None,
                  want_stats=False, keep_workspace=False, fp16_logits=False):
    """Fused global-to-local attention.  Returns (xcat_c [N_L,2W], cat3_c [N_L,3W], xcat_r, cat3_r, stats): first W columns
    of xcat = P@v, second W = own values; first W columns of cat3 = the sim_round2 aggregation (rest is filled by the
    caller's linear); stats (for backward) = softmax row sums [4,2,N_L] ++ sim_round2 denominators [2,N_L].
    ``keep_workspace``: also returns the buffer of normalised q / k / v rows for mca_backward(fwd_rows=...) -- a buffer of its own, so
    the (10x larger) scratch of the call is free again at once.  ``fp16_logits``: the reference-faithful rounding of the logits."""
    dev = q_c.device
    nt = tiles.shape[0]
    if splits is None:
        splits = mca_splits(nt)
    ws_bytes = lib.tscd_mca_workspace_bytes(N, N_L, W, splits)
    ws = temp((ws_bytes,), dtype=torch.uint8, device=dev)
    rows = temp((lib.tscd_mca_rows_bytes(N, N_L, W),), dtype=torch.uint8, device=dev) if keep_workspace else None
    mk = lambda c: temp((N_L, c), dtype=torch.float16, device=dev)
    xcat_c, cat3_c = mk(2 * W), mk(3 * W)
    xcat_r, cat3_r = (mk(2 * W), mk(3 * W)) if need_reg else (None, None)
    stats = temp((10 * N_L,), dtype=torch.float32, device=dev) if want_stats else None
    for t in (q_c, kv_c, q_r, kv_r, score):
        assert t.is_contiguous() and t.dtype == torch.float16
    call("tscd_mca_forward", q_c.data_ptr(), kv_c.data_ptr(), q_r.data_ptr(), kv_r.data_ptr(), score.data_ptr(),
         tiles.data_ptr(), nt, N, N_L, W, splits, int(need_reg), float(scale), float(sim_thresh), float(conf_thresh),
         ws.data_ptr(), ws_bytes, xcat_c.data_ptr(), _ptr(xcat_r), cat3_c.data_ptr(), _ptr(cat3_r), _ptr(stats), _ptr(rows),
         int(bool(fp16_logits)), _stream())
    if keep_workspace:
        return xcat_c, cat3_c, xcat_r, cat3_r, stats, rows
    return xcat_c, cat3_c, xcat_r, cat3_r, stats


@_profiled("mca_backward", lambda r, q_c, kv_c, q_r, kv_r, score, tiles, N, N_L, W, need_reg, *a, **k: _mca_work(N, N_L, W, need_reg, True))
def mca_backward(q_c, kv_c, q_r, kv_r, score, tiles, N, N_L, W, need_reg, scale, sim_thresh, conf_thresh, stats, dxcat_c, dxcat_r,
                 dcat3_c, dcat3_r, cat3_c, cat3_r, fwd_ws=None):
    """Gradients of the fused attention w.r.t. the projected rows: -> (dq_c [N_L,W], dkv_c [N,2W], dq_r, dkv_r, dscore [N]).
    dxcat_* [N_L,2W] contiguous fp16; dcat3_* / cat3_* [N_L,3W] (only their first W columns are read).  ``dxcat_c = dcat3_c = None``
    (with need_reg): no gradient reaches the class branch's outputs (agg_iou, whose class output is only used detached); the passes leave
    out what would only carry zeros -- the results are those of zero tensors, bit for bit."""
    dev = q_c.device
    nt = tiles.shape[0]
    S = mca_splits(nt)
    if (dxcat_c is None) != (dcat3_c is None) or (dxcat_c is None and not need_reg):
        raise ValueError("mca_backward: the class branch's gradients are both given or, with need_reg, both None")
    ws_bytes = lib.tscd_mca_backward_workspace_bytes(N, N_L, W, S, nt)
    ws = temp((ws_bytes,), dtype=torch.uint8, device=dev)
    dq_c = temp((N_L, W), dtype=torch.float16, device=dev)
    dq_r = temp((N_L, W), dtype=torch.float16, device=dev)
    dkv_c = temp((N, 2 * W), dtype=torch.float16, device=dev)
    dkv_r = temp((N, 2 * W), dtype=torch.float16, device=dev)
    dscore = temp((N,), dtype=torch.float32, device=dev)
    for t in ((dxcat_c, dcat3_c) if dxcat_c is not None else ()) + (cat3_c,) + ((dxcat_r, dcat3_r, cat3_r) if need_reg else ()):
        assert t.is_contiguous() and t.dtype == torch.float16
    call("tscd_mca_backward_fws", q_c.data_ptr(), kv_c.data_ptr(), q_r.data_ptr(), kv_r.data_ptr(), score.data_ptr(), tiles.data_ptr(), nt,
         N, N_L, W, S, int(need_reg), float(scale), float(sim_thresh), float(conf_thresh), stats.data_ptr(), _ptr(dxcat_c),
         _ptr(dxcat_r), _ptr(dcat3_c), _ptr(dcat3_r), 3 * W, cat3_c.data_ptr(), _ptr(cat3_r), 3 * W, ws.data_ptr(), ws_bytes,
         dq_c.data_ptr(), dkv_c.data_ptr(), dq_r.data_ptr(), dkv_r.data_ptr(), dscore.data_ptr(), _ptr(fwd_ws),
         0 if fwd_ws is None else fwd_ws.numel(), _stream())
    return dq_c, dkv_c, dq_r, dkv_r, dscore


def linear_dgrad(dy, weight):
    """dX = dY @ W for y = x @ W^T (W [out, in]): the same MFMA GEMM on W packed in data-gradient order (one pack kernel,
    no transpose copy); tiny prediction heads (out = 1 / 4 / NC) get their reduction axis padded to 8."""
    out_f, in_f = weight.shape
    n = dy.shape[0]
    kp = (out_f + 7) // 8 * 8
    if kp != out_f or not dy.is_contiguous():
        d2 = dy.new_zeros((n, kp))
        d2[:, :out_f] = dy
        dy = d2
    wpk = LINEAR_WEIGHTS.get(weight, mode=1)        # data-gradient packing, cached per parameter version like the forward copy
    return linear(dy, None, out=temp((n, in_f), dtype=torch.float16, device=dy.device), wpk=wpk)


# ----------------------------------------------------------------------------------------------------
# deferred weight gradients of the proposal stage's linears
# ----------------------------------------------------------------------------------------------------
# Nothing in a backward pass waits for a weight gradient.  The ~25 nn.Linear layers of the proposal stage (MCA projections, CAFM /
# TaskAligned attention projections, fc_reg_matcher, the prediction linears) have a few thousand rows each: launched one by one their
# dW GEMMs are launch- and latency-bound (tools/wgrad_census.py: 24 calls, 0.7 ms, < 90 TFLOP/s).  Inside a backward pass
# ``linear_bwd`` therefore only REGISTERS the weight gradient -- (x, dY) kept alive, dW allocated and handed to autograd -- and the
# pass's final callback computes all of them in ONE launch (tscd_linear_wgrad_batch_f16, conv_wgrad.hip) on the pass's stream.
# Conditions (as for param_grad_stream): the parameter has no gradient yet and this is its first contribution in this pass, so
# AccumulateGrad adopts the tensor without reading it; a gradient that was cloned instead (found at the flush) is repaired by a copy.
# Anything that must read the gradients before the pass ends (the flat-arena reducer of tscd_amd/ddp.py) calls
# ``flush_deferred_wgrads()`` first.  TSCD_DEFER_LINEAR_WGRAD=0 switches the deferral off.
import ctypes as _C

DEFER_LINEAR_WGRAD = _os.environ.get("TSCD_DEFER_LINEAR_WGRAD", "1") != "0"
FOREIGN_GRAD_HOOKS_OK = False      # set by tscd_amd.ddp.ArenaReducer (it flushes before it reads); torch's DDP reducer does not


class _WgradJobC(_C.Structure):
    _fields_ = [("x", _C.c_void_p), ("dy", _C.c_void_p), ("dw", _C.c_void_p), ("B", _C.c_int), ("H", _C.c_int), ("W", _C.c_int),
                ("cin", _C.c_int), ("cout", _C.c_int), ("ksize", _C.c_int), ("ldx", _C.c_int), ("lddy", _C.c_int), ("groups", _C.c_int),
                ("pad_", _C.c_int)]


class _ColsumJobC(_C.Structure):
    _fields_ = [("dy", _C.c_void_p), ("db", _C.c_void_p), ("M", _C.c_int), ("C", _C.c_int), ("lddy", _C.c_int), ("pad_", _C.c_int)]


class _WgradBatch:
    def __init__(self):
        self.jobs, self.keep, self.check, self.seen, self.task = [], [], [], set(), -1
        self.db_jobs = []

    def compute(self):
        """Launches the registered jobs on the current stream (the pass may go on afterwards: ``seen`` stays)."""
        jobs, self.jobs = self.jobs, []
        keep, self.keep = self.keep, []
        check, self.check = self.check, []
        db_jobs, self.db_jobs = self.db_jobs, []
        if not jobs and not db_jobs:
            return
        maxj = lib.tscd_wgrad_batch_max_jobs()
        for i in range(0, len(db_jobs), maxj):           # the bias gradients of the pass's linears: one launch
            part = db_jobs[i:i + maxj]
            arr = (_ColsumJobC * len(part))(*part)
            call("tscd_colsum_batch_f16", _C.cast(arr, _C.c_void_p), len(part), _stream())
        prof = KERNEL_PROFILE
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for i in range(0, len(jobs), maxj):
            part = jobs[i:i + maxj]
            arr = (_WgradJobC * len(part))(*part)
            arr_p = _C.cast(arr, _C.c_void_p)
            ws_bytes = lib.tscd_wgrad_batch_workspace_bytes(arr_p, len(part))
            ws = temp((ws_bytes,), dtype=torch.uint8, device=torch.device("cuda", _cur_device()))
            call("tscd_wgrad_batch_f16", arr_p, len(part), ws.data_ptr(), ws_bytes, _stream())
        if prof is not None:
            e1.record()
            fl = sum(2.0 * j.B * j.H * j.W * j.cin * j.cout * j.ksize * j.ksize * j.groups for j in jobs)
            by = sum(2.0 * j.B * j.H * j.W * (j.ldx + j.lddy) + 4.0 * j.groups * j.cout * j.cin * j.ksize * j.ksize for j in jobs)
            prof.setdefault("wgrad_batch", []).append((e0, e1, (fl, by)))
        for prm, dw in check:           # dw: the rows of the registered tensor that are this parameter's gradient
            g = prm.grad
            if g is not None and g.untyped_storage().data_ptr() != dw.untyped_storage().data_ptr():
                # AccumulateGrad cloned the (then unwritten) tensor instead of adopting it: overwrite the clone with the real gradient
                g.copy_(dw.reshape(g.shape))
        del keep

    def end_of_pass(self):
        self.compute()
        self.seen.clear()
        self.task = -1


_WB = {}


def _wgrad_batch(params):
    if not DEFER_LINEAR_WGRAD or PARAM_GRAD_STREAM:
        return None
    task = torch._C._current_graph_task_id()
    if task < 0:
        return None
    if not FOREIGN_GRAD_HOOKS_OK and torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1:
        return None
    dev = _cur_device()
    b = _WB.get(dev)
    if b is None:
        b = _WB[dev] = _WgradBatch()
    if b.task != task:
        if b.task >= 0:          # a pass that ended without its callback (an exception in backward)
            b.end_of_pass()
        torch.autograd.Variable._execution_engine.queue_callback(b.end_of_pass)
        b.task = task
    keys = [prm.data_ptr() for prm in params]
    if any(k in b.seen for k in keys):
        # a further contribution to a parameter of this pass: autograd ADDS it to the first one, which therefore has to exist now
        b.compute()
        return None
    if any(prm.grad is not None for prm in params):     # accumulation into a gradient of an earlier pass reads it: not deferred
        return None
    b.seen.update(keys)
    if len(b.jobs) >= lib.tscd_wgrad_batch_max_jobs():
        b.compute()             # a full launch is pending (a backbone's ~100 convs): go, and let go of the operands held so far
    return b


def flush_deferred_wgrads():
    """Computes every registered weight gradient now, on the current stream (no-op when none is pending)."""
    b = _WB.get(_cur_device())
    if b is not None:
        b.compute()


def linear_bwd(x, dy, weight, need_dx=True, need_dw=True, need_db=False, grad_params=None):
    """Backward of y = x W^T (+ b) in ONE foreign call -> (dx fp16 [n,in] | None, dw fp32 [out,in] | None, db fp32 [out] | None).
    x [n,in] fp16 (unit column stride), dy [n,out] fp16.  ``grad_params``: the parameters dw / db are the gradients OF (their
    dtype and shape) -- given, the call is split in two and the parameter half runs on the second stream (param_grad_stream)."""
    n, in_f = x.shape
    out_f = weight.shape[0]
    outp = (out_f + 7) // 8 * 8
    if outp != out_f or dy.stride(1) != 1 or dy.stride(0) % 8 != 0 or dy.data_ptr() % 16 != 0:
        d2 = dy.new_zeros((n, outp))
        d2[:, :out_f] = dy
        dy = d2
    dev = x.device
    dx = temp((n, in_f), dtype=torch.float16, device=dev) if need_dx else None
    dw = temp((outp, in_f), dtype=torch.float32, device=dev) if need_dw else None
    db = temp((outp,), dtype=torch.float32, device=dev) if need_db else None
    if n == 0:
        return (dx, None if dw is None else dw.zero_()[:out_f], None if db is None else db.zero_())
    ws_bytes = lib.tscd_linear_bwd_workspace_bytes(n, in_f, outp)
    ws = temp((ws_bytes,), dtype=torch.uint8, device=dev)
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    # the data-gradient copy of a PARAMETER is cached per version (and refreshed in the optimizer's batched re-pack): no pack launch here
    wdg = LINEAR_WEIGHTS.get(weight, mode=1) if need_dx and isinstance(weight, torch.nn.Parameter) and weight.is_contiguous() else None
    args = (x.data_ptr(), x.stride(0), dy.data_ptr(), dy.stride(0), w.data_ptr(), int(w.dtype == torch.float32), _ptr(wdg), n, in_f, out_f, outp)
    bias_param = grad_params[1] if grad_params is not None and len(grad_params) > 1 else None
    if grad_params is not None:
        grad_params = [p_ for p_ in grad_params if p_ is not None]
    batch = _wgrad_batch(grad_params) if (need_dw and grad_params is not None and x.stride(1) == 1 and x.stride(0) % 8 == 0
                                          and x.data_ptr() % 16 == 0 and in_f % 8 == 0) else None
    if batch is not None:
        # deferred: data gradient now, the weight-gradient GEMM with all the others at the end of the pass -- and the bias column sums
        # too when the caller named the bias parameter (grad_params = [weight, bias]); otherwise they are computed here
        db_deferred = need_db and bias_param is not None and bias_param.grad is None
        if need_dx or (need_db and not db_deferred):
            call("tscd_linear_bwd_packed_f16", *args, _ptr(dx), 0, 0 if db_deferred else _ptr(db), ws.data_ptr(), ws_bytes, _stream())
        if db_deferred:
            batch.db_jobs.append(_ColsumJobC(dy.data_ptr(), db.data_ptr(), n, outp, dy.stride(0), 0))
            batch.check.append((bias_param, db[:out_f]))
        batch.jobs.append(_WgradJobC(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 1, 1, n, in_f, outp, 1, x.stride(0), dy.stride(0), 1, 0))
        batch.keep.append((x, dy))
        batch.check.append((grad_params[0], dw[:out_f]))
        return dx, dw[:out_f], (None if db is None else db[:out_f])
    if (need_dw or need_db) and grad_params is not None:
        # data gradient on the pass's stream, parameter gradients on the second one (param_grad_stream)
        if need_dx:
            call("tscd_linear_bwd_packed_f16", *args, _ptr(dx), 0, 0, ws.data_ptr(), ws_bytes, _stream())
        with param_grad_stream(grad_params, x, dy) as side:
            if side:      # allocated under the second stream
                dw = temp((outp, in_f), dtype=torch.float32, device=dev) if need_dw else None
                db = temp((outp,), dtype=torch.float32, device=dev) if need_db else None
                ws = temp((ws_bytes,), dtype=torch.uint8, device=dev)
            call("tscd_linear_bwd_packed_f16", *args, 0, _ptr(dw), _ptr(db), ws.data_ptr(), ws_bytes, _stream())
    else:
        call("tscd_linear_bwd_packed_f16", *args, _ptr(dx), _ptr(dw), _ptr(db), ws.data_ptr(), ws_bytes, _stream())
    return dx, (None if dw is None else dw[:out_f]), (None if db is None else db[:out_f])


def bias_relu_bwd(dy, y=None):
    """Bias gradient of a biased conv / Linear, through the mask of its fused ReLU when ``y`` (the activation output) is given:
    -> (masked dy fp16 (dy itself without a mask), db fp32 [C]).  dy / y: [..., C] fp16 with unit channel stride, NHWC(-slice)."""
    C = dy.shape[-1]
    M = dy.numel() // C
    assert dy.dtype == torch.float16 and C % 8 == 0
    ldg = nhwc_ld(dy) if dy.dim() == 4 else dy.stride(0)
    db = temp((C,), dtype=torch.float32, device=dy.device)
    ws_bytes = lib.tscd_bias_relu_bwd_workspace_bytes(M, C)
    ws = temp((ws_bytes,), dtype=torch.uint8, device=dy.device)
    if y is None:
        call("tscd_bias_relu_bwd_f16", dy.data_ptr(), ldg, 0, 0, 0, 0, M, C, db.data_ptr(), ws.data_ptr(), ws_bytes, _stream())
        return dy, db
    gm = temp(dy.shape, dtype=torch.float16, device=dy.device)
    ldy = nhwc_ld(y) if y.dim() == 4 else y.stride(0)
    call("tscd_bias_relu_bwd_f16", dy.data_ptr(), ldg, y.data_ptr(), ldy, gm.data_ptr(), C, M, C, db.data_ptr(), ws.data_ptr(), ws_bytes,
         _stream())
    return gm, db


def linear_wgrad(x, dy):
    """dW [out, in] fp32 = dY^T @ X  (the pixel-axis GEMM of conv_wgrad.hip with 1x1 taps)."""
    n, cin = x.shape
    cout = dy.shape[1]
    cp = (cout + 7) // 8 * 8
    if cp != cout or not dy.is_contiguous():
        d2 = dy.new_zeros((n, cp))
        d2[:, :cout] = dy
        dy = d2
    dw = conv_wgrad(x.unsqueeze(0).unsqueeze(0), dy.unsqueeze(0).unsqueeze(0), cin, cp, 1)
    return dw[:cout, :, 0, 0]


# ----------------------------------------------------------------------------------------------------
# CAFM / TaskAligned pieces
# ----------------------------------------------------------------------------------------------------
_TILE_CACHE = {}


def attn_tiles(segments, device, rows=16):
    """segments: list of (row0, nrows, key0, key1) -> query tiles of `rows` rows (16: scalar kernels, 32: MFMA kernels).
    The (tiny) tile tables are cached per segment list: proposal counts repeat from clip to clip."""
    key = (tuple(segments), str(device), rows)
    hit = _TILE_CACHE.get(key)
    if hit is not None:
        return hit
    if len(_TILE_CACHE) > 512:
        _TILE_CACHE.clear()
    t = _attn_tiles(segments, device, rows)
    _TILE_CACHE[key] = t
    return t


def _attn_tiles(segments, device, rows):
    t = []
    for (r0, n, k0, k1) in segments:
        for r in range(0, n, rows):
            t.append((r0 + r, min(rows, n - r), k0, k1))
    return h2d(t, torch.int32, device).reshape(-1, 4)


def attn_key_tiles(segments, device):
    """The same segments seen from the keys: 32-key tiles (key row0, nkeys, query0, query1) for the key-stationary backward, or None
    when two segments share key rows (their contributions would have to be summed across tiles: the one-launch backward does that)."""
    key = (tuple(segments), str(device), "keys")
    if key in _TILE_CACHE:
        return _TILE_CACHE[key]
    spans = sorted((k0, k1) for (_, n, k0, k1) in segments if n > 0 and k1 > k0)
    disjoint = all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
    t = _attn_tiles([(k0, k1 - k0, r0, r0 + n) for (r0, n, k0, k1) in segments if n > 0 and k1 > k0], device, 32) if disjoint and spans else None
    _TILE_CACHE[key] = t
    return t


def cos_attn_uses_mfma(C, heads):
    return C % heads == 0 and bool(lib.tscd_cos_attn_mfma_supported(C // heads))


def cos_attn_mfma(q, k, v, heads, tiles32, full=False, out=None):
    """-> (out [n,C] fp16, lsum [n,heads] fp32).  ``full``: the tiles cover every query row (no zero fill needed); ``out``: a
    contiguous [n,C] fp16 destination (with ``full``)."""
    n, C = q.shape
    for t in (q, k, v):
        assert t.is_contiguous() and t.dtype == torch.float16 and t.shape[1] == C
    alloc = torch.empty if full else torch.zeros                          # rows outside every tile stay zero
    if out is None:
        out = alloc((n, C), dtype=torch.float16, device=q.device)
    else:
        assert full and out.is_contiguous() and out.dtype == torch.float16 and tuple(out.shape) == (n, C)
    lsum = alloc((n, heads), dtype=torch.float32, device=q.device)
    call("tscd_cos_attn_mfma_f16", q.data_ptr(), k.data_ptr(), v.data_ptr(), C, heads, tiles32.data_ptr(), tiles32.shape[0],
         out.data_ptr(), lsum.data_ptr(), _stream())
    return out, lsum


def cos_attn_mfma_bwd(q, k, v, d_out, out, lsum, heads, tiles32, full=False, outs=None, ktiles32=None, keys_full=False):
    """``outs``: (dq [n,C], dk [nk,C], dv [nk,C]) contiguous fp16 destinations (with ``full``).  ``ktiles32`` (attn_key_tiles): the
    two-launch backward without atomics; ``keys_full``: those tiles cover every key row (dk / dv need no zero fill)."""
    n, C = q.shape
    nk = k.shape[0]
    dev = q.device
    for t in (q, k, v, d_out, out):
        assert t.is_contiguous() and t.dtype == torch.float16 and t.shape[1] == C
    if outs is not None:
        dq, dk, dv = outs
        assert full and all(t.is_contiguous() and t.dtype == torch.float16 for t in outs) and tuple(dq.shape) == (n, C) \
            and tuple(dk.shape) == (nk, C) and tuple(dv.shape) == (nk, C)
        assert ktiles32 is None or keys_full
    else:
        dq = (torch.empty if full else torch.zeros)((n, C), dtype=torch.float16, device=dev)
        kalloc = torch.zeros if (ktiles32 is not None and not keys_full) else torch.empty
        dk, dv = (kalloc((nk, C), dtype=torch.float16, device=dev) for _ in range(2))
    if ktiles32 is not None:
        delta = temp((n * heads,), dtype=torch.float32, device=dev)
        call("tscd_cos_attn_mfma_bwd2_f16", q.data_ptr(), k.data_ptr(), v.data_ptr(), d_out.data_ptr(), out.data_ptr(), lsum.data_ptr(),
             n, nk, C, heads, tiles32.data_ptr(), tiles32.shape[0], ktiles32.data_ptr(), ktiles32.shape[0], delta.data_ptr(),
             dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), _stream())
        return dq, dk, dv
    ws = temp((2 * nk * C,), dtype=torch.float32, device=dev)
    call("tscd_cos_attn_mfma_bwd_f16", q.data_ptr(), k.data_ptr(), v.data_ptr(), d_out.data_ptr(), out.data_ptr(), lsum.data_ptr(),
         n, nk, C, heads, tiles32.data_ptr(), tiles32.shape[0], ws.data_ptr(), ws.numel() * 4, dq.data_ptr(), _stream())
    call("tscd_cos_attn_bwd_finish_f16", k.data_ptr(), ws.data_ptr(), nk, C, heads, dk.data_ptr(), dv.data_ptr(), _stream())
    return dq, dk, dv


def cos_attn(q, k, v, heads, tiles):
    n, C = q.shape
    out = temp((n, C), dtype=torch.float16, device=q.device)
    for t in (q, k, v):
        assert t.is_contiguous() and t.dtype == torch.float16 and t.shape[1] == C
    call("tscd_cos_attn_f16", q.data_ptr(), k.data_ptr(), v.data_ptr(), C, heads, tiles.data_ptr(), tiles.shape[0],
         out.data_ptr(), _stream())
    return out


def se_fuse(reg, edge, w0, w2, pos):
    n, C = reg.shape
    out = temp((n, C), dtype=torch.float16, device=reg.device)
    assert reg.stride(1) == 1 and edge.stride(1) == 1
    call("tscd_se_fuse_f16", reg.data_ptr(), reg.stride(0), edge.data_ptr(), edge.stride(0), w0.data_ptr(), w2.data_ptr(),
         _ptr(pos), n, C, out.data_ptr(), _stream())
    return out


def add_layernorm(a, b, gamma, beta, eps=1e-5):
    n, C = a.shape
    out = temp((n, C), dtype=torch.float16, device=a.device)
    assert a.is_contiguous() and (b is None or b.is_contiguous())
    call("tscd_add_layernorm_f16", a.data_ptr(), _ptr(b), gamma.data_ptr(), beta.data_ptr(), n, C, eps, out.data_ptr(), _stream())
    return out


def cos_attn_bwd(q, k, v, d_out, heads, tiles):
    """-> (dq, dk, dv) fp16 [n,C]: backward of cos_attn w.r.t. the raw (pre-normalisation) q / k and v."""
    n, C = q.shape
    nk = k.shape[0]
    dev = q.device
    for t in (q, k, v, d_out):
        assert t.is_contiguous() and t.dtype == torch.float16 and t.shape[1] == C
    assert d_out.shape == q.shape and v.shape == k.shape
    ws = temp((2 * nk * C,), dtype=torch.float32, device=dev)
    dq = ztemp((n, C), dtype=torch.float16, device=dev)        # rows outside every tile keep a zero gradient
    dk, dv = (temp((nk, C), dtype=torch.float16, device=dev) for _ in range(2))
    call("tscd_cos_attn_bwd_f16", q.data_ptr(), k.data_ptr(), v.data_ptr(), d_out.data_ptr(), n, nk, C, heads, tiles.data_ptr(),
         tiles.shape[0], ws.data_ptr(), ws.numel() * 4, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), _stream())
    return dq, dk, dv


def se_fuse_bwd(reg, edge, w0, w2, d_out, outs=None, acc=None):
    """-> (d_reg, d_edge fp16 [n,C], d_pos [C], d_w0 [32,2], d_w2 [2,32] fp32).  ``outs``: contiguous [n,C] fp16 destinations for
    d_reg / d_edge; ``acc``: a ZEROED (or to-be-added-to) fp32 [C + 128] block the parameter gradients are ADDED into (d_pos | d_w0 |
    d_w2): a chain of calls clears one arena once instead of one accumulator per call."""
    n, C = reg.shape
    dev = reg.device
    assert reg.stride(1) == 1 and edge.stride(1) == 1 and d_out.is_contiguous() and d_out.dtype == torch.float16
    if outs is not None:
        d_reg, d_edge = outs
        assert all(t.is_contiguous() and t.dtype == torch.float16 and tuple(t.shape) == (n, C) for t in outs)
    else:
        d_reg = temp((n, C), dtype=torch.float16, device=dev)
        d_edge = temp((n, C), dtype=torch.float16, device=dev)
    name = "tscd_se_fuse_bwd_f16"
    if acc is None:
        acc = temp((C + 128,), dtype=torch.float32, device=dev)      # zeroed by the entry point
    else:
        assert acc.dtype == torch.float32 and acc.is_contiguous() and acc.numel() == C + 128
        name = "tscd_se_fuse_bwd_acc_f16"
    call(name, reg.data_ptr(), reg.stride(0), edge.data_ptr(), edge.stride(0), w0.data_ptr(), w2.data_ptr(),
         d_out.data_ptr(), n, C, d_reg.data_ptr(), d_edge.data_ptr(), acc.data_ptr(), acc[C:].data_ptr(), acc[C + 64:].data_ptr(),
         _stream())
    return d_reg, d_edge, acc[:C], acc[C:C + 64].view(32, 2), acc[C + 64:].view(2, 32)


def add_layernorm_bwd(a, b, gamma, d_out, eps=1e-5, out=None, acc=None):
    """-> (dx fp16 [n,C] (= da = db), d_gamma [C], d_beta [C] fp32).  ``out``: contiguous [n,C] fp16 destination of dx; ``acc``: a zeroed
    (or to-be-added-to) fp32 [2,C] block d_gamma / d_beta are ADDED into."""
    n, C = a.shape
    dev = a.device
    assert a.is_contiguous() and (b is None or b.is_contiguous()) and d_out.is_contiguous() and d_out.dtype == torch.float16
    if out is not None:
        assert out.is_contiguous() and out.dtype == torch.float16 and tuple(out.shape) == (n, C)
        dx = out
    else:
        dx = temp((n, C), dtype=torch.float16, device=dev)
    name = "tscd_add_layernorm_bwd_f16"
    if acc is None:
        acc = temp((2, C), dtype=torch.float32, device=dev)          # zeroed by the entry point
    else:
        assert acc.dtype == torch.float32 and acc.is_contiguous() and tuple(acc.shape) == (2, C)
        name = "tscd_add_layernorm_bwd_acc_f16"
    call(name, a.data_ptr(), _ptr(b), gamma.data_ptr(), d_out.data_ptr(), n, C, eps, dx.data_ptr(),
         acc[0].data_ptr(), acc[1].data_ptr(), _stream())
    return dx, acc[0], acc[1]


# ---- the CAFM frame chain's row addressing inside se_fuse / add_layernorm (include/tscd_hip.h: TscdRowMap, TscdSeFuseEx, TscdLnEx) ----
import ctypes as _CT


class _RowMapC(_CT.Structure):
    _fields_ = [("idx", _CT.c_void_p), ("seg_rows", _CT.c_int), ("off0", _CT.c_int), ("off_stride", _CT.c_int)]


class _SeFuseExC(_CT.Structure):
    _fields_ = [("edge_map", _RowMapC), ("pos_seg_rows", _CT.c_int), ("pos_stride", _CT.c_int),
                ("add_reg", _CT.c_void_p * 3), ("add_reg_rows", _CT.c_int * 3),
                ("add_edge", _CT.c_void_p * 2), ("add_edge_rows", _CT.c_int * 2)]


class _LnExC(_CT.Structure):
    _fields_ = [("a_map", _RowMapC), ("dy_add", _CT.c_void_p), ("dx_mapped", _CT.c_void_p)]


def row_map(idx, seg_rows, off0, off_stride=0):
    """Row r (segment s = r // seg_rows, local l) -> s * seg_rows + idx[off0 + s * off_stride + l]; idx: int64 device table."""
    assert idx.dtype == torch.int64 and idx.is_contiguous() and seg_rows > 0
    return _RowMapC(idx.data_ptr(), seg_rows, off0, off_stride)


def _rows16(t, C):
    assert t.dtype == torch.float16 and t.is_contiguous() and t.shape[-1] == C
    return t.data_ptr()


def se_fuse_ex(reg, edge, w0, w2, pos, out, n, edge_map=None, pos_seg_rows=0, pos_stride=0):
    """se_fuse over ``n`` rows into ``out`` with the edge rows read through ``edge_map`` and one fp32 pos vector per ``pos_seg_rows`` rows
    (``pos`` then [segments, pos_stride])."""
    C = reg.shape[1]
    assert reg.stride(1) == 1 and edge.stride(1) == 1 and out.is_contiguous() and out.dtype == torch.float16 and out.shape[1] == C
    ex = _SeFuseExC()
    if edge_map is not None:
        ex.edge_map = edge_map
    ex.pos_seg_rows, ex.pos_stride = pos_seg_rows, pos_stride
    call("tscd_se_fuse_ex_f16", reg.data_ptr(), reg.stride(0), edge.data_ptr(), edge.stride(0), w0.data_ptr(), w2.data_ptr(),
         _ptr(pos), n, C, out.data_ptr(), _CT.addressof(ex), _stream())
    return out


def se_fuse_bwd_ex(reg, edge, w0, w2, d_out, d_reg, d_edge, n, dpos, dw0, dw2, edge_map=None, pos_seg_rows=0, pos_stride=0,
                   add_reg=(), add_edge=()):
    """se_fuse backward over ``n`` rows ADDING into the fp32 accumulators (dpos: one [C] vector per row segment, ``pos_stride`` floats
    apart).  ``edge_map``: the edge rows are read -- and their gradient written -- through the map; ``add_reg`` / ``add_edge``: up to 3 / 2
    (fp16 [rows, C] tensor, rows) addends summed into d_reg / d_edge before the one rounding (d_reg / d_edge may be one of them)."""
    C = reg.shape[1]
    assert reg.stride(1) == 1 and edge.stride(1) == 1 and d_out.is_contiguous() and d_out.dtype == torch.float16
    assert len(add_reg) <= 3 and len(add_edge) <= 2
    ex = _SeFuseExC()
    if edge_map is not None:
        ex.edge_map = edge_map
    ex.pos_seg_rows, ex.pos_stride = pos_seg_rows, pos_stride
    for k, (t, rows) in enumerate(add_reg):
        ex.add_reg[k], ex.add_reg_rows[k] = _rows16(t, C), rows
    for k, (t, rows) in enumerate(add_edge):
        ex.add_edge[k], ex.add_edge_rows[k] = _rows16(t, C), rows
    call("tscd_se_fuse_bwd_ex_f16", reg.data_ptr(), reg.stride(0), edge.data_ptr(), edge.stride(0), w0.data_ptr(), w2.data_ptr(),
         d_out.data_ptr(), n, C, _rows16(d_reg, C), _rows16(d_edge, C), dpos.data_ptr(), dw0.data_ptr(), dw2.data_ptr(),
         _CT.addressof(ex), _stream())


def add_layernorm_ex(a, b, gamma, beta, eps, out, n, a_map):
    """out[r] = LayerNorm(a[a_map(r)] (+ b[r])) over ``n`` rows."""
    C = out.shape[1]
    ex = _LnExC()
    ex.a_map = a_map
    call("tscd_add_layernorm_ex_f16", _rows16(a, C), None if b is None else _rows16(b, C), gamma.data_ptr(), beta.data_ptr(), n, C, eps,
         _rows16(out, C), _CT.addressof(ex), _stream())
    return out


def add_layernorm_bwd_ex(a, b, gamma, d_out, eps, dx, n, acc, a_map=None, dy_add=None, dx_mapped=None):
    """add_layernorm backward over ``n`` rows ADDING d_gamma / d_beta into ``acc`` [2, C]: ``a`` read through ``a_map``, dy = fp16(d_out +
    dy_add), dx written to ``dx`` (may be None) and to row a_map(r) of ``dx_mapped``."""
    C = d_out.shape[1]
    assert acc.dtype == torch.float32 and acc.is_contiguous() and tuple(acc.shape) == (2, C)
    ex = _LnExC()
    if a_map is not None:
        ex.a_map = a_map
    ex.dy_add = None if dy_add is None else _rows16(dy_add, C)
    ex.dx_mapped = None if dx_mapped is None else _rows16(dx_mapped, C)
    call("tscd_add_layernorm_bwd_ex_f16", _rows16(a, C), None if b is None else _rows16(b, C), gamma.data_ptr(), _rows16(d_out, C), n, C,
         eps, None if dx is None else _rows16(dx, C), acc[0].data_ptr(), acc[1].data_ptr(), _CT.addressof(ex), _stream())


COST_PAIRS_ONE_LAUNCH = _os.environ.get("TSCD_COST_PAIRS_ONE_LAUNCH", "1") != "0"


def match_costs_clip(reg, cls, pairs):
    """Every Hungarian cost matrix of a clip whose frames are matched against each other (no carried-over state): the [reg | cls]
    rows of ALL proposals are normalised once (two launches), each matrix is one GEMM over two row ranges of that buffer, written
    back to back into one flat fp16 buffer (what the single D2H copy takes).  reg / cls [n,C] fp16; pairs [(ref0, ref1, cur0, cur1)]
    row ranges.  -> (flat, [(nr, nc)]).  Same arithmetic as match_cost per pair."""
    n, C = reg.shape
    dev = reg.device
    assert reg.stride(1) == 1 and cls.stride(1) == 1 and (2 * C) % 32 == 0
    pad = 128                                   # the GEMM reads its B rows in tiles of up to 128: zero rows behind the last frame
    norm = temp((n + pad, 2 * C), dtype=torch.float16, device=dev)
    call("tscd_row_normalize_f16", reg.data_ptr(), reg.stride(0), n, n + pad, C, 1e-6, norm.data_ptr(), 2 * C, _stream())
    call("tscd_row_normalize_f16", cls.data_ptr(), cls.stride(0), n, n + pad, C, 1e-6, norm[:, C:].data_ptr(), 2 * C, _stream())
    shapes = [(r1 - r0, c1 - c0) for (r0, r1, c0, c1) in pairs]
    flat = temp((sum(a * b for a, b in shapes),), dtype=torch.float16, device=dev)
    if COST_PAIRS_ONE_LAUNCH and len(pairs) <= 16:
        # all matrices in ONE launch of 32 x 32 tiles (a launch per matrix was 4 x 23 us of 32 workgroups each, on the step's critical
        # path in front of the assignment chain); the same MFMA chain per output element: bit-identical costs
        import ctypes as C_
        tab = (C_.c_int * (4 * len(pairs)))(*[v for (r0, r1, c0, c1) in pairs for v in (r0, r1 - r0, c0, c1 - c0)])
        call("tscd_match_cost_pairs_f16", norm.data_ptr(), 2 * C, 2 * C, C_.cast(tab, C_.c_void_p), len(pairs), flat.data_ptr(), _stream())
        return flat, shapes
    o = 0
    for (r0, r1, c0, c1), (nr, nc) in zip(pairs, shapes):
        if nr * nc:
            conv2d(norm[r0:r1].unsqueeze(0).unsqueeze(0), norm[c0:], nc, 1, 1, EPI_COST, out=flat[o:o + nr * nc].view(1, 1, nr, nc))
        o += nr * nc
    return flat, shapes


def match_cost(ref_reg, cur_reg, ref_cls, cur_cls):
    """1 - (cos(reg_ref,reg_cur) + cos(cls_ref,cls_cur))/2 as ONE GEMM over [reg | cls] normalised rows -> [n_ref,n_cur] fp16."""
    nr, C = ref_reg.shape
    ncur = cur_reg.shape[0]
    dev = ref_reg.device
    bn = lib.tscd_conv_tile_n(ncur)
    rows_pad = (ncur + bn - 1) // bn * bn
    a = temp((nr, 2 * C), dtype=torch.float16, device=dev)
    b = temp((rows_pad, 2 * C), dtype=torch.float16, device=dev)
    for src, dst, n, npad in ((ref_reg, a[:, :C], nr, nr), (ref_cls, a[:, C:], nr, nr), (cur_reg, b[:, :C], ncur, rows_pad),
                              (cur_cls, b[:, C:], ncur, rows_pad)):
        assert src.stride(1) == 1
        call("tscd_row_normalize_f16", src.data_ptr(), src.stride(0), n, npad, C, 1e-6, dst.data_ptr(), 2 * C, _stream())
    cost = temp((nr, ncur), dtype=torch.float16, device=dev)
    assert (2 * C) % 32 == 0
    conv2d(a.unsqueeze(0).unsqueeze(0), b, ncur, 1, 1, EPI_COST, out=cost.unsqueeze(0).unsqueeze(0))
    return cost


# ----------------------------------------------------------------------------------------------------
# NMS / SimOTA
# ----------------------------------------------------------------------------------------------------
def batched_nms(det, frame_off, L, num_classes, conf_thresh, iou_thresh, cap_m, mode, cls_logit=None, obj_logit=None,
                reg_delta=None, max_rows=None, coord_trick=0):
    """-> (rows [L,capM,7] fp16, keep_idx [L,capM] int32, count [L] int32).  ``max_rows``: the largest number of det rows of one
    frame (default: all of them); ``coord_trick`` = n: torchvision's fp16 coordinate-offset arithmetic for frames with <= n candidates."""
    dev = det.device
    max_rows = int(max_rows) if max_rows else max(int(det.shape[0]), 1)
    ws_bytes = lib.tscd_nms_workspace_bytes(L, cap_m, num_classes, max_rows)
    ws = temp((ws_bytes,), dtype=torch.uint8, device=dev)
    out = temp((L, cap_m, 7), dtype=torch.float16, device=dev)
    idx = temp((L, cap_m), dtype=torch.int32, device=dev)
    cnt = ztemp((L,), dtype=torch.int32, device=dev)
    call("tscd_batched_nms", det.data_ptr(), det.shape[1], frame_off.data_ptr(), L, _ptr(cls_logit), _ptr(obj_logit),
         _ptr(reg_delta), num_classes, float(conf_thresh), float(iou_thresh), cap_m, max_rows, mode, int(coord_trick), ws.data_ptr(),
         ws_bytes, out.data_ptr(), idx.data_ptr(), cnt.data_ptr(), _stream())
    return out, idx, cnt


def simota_assign(outputs, labels, geo, num_classes, max_gt, topk=10, emu16=True, center_radius=4.5):
    """outputs [B,A,5+NC] fp16, labels [B,Lmax,5] (any float dtype).  -> fg_mask u8 [B,A], matched_gt i32 [B,A],
    pred_iou f32 [B,A], num_fg i32 [B], ngt i32 [B]."""
    B, A, _ = outputs.shape
    dev = outputs.device
    lab = labels.float().contiguous()
    G = min(max_gt, lab.shape[1])
    ws_bytes = lib.tscd_simota_workspace_bytes(B, A, G, topk)
    ws = temp((ws_bytes,), dtype=torch.uint8, device=dev)
    fg = temp((B, A), dtype=torch.uint8, device=dev)
    mg = temp((B, A), dtype=torch.int32, device=dev)
    pi = temp((B, A), dtype=torch.float32, device=dev)
    nfg = temp((B,), dtype=torch.int32, device=dev)
    ngt = temp((B,), dtype=torch.int32, device=dev)
    assert outputs.is_contiguous() and outputs.dtype == torch.float16
    call("tscd_simota_assign_r", outputs.data_ptr(), lab.data_ptr(), B, A, num_classes, lab.shape[1], G, geo.c_hw, geo.c_w,
         geo.c_stride, topk, int(emu16), float(center_radius), ws.data_ptr(), ws_bytes, fg.data_ptr(), mg.data_ptr(), pi.data_ptr(),
         nfg.data_ptr(), ngt.data_ptr(), _stream())
    return fg, mg, pi, nfg, ngt


# ----------------------------------------------------------------------------------------------------
# fused losses (loss.hip)
# ----------------------------------------------------------------------------------------------------
def loss_dense(raw_all, outputs, fg_mask, matched_gt, pred_iou, labels, num_classes, want_grad=True, geo=None):
    """Base-detector losses over [B,A,5+NC]: -> (sums fp32 [3] = IoU-loss / objectness-BCE / class-BCE sums (un-normalised),
    d_logit fp16 [B,A,1+NC] = d(obj_sum + cls_sum) / d raw obj|cls logits, or None)."""
    B, A, nch = raw_all.shape
    dev = raw_all.device
    assert raw_all.is_contiguous() and outputs.is_contiguous() and raw_all.dtype == torch.float16 and outputs.dtype == torch.float16
    assert labels.dtype == torch.float32 and labels.is_contiguous() and pred_iou.dtype == torch.float32
    ws_bytes = lib.tscd_loss_workspace_bytes(B, A, 0)
    ws = temp((ws_bytes,), dtype=torch.uint8, device=dev)
    sums = temp((3,), dtype=torch.float32, device=dev)
    d = temp((B, A, 1 + num_classes), dtype=torch.float16, device=dev) if want_grad else None
    if geo is not None and want_grad:      # base-detector training: also d(IoU-loss sum) / d raw box logits, [B,A,4]
        d_reg = temp((B, A, 4), dtype=torch.float16, device=dev)
        call("tscd_loss_dense_reg_f16", raw_all.data_ptr(), outputs.data_ptr(), fg_mask.data_ptr(), matched_gt.data_ptr(), pred_iou.data_ptr(),
             labels.data_ptr(), B, A, num_classes, labels.shape[1], geo.c_hw, geo.c_stride, ws.data_ptr(), ws_bytes, sums.data_ptr(), _ptr(d),
             d_reg.data_ptr(), _stream())
        return sums, d, d_reg
    call("tscd_loss_dense_f16", raw_all.data_ptr(), outputs.data_ptr(), fg_mask.data_ptr(), matched_gt.data_ptr(), pred_iou.data_ptr(),
         labels.data_ptr(), B, A, num_classes, labels.shape[1], ws.data_ptr(), ws_bytes, sums.data_ptr(), _ptr(d), _stream())
    return (sums, d) if geo is None else (sums, d, None)


def loss_rows(row_b, row_a, A, num_classes, fg_mask, matched_gt, pred_iou, labels, cls_logit=None, obj_logit=None, reg_delta=None,
              base=None, want_grad=True):
    """Refined-head losses over the N local proposal rows (row n = anchor row_a[n] of frame row_b[n], int32): -> (sums fp32 [3] =
    class BCE / matched-objectness BCE / SmoothL1 sums of the parts whose logits are given, d_cls, d_obj, d_reg (fp16 or None))."""
    N = row_b.shape[0]
    dev = row_b.device
    part = (1 if cls_logit is not None else 0) | (2 if obj_logit is not None else 0) | (4 if reg_delta is not None else 0)
    for t in (cls_logit, obj_logit, reg_delta):
        assert t is None or (t.is_contiguous() and t.dtype == torch.float16)
    assert row_b.dtype == torch.int32 and row_a.dtype == torch.int32 and row_b.is_contiguous() and row_a.is_contiguous()
    ws_bytes = lib.tscd_loss_workspace_bytes(0, 0, N)
    ws = temp((ws_bytes,), dtype=torch.uint8, device=dev)
    sums = temp((3,), dtype=torch.float32, device=dev)
    mk = lambda t: torch.empty_like(t) if (t is not None and want_grad) else None
    d_cls, d_obj, d_reg = mk(cls_logit), mk(obj_logit), mk(reg_delta)
    ndet = 0
    if base is not None:
        assert base.dtype == torch.float16 and base.stride(1) == 1
        ndet = base.stride(0)
    call("tscd_loss_rows_f16", row_b.data_ptr(), row_a.data_ptr(), N, A, num_classes, labels.shape[1], fg_mask.data_ptr(),
         matched_gt.data_ptr(), pred_iou.data_ptr(), labels.data_ptr(), _ptr(cls_logit), _ptr(obj_logit), _ptr(reg_delta), _ptr(base), ndet,
         part, ws.data_ptr(), ws_bytes, sums.data_ptr(), _ptr(d_cls), _ptr(d_obj), _ptr(d_reg), _stream())
    return sums, d_cls, d_obj, d_reg


def cls_score_gather(raw_all, row_b, row_a, pred_rows, cls_col=6):
    """sigmoid(raw_all[row_b, row_a, 5 + class]) per proposal row -> [N] fp16; raw_all [B,A,nch] fp16 contiguous."""
    B, A, nch = raw_all.shape
    N = row_b.shape[0]
    assert raw_all.is_contiguous() and raw_all.dtype == torch.float16 and pred_rows.dtype == torch.float16 and pred_rows.stride(1) == 1
    out = temp((N,), dtype=torch.float16, device=raw_all.device)
    call("tscd_cls_score_gather_f16", raw_all.data_ptr(), B, A, nch, row_b.data_ptr(), row_a.data_ptr(), pred_rows.data_ptr(),
         pred_rows.stride(0), cls_col, N, out.data_ptr(), _stream())
    return out


def cls_score_scatter(g, s, shape, row_b, row_a, pred_rows, cls_col=6):
    """Backward of cls_score_gather -> d raw_all [B,A,nch] fp16 (zero outside the proposals' own class logits)."""
    B, A, nch = shape
    d = temp(shape, dtype=torch.float16, device=g.device)
    assert g.is_contiguous() and g.dtype == torch.float16 and s.is_contiguous()
    call("tscd_cls_score_scatter_f16", g.data_ptr(), s.data_ptr(), B, A, nch, row_b.data_ptr(), row_a.data_ptr(), pred_rows.data_ptr(),
         pred_rows.stride(0), cls_col, g.shape[0], d.data_ptr(), _stream())
    return d


def pred_grad_split(d_raw, nc):
    """d_raw [B,h,w,5+nc] fp16 (a level's slice of the clip gradient: rows of a frame contiguous, any batch stride) ->
    (dY objectness [B,h,w,8], dY classes [B,h,w,ncp] zero-padded fp16, db [5+nc] fp32 with entries 4.. valid)."""
    B, h, w, nch = d_raw.shape
    assert nch == 5 + nc and d_raw.dtype == torch.float16 and d_raw.stride(3) == 1 and d_raw.stride(2) == nch and d_raw.stride(1) == w * nch
    ncp = (nc + 7) // 8 * 8
    dev = d_raw.device
    M = B * h * w
    dy_obj = temp((B, h, w, 8), dtype=torch.float16, device=dev)
    dy_cls = temp((B, h, w, ncp), dtype=torch.float16, device=dev)
    db = temp((nch,), dtype=torch.float32, device=dev)
    ws_bytes = lib.tscd_pred_grad_split_workspace_bytes(M, nch)
    ws = temp((ws_bytes,), dtype=torch.uint8, device=dev)
    call("tscd_pred_grad_split_f16", d_raw.data_ptr(), d_raw.stride(0) if B > 1 else h * w * nch, B, h * w, nch, ncp, dy_obj.data_ptr(),
         dy_cls.data_ptr(), db.data_ptr(), ws.data_ptr(), ws_bytes, _stream())
    return dy_obj, dy_cls, db


def scale_grad(d, g, i0, i1=None, split=0, pad=0):
    """Backward of a fused loss: fp16(d.float() * g[i]) in ONE launch (product in fp32, rounded once: g = loss scale x weight / num_fg
    overflows fp16 on its own at GradScaler's initial scale).  d fp16 [..., C] contiguous; g fp32 device vector; columns < split take
    g[i0], the others g[i1]; ``pad`` zero columns are put in front."""
    assert d.is_contiguous() and d.dtype == torch.float16 and g.dtype == torch.float32 and g.is_contiguous()
    C = d.shape[-1] if d.dim() > 1 else 1
    rows = d.numel() // C
    out = temp(d.shape[:-1] + (C + pad,) if d.dim() > 1 else d.shape, dtype=torch.float16, device=d.device)
    i1 = i0 if i1 is None else i1
    call("tscd_scale_grad_f16", d.data_ptr(), g.data_ptr(), i0, i1, split if i1 != i0 else C, rows, C, pad, out.data_ptr(), _stream())
    return out


# ----------------------------------------------------------------------------------------------------
# backward kernels of the trainable head convs
# ----------------------------------------------------------------------------------------------------
def conv_wgrad(x, dy, cin, cout, ksize, groups=1, variant=1, grad_params=None):
    """dW [groups*cout,cin,k,k] fp32 of stride-1 'same' convs; x [B,H,W,groups*cin], dy [B,H,W,groups*cout] fp16 NHWC(-slices).
    ``groups`` > 1: group g's gradient from channels [g*cin, ...) of x and [g*cout, ...) of dy, stacked along dim 0.
    ``variant`` (tests): 0 = the register-staged cross-check loop.
    ``grad_params``: the parameters whose gradients the rows of dW are, in row order (each takes its own shape[0] rows; the caller hands
    plain row slices of the result to autograd) -- inside a backward pass the GEMM is then only REGISTERED and runs with the pass's
    other weight gradients in one launch (deferred weight gradients, above)."""
    B, H, W, _ = x.shape
    M = B * H * W
    dw = temp((groups * cout, cin, ksize, ksize), dtype=torch.float32, device=x.device)
    if grad_params is not None and variant == 1 and M > 0:
        ldx, lddy = nhwc_ld(x), nhwc_ld(dy)
        rows = sum(p.shape[0] for p in grad_params)
        same_cols = all(tuple(p.shape[1:]) == (cin, ksize, ksize) for p in grad_params)
        if same_cols and rows <= groups * cout and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0 \
                and lib.tscd_wgrad_batch_supported(B, H, W, cin, cout, ksize, ldx, lddy, groups):
            batch = _wgrad_batch(grad_params)
            if batch is not None:
                batch.jobs.append(_WgradJobC(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W, cin, cout, ksize, ldx, lddy, groups, 0))
                batch.keep.append((x, dy))
                r0 = 0
                for p in grad_params:
                    batch.check.append((p, dw[r0:r0 + p.shape[0]]))
                    r0 += p.shape[0] if groups == 1 or p.shape[0] == cout else cout
                return dw
    return _conv_wgrad_now(x, dy, dw, cin, cout, ksize, groups, variant)


# timed only when the GEMM is launched here: a deferred job's time is the "wgrad_batch" entry of its pass
@_profiled("conv_wgrad", lambda r, x, dy, dw, cin, cout, ksize, groups, variant:
           (2.0 * x.shape[0] * x.shape[1] * x.shape[2] * cin * cout * ksize * ksize * groups,
            2.0 * (x.numel() + dy.numel()) + 4.0 * groups * cout * cin * ksize * ksize))
def _conv_wgrad_now(x, dy, dw, cin, cout, ksize, groups, variant):
    B, H, W, _ = x.shape
    M = B * H * W
    if groups == 1 and variant == 1:
        ws_bytes = lib.tscd_conv_wgrad_workspace_bytes(M, cout, cin, ksize)
        ws = temp((ws_bytes,), dtype=torch.uint8, device=x.device)
        call("tscd_conv_wgrad_nhwc_f16", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W, cin, cout, ksize, nhwc_ld(x), nhwc_ld(dy),
             ws.data_ptr(), ws_bytes, _stream())
        return dw
    ws_bytes = lib.tscd_conv_wgrad_workspace_bytes_ex(M, cout, cin, ksize, groups)
    ws = temp((ws_bytes,), dtype=torch.uint8, device=x.device)
    call("tscd_conv_wgrad_nhwc_f16_ex", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W, cin, cout, ksize, nhwc_ld(x), nhwc_ld(dy),
         groups, variant, ws.data_ptr(), ws_bytes, _stream())
    return dw


def conv_dgrad(dy, weight):
    """dX of a stride-1 'same' conv = the forward MFMA kernel on dY with the weights packed in dgrad order."""
    cout, cin, k, _ = weight.shape
    wpk = WEIGHTS.get(weight, None, 1)
    dx, _ = conv2d(dy, wpk, cin, k, 1, EPI_RAW)
    return dx


def bn_silu_bwd(dy, raw, scale_shift, mean, invstd, gamma):
    """-> (draw fp16 like raw, dgamma [C], dbeta [C]) for y = SiLU(BN_train(raw))."""
    B, H, W, C = raw.shape
    M = B * H * W
    dy = dy.contiguous()
    assert raw.is_contiguous()
    ws_bytes = lib.tscd_bn_silu_bwd_workspace_bytes(M, C)
    ws = temp((ws_bytes,), dtype=torch.uint8, device=raw.device)
    draw = torch.empty_like(raw)
    sums = temp((2, C), dtype=torch.float32, device=raw.device)
    call("tscd_bn_silu_bwd_f16", dy.data_ptr(), raw.data_ptr(), scale_shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
         gamma.data_ptr(), M, C, draw.data_ptr(), sums.data_ptr(), ws.data_ptr(), ws_bytes, _stream())
    return draw, sums[1], sums[0]


@_profiled("bn_silu_bwd", lambda r, dys, raw, *a, **k: (0.0, 2.0 * raw.numel() * 5))       # two passes over dy and raw + draw written
def bn_silu_bwd_multi(dys, raw, scale_shift, mean, invstd, gammas):
    """The same over the channels of SEVERAL BaseConvs that shared a conv launch: dys = 1 or 2 gradient tensors [B,H,W,c_i]
    (NHWC(-slices), sum c_i = C), raw [B,H,W,C] a channel slice, gammas = 1 or 2 fp32 vectors covering the C channels.
    -> (draw [B,H,W,C] contiguous fp16, dgamma [C], dbeta [C])."""
    B, H, W, C = raw.shape
    M = B * H * W
    assert len(dys) in (1, 2) and sum(d.shape[-1] for d in dys) == C and all(d.dtype == torch.float16 for d in dys)
    assert len(gammas) in (1, 2) and sum(g.shape[0] for g in gammas) == C and all(g.dtype == torch.float32 for g in gammas)
    ws_bytes = lib.tscd_bn_silu_bwd_workspace_bytes(M, C)
    ws = temp((ws_bytes,), dtype=torch.uint8, device=raw.device)
    draw = empty_nhwc(B, H, W, C, raw.device)
    sums = temp((2, C), dtype=torch.float32, device=raw.device)
    d1 = dys[1] if len(dys) == 2 else None
    call("tscd_bn_silu_bwd_f16_ex", dys[0].data_ptr(), nhwc_ld(dys[0]), _ptr(d1), nhwc_ld(d1) if d1 is not None else 0, dys[0].shape[-1],
         raw.data_ptr(), nhwc_ld(raw), scale_shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gammas[0].data_ptr(),
         _ptr(gammas[1] if len(gammas) > 1 else None), gammas[0].shape[0], M, C, draw.data_ptr(), C, sums.data_ptr(), ws.data_ptr(),
         ws_bytes, _stream())
    return draw, sums[1], sums[0]


# ----------------------------------------------------------------------------------------------------
# ModelEMA.update over every fp32 state tensor in one launch (elementwise.hip: tscd_ema_update_f32)
# ----------------------------------------------------------------------------------------------------
_EMA_TABLES = {}


def ema_update(dst, src, decay, tag=None):
    """dst[i] = decay * dst[i] + (1 - decay) * src[i] for lists of same-shaped contiguous fp32 CUDA tensors, one launch
    (decay 0 with finite dst = a multi-tensor copy).  ``tag`` keeps the cached pointer tables of different callers apart.
    The chunk table (device pointers) is cached and validated against EVERY address on each call (parameters and buffers are
    updated in place, so the addresses only change when a tensor is replaced, moved or cast -- the table is rebuilt then;
    ~0.1 ms of host time for ~1300 addresses).  The kernel writes through raw pointers, which autograd's version counters do
    not see: the destinations' versions are bumped afterwards so that caches keyed on them (the packed conv / linear weights
    of WeightCache) notice the update."""
    if not dst:
        return
    dev = dst[0].device
    sig = tuple(t.data_ptr() for t in dst) + tuple(t.data_ptr() for t in src)
    key = (len(dst), dev, tag)
    hit = _EMA_TABLES.get(key)
    if hit is None or hit[0] != sig:
        import numpy as np
        ch = lib.tscd_ema_chunk_elems()
        rec = []
        for a, b in zip(dst, src):
            assert a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.is_contiguous() \
                and b.is_contiguous() and a.numel() == b.numel(), "ema_update: contiguous fp32 CUDA tensors of equal size"
            pa, pb, n = a.data_ptr(), b.data_ptr(), a.numel()
            for o in range(0, n, ch):
                rec.append((pa + 4 * o, pb + 4 * o, min(ch, n - o)))
        table = torch.from_numpy(np.array(rec, dtype=np.int64).reshape(-1, 3)).to(dev)
        if len(_EMA_TABLES) > 16:
            _EMA_TABLES.clear()
        hit = _EMA_TABLES[key] = (sig, table, len(rec))
    call("tscd_ema_update_f32", hit[1].data_ptr(), hit[2], float(decay), float(1.0 - decay), _stream())
    torch.autograd.graph.increment_version(dst)
