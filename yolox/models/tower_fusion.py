"""The decoupled head's towers as fused launches.

Reference: TSCDHead.forward per-level block, yolox/models/tscd_head.py:315-324 -- four towers of two BaseConv(3x3) each
(``reg_convs``, ``cls_convs`` frozen; ``cls_convs2``, ``reg_convs2`` trained) all read the SAME stem output:

    first convs   several towers' first convs run as ONE dense conv over the concatenated output channels: the stem output is staged
                  once for n*W output channels (4x the workgroups of one conv on the 36x36 / 18x18 levels, which do not fill 256 CUs
                  otherwise; on the 72x72 level the partly filled last round shrinks from 16 % to < 4 % of the launch);
    second convs  each tower's second conv reads its own W-channel slice of that buffer: ONE grouped launch
                  (tscd_conv2d_nhwc_f16_ex, groups = n), writing the towers' outputs as channel slices of one [B,h,w,n*W] buffer.

Each BaseConv keeps its own BatchNorm (gamma / beta / running statistics stay the module's own tensors: state_dict untouched); one
statistics launch and one apply launch cover all of them.  The trainable pair has a matching backward: BatchNorm + SiLU backward
over both towers in one pass, ONE grouped weight-gradient launch for the second convs, ONE grouped data-gradient launch, and ONE dense
weight-gradient launch (Cout = 2W) for the first convs, whose input (the frozen stem's output) needs no gradient.

Packed weights: the fused matrices are ONE buffer whose row blocks are registered in ``ops.WEIGHTS`` as the packed copies of the
individual parameters, so the optimizer's batched in-place re-pack (ops.refresh_packed_weights) keeps the fused matrix current with
no extra launch, and single-conv calls on those parameters read the same storage.
"""
import weakref

import torch

from tscd_amd import ops

from . import network_blocks as nb

_FUSED = {}       # (ids of the parameters, mode, layout) -> (big packed buffer, row offsets)


def fusable(convs, x=None):
    """The fused path needs every conv's packed row block to be exactly its Cout rows (Cout a multiple of the conv tile width it
    selects: 32 / 64 / 128-multiples) so the blocks tile one matrix; 3x3 stride-1 BaseConvs of one width on CUDA."""
    w0 = convs[0].conv.weight
    cout, cin, k, _ = w0.shape
    if not w0.is_cuda or cout % ops.lib.tscd_conv_tile_n(cout) != 0:
        return False
    mode0 = batch_stats(convs[0])
    for c in convs:
        w = c.conv.weight
        if tuple(w.shape) != (cout, cin, k, k) or c.stride != 1 or w.dtype not in (torch.float32, torch.float16) or not w.is_contiguous():
            return False
        # one launch normalises all of them: they must agree on batch vs running statistics (fix_bn puts BatchNorms in eval() inside a
        # training model, ovis_tscd_large.py:153-160), eps and momentum
        if batch_stats(c) != mode0 or c.bn.eps != convs[0].bn.eps or c.bn.momentum != convs[0].bn.momentum:
            return False
    return True


def batch_stats(conv):
    """BaseConv.forward's rule (network_blocks.py:58-61 over nn.BatchNorm2d): batch statistics iff the module AND its BatchNorm train."""
    return bool(conv.training and conv.bn.training)


def _fused_weights(params, mode, layout):
    """One packed matrix holding the packed copies of ``params`` as row blocks.
    layout "dense" (shared input, mode 0): block i = rows [i*Cout, (i+1)*Cout) -- with the tail padded to the dense conv's tile width;
    layout "grouped": block g = the g-th group's own padded matrix (rows_pad each), forward (mode 0) or data-gradient (mode 1) order."""
    key = tuple(id(p) for p in params) + (mode, layout)
    cache = ops.WEIGHTS
    ent = _FUSED.get(key)
    if ent is not None:
        big, offs, refs = ent
        ok = True
        for p, off, r in zip(params, offs, refs):
            e = cache._c.get((id(p), mode, None))
            if r() is not p or e is None or e[2]() is not p or e[0] != cache._tag(p) or e[1].data_ptr() != big.data_ptr() + off:
                ok = False
                break
        if ok:
            return big
    w0 = params[0]
    Cout, Cin, inner, KH, KW, rows_pad, kpad = ops._pack_dims(w0.shape, None, mode)
    n = len(params)
    rows = Cout if mode == 0 else Cin                   # rows of one parameter's block
    assert rows_pad == rows, "fusable() admits only widths whose packed block has no padding rows"
    if layout == "dense":
        assert mode == 0
        bn = ops.lib.tscd_conv_tile_n(n * rows)
        total = (n * rows + bn - 1) // bn * bn
    else:
        total = n * rows_pad
    big = torch.zeros((total, kpad), dtype=torch.float16, device=w0.device)
    offs = []
    for i, p in enumerate(params):
        blk = big[i * rows:(i + 1) * rows]
        w = p.detach()
        ops.call("tscd_pack_conv_weight", w.data_ptr(), int(w.dtype == torch.float32), blk.data_ptr(), Cout, Cin, inner, KH, KW, rows, kpad,
                 mode, ops._stream())
        # the block IS this parameter's packed copy from now on (WeightCache entry layout: tag, packed, weakref, shape, cin_pad, mode)
        cache._c[(id(p), mode, None)] = (cache._tag(p), blk, weakref.ref(p), tuple(p.shape), None, mode)
        offs.append(i * rows * kpad * 2)
    ops._REFRESH_TABLES.clear()
    if key in _FUSED:
        # the fused buffer of a live key is being replaced: captured graphs of the frozen front hold raw pointers into the old one
        from tscd_amd import graphs
        graphs.clear()
    _FUSED[key] = (big, offs, [weakref.ref(p) for p in params])
    if len(_FUSED) > 64:
        _evict_dead()
    return big


def _evict_dead():
    """Drop the entries whose parameters are gone (discarded models, EMA copies): per key, never the live ones -- a live entry owns
    the packed storage that ops.WEIGHTS and captured graphs point into."""
    for k in [k for k, (_, _, refs) in _FUSED.items() if any(r() is None for r in refs)]:
        del _FUSED[k]


def _bn_tuple(conv):
    bn = conv.bn
    f = lambda t: t if t.dtype == torch.float32 else t.float()
    return (f(bn.weight), f(bn.bias), f(bn.running_mean), f(bn.running_var))


def _tick(convs):
    for c in convs:
        bn = c.bn
        if bn.track_running_stats:
            if nb._DEFERRED_TICKS is not None:
                nb._DEFERRED_TICKS.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked += 1


def _folded(convs):
    """Concatenated eval-mode scale / shift [2, n*C] of several BaseConvs, cached on the first module per state of all BN tensors."""
    c0 = convs[0]
    ver = tuple(t._version for c in convs for t in (c.bn.weight, c.bn.bias, c.bn.running_mean, c.bn.running_var)) + \
        tuple(c.bn.weight.data_ptr() for c in convs)
    d = c0.__dict__
    hit = d.get("_fused_folded")
    if hit is not None and hit[0] == ver and hit[1] == len(convs):
        return hit[2]
    ss = torch.cat([ops.bn_fold(c.bn.weight.float(), c.bn.bias.float(), c.bn.running_mean.float(), c.bn.running_var.float(), c.bn.eps)
                    for c in convs], 1).contiguous()
    d["_fused_folded"] = (ver, len(convs), ss)
    return ss


def towers_nograd(x, firsts, seconds):
    """n towers (first conv, second conv) on a shared input x [B,h,w,Cin] without autograd -> [B,h,w,n*W]; tower i = channels
    [i*W, (i+1)*W).  Training-mode modules take batch statistics (running statistics updated, like the reference's frozen towers under
    model.train()); evaluation-mode modules take the folded BatchNorm in the conv epilogue."""
    n = len(firsts)
    W = firsts[0].conv.out_channels
    wa = _fused_weights([c.conv.weight for c in firsts], 0, "dense")
    wb = _fused_weights([c.conv.weight for c in seconds], 0, "grouped")
    k = firsts[0].conv.kernel_size[0]
    training = batch_stats(firsts[0])
    if training:
        _tick(firsts)
        _tick(seconds)
        eps, mom = firsts[0].bn.eps, firsts[0].bn.momentum
        a = ops.conv_bn_act_train_multi(x, wa, W, [_bn_tuple(c) for c in firsts], k, 1, eps, mom)[0]
        return ops.conv_bn_act_train_multi(a, wb, W, [_bn_tuple(c) for c in seconds], k, n, eps, mom)[0]
    ssa, ssb = _folded(firsts), _folded(seconds)
    a, _ = ops.conv2d(x, wa, n * W, k, 1, ops.EPI_BN_SILU, scale_shift=ssa)
    b, _ = ops.conv2d(a, wb, W, k, 1, ops.EPI_BN_SILU, scale_shift=ssb, groups=n)
    return b


def shared_input_pair_nograd(x, c0, c1):
    """Two 1x1 / 3x3 BaseConvs of one width on the SAME input, without autograd, as ONE conv launch over the concatenated output channels
    (+ one statistics and one apply launch in training mode): CSPLayer's conv1 || conv2 (network_blocks.py:163-176 / reference
    network_blocks.py:134-165).  -> [B,h,w,2*Cout]: c0's output in channels [0, Cout), c1's in [Cout, 2*Cout).  The conv outputs are the two separate
    launches' bits (every output channel sums its K in the same order); in training mode the batch statistics are summed over the wider
    launch's M-tiles (another tile height), so they agree to fp32 rounding, not bit for bit."""
    C = c0.conv.out_channels
    wa = _fused_weights([c0.conv.weight, c1.conv.weight], 0, "dense")
    k = c0.conv.kernel_size[0]
    if batch_stats(c0):
        _tick([c0, c1])
        return ops.conv_bn_act_train_multi(x, wa, C, [_bn_tuple(c0), _bn_tuple(c1)], k, 1, c0.bn.eps, c0.bn.momentum)[0]
    out, _ = ops.conv2d(x, wa, 2 * C, k, 1, ops.EPI_BN_SILU, scale_shift=_folded([c0, c1]))
    return out


class _TowerPairTrain(torch.autograd.Function):
    """Two trainable towers on the (gradient-free) stem output, training mode.  Inputs after x: the 4 conv weights, 4 gammas, 4 betas
    (first convs, then second convs); ``mods`` carries the BatchNorm buffers.  Returns the two towers' outputs (views of one buffer)."""

    @staticmethod
    def forward(ctx, x, mods, wa0, wa1, wb0, wb1, ga0, ga1, gb0, gb1, ba0, ba1, bb0, bb1):
        firsts, seconds = mods
        W = wa0.shape[0]
        k = wa0.shape[-1]
        wa = _fused_weights([wa0, wa1], 0, "dense")
        wb = _fused_weights([wb0, wb1], 0, "grouped")
        f = lambda t: t if t.dtype == torch.float32 else t.float()
        bna = [(f(g), f(b), f(c.bn.running_mean), f(c.bn.running_var)) for g, b, c in ((ga0, ba0, firsts[0]), (ga1, ba1, firsts[1]))]
        bnb = [(f(g), f(b), f(c.bn.running_mean), f(c.bn.running_var)) for g, b, c in ((gb0, bb0, seconds[0]), (gb1, bb1, seconds[1]))]
        eps, mom = firsts[0].bn.eps, firsts[0].bn.momentum
        a_out, a_raw, a_ss, a_mean, a_inv = ops.conv_bn_act_train_multi(x, wa, W, bna, k, 1, eps, mom, keep_raw=True)
        b_out, b_raw, b_ss, b_mean, b_inv = ops.conv_bn_act_train_multi(a_out, wb, W, bnb, k, 2, eps, mom, keep_raw=True)
        ctx.save_for_backward(x, a_out, a_raw, a_ss, a_mean, a_inv, b_raw, b_ss, b_mean, b_inv, wa0, wa1, wb0, wb1,
                              bna[0][0], bna[1][0], bnb[0][0], bnb[1][0])
        ctx.pdtype = ga0.dtype
        return b_out[..., :W], b_out[..., W:]

    @staticmethod
    def backward(ctx, g0, g1):
        (x, a_out, a_raw, a_ss, a_mean, a_inv, b_raw, b_ss, b_mean, b_inv, wa0, wa1, wb0, wb1, ga0, ga1, gb0, gb1) = ctx.saved_tensors
        W = wa0.shape[0]
        k = wa0.shape[-1]
        B, H, Wd, _ = b_raw.shape
        h16 = lambda g: (ops.ztemp((B, H, Wd, W), dtype=torch.float16, device=b_raw.device) if g is None else
                         (g if g.dtype == torch.float16 else g.to(torch.float16)))
        g0, g1 = h16(g0), h16(g1)
        try:
            ops.nhwc_ld(g0), ops.nhwc_ld(g1)
        except (ValueError, AssertionError):
            g0, g1 = g0.contiguous(), g1.contiguous()
        # second convs: BatchNorm + SiLU backward of both towers in one pass, grouped weight gradient, grouped data gradient
        draw_b, dgam_b, dbet_b = ops.bn_silu_bwd_multi([g0, g1], b_raw, b_ss, b_mean, b_inv, [gb0, gb1])
        wd = wa0.dtype
        with ops.param_grad_stream([wb0, wb1], a_out, draw_b):      # weight gradients run beside the data-gradient chain
            dwb = ops.conv_wgrad(a_out, draw_b, W, W, k, groups=2, grad_params=[wb0, wb1]).to(wd)
        wdg = _fused_weights([wb0, wb1], 1, "grouped")
        da, _ = ops.conv2d(draw_b, wdg, W, k, 1, ops.EPI_RAW, groups=2)
        # first convs: one BatchNorm + SiLU backward over the 2W channels, one dense weight gradient (the stem output needs none)
        draw_a, dgam_a, dbet_a = ops.bn_silu_bwd_multi([da], a_raw, a_ss, a_mean, a_inv, [ga0, ga1])
        with ops.param_grad_stream([wa0, wa1], x, draw_a):
            dwa = ops.conv_wgrad(x, draw_a, x.shape[-1], 2 * W, k, grad_params=[wa0, wa1]).to(wd)
        pd = ctx.pdtype
        c = lambda t: t if t.dtype == pd else t.to(pd)
        return (None, None, dwa[:W], dwa[W:], dwb[:W], dwb[W:],
                c(dgam_a[:W]), c(dgam_a[W:]), c(dgam_b[:W]), c(dgam_b[W:]), c(dbet_a[:W]), c(dbet_a[W:]), c(dbet_b[:W]), c(dbet_b[W:]))


def tower_pair_train(x, tower0, tower1):
    """(tower0(x), tower1(x)) for two trainable towers (nn.Sequential of two BaseConv) in training mode, x without gradient."""
    firsts, seconds = (tower0[0], tower1[0]), (tower0[1], tower1[1])
    _tick(firsts)
    _tick(seconds)
    return _TowerPairTrain.apply(x, (firsts, seconds),
                                 firsts[0].conv.weight, firsts[1].conv.weight, seconds[0].conv.weight, seconds[1].conv.weight,
                                 firsts[0].bn.weight, firsts[1].bn.weight, seconds[0].bn.weight, seconds[1].bn.weight,
                                 firsts[0].bn.bias, firsts[1].bn.bias, seconds[0].bn.bias, seconds[1].bn.bias)
