"""TSCDHead on the MI355X kernels.

Interface mirror of the reference's yolox/models/tscd_head.py:26-1916 for the flag set the TSCD exps use
(agg_type='mca', decouple_reg, reconf, ota_mode, use_pre_nms=False, cat_ota_fg=False, cls_ota):
same constructor (including its misspelt keyword names), attribute names and state_dict keys, same forward
contract -- training returns the reference's 9-tuple of losses, eval returns (result, result_ori).

What changed is where the work happens.  The reference drives the proposal stage from the host (per-image SimOTA
loops with .item() syncs, a per-proposal torch.where loop, per-frame attention loops); here every stage is a
device kernel over the whole clip and the host only learns (a) the per-frame proposal counts (one D2H, they size
the ragged tensors), (b) the Hungarian cost matrix per local frame (scipy runs on the host as in the reference)
and (c) in eval the per-frame detection counts.  Losses are evaluated densely over all anchors with device-side
masks instead of boolean-mask gathers, so no further syncs occur.
"""
import math

import contextlib
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from tscd_amd import functional as TF
from tscd_amd import graphs, ops
from tscd_amd.autograd import eager_branch

from . import tower_fusion as TW
from .network_blocks import BaseConv
from .post_process import nms_frames
from .post_trans import MCA_tscd_g2l_reg
from .surrounding_extraction import WaveletsHFBlock
from .tscd_matching import AwarePositionRegMatcher, TaskAligned, linear as hip_linear


class _GatherRows(torch.autograd.Function):
    """find_feature_score's row gather; backward = scatter into zero-initialised per-level gradient maps."""

    @staticmethod
    def forward(ctx, geo, sel_idx, frame_off, N, W, nmaps, *maps):
        per_feature = [list(maps[f * 3:(f + 1) * 3]) for f in range(nmaps)]
        outs = ops.gather_rows(per_feature, geo, sel_idx, frame_off, N, W)
        ctx.geo, ctx.sel_idx, ctx.frame_off, ctx.N, ctx.W, ctx.nmaps = geo, sel_idx, frame_off, N, W, nmaps
        ctx.shapes = [m.shape for m in maps]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        zeros = [ops.ztemp(s, dtype=torch.float16, device=grads[0].device) for s in ctx.shapes]
        per_feature = [zeros[f * 3:(f + 1) * 3] for f in range(ctx.nmaps)]
        gs = [g if g is not None else ops.ztemp((ctx.N, ctx.W), dtype=torch.float16, device=zeros[0].device) for g in grads]
        ops.scatter_rows([g.to(torch.float16) for g in gs], per_feature, ctx.geo, ctx.sel_idx, ctx.frame_off, ctx.N, ctx.W)
        return (None,) * 6 + tuple(zeros)


class _TapGather(torch.autograd.Function):
    """The row gather of the trained towers' outputs (find_feature_score) TOGETHER with the regression maps' other consumer: returns
    (f_cls, f_reg, reg_tap[0..2]) where reg_tap[k] aliases reg_maps[k] and feeds the edge block.  Backward then receives the edge
    block's gradient of reg_maps[k] and the gathered rows' gradient in ONE call and adds the rows into the former in place (distinct
    anchors: no two rows meet) -- the separate path zero-filled a map per level, scattered into it and left the sum to a tensor add."""

    @staticmethod
    def forward(ctx, geo, sel_idx, frame_off, N, W, *maps):
        cls_maps, reg_maps = list(maps[:3]), list(maps[3:6])
        f_cls, f_reg = ops.gather_rows([cls_maps, reg_maps], geo, sel_idx, frame_off, N, W)
        ctx.geo, ctx.sel_idx, ctx.frame_off, ctx.N, ctx.W = geo, sel_idx, frame_off, N, W
        ctx.shapes = [m.shape for m in maps]
        return (f_cls, f_reg) + tuple(m.view(m.shape) for m in reg_maps)

    @staticmethod
    def backward(ctx, g_cls, g_reg, *g_taps):
        dev = ctx.sel_idx.device
        zc = [ops.ztemp(s, dtype=torch.float16, device=dev) for s in ctx.shapes[:3]]
        ok = all(g is not None and g.dtype == torch.float16 and g.is_contiguous() and g.shape == s for g, s in zip(g_taps, ctx.shapes[3:]))
        zr = list(g_taps) if ok else [ops.ztemp(s, dtype=torch.float16, device=dev) for s in ctx.shapes[3:]]
        rows = [g.to(torch.float16) if g is not None else ops.ztemp((ctx.N, ctx.W), dtype=torch.float16, device=dev) for g in (g_cls, g_reg)]
        ops.scatter_rows(rows, [zc, zr], ctx.geo, ctx.sel_idx, ctx.frame_off, ctx.N, ctx.W, add=(False, ok))
        if not ok:
            zr = [z if g is None else z + g.to(torch.float16) for z, g in zip(zr, g_taps)]
        return (None,) * 5 + tuple(zc) + tuple(zr)


class HeadPrefix:
    """The part of the head that depends on nothing the optimizer steps (reference: ovis_tscd_large.py:141-149 freezes ``stems``,
    ``reg_convs``, ``cls_convs`` and ``reg_preds``): per pyramid level (x = stem output [B,h,w,W], fz = [reg_feat | cls_feat]
    [B,h,w,2W], reg_o = box logits [B,h,w,4]), all NHWC fp16.  TSCD computes it together with the backbone pass -- for the NEXT clip
    inside this clip's assignment window, as one hipGraph replay (yolox/models/tscd.py) -- and hands it to TSCDHead.forward in place of
    the pyramid."""
    __slots__ = ("levels",)

    def __init__(self, levels):
        self.levels = levels

    def tensors(self):
        return tuple(t for lvl in self.levels for t in lvl)

    @staticmethod
    def from_tensors(ts):
        return HeadPrefix([tuple(ts[i:i + 3]) for i in range(0, len(ts), 3)])


# Streams of the trained towers: "0" all on the main stream; "1" the 36x36 / 18x18 levels on a side stream each, the 72x72 level on
# the main stream behind the copy of the proposal counts; "2" the 72x72 level on one side stream, the two small levels on another;
# "3" one side stream per level; "4" the towers on the main stream and the prediction convs / decode / selection / counts copy on
# ONE side stream.  Default "0": with the assignment chain joined on the device (tscd_matching.DEFER_CHAIN) the host runs far ahead
# of the device and a single in-order queue executes back to back; every layout with a side stream measured 0.1-0.2 ms slower in
# the backward pass (the engine's cross-stream event waits) than it gained in the forward (profiles/r03_stream_layouts.txt).  HIP multiplexes a process's streams onto a few hardware queues (4 by default): with more streams
# than that two of them share an in-order queue, and the main stream's work was seen queued behind the prefetched backbone graph.
LEVEL_STREAM_LAYOUT = os.environ.get("TSCD_LEVEL_STREAMS", "0")
LEVEL_STREAMS = LEVEL_STREAM_LAYOUT != "0"
FUSE_TOWERS = os.environ.get("TSCD_FUSE_TOWERS", "1") != "0"
EAGER_CLASS_BRANCH = os.environ.get("TSCD_EAGER_CLASS_BRANCH", "1") != "0"
HOOK_EARLY = os.environ.get("TSCD_PREFETCH_EARLY", "1") != "0"      # the next clip's frozen pass right behind the cost matrices' copy
# the trained towers' row gather and the edge block share ONE autograd node on the regression maps (_TapGather: the rows are added into the
# edge block's gradient map in place); "0" = separate gather (zero-filled maps + a tensor add per level)
TAP_GATHER = os.environ.get("TSCD_TAP_GATHER", "1") != "0"


_SIDE_STREAMS = {}      # device -> [torch.cuda.Stream]; process-wide (module attributes must stay deep-copyable for ModelEMA)


_EAGER_SEED = {}        # device -> 0-dim fp32 tensor: loss scale the eager class branch is seeded with when no scaler scale is given


def _eager_seed(dev):
    t = _EAGER_SEED.get(dev)
    if t is None:
        t = _EAGER_SEED[dev] = torch.full((), 65536.0, dtype=torch.float32, device=dev)      # GradScaler's default init_scale
    return t


def _side_streams(dev, n):
    st = _SIDE_STREAMS.get(dev)
    if st is None or len(st) < n:
        st = _SIDE_STREAMS[dev] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return st


class TSCDHead(nn.Module):
    def __init__(self, num_classes, width=1.0, strides=[8, 16, 32], in_channels=[256, 512, 1024], act="silu", depthwise=False,
                 heads=4, drop=0.0, use_score=True, defualt_p=30, sim_thresh=0.75, pre_nms=0.75, ave=True, defulat_pre=750,
                 test_conf=0.001, use_mask=False, gmode=True, lmode=True, both_mode=False, localBlocks=1, **kwargs):
        super().__init__()
        if depthwise:
            raise NotImplementedError("depthwise convs are not on the TSCD path")
        unsupported = (kwargs.get("agg_type", "localagg") != "mca" or not kwargs.get("ota_mode", False)
                       or not kwargs.get("decouple_reg", False) or not kwargs.get("reconf", False)
                       or kwargs.get("use_pre_nms", True) or kwargs.get("cat_ota_fg", True) or kwargs.get("vid_ota", False)
                       or not kwargs.get("cls_ota", True) or not kwargs.get("vid_cls", True) or not kwargs.get("vid_reg", True)
                       or both_mode or not ave or use_mask)
        if unsupported:
            raise NotImplementedError(
                "only the flag set of exps/TSCD_OVIS/ovis_tscd_{base,large}.py / TSCD_VID/vid_tscd_large.py is built: "
                "agg_type='mca', ota_mode, decouple_reg, reconf, use_pre_nms=False, cat_ota_fg=False (SURVEY.md T4)")
        self.Afternum = defualt_p
        self.Prenum = defulat_pre
        self.simN = defualt_p
        self.nms_thresh = pre_nms
        self.n_anchors = 1
        self.use_score = use_score
        self.num_classes = num_classes
        self.decode_in_inference = True
        self.gmode, self.lmode, self.both_mode = gmode, lmode, both_mode
        self.width = int(256 * width)
        self.sim_thresh, self.ave, self.use_mask = sim_thresh, ave, use_mask
        W = self.width

        self.cls_convs = nn.ModuleList()
        self.reg_convs = nn.ModuleList()
        self.edge_enhance_reg = nn.ModuleList()
        self.cls_preds = nn.ModuleList()
        self.reg_preds = nn.ModuleList()
        self.obj_preds = nn.ModuleList()
        self.cls_convs2 = nn.ModuleList()
        self.reg_convs2 = nn.ModuleList()
        self.agg = MCA_tscd_g2l_reg(dim=W, out_dim=4 * W, num_heads=heads, attn_drop=drop, reconf=True)
        self.agg_iou = MCA_tscd_g2l_reg(dim=W, out_dim=4 * W, num_heads=heads, attn_drop=drop, reconf=True)
        self.local_reg_matcher = AwarePositionRegMatcher(hidden_channel=W, feedforward_channel=W, num_head=8,
                                                         decoder_layer_num=1, act=act)
        self.fc_reg_matcher = nn.Linear(W, 4 * W)
        self.task_aligned = TaskAligned(hidden_channel=4 * W, num_head=8, decoder_layer_num=1, act=act)
        self.cls_pred = nn.Linear(4 * W, num_classes)
        self.matcher_obj_pred = nn.Linear(4 * W, 1)
        self.matcher_reg_pred = nn.Linear(4 * W, 4)
        self.stems = nn.ModuleList()
        self.kwargs = kwargs

        tower = lambda: nn.Sequential(BaseConv(W, W, 3, 1, act=act), BaseConv(W, W, 3, 1, act=act))
        for i in range(len(in_channels)):
            self.stems.append(BaseConv(int(in_channels[i] * width), W, 1, 1, act=act))
            self.cls_convs.append(tower())
            self.reg_convs.append(tower())
            self.edge_enhance_reg.append(nn.Sequential(WaveletsHFBlock(in_channels=W)))
            self.cls_preds.append(nn.Conv2d(W, self.n_anchors * num_classes, 1, 1, 0))
            self.reg_preds.append(nn.Conv2d(W, 4, 1, 1, 0))
            self.obj_preds.append(nn.Conv2d(W, self.n_anchors * 1, 1, 1, 0))
            self.cls_convs2.append(tower())
            self.reg_convs2.append(tower())

        self.use_l1 = False
        self.l1_loss = nn.L1Loss(reduction="none")
        self.bcewithlog_loss = nn.BCEWithLogitsLoss(reduction="none")
        from .losses import IOUloss
        self.iou_loss = IOUloss(reduction="none")      # attribute kept for the reference's module tree; the head's IoU loss runs in loss.hip
        self.smooth_l1_loss = nn.SmoothL1Loss(reduction="none")
        self.strides = strides
        self.ota_mode = True
        self.max_gt = 128          # SimOTA cost planes are sized for this many boxes per frame
        self.debug = None          # tests set a dict here to capture intermediates
        self.loss_scale = None     # 0-dim fp32 device tensor: the GradScaler's current scale (set by the training loop per step)

    def initialize_biases(self, prior_prob):
        for convs in (self.cls_preds, self.obj_preds):
            for conv in convs:
                b = conv.bias.view(self.n_anchors, -1)
                b.data.fill_(-math.log((1 - prior_prob) / prior_prob))
                conv.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    # ------------------------------------------------------------------------------------------------
    def _dbg(self, **kw):
        if self.debug is not None:
            self.debug.update(kw)

    @torch.no_grad()
    def frozen_prefix(self, xin):
        """stems + the base towers (+ the box prediction convs when they are frozen too) of every level, without autograd -- the
        reference's frozen half of tscd_head.py:315-373 (its BatchNorms still take batch statistics in training mode).  xin: the
        pyramid as NCHW-logical maps.  The two base towers run as fused launches (tower_fusion.py)."""
        return HeadPrefix([self.frozen_prefix_level(k, xf.permute(0, 2, 3, 1)) for k, xf in enumerate(xin)])

    def frozen_prefix_level(self, k, f):
        """Level k of frozen_prefix on its NHWC pyramid map -> (stem output, [reg_feat | cls_feat], box logits or None)."""
        W = self.width
        f = f if f.dtype == torch.float16 else f.half()
        x = self.stems[k](f)
        firsts, seconds = [self.reg_convs[k][0], self.cls_convs[k][0]], [self.reg_convs[k][1], self.cls_convs[k][1]]
        if FUSE_TOWERS and TW.fusable(firsts + seconds):
            fz = TW.towers_nograd(x, firsts, seconds)                 # [reg_feat | cls_feat]
        else:
            fz = ops.empty_nhwc(x.shape[0], x.shape[1], x.shape[2], 2 * W, x.device)
            self.reg_convs[k][1](self.reg_convs[k][0](x), out=fz[..., :W])
            self.cls_convs[k][1](self.cls_convs[k][0](x), out=fz[..., W:])
        rp = self.reg_preds[k]
        reg_o = None if rp.weight.requires_grad else TF.conv_bias(fz[..., :W], rp.weight, rp.bias)
        return (x, fz, reg_o)

    def prefix_modules(self):
        """The modules frozen_prefix runs (what a captured graph of it depends on)."""
        return [self.stems, self.reg_convs, self.cls_convs, self.reg_preds]

    def forward(self, xin, labels=None, imgs=None, time_embedding=None, nms_thresh=0.5, lframe=0, gframe=32, resume=False):
        nc, W = self.num_classes, self.width
        nch = 5 + nc
        training = self.training
        grad = torch.is_grad_enabled() and training
        pre = xin if isinstance(xin, HeadPrefix) else self.frozen_prefix(xin)
        nlev = len(pre.levels)
        x0 = pre.levels[0][0]
        B, dev = x0.shape[0], x0.device
        geo = ops.LevelGeometry([lvl[0].shape[1:3] for lvl in pre.levels], self.strides)
        self.hw = [tuple(lvl[0].shape[1:3]) for lvl in pre.levels]

        raw_levels, cls_maps, reg_maps = [None] * nlev, [None] * nlev, [None] * nlev

        def towers(k):
            """The trained towers cls_convs2 / reg_convs2 of level k on the stem output (tscd_head.py:321-324)."""
            x = pre.levels[k][0]
            t0, t1 = self.cls_convs2[k], self.reg_convs2[k]
            fus = FUSE_TOWERS and not x.requires_grad and TW.fusable([t0[0], t1[0], t0[1], t1[1]])
            if fus and grad and TW.batch_stats(t0[0]) and all(p.requires_grad for p in graphs.parameters_of(t0) + graphs.parameters_of(t1)):
                cls_maps[k], reg_maps[k] = TW.tower_pair_train(x, t0, t1)
            elif fus and not grad:                   # (towers_nograd follows the modules' own BatchNorm mode)
                b2 = TW.towers_nograd(x, [t0[0], t1[0]], [t0[1], t1[1]])
                cls_maps[k], reg_maps[k] = b2[..., :W], b2[..., W:]
            else:
                cls_maps[k], reg_maps[k] = t0[1](t0[0](x)), t1[1](t1[0](x))

        def preds(k):
            x, fz, reg_o = pre.levels[k]
            reg_feat, cls_feat = fz[..., :W], fz[..., W:]
            if reg_o is None:
                reg_o = TF.conv_bias(reg_feat, self.reg_preds[k].weight, self.reg_preds[k].bias)
            raw_levels[k] = TF.pred_level(reg_feat, cls_feat, reg_o, self.obj_preds[k], self.cls_preds[k])   # [B,h,w,5+NC], autograd-tracked

        # Order of issue.  What the proposal COUNTS depend on (prediction convs on the frozen features, decode, selection, the
        # asynchronous copy of the counts) and the trained towers (which the counts do not depend on) are independent until the feature
        # gather; layout "4" runs the former on a side stream and the towers on the main stream: the copy leaves early, the
        # small launches fill in underneath the towers, and in backward only the prediction convs' (leaf) gradients run off the main
        # stream.  Layouts "1"-"3" put towers on side streams instead (see LEVEL_STREAM_LAYOUT).  Under DistributedDataParallel the
        # reducer keeps every parameter's AccumulateGrad node alive from its construction (on the main stream), autograd orders a
        # gradient produced on a side stream in front of that node with an event wait, and the reducer syncs a bucket's all-reduce with
        # the stream its hook runs on -- the main one (TSCD_ONE_GRAD_STREAM=1 keeps every gradient-producing kernel on the main stream).
        use_side = LEVEL_STREAMS and nlev > 1 and dev.type == "cuda" and not (grad and os.environ.get("TSCD_ONE_GRAD_STREAM") == "1")
        lay = LEVEL_STREAM_LAYOUT if use_side else "0"
        joins, on_main, head_stream = [], list(range(nlev)), None
        if use_side:
            main = torch.cuda.current_stream()
            fork = torch.cuda.Event()
            fork.record(main)
        if use_side and lay == "4":
            head_stream = _side_streams(dev, 1)[0]
            head_stream.wait_event(fork)
        elif use_side:
            where = {"1": [None] + list(range(nlev - 1)), "2": [0] + [1] * (nlev - 1)}.get(lay, list(range(nlev)))   # level -> side stream
            side = _side_streams(dev, 1 + max(w for w in where if w is not None))
            for i in sorted({w for w in where if w is not None}):
                side[i].wait_event(fork)
                with torch.cuda.stream(side[i]):
                    for k in range(nlev):
                        if where[k] == i:
                            # stream-safety rule (tscd_amd/ops.py): the level's stem output was allocated under the main stream and
                            # is read here; the towers' outputs are allocated under the side stream and read on the main one
                            ops.share(pre.levels[k][0], side[i])
                            towers(k)
                            ops.share(cls_maps[k], main)
                            ops.share(reg_maps[k], main)
                    ev = torch.cuda.Event()
                    ev.record(side[i])
                    joins.append(ev)
            on_main = [k for k in range(nlev) if where[k] is None]
        with (torch.cuda.stream(head_stream) if head_stream is not None else contextlib.nullcontext()):
            if head_stream is not None:          # the prefix tensors cross from the main stream (see the rule above)
                for lvl in pre.levels:
                    for t in lvl:
                        if t is not None:
                            ops.share(t, head_stream)
            for k in range(nlev):
                preds(k)
            with torch.no_grad():
                outputs, det, score = ops.decode_score([r.detach().contiguous() for r in raw_levels], geo, nc, training)
                sel_idx, sel_cnt = ops.select_proposals(score, 0.001, self.kwargs.get("minimal_limit", 0),
                                                        self.kwargs.get("maximal_limit", 0))
                counts_host = torch.empty((B,), dtype=torch.int32, pin_memory=True) if dev.type == "cuda" else None
                if counts_host is not None:
                    counts_host.copy_(sel_cnt, non_blocking=True)
                    counts_ready = torch.cuda.Event()
                    counts_ready.record()
            if head_stream is not None:
                joins.append(counts_ready)
                for t in list(raw_levels) + [outputs, det, score, sel_idx, sel_cnt]:      # ... and the results cross back
                    if t is not None:
                        ops.share(t, main)
        for k in on_main:                      # they run while the host waits for the counts
            towers(k)
        for ev in joins:
            main.wait_event(ev)
        if counts_host is not None:
            counts_ready.synchronize()                                 # sync #1: ragged sizes (the towers run meanwhile)
            ppf = counts_host.tolist()
        else:
            ppf = sel_cnt.cpu().tolist()
        N, nl = sum(ppf), sum(ppf[:lframe])
        self._dbg(outputs=outputs, det=det, score=score, sel_idx=sel_idx, preds_per_frame=ppf)
        if N == 0 and training:
            return torch.tensor(0), 0, 0, 0, 0, 1, 0, 0, 0
        off = [0]
        for n in ppf:
            off.append(off[-1] + n)
        frame_off = ops.h2d(off, torch.int32, dev)
        # pred_result rows, frame-major [N,7+NC], and the int32 (frame, anchor) of every proposal row
        pred_rows, b_idx32, a_idx32 = ops.gather_det(det, sel_idx, frame_off, N, rows=True)
        if not training and B == 1:
            pr = [pred_rows[off[i]:off[i + 1]] for i in range(B)]
            return pr, pr

        def gather(feature_maps):
            maps = list(feature_maps)               # [B,h,w,W] maps or W-channel slices of the towers' fused output buffers
            nf = len(maps) // 3
            if grad and any(m.requires_grad for m in maps):
                return _GatherRows.apply(geo, sel_idx, frame_off, N, W, nf, *maps)
            return ops.gather_rows([maps[3 * f:3 * f + 3] for f in range(nf)], geo, sel_idx, frame_off, N, W)

        if grad and TAP_GATHER and len(cls_maps) == 3 and all(m.requires_grad for m in cls_maps + reg_maps):
            tapped = _TapGather.apply(geo, sel_idx, frame_off, N, W, *cls_maps, *reg_maps)
            f_cls, f_reg, reg_maps = tapped[0], tapped[1], list(tapped[2:])      # the edge block reads the taps
        else:
            f_cls, f_reg = gather(cls_maps + reg_maps)
        if training:
            raw_all = torch.cat([r.reshape(B, -1, nch) for r in raw_levels], 1)              # [B,A,5+NC] logits with grad
        if grad:
            cls_scores = TF.cls_score(raw_all, b_idx32, a_idx32, pred_rows)                  # == pred_rows[:,5], differentiable
        else:
            cls_scores = pred_rows[:, 5]

        kw = dict(preds_per_frame=ppf, lframe=lframe, gframe=gframe, conf_sim_thresh=self.kwargs.get("conf_sim_thresh", 0.99))
        features_cls, features_reg = self.agg_iou(f_cls, f_reg, cls_scores, None, sim_thresh=self.sim_thresh, ave=self.ave,
                                                  use_mask=self.use_mask, **kw)
        # The matcher's cost matrices go to the host now; everything queued between prepare() and finish() -- the Haar edge
        # block, its gather, the first MCA and the class predictions, none of which depend on the assignments -- runs on
        # the device while the host solves them.
        prep = self.local_reg_matcher.prepare(features_reg, features_cls, ppf[:lframe], resume=resume)
        hook = getattr(self, "window_hook", None)
        if hook is not None and HOOK_EARLY:
            # the next clip's frozen pass (tscd.py: a graph replay on a low-priority stream) starts as soon as the cost matrices are on
            # their way: it then runs underneath the window's own small launches instead of behind them, and is over before the
            # backward starts (its tail used to overlap the first convs of the backward: 237 vs 155 us for the wavelet block's wgrad)
            hook()
            hook = None
        if training:
            # the label assignment only feeds the losses: queued behind the cost matrices, it runs inside the assignment window
            with torch.no_grad():
                fg_mask, matched_gt, pred_iou, num_fg_b, ngt_b = ops.simota_assign(outputs, labels, geo, nc, self.max_gt)
        edge_maps = [self.edge_enhance_reg[k][0](reg_maps[k]) for k in range(len(reg_maps))]
        (f_edge,) = gather(edge_maps)
        if self.debug is not None:
            self._dbg(agg_idx=[sel_idx[b, :ppf[b]] for b in range(B)], pred_rows=pred_rows, f_cls=f_cls, f_reg=f_reg,
                      f_edge=f_edge, edge_maps=edge_maps, reg_maps=reg_maps)
        def class_branch(fc, fr, cs):
            fcm, _ = self.agg(fc, fr, cs, None, sim_thresh=self.sim_thresh, ave=self.ave, use_mask=self.use_mask,
                              need_reg=False, **kw)
            return fcm, hip_linear(fcm, self.cls_pred)

        eager = grad and EAGER_CLASS_BRANCH
        if not eager:
            features_cls_matched, cls_preds = class_branch(f_cls, f_reg, cls_scores)
        if training:
            # ---------------- losses (get_matcher_refined_losses, tscd_head.py:1008-1203), dense + masked: everything that
            # does not involve the matcher's outputs is queued here, inside the assignment window ----------------
            # fused loss kernels (tscd_amd/csrc/loss.hip): sums + logit gradients in one pass each
            fg = fg_mask.bool()
            lab = labels.float().contiguous()
            dense = TF.dense_loss(raw_all.contiguous(), outputs, fg_mask, matched_gt, pred_iou, lab, nc)   # IoU / obj / cls sums
            # refined heads: local-frame proposals that SimOTA marked foreground
            row_b, row_a = b_idx32[:nl].contiguous(), a_idx32[:nl].contiguous()
            num_fg_local = num_fg_b[:lframe].sum().float()
            rowargs = (row_b, row_a, geo.A, nc, fg_mask, matched_gt, pred_iou, lab)

            def class_loss(cp):
                return TF.row_loss(*rowargs, cls_logit=cp)[0] / num_fg_local

            if eager:
                # first MCA -> cls_pred -> refined class loss, forward AND backward, queued inside the assignment window
                # (nothing downstream of the matcher feeds this branch); the main backward only rescales its gradients
                def branch(fc, fr, cs):
                    fcm, cp = class_branch(fc, fr, cs)
                    return class_loss(cp), (fcm.detach(), cp.detach())

                # seed = the loss scale of this step: `self.loss_scale` when the training loop provides it (a device tensor,
                # e.g. scaler.scale(ones)); otherwise a tracked copy of the upstream gradient of the previous step
                seed, track = self.loss_scale, False
                if seed is None:
                    seed, track = _eager_seed(dev), True
                loss_r_cls, features_cls_matched, cls_preds = eager_branch(
                    branch, [f_cls, f_reg, cls_scores], list(graphs.parameters_of(self.agg) + graphs.parameters_of(self.cls_pred)), seed, track)
            else:
                loss_r_cls = class_loss(cls_preds)
        if hook is not None:          # queued behind the window's own work: the next clip's frozen backbone pass (tscd.py)
            hook()
        matched = self.local_reg_matcher.finish(prep, f_reg, f_edge, time_embedding[:lframe])
        matched = f_reg[:nl] if matched is None else matched[-1]
        if self.debug is not None:
            self._dbg(agg_cls=features_cls_matched, agg_iou_cls=features_cls, agg_iou_obj=features_reg, cafm=matched,
                      hungarian=list(self.local_reg_matcher.last_hungarian))
        matched4 = hip_linear(matched, self.fc_reg_matcher)
        obj_refined = self.task_aligned(matched4, features_reg, ppf[:lframe])
        if obj_refined is None:
            obj_refined = features_reg
        m_obj = hip_linear(obj_refined, self.matcher_obj_pred)
        m_reg = hip_linear(matched4, self.matcher_reg_pred)
        if self.debug is not None:
            self._dbg(cls_preds=cls_preds, matcher_obj_preds=m_obj, matcher_reg_preds=m_reg, cls_scores=cls_scores, matched4=matched4,
                      obj_refined=obj_refined, hungarian_rows=list(self.local_reg_matcher.last_hungarian_rows))

        if not training:
            res, res_ori, keep = nms_frames(pred_rows[:nl], ppf[:lframe], nc, cls_preds, m_obj, m_reg, 0.001, nms_thresh,
                                            return_idx=True)
            self._dbg(nms_keep=keep)
            return res, res_ori

        # ---------------- refined-head losses (the rest was queued inside the assignment window) ----------------
        rows = TF.row_loss(*rowargs, obj_logit=m_obj.reshape(-1), reg_delta=m_reg, base=pred_rows[:nl])
        # normalisation by the foreground counts, the clip of the matched-objectness loss at 15 and the weighted total
        # (tscd_head.py:1157-1203) in one launch: st = [total, 3 iou, obj, cls, num_fg / num_gts, refined cls, matched obj, 6 matched iou]
        loss, st = TF.combine_losses(dense, rows, loss_r_cls, num_fg_b, ngt_b, lframe)
        if self.debug is not None:
            self._dbg(fg_mask=fg_mask, matched_gt=matched_gt, pred_iou=pred_iou, num_fg=num_fg_b,
                      refined_fg_rows=fg[b_idx32[:nl].long(), a_idx32[:nl].long()].sum())
        return (loss, st[1], st[2], st[3], 0.0, st[4], loss_r_cls, st[6], st[7])
