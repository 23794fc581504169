"""Autograd-aware building blocks on top of tscd_amd.ops.

Forward AND backward of the trainable head conv blocks are HIP kernels (conv_igemm.hip, conv_wgrad.hip): data
gradients reuse the forward MFMA kernel with dgrad-packed weights, weight gradients use the pixel-axis GEMM,
BatchNorm+SiLU backward is a fused two-pass kernel.

The frozen backbone / base head never build a graph (the reference freezes them: ovis_tscd_large.py:116-149).
"""
import torch

from . import ops


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def _nchw(t):
    """NHWC-logical tensor -> NCHW-logical view (channels_last memory), for aten conv backward."""
    return t.permute(0, 3, 1, 2)


def _pad8(dy):
    """dY with its channels zero-padded to a multiple of 8 (rows stay 16-byte aligned); dy itself when already aligned."""
    c = dy.shape[-1]
    if c % 8 == 0:
        return dy.contiguous()
    dyp = dy.new_zeros(dy.shape[:-1] + ((c + 7) // 8 * 8,))
    dyp[..., :c] = dy
    return dyp


def _conv_backward(dy, x, weight, stride, need_dx, need_dw, dy_padded=None):
    """Stride-1 'same' conv gradients on the HIP kernels: dgrad = the forward MFMA kernel with dgrad-packed weights,
    wgrad = conv_wgrad.hip (pixel-axis GEMM, transposing LDS reads, deterministic split-K)."""
    cout, cin, k, _ = weight.shape
    dy = dy.to(torch.float16)
    if stride == 2:
        # 3x3 / stride 2 / pad 1 (the downsampling convs of CSPDarknet and PAFPN, row N4): y[o] = sum_k x[2o+k-1] w[k], so dX is the
        # stride-1 data gradient of dY with zeros inserted between its pixels and dW the stride-1 weight gradient against it
        assert k == 3 and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 and cout % 8 == 0, "stride-2 backward: 3x3 convs on even extents"
        dy = ops.zero_insert2(dy.contiguous())
    else:
        assert stride == 1, "conv backward: stride 1 or 2"
    # prediction convs (Cout = 25 / 4 / 1): dY's channels zero-padded so rows stay 16-byte aligned
    dyp = dy_padded
    if dyp is None and cout % 8 and (need_dx or need_dw):
        dyp = _pad8(dy)
    dx = None
    if need_dx:
        if cout % 8:      # zero weight rows for the padded channels; packed per call (not a cached parameter)
            wp = weight.new_zeros((dyp.shape[-1], cin, k, k))
            wp[:cout] = weight
            dx, _ = ops.conv2d(dyp, ops.pack_conv_weight(wp, None, 1), cin, k, 1, ops.EPI_RAW)
        else:
            dx = ops.conv_dgrad(dy, weight)
    dw = None
    if need_dw:
        dyp = dy.contiguous() if dyp is None else dyp
        # x may carry zero-padded channels (Focus: 12 -> 16): their weight gradient is computed and dropped
        with ops.param_grad_stream([weight], x, dyp):
            # (registered for the pass's batched launch when the result's rows ARE the parameter's gradient: no padded input channels)
            gp = [weight] if (x.shape[-1] == cin and weight.dtype == torch.float32) else None
            dw = ops.conv_wgrad(x, dyp, x.shape[-1], dyp.shape[-1], k, grad_params=gp)[:cout, :cin].to(weight.dtype).contiguous()
    return dx, dw


class _ConvBnSiluTrain(torch.autograd.Function):
    """BaseConv in training mode: conv -> batch-stat BN (running stats updated) -> SiLU."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, stride, eps, momentum):
        wpk = ops.WEIGHTS.get(weight, x.shape[-1] if x.shape[-1] != weight.shape[1] else None)     # Focus: 12 -> 16 input channels
        cout, k = weight.shape[0], weight.shape[-1]
        out, raw, ss, mean, invstd = ops.conv_bn_act_train(x, wpk, cout, k, stride, gamma, beta, running_mean, running_var, eps,
                                                           momentum, ops.ACT_SILU, keep_raw=True)
        ctx.save_for_backward(x, weight, gamma, raw, ss, mean, invstd)
        ctx.stride = stride
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight, gamma, raw, ss, mean, invstd = ctx.saved_tensors
        draw, dgamma, dbeta = ops.bn_silu_bwd(dy.to(torch.float16), raw, ss, mean, invstd, gamma.float())
        dx, dw = _conv_backward(draw, x, weight, ctx.stride, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dx, dw, dgamma.to(gamma.dtype), dbeta.to(gamma.dtype), None, None, None, None, None


class _MaxPool5(torch.autograd.Function):
    """Stride-1 5x5 max pool (SPP), forward ops.maxpool, backward tscd_maxpool5_bwd (first maximum in scan order takes the gradient)."""

    @staticmethod
    def forward(ctx, x):
        B, H, W, C = x.shape
        y = ops.maxpool(x, ops.empty_nhwc(B, H, W, C, x.device), 5)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool5_bwd(x, dy.to(torch.float16).contiguous())


def maxpool5(x):
    return _MaxPool5.apply(x) if _needs_grad(x) else ops.maxpool(x, ops.empty_nhwc(*x.shape[:3], x.shape[3], x.device), 5)


class _Upsample2x(torch.autograd.Function):
    """Nearest-neighbour x2 upsample (PAFPN), forward copy_slice(upsample=True), backward the 2x2 block sums."""

    @staticmethod
    def forward(ctx, x):
        B, H, W, C = x.shape
        y = ops.empty_nhwc(B, 2 * H, 2 * W, C, x.device)
        ops.copy_slice(x, y, upsample=True)
        return y

    @staticmethod
    def backward(ctx, dy):
        return ops.upsample2_bwd(dy.to(torch.float16).contiguous())


def upsample2x(x):
    return _Upsample2x.apply(x)


def conv_bn_silu(x, weight, gamma, beta, running_mean, running_var, stride, training, out=None, residual=None,
                 eps=ops.BN_EPS, momentum=ops.BN_MOMENTUM, folded=None):
    """BaseConv forward (network_blocks.py:36-61).  x NHWC fp16.  ``out`` / ``residual`` (channel slices) are only
    allowed on the no-grad path.  ``folded``: eval-mode scale / shift [2,C] computed ahead of time (yolox.utils.fuse_model)."""
    if training and _needs_grad(x, weight, gamma, beta):
        assert out is None and residual is None
        return _ConvBnSiluTrain.apply(x, weight, gamma, beta, running_mean, running_var, stride, eps, momentum)
    cout, k = weight.shape[0], weight.shape[-1]
    cin_pad = x.shape[-1] if x.shape[-1] != weight.shape[1] else None   # Focus: 12 -> 16 channels
    wpk = ops.WEIGHTS.get(weight, cin_pad)
    if training:
        return ops.conv_bn_act_train(x, wpk, cout, k, stride, gamma, beta, running_mean, running_var, eps, momentum, ops.ACT_SILU,
                                     residual=residual, out=out)[0]
    ss = folded if folded is not None else ops.bn_fold(gamma, beta, running_mean, running_var, eps)
    if residual is None:
        y, _ = ops.conv2d(x, wpk, cout, k, stride, ops.EPI_BN_SILU, scale_shift=ss, out=out)
        return y
    if out is not None and out.data_ptr() == residual.data_ptr():      # writing back over the residual: raw output in its own buffer
        y, _ = ops.conv2d(x, wpk, cout, k, stride, ops.EPI_RAW)
        return ops.bn_act(y, ss, ops.ACT_SILU, residual=residual, out=out)
    y, _ = ops.conv2d(x, wpk, cout, k, stride, ops.EPI_RAW, out=out)
    return ops.bn_act(y, ss, ops.ACT_SILU, residual=residual)


class _ConvBias(torch.autograd.Function):
    """nn.Conv2d with bias (+ optional ReLU): prediction 1x1 convs and the wavelet block's filters."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        wpk = ops.WEIGHTS.get(weight)
        y, _ = ops.conv2d(x, wpk, weight.shape[0], weight.shape[-1], 1, ops.EPI_BIAS_RELU if relu else ops.EPI_BIAS,
                          bias=ops.f32(bias))
        ctx.save_for_backward(x, weight, y)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = dy.to(torch.float16)
        C = dy.shape[-1]
        db = dyp = None
        if C % 8 == 0 and (ctx.relu or ctx.needs_input_grad[2]):
            # ReLU mask and bias gradient in one pass over dy (one kernel instead of compare, multiply, cast and reduce)
            dy, db = ops.bias_relu_bwd(dy.contiguous(), y if ctx.relu else None)
        else:
            if ctx.relu:
                dy = dy * (y > 0)
            if ctx.needs_input_grad[2]:
                # prediction convs (C = 25 / 4 / 1): column sums on the zero-padded copy the weight gradient needs anyway
                dyp = _pad8(dy)
                db = ops.bias_relu_bwd(dyp)[1][:C]
        dx, dw = _conv_backward(dy.contiguous(), x, weight, 1, ctx.needs_input_grad[0], ctx.needs_input_grad[1], dy_padded=dyp)
        if db is not None:
            db = db.to(weight.dtype) if ctx.needs_input_grad[2] else None
        return dx, dw, db, None


class _PredLevel(torch.autograd.Function):
    """One pyramid level's raw prediction map cat([reg, obj, cls], -1) (tscd_head.py:213-239, 369-373) with the box branch given
    (frozen reg_preds, computed by the prefetched prefix) and the objectness / class 1x1 convs trainable on features that need no
    gradient: the backward splits the map's gradient once (ops.pred_grad_split: padded dY's + both bias gradients) and runs the two
    weight-gradient GEMMs beside the data-gradient chain -- instead of slicing, padding, summing and copying per conv."""

    @staticmethod
    def forward(ctx, reg_feat, cls_feat, reg_o, w_obj, b_obj, w_cls, b_cls):
        obj_o, _ = ops.conv2d(reg_feat, ops.WEIGHTS.get(w_obj), w_obj.shape[0], 1, 1, ops.EPI_BIAS, bias=ops.f32(b_obj))
        cls_o, _ = ops.conv2d(cls_feat, ops.WEIGHTS.get(w_cls), w_cls.shape[0], 1, 1, ops.EPI_BIAS, bias=ops.f32(b_cls))
        ctx.save_for_backward(reg_feat, cls_feat, w_obj, w_cls)
        ctx.bias_dtype = b_obj.dtype
        return torch.cat([reg_o, obj_o, cls_o], -1)

    @staticmethod
    def backward(ctx, d_raw):
        reg_feat, cls_feat, w_obj, w_cls = ctx.saved_tensors
        nc = w_cls.shape[0]
        d_raw = d_raw if d_raw.dtype == torch.float16 else d_raw.to(torch.float16)
        B, h, w, nch = d_raw.shape
        if not (d_raw.stride(3) == 1 and d_raw.stride(2) == nch and d_raw.stride(1) == w * nch):
            d_raw = d_raw.contiguous()
        dy_obj, dy_cls, db = ops.pred_grad_split(d_raw, nc)
        with ops.param_grad_stream([w_obj], reg_feat, dy_obj):
            dw_obj = ops.conv_wgrad(reg_feat, dy_obj, reg_feat.shape[-1], 8, 1, grad_params=[w_obj] if w_obj.dtype == torch.float32 else None)[:1].to(w_obj.dtype)
        with ops.param_grad_stream([w_cls], cls_feat, dy_cls):
            dw_cls = ops.conv_wgrad(cls_feat, dy_cls, cls_feat.shape[-1], dy_cls.shape[-1], 1, grad_params=[w_cls] if w_cls.dtype == torch.float32 else None)[:nc].to(w_cls.dtype)
        bd = ctx.bias_dtype
        return None, None, None, dw_obj, db[4:5].to(bd), dw_cls, db[5:5 + nc].to(bd)


def pred_level(reg_feat, cls_feat, reg_o, obj_conv, cls_conv):
    """raw [B,h,w,5+NC] = cat(reg_o, obj_conv(reg_feat), cls_conv(cls_feat)); the fused backward when only the two convs' parameters
    need gradients, the per-conv path otherwise."""
    ps = (obj_conv.weight, obj_conv.bias, cls_conv.weight, cls_conv.bias)
    fused = (torch.is_grad_enabled() and all(p is not None and p.requires_grad for p in ps)
             and not (reg_feat.requires_grad or cls_feat.requires_grad or reg_o.requires_grad)
             and obj_conv.weight.shape[0] == 1 and obj_conv.weight.shape[-1] == 1 and cls_conv.weight.shape[-1] == 1 and cls_conv.weight.shape[0] <= 35)
    if fused:
        return _PredLevel.apply(reg_feat, cls_feat, reg_o, *ps)
    return torch.cat([reg_o, conv_bias(reg_feat, obj_conv.weight, obj_conv.bias), conv_bias(cls_feat, cls_conv.weight, cls_conv.bias)], -1)


class _ClsScore(torch.autograd.Function):
    """sigmoid of every selected proposal's own class logit, gathered from the clip's raw map [B,A,5+NC] (== pred_rows[:, 5], with
    the gradient the reference keeps through it, tscd_head.py:1560-1600): one gather forward, one scatter backward -- torch's
    advanced-indexing backward sorts the indices first (two dozen small launches)."""

    @staticmethod
    def forward(ctx, raw_all, row_b, row_a, pred_rows):
        s = ops.cls_score_gather(raw_all, row_b, row_a, pred_rows)
        ctx.save_for_backward(s, row_b, row_a, pred_rows)
        ctx.shape = tuple(raw_all.shape)
        return s

    @staticmethod
    def backward(ctx, g):
        s, row_b, row_a, pred_rows = ctx.saved_tensors
        g = g if g.dtype == torch.float16 else g.to(torch.float16)
        return ops.cls_score_scatter(g.contiguous(), s, ctx.shape, row_b, row_a, pred_rows), None, None, None


def cls_score(raw_all, row_b, row_a, pred_rows):
    return _ClsScore.apply(raw_all.contiguous(), row_b, row_a, pred_rows)


def conv_bias(x, weight, bias, relu=False):
    if _needs_grad(x, weight, bias):
        return _ConvBias.apply(x, weight, bias, relu)
    wpk = ops.WEIGHTS.get(weight)
    y, _ = ops.conv2d(x, wpk, weight.shape[0], weight.shape[-1], 1, ops.EPI_BIAS_RELU if relu else ops.EPI_BIAS,
                      bias=ops.f32(bias))
    return y


class _HaarHF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.haar_hf(x)

    @staticmethod
    def backward(ctx, d):
        # adjoint of the analysis step = synthesis with LL = 0 (DWT_Function.backward, surrounding_extraction.py:30-41)
        return ops.haar_idwt(d.to(torch.float16).contiguous())


class _HaarIdwtMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hf, content):
        hf = hf.contiguous()
        ctx.save_for_backward(hf, content)
        return ops.haar_idwt_mul(hf, content)

    @staticmethod
    def backward(ctx, d):
        hf, content = ctx.saved_tensors
        dhf, dcontent = ops.haar_idwt_mul_bwd(d.to(torch.float16).contiguous(), hf, content)
        return dhf, dcontent


def haar_hf(x):
    return _HaarHF.apply(x) if _needs_grad(x) else ops.haar_hf(x)


def haar_idwt_mul(hf, content):
    return _HaarIdwtMul.apply(hf, content) if _needs_grad(hf, content) else ops.haar_idwt_mul(hf.contiguous(), content)


# ----------------------------------------------------------------------------------------------------
# fused losses (tscd_amd/csrc/loss.hip): forward sums and the gradients w.r.t. the logits come out of ONE pass; backward only
# scales the stored gradients by the upstream scalars
# ----------------------------------------------------------------------------------------------------
class _DenseLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw_all, outputs, fg_mask, matched_gt, pred_iou, labels, num_classes, geo):
        want = ctx.needs_input_grad[0]
        if geo is None:
            sums, d = ops.loss_dense(raw_all, outputs, fg_mask, matched_gt, pred_iou, labels, num_classes, want_grad=want)
            d_reg = None
        else:
            sums, d, d_reg = ops.loss_dense(raw_all, outputs, fg_mask, matched_gt, pred_iou, labels, num_classes, want_grad=want, geo=geo)
        ctx.save_for_backward(d, d_reg)
        ctx.nc = num_classes
        return sums

    @staticmethod
    def backward(ctx, g):
        d, d_reg = ctx.saved_tensors
        if d is None:
            return (None,) * 8
        nc = ctx.nc
        # channel 0 (objectness) scales with d sums[1], channels 1.. (classes) with d sums[2]; the 4 box channels get the IoU-loss
        # gradient (d sums[0]) when the box branch trains (geo given), none otherwise
        # (the products are formed in fp32 and rounded once -- ops.scale_grad: g = loss scale x weight / num_fg overflows fp16 by itself
        # at GradScaler's initial 65536 with a handful of foreground anchors, the product does not)
        g = g.float().contiguous()
        if d_reg is None:
            return ops.scale_grad(d, g, 1, 2, split=1, pad=4), None, None, None, None, None, None, None
        return torch.cat((ops.scale_grad(d_reg, g, 0), ops.scale_grad(d, g, 1, 2, split=1)), dim=-1), None, None, None, None, None, None, None


def dense_loss(raw_all, outputs, fg_mask, matched_gt, pred_iou, labels, num_classes, geo=None):
    """-> fp32 [3]: IoU-loss sum, objectness BCE sum, class BCE sum.  The IoU loss has a gradient (to the raw box logits, through the
    decode) only when ``geo`` (ops.LevelGeometry) is given: the TSCD head's boxes are detached, the base detector's train (row N4)."""
    return _DenseLoss.apply(raw_all, outputs, fg_mask, matched_gt, pred_iou, labels, num_classes, geo)


class _RowLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_logit, obj_logit, reg_delta, base, row_b, row_a, A, num_classes, fg_mask, matched_gt, pred_iou, labels):
        c = lambda t: None if t is None else t.to(torch.float16).contiguous()
        cls_l, obj_l, reg_d = c(cls_logit), c(obj_logit), c(reg_delta)
        want = any(ctx.needs_input_grad[:3])
        sums, d_cls, d_obj, d_reg = ops.loss_rows(row_b, row_a, A, num_classes, fg_mask, matched_gt, pred_iou, labels, cls_l, obj_l, reg_d,
                                                  base, want_grad=want)
        ctx.save_for_backward(d_cls, d_obj, d_reg)
        ctx.shapes = [None if t is None else t.shape for t in (cls_logit, obj_logit, reg_delta)]
        return sums

    @staticmethod
    def backward(ctx, g):
        d_cls, d_obj, d_reg = ctx.saved_tensors
        g = g.float().contiguous()          # product in fp32, rounded once (see _DenseLoss.backward)
        out = [None if d is None else ops.scale_grad(d, g, i).reshape(shp) for i, (d, shp) in enumerate(zip((d_cls, d_obj, d_reg), ctx.shapes))]
        return (out[0], out[1], out[2]) + (None,) * 9


def row_loss(row_b, row_a, A, num_classes, fg_mask, matched_gt, pred_iou, labels, cls_logit=None, obj_logit=None, reg_delta=None, base=None):
    """-> fp32 [3]: refined class BCE sum (rows that are SimOTA foreground), matched-objectness BCE sum (all rows), SmoothL1 sum
    of the box deltas (foreground rows); a part whose logits are None is 0."""
    return _RowLoss.apply(cls_logit, obj_logit, reg_delta, base, row_b, row_a, A, num_classes, fg_mask, matched_gt, pred_iou, labels)


class _CombineLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dense, rows, r_cls, num_fg_b, ngt_b, lframe):
        dev = dense.device
        out = ops.temp(8, dtype=torch.float32, device=dev)
        w = ops.temp(6, dtype=torch.float32, device=dev)
        rc = r_cls.detach().float().reshape(1)
        ops.call("tscd_loss_combine_f32", dense.data_ptr(), rows.data_ptr(), rc.data_ptr(), num_fg_b.data_ptr(), ngt_b.data_ptr(),
                 num_fg_b.numel(), int(lframe), out.data_ptr(), w.data_ptr(), ops._stream())
        ctx.save_for_backward(w)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g, _g_stats):
        (w,) = ctx.saved_tensors
        d = g * w
        return d[:3], d[3:], g, None, None, None


def combine_losses(dense, rows, r_cls, num_fg_b, ngt_b, lframe):
    """The scalar tail of get_matcher_refined_losses (tscd_head.py:1157-1203) -> (total loss (differentiable w.r.t. dense, rows, r_cls),
    stats fp32 [8] = total, 3 iou, obj, cls, num_fg / num_gts, refined cls, matched obj (clipped), 6 matched iou)."""
    return _CombineLosses.apply(dense.contiguous(), rows.contiguous(), r_cls, num_fg_b.contiguous(), ngt_b.contiguous(), lframe)
