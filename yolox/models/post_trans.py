"""MCA: global-to-local multi-head cosine attention (interface of the reference's yolox/models/post_trans.py
Attention_mca_g2l :550-714 and MCA_tscd_g2l_reg :1109-1162) on the fused MI355X kernel (tscd_amd/csrc/mca.hip).

Differences from the reference are internal: all local frames of the clip go through ONE fused launch (the
reference loops over frames and re-projects the shared global keys each time), no [heads, N1, N2] tensor is
materialised, the four projections and the output linears are MFMA GEMMs writing straight into the concat buffers.
"""
import torch
import torch.nn as nn

import os

from tscd_amd import ops

# Reference-faithful switch for the attention logits: the reference forms q.k^T, x 25 and x class score as fp16 tensors
# under autocast / .half() (post_trans.py:655-657) and feeds the rounded values to the softmax; the product's default keeps them in fp32
# (tests/test_deviation_bounds.py bounds the difference at < 1 % of the outputs' scale).  FP16_LOGITS = True (or
# TSCD_MCA_FP16_LOGITS=1) selects the reference's rounding in the forward kernels -- evaluation and training forward; the backward
# differentiates the unrounded expression (straight-through).
FP16_LOGITS = os.environ.get("TSCD_MCA_FP16_LOGITS", "0") == "1"


class Attention_mca_g2l(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., scale=25, reconf=False):
        super().__init__()
        if num_heads != 4:
            raise NotImplementedError("the fused MCA kernel maps the 4 heads of TSCD onto the 4 waves of a workgroup")
        self.num_heads = num_heads
        self.scale = scale
        self.reconf = reconf
        self.q_cls_local = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv_cls = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.q_reg_local = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv_reg = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.linear = nn.Linear(2 * dim, 2 * dim)
        if self.reconf:
            self.linear_reg = nn.Linear(2 * dim, 2 * dim)


class _MCAFunction(torch.autograd.Function):
    """Forward and backward of MCA_tscd_g2l_reg on the HIP kernels (mca.hip / mca_bwd.hip + MFMA GEMMs)."""

    @staticmethod
    def forward(ctx, cfg, x_cls, x_reg, score, wq_c, wkv_c, wq_r, wkv_r, w_lin, b_lin, w_linr, b_linr, w_out, b_out, w_obj, b_obj):
        ppf, lframe, scale, sim_thresh, conf_thresh, need_reg = cfg[:6]
        f16l = bool(cfg[6]) if len(cfg) > 6 else False
        ctx.set_materialize_grads(False)     # an unused output (agg_iou's cls branch is only used detached) gives None
        N, W = x_cls.shape
        nl = sum(ppf[:lframe])
        xc, xr, sc = x_cls.contiguous(), x_reg.contiguous(), score.to(torch.float16).contiguous()
        # the four input projections do not depend on each other: one job-table launch (ops.linear_jobs)
        q_c, kv_c, q_r, kv_r = ops.linear_jobs([(xc[:nl], wq_c, None, None), (xc, wkv_c, None, None), (xr[:nl], wq_r, None, None),
                                                (xr, wkv_r, None, None)])
        tiles = ops.mca_tiles(ppf, lframe, xc.device)
        want = any(ctx.needs_input_grad)
        # with a backward to come the forward's workspace stays alive: its normalised q / k / v rows are what the backward's first launch
        # would re-derive
        res = ops.mca_attention(q_c, kv_c, q_r, kv_r, sc, tiles, N, nl, W, need_reg, scale, sim_thresh, conf_thresh, want_stats=want,
                                keep_workspace=want, fp16_logits=f16l)
        xcat_c, cat3_c, xcat_r, cat3_r, stats = res[:5]
        fwd_ws = res[5] if want else None
        t_obj = None
        if need_reg:       # the class and the regression branch run side by side: two launches of two jobs instead of four
            ops.linear_jobs([(xcat_c, w_lin, b_lin, cat3_c[:, W:]), (xcat_r, w_linr, b_linr, cat3_r[:, W:])])
            t_cls, t_obj = ops.linear_jobs([(cat3_c, w_out, b_out, None), (cat3_r, w_obj, b_obj, None)])
        else:
            ops.linear(xcat_c, w_lin, b_lin, out=cat3_c[:, W:])
            t_cls = ops.linear(cat3_c, w_out, b_out)
        if want:
            ctx.cfg = cfg
            ctx.biases = (b_lin, b_linr, b_out, b_obj)     # parameter objects: name the bias gradients for the pass's batched launch
            ctx.save_for_backward(xc, xr, sc, q_c, kv_c, q_r, kv_r, tiles, stats, xcat_c, cat3_c, xcat_r, cat3_r, wq_c, wkv_c, wq_r, wkv_r,
                                  w_lin, w_linr, w_out, w_obj, fwd_ws)
        return (t_cls, t_obj) if need_reg else t_cls

    @staticmethod
    def backward(ctx, g_cls, g_obj=None):
        ppf, lframe, scale, sim_thresh, conf_thresh, need_reg = ctx.cfg[:6]
        (xc, xr, sc, q_c, kv_c, q_r, kv_r, tiles, stats, xcat_c, cat3_c, xcat_r, cat3_r, wq_c, wkv_c, wq_r, wkv_r, w_lin, w_linr, w_out,
         w_obj, fwd_ws) = ctx.saved_tensors
        N, W = xc.shape
        nl = q_c.shape[0]
        z16 = lambda *shape: ops.ztemp(shape, dtype=torch.float16, device=xc.device)

        b_lin, b_linr, b_out, b_obj = ctx.biases

        def head(g, cat3, xcat, w_o, w_l, b_o, b_l):
            """through linear(3W->4W) and mca.linear(2W->2W): -> d cat3 [nl,3W], d xcat [nl,2W], dW_o, db_o, dW_l, db_l"""
            if g is None:      # this output was not used downstream: no gradient for its two linears
                if need_reg and g_obj is not None and cat3 is cat3_c:
                    return None, None, None, None, None, None      # ... and none for its attention outputs: mca_backward leaves it out
                return z16(nl, 3 * W), z16(nl, 2 * W), None, None, None, None
            g = g.to(torch.float16).contiguous()
            d_cat3, dw_o, db_o = ops.linear_bwd(cat3, g, w_o, True, True, True, grad_params=[w_o, b_o])
            d_xcat, dw_l, db_l = ops.linear_bwd(xcat, d_cat3[:, W:], w_l, True, True, True, grad_params=[w_l, b_l])    # dY = a column slice of d_cat3
            return d_cat3, d_xcat, dw_o, db_o, dw_l, db_l

        dc3_c, dxc_c, dw_out, db_out, dw_lin, db_lin = head(g_cls, cat3_c, xcat_c, w_out, w_lin, b_out, b_lin)
        if need_reg:
            dc3_r, dxc_r, dw_obj, db_obj, dw_linr, db_linr = head(g_obj, cat3_r, xcat_r, w_obj, w_linr, b_obj, b_linr)
        else:
            dc3_r = dxc_r = dw_obj = db_obj = dw_linr = db_linr = None
        dq_c, dkv_c, dq_r, dkv_r, dscore = ops.mca_backward(q_c, kv_c, q_r, kv_r, sc, tiles, N, nl, W, need_reg, scale, sim_thresh,
                                                            conf_thresh, stats, dxc_c, dxc_r, dc3_c, dc3_r, cat3_c, cat3_r, fwd_ws=fwd_ws)
        dx_c, dw_kv_c, _ = ops.linear_bwd(xc, dkv_c, wkv_c, grad_params=[wkv_c])
        dxq_c, dw_q_c, _ = ops.linear_bwd(xc[:nl], dq_c, wq_c, grad_params=[wq_c])
        dx_c[:nl] += dxq_c
        dx_r, dw_kv_r, _ = ops.linear_bwd(xr, dkv_r, wkv_r, grad_params=[wkv_r])
        dxq_r, dw_q_r, _ = ops.linear_bwd(xr[:nl], dq_r, wq_r, grad_params=[wq_r])
        dx_r[:nl] += dxq_r
        like = lambda g, w: None if g is None else g.to(w.dtype).reshape(w.shape)
        grads = (None, dx_c, dx_r, dscore.to(sc.dtype), like(dw_q_c, wq_c), like(dw_kv_c, wkv_c), like(dw_q_r, wq_r), like(dw_kv_r, wkv_r),
                 like(dw_lin, w_lin), db_lin, like(dw_linr, w_linr) if need_reg else None, db_linr,
                 like(dw_out, w_out), db_out, like(dw_obj, w_obj) if need_reg else None, db_obj)
        return grads


class MCA_tscd_g2l_reg(nn.Module):
    def __init__(self, dim, out_dim, num_heads=4, qkv_bias=False, attn_drop=0., scale=25, reconf=False):
        super().__init__()
        self.reconf = reconf
        self.mca = Attention_mca_g2l(dim, num_heads, qkv_bias, attn_drop, scale=scale, reconf=reconf)
        self.linear = nn.Linear(3 * dim, out_dim)
        if reconf:
            self.linear_obj = nn.Linear(3 * dim, out_dim)

    def forward(self, x_cls, x_reg, cls_score=None, fg_score=None, sim_thresh=0.75, ave=True, use_mask=False, **kwargs):
        """x_cls / x_reg: [1,N,W] (reference layout) or [N,W]; the first sum(preds_per_frame[:lframe]) rows are the local
        frames' proposals.  Returns (trans_cls [N_L,out], trans_obj [N_L,out] or None).  ``need_reg=False`` (private
        kwarg used by the head for ``agg``, whose second output the reference discards) skips the regression branch."""
        if not ave or use_mask:
            raise NotImplementedError("TSCD runs MCA with ave=True, use_mask=False (tscd_base.py:49,61)")
        ppf, lframe = list(kwargs.get("preds_per_frame")), kwargs.get("lframe")
        need_reg = bool(kwargs.get("need_reg", True)) and self.reconf
        conf_thresh = kwargs.get("conf_sim_thresh", 0.99)
        x_cls = x_cls[0] if x_cls.dim() == 3 else x_cls
        x_reg = x_reg[0] if x_reg.dim() == 3 else x_reg
        m = self.mca
        params = [m.q_cls_local.weight, m.kv_cls.weight, m.q_reg_local.weight, m.kv_reg.weight, m.linear.weight, m.linear.bias,
                  m.linear_reg.weight if need_reg else None, m.linear_reg.bias if need_reg else None,
                  self.linear.weight, self.linear.bias, self.linear_obj.weight if need_reg else None,
                  self.linear_obj.bias if need_reg else None]
        cfg = (ppf, lframe, float(m.scale), float(sim_thresh), float(conf_thresh), need_reg, bool(kwargs.get("fp16_logits", FP16_LOGITS)))
        out = _MCAFunction.apply(cfg, x_cls.to(torch.float16), x_reg.to(torch.float16), cls_score, *params)
        return out if need_reg else (out, None)
