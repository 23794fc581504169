"""CAFM (AwarePositionRegMatcher) and TaskAligned on the MI355X kernels.

Interface mirror of the reference's yolox/models/tscd_matching.py (PositionMHAttention :11-137, MHAttention :140-181,
SEModule :264-283, SelfAttentionLayer :286-357, CrossAttentionLayer :394-458, FFNLayer :461-505,
ReferringCrossAttentionLayer :535-636, AwarePositionRegMatcher :639-1080, TaskAligned :1083-1139): same submodule
tree (including the layers the reference constructs but never calls -- they carry state_dict keys), same stateful
frame-to-frame behaviour.  Device work per local frame: one MFMA GEMM for the Hungarian cost, the SE gate,
q/k/v GEMMs, the cosine attention kernel and the fused residual+LayerNorm; the assignment itself stays on the host like
the reference's scipy call, solved by tscd_amd/csrc_host/lap.cpp (same algorithm and tie rules, SIMD inner loop) -- all
cost matrices of a clip reach the host with ONE D2H.
"""
import functools
from concurrent.futures import ThreadPoolExecutor
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from tscd_amd import ops
from tscd_amd._host import KIND_PREV, KIND_SELF, KIND_STATE, LATCH_WAIT_CB, Latch, lap_chain_f16, lap_f16


KEY_STATIONARY_BWD = __import__("os").environ.get("TSCD_ATTN_KEY_STATIONARY_BWD", "1") != "0"   # cosine attention backward: two launches, no atomics
FUSED_CHAIN = __import__("os").environ.get("TSCD_FUSED_CAFM", "1") != "0"     # _CafmChain: the clamped training regime as one autograd node


# the clip's chain of host assignments runs on this worker while the main thread keeps queueing device work
_ASSIGN_WORKER = ThreadPoolExecutor(max_workers=1, thread_name_prefix="tscd-lap")


_PINNED = {}            # device index -> pinned fp16 staging buffer of the cost matrices (grow-only)
# The clamped training regime does not wait for the assignments on the host: the stream is parked on a latch the worker raises after
# writing the clip's index table to pinned memory, and CAFM, the refined heads and the whole backward are queued behind it while the
# chain is still being solved (TSCD_DEFER_ASSIGNMENTS=0: join the chain on the host first, as every other regime does).
DEFER_CHAIN = __import__("os").environ.get("TSCD_DEFER_ASSIGNMENTS", "1") != "0"
_TABLES = {}            # device index -> pinned int64 index table the worker writes (grow-only; always holds in-range rows)
_TABLE_N = {}           # device index -> frame size n the table's rows are known to be in range for
_LATCH = None           # one latch for the process (generations only grow)
_TIMEOUTS_SEEN = 0      # latch timeouts already reported by _join_pending


def _latch():
    global _LATCH
    if _LATCH is None:
        _LATCH = Latch()
    return _LATCH


def _wait_copy(device_index, event):
    """Runs on the worker: bind the thread to the rank's GPU (a fresh thread defaults to device 0) and wait for the copy."""
    torch.cuda.set_device(device_index)
    event.synchronize()


def _f32(t):
    return ops.f32(t)


# ------------------------------------------------------------------------------------------------- device ops
class _Linear(torch.autograd.Function):
    """nn.Linear on the MFMA GEMM, forward and backward: dX = dY W (same kernel, transposed weight), dW = dY^T X (pixel-axis
    GEMM of conv_wgrad.hip), db = column sums."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.bias_param = b               # (the parameter object: names the bias gradient for the pass's batched launch)
        return ops.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        if g.dtype != torch.float16:
            g = g.to(torch.float16)
        need = ctx.needs_input_grad
        dx, dw, db = ops.linear_bwd(x, g, w, need[0], need[1], ctx.has_bias and need[2], grad_params=[w, ctx.bias_param if ctx.has_bias and need[2] else None])
        if dw is not None and dw.dtype != w.dtype:
            dw = dw.to(w.dtype)
        if db is not None and db.dtype != w.dtype:
            db = db.to(w.dtype)
        return dx, dw, db


class _LinearJobs(torch.autograd.Function):
    """Several independent nn.Linear calls (x_i, w_i, b_i) whose forward GEMMs run as ONE job-table launch (ops.linear_jobs); the backward
    is each layer's own (the weight / bias gradients join the pass's batched launch as for _Linear)."""

    @staticmethod
    def forward(ctx, n, *args):
        xs = [a.contiguous() for a in args[:n]]
        ws, bs = args[n:2 * n], args[2 * n:3 * n]
        ctx.n = n
        ctx.bias_params = bs
        ctx.save_for_backward(*xs, *ws)
        return tuple(ops.linear_jobs([(x, w, b, None) for x, w, b in zip(xs, ws, bs)]))

    @staticmethod
    def backward(ctx, *gs):
        n = ctx.n
        xs, ws = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        need = ctx.needs_input_grad
        dxs, dws, dbs = [], [], []
        for i in range(n):
            g, x, w, b = gs[i], xs[i], ws[i], ctx.bias_params[i]
            if g is None:
                dxs.append(None); dws.append(None); dbs.append(None)
                continue
            if g.dtype != torch.float16:
                g = g.to(torch.float16)
            want_b = b is not None and need[1 + 2 * n + i]
            dx, dw, db = ops.linear_bwd(x, g.contiguous(), w, need[1 + i], need[1 + n + i], want_b, grad_params=[w, b if want_b else None])
            dxs.append(dx)
            dws.append(dw if dw is None or dw.dtype == w.dtype else dw.to(w.dtype))
            dbs.append(db if db is None or db.dtype == w.dtype else db.to(w.dtype))
        return (None, *dxs, *dws, *dbs)


def linears(pairs):
    """[(x, nn.Linear)] -> outputs; independent layers, one forward launch where the shapes allow (ops.linear_jobs)."""
    n = len(pairs)
    xs, ws, bs = [x for x, _ in pairs], [l.weight for _, l in pairs], [l.bias for _, l in pairs]
    if torch.is_grad_enabled() and any(t.requires_grad for t in xs + ws):
        return _LinearJobs.apply(n, *xs, *ws, *bs)
    return tuple(ops.linear_jobs([(x.contiguous(), w, b, None) for x, w, b in zip(xs, ws, bs)]))


def linear(x, lin):
    """nn.Linear on the MFMA GEMM (fp16 in/out)."""
    w, b = lin.weight, lin.bias
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        return _Linear.apply(x, w, b)
    return ops.linear(x.contiguous(), w, b)


class _CosAttention(torch.autograd.Function):
    """Scalar-kernel path (head dims that are not a multiple of 16)."""

    @staticmethod
    def forward(ctx, q, k, v, heads, tiles):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        ctx.save_for_backward(q, k, v, tiles)
        ctx.heads = heads
        return ops.cos_attn(q, k, v, heads, tiles)

    @staticmethod
    def backward(ctx, g):
        q, k, v, tiles = ctx.saved_tensors
        dq, dk, dv = ops.cos_attn_bwd(q, k, v, g.to(torch.float16).contiguous(), ctx.heads, tiles)
        return dq, dk, dv, None, None


class _CosAttentionMFMA(torch.autograd.Function):
    """Matrix-core path (attn_mfma.hip): the forward also returns the softmax denominators the backward re-uses."""

    @staticmethod
    def forward(ctx, q, k, v, heads, tiles32, full, ktiles32=None, keys_full=False):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out, lsum = ops.cos_attn_mfma(q, k, v, heads, tiles32, full)
        ctx.save_for_backward(q, k, v, tiles32, out, lsum)
        ctx.heads, ctx.full, ctx.ktiles, ctx.keys_full = heads, full, ktiles32, keys_full
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, tiles32, out, lsum = ctx.saved_tensors
        dq, dk, dv = ops.cos_attn_mfma_bwd(q, k, v, g.to(torch.float16).contiguous(), out, lsum, ctx.heads, tiles32, ctx.full,
                                           ktiles32=ctx.ktiles, keys_full=ctx.keys_full)
        return dq, dk, dv, None, None, None, None, None


def cos_attention(q, k, v, heads, segments):
    """softmax(q^ k^T) v per head, rows grouped in segments [(row0, nrows, key0, key1)]."""
    grad = torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)
    if ops.cos_attn_uses_mfma(q.shape[1], heads):
        tiles = ops.attn_tiles(segments, q.device, rows=32)
        full = sum(seg[1] for seg in segments) == q.shape[0]      # every query row is in some tile: outputs need no zero fill
        if grad:
            # key-stationary backward (no atomics) when no key row is shared between segments
            ktiles = ops.attn_key_tiles(segments, q.device) if KEY_STATIONARY_BWD else None
            keys_full = ktiles is not None and sum(seg[3] - seg[2] for seg in segments if seg[1] > 0) == k.shape[0]
            return _CosAttentionMFMA.apply(q, k, v, heads, tiles, full, ktiles, keys_full)
        return ops.cos_attn_mfma(q.contiguous(), k.contiguous(), v.contiguous(), heads, tiles, full)[0]
    tiles = ops.attn_tiles(segments, q.device)
    if grad:
        return _CosAttention.apply(q, k, v, heads, tiles)
    return ops.cos_attn(q.contiguous(), k.contiguous(), v.contiguous(), heads, tiles)


class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, gamma, beta, eps):
        a = a.contiguous()
        b = None if b is None else b.contiguous()
        g32 = _f32(gamma).contiguous()
        ctx.save_for_backward(a, b, g32)
        ctx.eps, ctx.pdtype = eps, gamma.dtype
        return ops.add_layernorm(a, b, g32, _f32(beta).contiguous(), eps)

    @staticmethod
    def backward(ctx, g):
        a, b, g32 = ctx.saved_tensors
        dx, dgamma, dbeta = ops.add_layernorm_bwd(a, b, g32, g.to(torch.float16).contiguous(), ctx.eps)
        need = ctx.needs_input_grad
        return (dx if need[0] else None, dx if b is not None and need[1] else None,
                dgamma.to(ctx.pdtype) if need[2] else None, dbeta.to(ctx.pdtype) if need[3] else None, None)


def add_layernorm(a, b, norm):
    return _AddLayerNorm.apply(a, b, norm.weight, norm.bias, norm.eps)


class _SEFuse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, reg, edge, pos, w0, w2):
        w0f, w2f = _f32(w0).contiguous(), _f32(w2).contiguous()
        ctx.save_for_backward(reg, edge, w0f, w2f)
        ctx.pos_shape, ctx.pos_dtype, ctx.wdtype = pos.shape, pos.dtype, w0.dtype
        return ops.se_fuse(reg, edge, w0f, w2f, _f32(pos).reshape(-1).contiguous())

    @staticmethod
    def backward(ctx, g):
        reg, edge, w0f, w2f = ctx.saved_tensors
        d_reg, d_edge, d_pos, d_w0, d_w2 = ops.se_fuse_bwd(reg, edge, w0f, w2f, g.to(torch.float16).contiguous())
        need = ctx.needs_input_grad
        return (d_reg if need[0] else None, d_edge if need[1] else None,
                d_pos.reshape(ctx.pos_shape).to(ctx.pos_dtype) if need[2] else None,
                d_w0.to(ctx.wdtype) if need[3] else None, d_w2.to(ctx.wdtype) if need[4] else None)


class MHAttention(nn.Module):
    def __init__(self, d_model, num_heads=8, dropout=0., qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.q_reg = nn.Linear(d_model, d_model, bias=qkv_bias)
        self.k_reg = nn.Linear(d_model, d_model, bias=qkv_bias)
        self.v_reg = nn.Linear(d_model, d_model, bias=qkv_bias)
        self.attn_drop = nn.Dropout(dropout)

    def attend(self, q_in, k_in, v_in, segments):
        q, k, v = linears([(q_in, self.q_reg), (k_in, self.k_reg), (v_in, self.v_reg)])
        return cos_attention(q, k, v, self.num_heads, segments)


class PositionMHAttention(MHAttention):
    """Only the boxes=None path is live on the TSCD route (tscd_matching.py:49-50); position_embedding is a parameter
    holder."""

    def __init__(self, d_model, num_heads=8, dropout=0., qkv_bias=False):
        super().__init__(d_model, num_heads, dropout, qkv_bias)
        self.position_embedding = nn.Conv2d(64, num_heads, kernel_size=1, stride=1, padding=0)


class SEModule(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(channels, channels * 16, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(channels * 16, channels, bias=False), nn.Sigmoid())

    def fuse(self, reg, edge, pos):
        """reg*g0 + edge*g1 + pos  (SE gate per (proposal, channel) + with_pos_embed), reg/edge [n,C], pos [1,C] or [C]."""
        return _SEFuse.apply(reg, edge, pos, self.fc[0].weight, self.fc[2].weight)


def _xavier(mod):
    for p in mod.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)


class SelfAttentionLayer(nn.Module):
    """Constructed by the reference but never called on the TSCD path -- parameter holder."""

    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        self.self_attn = PositionMHAttention(d_model, nhead, dropout=dropout)
        self.norm = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self.CA = SEModule(2)
        _xavier(self)


class FFNLayer(nn.Module):
    """Constructed by the reference but never called on the TSCD path -- parameter holder."""

    def __init__(self, d_model, dim_feedforward=2048, dropout=0.0, activation="silu", normalize_before=False):
        super().__init__()
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)
        _xavier(self)


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0, activation="silu", normalize_before=False):
        super().__init__()
        assert not normalize_before
        self.multihead_attn = MHAttention(d_model, nhead, dropout=dropout)
        self.norm = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        _xavier(self)

    def forward(self, tgt, memory, segments):
        return add_layernorm(tgt, self.multihead_attn.attend(tgt, memory, memory, segments), self.norm)


class ReferringCrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        assert not normalize_before
        self.multihead_attn = PositionMHAttention(d_model, nhead, dropout=dropout)
        self.norm = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self.CA = SEModule(2)
        _xavier(self)

    def forward(self, identity, tgt, memory, pos, query_pos, edge, query_edge):
        n = tgt.shape[0]
        q_in = self.CA.fuse(tgt, query_edge, query_pos)
        k_in = self.CA.fuse(memory, edge, pos)
        t2 = self.multihead_attn.attend(q_in, k_in, memory, [(0, n, 0, memory.shape[0])])
        return add_layernorm(identity, t2, self.norm)


class _PermuteRows(torch.autograd.Function):
    """t[idx] for idx a permutation of all rows of t, idx_inv its inverse: backward = grad[idx_inv]."""

    @staticmethod
    def forward(ctx, t, idx, idx_inv):
        ctx.save_for_backward(idx_inv)
        return t.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx_inv,) = ctx.saved_tensors
        return g.index_select(0, idx_inv), None, None


class _TableFeed:
    """The deferred join, frame by frame: ``wait(j)`` parks the current stream on the worker's latch until matrix j of the clip's chain
    is solved (tscd_lap_chain_f16_table raises one generation per matrix) and queues the copy of that frame's part of the index table
    from pinned memory -- [c_j | inv_j], with inv_0 riding along with frame 1 (or alone for a one-frame clip)."""

    def __init__(self, table, dev_idx, latch, gen0, n, L):
        self.table, self.dev_idx, self.latch, self.gen0, self.n, self.L = table, dev_idx, latch, gen0, n, L

    def wait(self, j):
        n = self.n
        a, b = ((0, n) if self.L == 1 else (0, 3 * n)) if j <= 1 else ((2 * j - 1) * n, (2 * j + 1) * n)
        ops.call("tscd_stream_host_func", ops._stream(), LATCH_WAIT_CB, self.latch.ticket(self.gen0 + j))
        self.dev_idx[a:b].copy_(self.table[a:b], non_blocking=True)


class _CafmChain(torch.autograd.Function):
    """The whole frame-to-frame chain of AwarePositionRegMatcher.forward (tscd_matching.py:761-888) for a clip whose local frames all
    carry the same number of proposals (the clamped regime: every frame at ``maximal_limit``) as ONE autograd node: the per-frame
    SE gates, q / k / v projections, cosine attention and residual LayerNorms are the same kernels the per-op path launches, but the
    k / v projections of all frames run as one GEMM each (they do not depend on the recurrence), the row permutations are written
    straight into their destinations, and the backward walks the chain in reverse without autograd's bookkeeping (its ~50 small
    nodes per clip -- slices, cats, index gradients, dtype casts -- were host time on the critical path behind the assignments).

    inputs: feat / edge [nl, W] fp16 (rows of the local frames, frame-major), temb [L, 256], index table (int64, device) with per
    frame c (assignment order) and inv (its inverse), then the parameters.  ``meta`` = (n per frame, L, heads, offsets of c / inv in
    the index table, eps of the two LayerNorms)."""

    @staticmethod
    def forward(ctx, meta, idx, feat, edge, temb, w_pos, b_pos, se_w0, se_w2, wq, wk, wv, ln_g, ln_b, dec_g, dec_b):
        n, L, heads, slots, eps_ln, eps_dec, feed = meta
        W = feat.shape[1]
        dev = feat.device
        f32 = lambda t: t if t.dtype == torch.float32 else t.float()
        w0f, w2f = f32(se_w0).contiguous(), f32(se_w2).contiguous()
        g_ln, b_ln, g_dec, b_dec = f32(ln_g).contiguous(), f32(ln_b).contiguous(), f32(dec_g).contiguous(), f32(dec_b).contiguous()
        feat, edge = feat.contiguous(), edge.contiguous()
        nl = n * L
        temb16 = temb.to(torch.float16).contiguous()
        t_emb = ops.linear(temb16, w_pos, b_pos).float()                   # [L, W] fp32 rows (the SE gate takes fp32 position vectors)
        # The row permutations (assignment order c_j, its inverse inv_j) are not materialised: the kernels that consume / produce the
        # rows read / write them through a row map into the clip's index table (ops.row_map; 18 gather launches per step before).
        c_off, inv_off = [sl[0] for sl in slots], [sl[1] for sl in slots]
        inv_stride = inv_off[1] - inv_off[0] if L > 1 else 0
        assert all(inv_off[j] == inv_off[0] + j * inv_stride for j in range(L)), "index table: the inverse permutations are not evenly spaced"
        cmap = lambda j: ops.row_map(idx, n, c_off[j])
        # key side of every frame: k_in = SE(cur, cur_edge) + t_j (one launch: a position vector per frame), then ONE projection each
        # for k and v over all local rows
        k_in = ops.temp((nl, W), dtype=torch.float16, device=dev)
        ops.se_fuse_ex(feat, edge, w0f, w2f, t_emb, k_in, nl, pos_seg_rows=n, pos_stride=W)
        k_all, v_all = ops.linear_jobs([(k_in, wk, None, None), (feat, wv, None, None)])
        mfma = ops.cos_attn_uses_mfma(W, heads)
        tiles = ops.attn_tiles([(0, n, 0, n)], dev, rows=32 if mfma else 16)
        ctx.ktiles = ops.attn_key_tiles([(0, n, 0, n)], dev) if mfma and KEY_STATIONARY_BWD else None
        q_in = ops.temp((nl, W), dtype=torch.float16, device=dev)
        q_all = ops.temp((nl, W), dtype=torch.float16, device=dev)
        t2_all = ops.temp((nl, W), dtype=torch.float16, device=dev)
        out_all = ops.temp((nl, W), dtype=torch.float16, device=dev)    # per-frame outputs in assignment order (= the next frame's tgt)
        lsums = []
        for j in range(L):
            s0, s1 = j * n, (j + 1) * n
            if feed is not None and j >= 1:
                feed.wait(j)       # frame j's assignment order: the stream goes on as soon as ITS matrix is solved (frame 0 needs none)
            # query side: tgt = the previous frame's output (assignment order), its edge rows = the previous frame's in ITS assignment
            # order (frame 1 reads frame 0's in place: frame 0 keeps its own order)
            tgt = feat[:n] if j == 0 else out_all[s0 - n:s0]
            qedge = edge[:n] if j <= 1 else edge[s0 - n:s0]
            ops.se_fuse_ex(tgt, qedge, w0f, w2f, t_emb[max(j - 1, 0)], q_in[s0:s1], n, edge_map=cmap(j - 1) if j >= 2 else None)
            ops.linear(q_in[s0:s1], wq, out=q_all[s0:s1])
            if mfma:
                lsums.append(ops.cos_attn_mfma(q_all[s0:s1], k_all[s0:s1], v_all[s0:s1], heads, tiles, True, out=t2_all[s0:s1])[1])
            else:
                t2_all[s0:s1].copy_(ops.cos_attn(q_all[s0:s1], k_all[s0:s1], v_all[s0:s1], heads, tiles))
            # residual input: cur (first frame) / cur[c_j]
            if j == 0:
                ops.call("tscd_add_layernorm_f16", feat.data_ptr(), t2_all.data_ptr(), g_ln.data_ptr(), b_ln.data_ptr(), n, W, eps_ln,
                         out_all.data_ptr(), ops._stream())
            else:
                ops.add_layernorm_ex(feat[s0:s1], t2_all[s0:s1], g_ln, b_ln, eps_ln, out_all[s0:s1], n, cmap(j))
        if feed is not None and L == 1:
            feed.wait(0)
        # back to proposal order inside the decoder norm (row-wise, so it commutes with the permutation): row i of frame j reads
        # out_j[inv_j[i]] (after the loop: frame 0 then never waits for its own self-assignment)
        matched = ops.temp((nl, W), dtype=torch.float16, device=dev)
        ops.add_layernorm_ex(out_all, None, g_dec, b_dec, eps_dec, matched, nl, ops.row_map(idx, n, inv_off[0], inv_stride))
        ctx.meta = meta[:6]
        ctx.mfma = mfma
        ctx.dtypes = (w_pos.dtype, se_w0.dtype, wq.dtype, ln_g.dtype, dec_g.dtype)
        ctx.save_for_backward(idx, feat, edge, temb16, t_emb, w_pos, w0f, w2f, wq, wk, wv, g_ln, g_dec, k_in, k_all, v_all, q_in, q_all,
                              t2_all, out_all, tiles, *lsums)
        return matched

    @staticmethod
    def backward(ctx, d_matched):
        n, L, heads, slots, eps_ln, eps_dec = ctx.meta
        (idx, feat, edge, temb16, t_emb, w_pos, w0f, w2f, wq, wk, wv, g_ln, g_dec, k_in, k_all, v_all, q_in, q_all, t2_all,
         out_all, tiles, *lsums) = ctx.saved_tensors
        W = feat.shape[1]
        dev = feat.device
        nl = n * L
        c_off, inv_off = [sl[0] for sl in slots], [sl[1] for sl in slots]
        inv_stride = inv_off[1] - inv_off[0] if L > 1 else 0
        cmap = lambda j: ops.row_map(idx, n, c_off[j])
        h16 = lambda t: t if t.dtype == torch.float16 else t.to(torch.float16)
        mk = lambda: ops.temp((nl, W), dtype=torch.float16, device=dev)
        dk_all, dv_all, dq_all = mk(), mk(), mk()
        d_out_all = mk()                                     # d loss / d out_j of every frame, in assignment order
        d_idn = mk()                                         # gradient of the residual inputs, in proposal order (added to d_feat at the end)
        d_edge = mk()                                        # query-side contributions to the edge rows, in proposal order
        # ONE zeroed arena for every fp32 parameter-gradient accumulator of the chain (the kernels ADD into their part): per frame the
        # query-side gate's [d pos | d w0 | d w2], the key-side gate's, the residual LayerNorm's [d gamma, d beta]; the decoder norm's
        SW = W + 128
        arena = ops.ztemp((2 * L * SW + (L + 1) * 2 * W,), dtype=torch.float32, device=dev)
        A, K = arena[:L * SW].view(L, SW), arena[L * SW:2 * L * SW].view(L, SW)
        ln_acc = arena[2 * L * SW:].view(L + 1, 2, W)
        # decoder norm: matched[i of frame j] = LN(out_j[inv_j[i]]) -- dx goes straight back to row inv_j[i] of frame j
        ops.add_layernorm_bwd_ex(out_all, None, g_dec, h16(d_matched).contiguous(), eps_dec, None, nl, ln_acc[L],
                                 a_map=ops.row_map(idx, n, inv_off[0], inv_stride), dx_mapped=d_out_all)
        dg_dec, db_dec = ln_acc[L, 0], ln_acc[L, 1]
        carry_out = None        # d loss / d out_j (assignment order) arriving from frame j+1's query path
        d_tgt0 = d_qedge0 = None
        for j in range(L - 1, -1, -1):
            s0, s1 = j * n, (j + 1) * n
            # residual input: cur (first frame: its gradient is already in proposal order) or cur[c_j] (dx also lands at row c_j[r] of d_idn);
            # d out_j = the decoder norm's part + what frame j+1's query path sent back
            dx = d_idn[s0:s1] if j == 0 else ops.temp((n, W), dtype=torch.float16, device=dev)
            ops.add_layernorm_bwd_ex(feat[s0:s1], t2_all[s0:s1], g_ln, d_out_all[s0:s1], eps_ln, dx, n, ln_acc[j],
                                     a_map=cmap(j) if j > 0 else None, dy_add=carry_out, dx_mapped=d_idn[s0:s1] if j > 0 else None)
            if ctx.mfma:
                ops.cos_attn_mfma_bwd(q_all[s0:s1], k_all[s0:s1], v_all[s0:s1], dx, t2_all[s0:s1], lsums[j], heads, tiles, True,
                                      outs=(dq_all[s0:s1], dk_all[s0:s1], dv_all[s0:s1]), ktiles32=ctx.ktiles, keys_full=True)
            else:
                dq, dk, dv = ops.cos_attn_bwd(q_all[s0:s1], k_all[s0:s1], v_all[s0:s1], dx, heads, tiles)
                dq_all[s0:s1].copy_(dq)
                dk_all[s0:s1].copy_(dk)
                dv_all[s0:s1].copy_(dv)
            # query path: q = Wq q_in, q_in = SE(tgt, qedge) + qpos  (Wq's own gradient is one GEMM over all frames below)
            dq_in = ops.linear_dgrad(dq_all[s0:s1], wq)
            tgt = feat[:n] if j == 0 else out_all[s0 - n:s0]
            d_tgt = ops.temp((n, W), dtype=torch.float16, device=dev)
            if j == 0:
                d_qedge0 = ops.temp((n, W), dtype=torch.float16, device=dev)
                ops.se_fuse_bwd_ex(tgt, edge[:n], w0f, w2f, dq_in, d_tgt, d_qedge0, n, A[0], A[0, W:], A[0, W + 64:])
                d_tgt0, carry_out = d_tgt, None
            else:
                # frame j's query-side edge rows are frame j-1's in assignment order (frame 0 keeps its own): their gradient goes
                # straight to its rows of d_edge
                ops.se_fuse_bwd_ex(tgt, edge[s0 - n:s0], w0f, w2f, dq_in, d_tgt, d_edge[s0 - n:s0], n, A[j], A[j, W:], A[j, W + 64:],
                                   edge_map=cmap(j - 1) if j >= 2 else None)
                carry_out = d_tgt
        # key / value side of all frames at once
        dk_in, dwk, _ = ops.linear_bwd(k_in, dk_all, wk, True, True, False, grad_params=[wk])
        d_feat, dwv, _ = ops.linear_bwd(feat, dv_all, wv, True, True, False, grad_params=[wv])
        _, dwq, _ = ops.linear_bwd(q_in, dq_all, wq, False, True, False, grad_params=[wq])
        # key-side gate of all frames in ONE launch (a d pos vector per frame), which also sums up d feat and d edge: value path +
        # residual inputs + key-side gate (+ frame 0's query-side gate); the last frame's edge rows feed no later query gate
        ops.se_fuse_bwd_ex(feat, edge, w0f, w2f, dk_in, d_feat, d_edge, nl, K, K[0, W:], K[0, W + 64:], pos_seg_rows=n, pos_stride=SW,
                           add_reg=[(d_feat, nl), (d_idn, nl), (d_tgt0, n)],
                           add_edge=([(d_edge, (L - 1) * n)] if L > 1 else []) + [(d_qedge0, n)])
        d_temb = K[:, :W].clone()                                   # key-side gate of frame j adds t_j
        d_temb[0] += A[0, :W]                                       # query-side gate: t_0 for the first frame, t_{j-1} for frame j
        if L > 1:
            d_temb[:L - 1] += A[1:, :W]
        dse = (A[:, W:] + K[:, W:]).sum(0)
        dw0, dw2 = dse[:64].view(32, 2), dse[64:].view(2, 32)
        ln_sum = ln_acc[:L].sum(0)
        dg_ln, db_ln = ln_sum[0], ln_sum[1]
        _, dw_pos, db_pos = ops.linear_bwd(temb16, d_temb.to(torch.float16), w_pos, False, True, True)
        dt_pos, dt_se, dt_q, dt_ln, dt_dec = ctx.dtypes
        cast = lambda t, dt: None if t is None else (t if t.dtype == dt else t.to(dt))
        return (None, None, d_feat, d_edge, None, cast(dw_pos, dt_pos), cast(db_pos, dt_pos),
                cast(dw0, dt_se), cast(dw2, dt_se), cast(dwq, dt_q), cast(dwk, dt_q), cast(dwv, dt_q), cast(dg_ln, dt_ln), cast(db_ln, dt_ln),
                cast(dg_dec, dt_dec), cast(db_dec, dt_dec))


class AwarePositionRegMatcher(nn.Module):
    def __init__(self, hidden_channel=512, feedforward_channel=2048, num_head=8, decoder_layer_num=6, act="silu"):
        super().__init__()
        if decoder_layer_num != 1:
            raise NotImplementedError("TSCD builds the matcher with decoder_layer_num=1 (tscd_head.py:120-122)")
        self.num_heads = num_head
        self.num_layers = decoder_layer_num
        self.transformer_self_attention_layers = nn.ModuleList(
            [SelfAttentionLayer(hidden_channel, num_head, 0.0, act, False)])
        self.transformer_aware_cross_attention_layers = nn.ModuleList(
            [ReferringCrossAttentionLayer(hidden_channel, num_head, 0.0, act, False)])
        self.transformer_ffn_layers = nn.ModuleList([FFNLayer(hidden_channel, feedforward_channel, 0.0, act, False)])
        self.absolute_position_embedding = nn.Linear(256, hidden_channel)
        self.edge_feature_embedding = nn.Linear(int(hidden_channel / 4), hidden_channel)
        self.decoder_norm = nn.LayerNorm(hidden_channel)
        self._pending = None
        self._clear_state()
        self.last_hungarian = []     # test hook: the assignment's column indices per processed frame
        self.last_hungarian_rows = []   # ... and its row indices

    # test hooks: the assignment's column / row indices per processed frame.  With the deferred chain they are filled in when first
    # read (joining the worker) -- the train step itself never reads them.
    @property
    def last_hungarian(self):
        self._join_pending()
        return self._last_hungarian

    @last_hungarian.setter
    def last_hungarian(self, v):
        self._last_hungarian = v

    @property
    def last_hungarian_rows(self):
        self._join_pending()
        return self._last_hungarian_rows

    @last_hungarian_rows.setter
    def last_hungarian_rows(self, v):
        self._last_hungarian_rows = v

    def _join_pending(self):
        """Joins a chain whose result the device consumed without the host (deferred regime): raises what the worker raised."""
        pend = self.__dict__.get("_pending")
        if pend is None:
            return
        self._pending = None
        solved = pend.result()
        self._last_hungarian = [np.asarray(col_raw) for (_, col_raw, _) in solved]
        self._last_hungarian_rows = [np.asarray(row) for (row, _, _) in solved]
        global _TIMEOUTS_SEEN
        if _LATCH is not None and _LATCH.timeouts > _TIMEOUTS_SEEN:
            seen, _TIMEOUTS_SEEN = _TIMEOUTS_SEEN, _LATCH.timeouts          # reported once per occurrence
            raise RuntimeError("the device's wait for the assignment chain timed out (%d new time(s)): the stream went on with the index "
                               "table's previous contents" % (_LATCH.timeouts - seen))

    def _clear_state(self):
        self.last_outputs = None              # reference name; here the previous frame's output rows [n,C]
        self.last_frame_reg_embeds = None
        self.last_frame_cls_embeds = None
        self.last_frame_time_embedding = None
        self.last_frame_edge_embeds = None
        self.last_frame_n = 0

    @staticmethod
    @torch.no_grad()
    def _cost_matrices_async(pairs, pinned=None):
        """pairs: list of (ref_reg, cur_reg, ref_cls, cur_cls).  One MFMA GEMM each (1 - mean cosine, fp16 like the
        reference's torch.mm under autocast/.half(), tscd_matching.py:919-930); all of them are copied to ONE pinned host
        buffer asynchronously.  Returns (shapes, pinned buffer, event) -- `_cost_matrices_wait` blocks on the event."""
        costs = [ops.match_cost(a.detach().contiguous(), b.detach().contiguous(), c.detach().contiguous(), d.detach().contiguous())
                 for (a, b, c, d) in pairs]
        if not costs:
            return [], None, None
        flat = torch.cat([c.reshape(-1) for c in costs])
        return AwarePositionRegMatcher._copy_costs_async(flat, [tuple(c.shape) for c in costs], pinned)

    @staticmethod
    def _copy_costs_async(flat, shapes, pinned):
        if pinned is None or pinned.numel() < flat.numel():
            pinned = torch.empty((max(flat.numel(), 1 << 20),), dtype=torch.float16, pin_memory=True)
        host = pinned[:flat.numel()]
        host.copy_(flat, non_blocking=True)               # the only device->host transfer of the matcher
        ev = torch.cuda.Event()
        ev.record()
        return list(shapes), pinned, ev

    @staticmethod
    def _cost_matrices_wait(shapes, pinned, ev):
        if not shapes:
            return []
        ev.synchronize()
        host = pinned.numpy().view(np.uint16)
        out, o = [], 0
        for shp in shapes:
            n = shp[0] * shp[1]
            out.append(host[o:o + n].reshape(shp).copy())
            o += n
        return out

    @staticmethod
    def _cost_matrices(pairs):
        return AwarePositionRegMatcher._cost_matrices_wait(*AwarePositionRegMatcher._cost_matrices_async(pairs))

    def forward(self, features, features_reg, features_cls, features_edge, preds_per_frame, time_embedding, resume=False):
        """features / features_edge [N,W] (first sum(preds_per_frame) rows used), features_reg / features_cls [N_L,4W],
        time_embedding [lframe,256].  Returns [1, N_L, W] like the reference (the head takes [-1]) or None."""
        return self.finish(self.prepare(features_reg, features_cls, preds_per_frame, resume), features, features_edge,
                           time_embedding)

    def prepare(self, features_reg, features_cls, preds_per_frame, resume=False):
        """First half of forward: plans the frame-to-frame matching and launches every Hungarian cost matrix of the clip plus
        their (asynchronous) copy to the host.  The caller may queue device work that does not depend on the matcher's
        output before calling `finish`: it runs while the host solves the assignments.

        The Hungarian cost of frame i is a row permutation of cost(MCA features of frame i-1, frame i): the reference's
        ``last_frame_*_embeds`` are the previous frame's MCA rows re-ordered by the previous assignment.  So every cost
        matrix of the clip is computed up front (no dependence on the attention outputs), copied to the host once, and
        the assignments are solved by the host solver (tscd_amd/csrc_host/lap.cpp, bit-identical to the scipy call of the
        reference) on the permuted rows while the device runs the previous frame's attention."""
        # ---- plan: which (reference, current) pair each non-empty frame is matched against ----
        spans, start = [], 0
        for n in preds_per_frame:
            spans.append((start, start + n))
            start += n
        plan, prev, have_state = [], None, self.last_outputs is not None
        for i, n in enumerate(preds_per_frame):
            if n == 0:
                if i == 0 and resume is False:
                    have_state, prev = False, None
                continue
            first = (i == 0 and resume is False) or (prev is None and not have_state)
            plan.append((i, "self" if first else ("state" if prev is None else "prev"), prev))
            prev = i
        pairs = []
        for (i, kind, pi) in plan:
            s0, s1 = spans[i]
            cur_reg, cur_cls = features_reg[s0:s1], features_cls[s0:s1]
            if kind == "self":
                pairs.append((cur_reg, cur_reg, cur_cls, cur_cls))
            elif kind == "state":
                pairs.append((self.last_frame_reg_embeds, cur_reg, self.last_frame_cls_embeds, cur_cls))
            else:
                p0, p1 = spans[pi]
                pairs.append((features_reg[p0:p1], cur_reg, features_cls[p0:p1], cur_cls))
        dev_index = features_reg.device.index if features_reg.device.index is not None else torch.cuda.current_device()
        if plan and all(kind != "state" for (_, kind, _) in plan):
            # every matrix pairs two row ranges of this clip's MCA outputs: normalise all rows once, one GEMM per matrix straight into
            # the flat buffer the copy takes
            ranges = [(spans[i] if kind == "self" else spans[pi]) + spans[i] for (i, kind, pi) in plan]
            with torch.no_grad():
                flat, shapes = ops.match_costs_clip(features_reg.detach(), features_cls.detach(), ranges)
            shapes, pinned, ev = self._copy_costs_async(flat, shapes, _PINNED.get(dev_index))
        else:
            shapes, pinned, ev = self._cost_matrices_async(pairs, _PINNED.get(dev_index))
        _PINNED[dev_index] = pinned
        kinds = [{"self": KIND_SELF, "state": KIND_STATE, "prev": KIND_PREV}[kind] for (_, kind, _) in plan]
        chain, deferred = None, None
        self._join_pending()          # a failure of the previous clip's chain surfaces here at the latest
        if shapes:
            kw = {}
            if DEFER_CHAIN and features_reg.is_cuda and self._fused_regime(plan, list(preds_per_frame), resume):
                n, L = preds_per_frame[0], len(preds_per_frame)
                tlen = (2 * L - 1) * n
                table = _TABLES.get(dev_index)
                if table is None or table.numel() < tlen:
                    table = _TABLES[dev_index] = torch.zeros((max(tlen, 1 << 14),), dtype=torch.int64, pin_memory=True)
                if _TABLE_N.get(dev_index) != n:
                    # a wait that gives up (30 s) lets the stream go on with the table's previous contents: they must be in range for
                    # THIS clip's frame size.  (The device's copy of the previous clip's table is long done: its counts were on the host.)
                    table[:tlen] = torch.arange(tlen, dtype=torch.int64) % n
                    _TABLE_N[dev_index] = n
                latch = _latch()
                gens = [latch.new_generation() for _ in range(L)]          # one per frame: the worker raises them one solve at a time
                deferred = dict(table=table, tlen=tlen, latch=latch, gen0=gens[0])
                kw = dict(table_ptr=table.data_ptr(), table_len=tlen, latch=latch, gen=gens[-1])
            chain = _ASSIGN_WORKER.submit(lap_chain_f16, pinned.data_ptr(), shapes, kinds,
                                          functools.partial(_wait_copy, dev_index, ev), **kw)
        return dict(plan=plan, spans=spans, chain=chain, ppf=list(preds_per_frame), resume=resume,
                    features_reg=features_reg, features_cls=features_cls, deferred=deferred)

    def _fused_regime(self, plan, ppf, resume):
        """What _CafmChain needs of the clip and that is known before the assignments: gradients wanted, no state carried in, every
        local frame non-empty with the same number of proposals, bias-free q / k / v projections."""
        if not FUSED_CHAIN or not torch.is_grad_enabled() or resume is not False or not ppf or len(plan) != len(ppf):
            return False
        if ppf[0] == 0 or any(m != ppf[0] for m in ppf) or [k for (_, k, _) in plan] != ["self"] + ["prev"] * (len(ppf) - 1):
            return False
        attn = self.transformer_aware_cross_attention_layers[0].multihead_attn
        return attn.q_reg.bias is None and attn.k_reg.bias is None and attn.v_reg.bias is None

    def _finish_fused(self, plan, slots, dev_idx, ppf, resume, features, features_edge, time_embedding, layer, solved, feed=None):
        """The clamped training regime -- every local frame with the same number of proposals, no state carried in, gradients wanted --
        as ONE autograd node (_CafmChain).  Returns None when the clip does not qualify (the per-op path below handles every case).
        ``solved`` None: the deferred form -- ``slots`` / ``dev_idx`` describe the table the device receives once the chain is done."""
        if not self._fused_regime(plan, ppf, resume) or not (features.requires_grad or features_edge.requires_grad):
            return None
        n, L = ppf[0], len(ppf)
        attn = layer.multihead_attn
        self._clear_state()
        if solved is not None:
            self.last_hungarian = [np.asarray(col_raw) for (_, col_raw, _) in solved]
            self.last_hungarian_rows = [np.asarray(row) for (row, _, _) in solved]
        # offsets of c (assignment order; -1 for the first frame, which keeps its own order) and inv in the clip's index table
        sl = tuple((rec["c"][0] if "c" in rec else -1, rec["inv"][0]) for rec in slots)
        meta = (n, L, attn.num_heads, sl, layer.norm.eps, self.decoder_norm.eps, feed)
        nl = n * L
        out = _CafmChain.apply(meta, dev_idx, features[:nl], features_edge[:nl], time_embedding, self.absolute_position_embedding.weight,
                               self.absolute_position_embedding.bias, layer.CA.fc[0].weight, layer.CA.fc[2].weight, attn.q_reg.weight,
                               attn.k_reg.weight, attn.v_reg.weight, layer.norm.weight, layer.norm.bias, self.decoder_norm.weight,
                               self.decoder_norm.bias)
        # training always passes resume=False (tscd_trainer.py:168-169): the state this clip would leave behind is cleared by the next call
        return out.unsqueeze(0)

    def finish(self, prep, features, features_edge, time_embedding):
        dev = features.device
        layer = self.transformer_aware_cross_attention_layers[0]
        plan, spans, preds_per_frame, resume = prep["plan"], prep["spans"], prep["ppf"], prep["resume"]
        features_reg, features_cls = prep["features_reg"], prep["features_cls"]
        self._pending = None
        self.last_hungarian, self.last_hungarian_rows = [], []
        d = prep.get("deferred")
        if d is not None and (features.requires_grad or features_edge.requires_grad):
            # Deferred join: the stream waits for the worker (a host function parked on the latch), then copies the index table the
            # worker wrote ([inv_0 | c_1 | inv_1 | ...], tscd_lap_chain_f16_table) -- the host goes on queueing.
            n, L = preds_per_frame[0], len(preds_per_frame)
            dev_idx = ops.temp((d["tlen"],), dtype=torch.int64, device=dev)
            slots = [{"inv": (0, n)}] + [{"c": ((2 * j - 1) * n, n), "inv": (2 * j * n, n)} for j in range(1, L)]
            fused = self._finish_fused(plan, slots, dev_idx, preds_per_frame, resume, features, features_edge, time_embedding, layer, None,
                                       feed=_TableFeed(d["table"], dev_idx, d["latch"], d["gen0"], n, L))
            self._pending = prep["chain"]
            if fused is not None:
                return fused
            self._pending = None       # cannot happen (same predicate as prepare); fall through to the joined path
        solved = prep["chain"].result() if prep["chain"] is not None else []

        # every index table of the clip (column orders, unmatched / surviving rows, inverse permutations) goes to the device
        # in ONE copy; the per-frame loop below only slices views of it
        host_idx, slots = [], []
        last_n = self.last_frame_n if not (len(preds_per_frame) and resume is False) else 0
        for (i, kind, pi), (row, col_raw, perm) in zip(plan, solved):
            n = preds_per_frame[i]
            ent = {}
            if kind == "self":
                ret = col_raw
            else:
                ret = perm
                if last_n < n:
                    ent["nm"] = perm[len(col_raw):]
                elif last_n > n:
                    ent["r"] = row
                ent["c"] = perm
            ent["inv"] = np.argsort(ret)
            last_n = n
            rec = {}
            for k, a in ent.items():
                a = np.ascontiguousarray(a, dtype=np.int64)
                rec[k] = (sum(len(h) for h in host_idx), len(a))
                host_idx.append(a)
            slots.append(rec)
        dev_idx = ops.h2d(np.concatenate(host_idx), torch.long, dev) if host_idx else None
        fused = self._finish_fused(plan, slots, dev_idx, preds_per_frame, resume, features, features_edge, time_embedding, layer, solved)
        if fused is not None:
            return fused
        t_emb = linear(time_embedding.to(torch.float16).contiguous(), self.absolute_position_embedding)
        # fp32 rows (the SE gate takes fp32 position vectors: one cast here instead of one per use and per gradient); unbind:
        # one stack in backward instead of a zero-filled slice gradient per use
        t_rows = t_emb.float().unbind(0)
        view = lambda rec, k: dev_idx[rec[k][0]:rec[k][0] + rec[k][1]]
        # row gathers by index_select: its backward is one index_add into zeros (the indices are distinct rows), where
        # advanced indexing would sort the indices first (8 small launches per gather on a launch-bound stretch)
        rows = lambda t, i: t.index_select(0, i)
        # a gather by a PERMUTATION of all rows (the assignment order `c`, its inverse `inv`): the gradient is the gather by the inverse
        # permutation -- one launch, where index_select's backward zero-fills a buffer and index_adds into it
        prows = lambda t, i, i_inv: _PermuteRows.apply(t, i, i_inv)
        # per-frame views by ONE split (its backward is one cat) instead of a zero-filled [N, W] slice gradient per use
        n_loc = sum(preds_per_frame)
        f_loc = features[:n_loc].split(list(preds_per_frame))
        e_loc = features_edge[:n_loc].split(list(preds_per_frame))

        outs = []
        for (i, kind, pi), (row, col_raw, perm), rec in zip(plan, solved, slots):
            n = preds_per_frame[i]
            s0, s1 = spans[i]
            if i == 0 and resume is False:
                self._clear_state()
            cur, cur_edge = f_loc[i], e_loc[i]
            cur_reg, cur_cls = features_reg[s0:s1], features_cls[s0:s1]
            t_i = t_rows[i].unsqueeze(0)
            if kind == "self":
                self.last_frame_reg_embeds, self.last_frame_cls_embeds, self.last_frame_n = cur_reg, cur_cls, n
                self.last_hungarian.append(np.asarray(col_raw))
                self.last_hungarian_rows.append(np.asarray(row))
                out = layer(cur, cur, cur, t_i, t_i, cur_edge, cur_edge)
                new_edge = cur_edge
            else:
                # `perm`: the assignment's columns, followed by the unmatched current proposals (ascending) when the
                # reference frame had fewer rows -- the order in which the reference keeps this frame from now on
                self.last_hungarian.append(np.asarray(col_raw))
                self.last_hungarian_rows.append(np.asarray(row))
                if "nm" in rec:
                    nm = view(rec, "nm")
                    last_feat = torch.cat((self.last_outputs, rows(cur, nm)), 0)
                    last_edge = torch.cat((self.last_frame_edge_embeds, rows(cur_edge, nm)), 0)
                elif "r" in rec:
                    r = view(rec, "r")
                    last_feat, last_edge = rows(self.last_outputs, r), rows(self.last_frame_edge_embeds, r)
                else:
                    last_feat, last_edge = self.last_outputs, self.last_frame_edge_embeds
                c, c_inv = view(rec, "c"), view(rec, "inv")          # inv = argsort(perm): the inverse of c
                self.last_frame_reg_embeds, self.last_frame_cls_embeds, self.last_frame_n = prows(cur_reg, c, c_inv), prows(cur_cls, c, c_inv), n
                out = layer(prows(cur, c, c_inv), last_feat, cur, t_i, self.last_frame_time_embedding, cur_edge, last_edge)
                new_edge = prows(cur_edge, c, c_inv)
            self.last_outputs, self.last_frame_time_embedding, self.last_frame_edge_embeds = out, t_i, new_edge
            if kind == "self":
                outs.append(rows(out, view(rec, "inv")))
            else:
                outs.append(prows(out, view(rec, "inv"), view(rec, "c")))
        if not plan and len(preds_per_frame) and preds_per_frame[0] == 0 and resume is False:
            self._clear_state()
        if not outs:
            return None
        o = torch.cat(outs, 0)
        return add_layernorm(o, None, self.decoder_norm).unsqueeze(0)


class TaskAligned(nn.Module):
    def __init__(self, hidden_channel=512, num_head=8, decoder_layer_num=6, act="silu"):
        super().__init__()
        if decoder_layer_num != 1:
            raise NotImplementedError("TSCD builds TaskAligned with decoder_layer_num=1 (tscd_head.py:128-129)")
        self.num_heads = num_head
        self.num_layers = decoder_layer_num
        self.transformer_cross_attention_layers = nn.ModuleList([CrossAttentionLayer(hidden_channel, num_head, 0.0, act, False)])
        self.decoder_norm = nn.LayerNorm(hidden_channel)

    def forward(self, features_reg, features_obj, preds_per_frame):
        """All frames in one launch per op: segments keep each frame's queries on its own keys."""
        segs, start = [], 0
        for n in preds_per_frame:
            if n:
                segs.append((start, n, start, start + n))
            start += n
        if not segs:
            return None
        out = self.transformer_cross_attention_layers[0](features_obj, features_reg, segs)
        return add_layernorm(out, None, self.decoder_norm)
