// Fused TSCD losses (reference: TSCDHead.get_matcher_refined_losses yolox/models/tscd_head.py:1008-1203, IOUloss
// yolox/models/losses.py:9-53 (loss_type "iou": 1 - IoU^2), encode_reg_preds tscd_head.py:951-974).
//
// The reference evaluates the losses with boolean-mask gathers over per-image lists; the first port of this repository used
// ~100 small device-PyTorch launches per step (where / one_hot / gather / mul / div / sum ...).  Here:
//   loss_dense   one thread per (frame, anchor) over [B, A, 5+NC]:  IoU loss of the SimOTA-foreground anchors against their
//                matched box, objectness BCE of every anchor (target = foreground), class BCE of the foreground anchors
//                (target = onehot(matched class) * matched IoU) -- forward sums AND the gradients w.r.t. the raw obj / cls
//                logits in the same pass (sigmoid(x) - t, masked), nothing intermediate in HBM;
//   loss_rows    one wave per refined local proposal row:  class BCE of the refined class logits (target gathered from the
//                same assignment through the row's (frame, anchor)), objectness BCE of the matched-objectness logit (target =
//                "this proposal is a foreground anchor"), SmoothL1 of the box deltas against encode_reg_preds(gt, base box);
//                sums + gradients in one pass;
//   reduce       fixed-order reduction of the per-block partial sums (deterministic: two runs give bit-identical losses).
// All arithmetic in fp32 on the fp16 inputs (the documented deviation of DESIGN.md section 2: the reference's fp16 prod
// overflows for boxes above 65504 px^2).  Normalisation (1 / num_fg, the clip of the matched-objectness loss at 15, the loss
// weights 3 / 6) is applied by the caller from the returned sums -- they are scalars on the device.
#include "common.h"

__device__ __forceinline__ float bce_logits(float x, float t) {
  // max(x,0) - x t + log1p(exp(-|x|))  (torch.nn.functional.binary_cross_entropy_with_logits)
  return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int tid = threadIdx.x;
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// partial [gridDim.x][3]: iou, obj, cls sums of the block.  d_logit [B][A][1+NC] fp16: d(obj_sum + cls_sum) / d raw logit
// (channel 0 = objectness, 1.. = classes), un-normalised.
__global__ __launch_bounds__(256) void loss_dense_kernel(const f16* __restrict__ raw, const f16* __restrict__ outputs,
                                                        const unsigned char* __restrict__ fg, const int* __restrict__ matched_gt,
                                                        const float* __restrict__ pred_iou, const float* __restrict__ labels,
                                                        int B, int A, int NC, int Lmax, float* __restrict__ partial,
                                                        f16* __restrict__ d_logit, f16* __restrict__ d_reg, int hw0, int hw1, float st0,
                                                        float st1, float st2) {
  __shared__ float red[4];
  const long total = (long)B * A;
  const int nch = 5 + NC;
  float s_iou = 0.f, s_obj = 0.f, s_cls = 0.f;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int b = (int)(idx / A);
    const f16* r = raw + idx * nch;
    const bool f = fg[idx] != 0;
    const float xo = (float)r[4];
    s_obj += bce_logits(xo, f ? 1.f : 0.f);
    f16* d = d_logit ? d_logit + idx * (1 + NC) : nullptr;
    if (d) d[0] = (f16)(1.0f / (1.0f + expf(-xo)) - (f ? 1.f : 0.f));
    if (f) {
      const float* L = labels + ((long)b * Lmax + matched_gt[idx]) * 5;
      const f16* o = outputs + idx * nch;
      // IOUloss: boxes as (cx, cy, w, h)
      const float px = (float)o[0], py = (float)o[1], pw = (float)o[2], ph = (float)o[3];
      const float gx = L[1], gy = L[2], gw = L[3], gh = L[4];
      const float tlx = fmaxf(px - pw * 0.5f, gx - gw * 0.5f), tly = fmaxf(py - ph * 0.5f, gy - gh * 0.5f);
      const float brx = fminf(px + pw * 0.5f, gx + gw * 0.5f), bry = fminf(py + ph * 0.5f, gy + gh * 0.5f);
      const float en = (tlx < brx && tly < bry) ? 1.f : 0.f;
      const float ai = (brx - tlx) * (bry - tly) * en;
      const float iou = ai / (pw * ph + gw * gh - ai + 1e-16f);
      s_iou += 1.0f - iou * iou;
      if (d_reg) {
        // d(1 - IoU^2) / d raw box logits (base-detector training, row N4): the box is decoded as cx = (r0 + grid) * stride,
        // w = exp(r2) * stride, so d/dr0 = stride * d/dcx and d/dr2 = w * d/dw; min / max pass the gradient to the active side
        // (half to each on an exact tie, like torch)
        const int a = (int)(idx - (long)b * A);
        const float st = a < hw0 ? st0 : (a < hw0 + hw1 ? st1 : st2);
        auto side = [](float p, float g, bool want_greater) { return p == g ? 0.5f : ((want_greater ? p > g : p < g) ? 1.f : 0.f); };
        const float wax = side(px - pw * 0.5f, gx - gw * 0.5f, true), wbx = side(px + pw * 0.5f, gx + gw * 0.5f, false);
        const float way = side(py - ph * 0.5f, gy - gh * 0.5f, true), wby = side(py + ph * 0.5f, gy + gh * 0.5f, false);
        const float iw = brx - tlx, ih = bry - tly;
        const float U = pw * ph + gw * gh - ai + 1e-16f;
        // d ai: x-centre, y-centre, width, height of the predicted box
        const float dai[4] = {en * ih * (wbx - wax), en * iw * (wby - way), en * ih * 0.5f * (wbx + wax), en * iw * 0.5f * (wby + way)};
        const float dap[4] = {0.f, 0.f, ph, pw};
        const float chain[4] = {st, st, pw, ph};
        f16* dr = d_reg + idx * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float diou = (dai[q] * (U + ai) - ai * dap[q]) / (U * U);
          dr[q] = (f16)(-2.0f * iou * diou * chain[q]);
        }
      }
      int gc = (int)L[0];
      gc = gc < 0 ? 0 : (gc >= NC ? NC - 1 : gc);
      const float tiou = pred_iou[idx];
      for (int c = 0; c < NC; ++c) {
        const float x = (float)r[5 + c], t = c == gc ? tiou : 0.f;
        s_cls += bce_logits(x, t);
        if (d) d[1 + c] = (f16)(1.0f / (1.0f + expf(-x)) - t);
      }
    } else {
      if (d)
        for (int c = 0; c < NC; ++c) d[1 + c] = (f16)0.f;
      if (d_reg) *(f16x4*)(d_reg + idx * 4) = f16x4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    }
  }
  const float a = block_sum_256(s_iou, red), o2 = block_sum_256(s_obj, red), c2 = block_sum_256(s_cls, red);
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 3 + 0] = a;
    partial[blockIdx.x * 3 + 1] = o2;
    partial[blockIdx.x * 3 + 2] = c2;
  }
}

// One wave per refined row n (a local-frame proposal): (b, a) = (row_b[n], row_a[n]).
//   part bit 0: class BCE of cls_logit[n][NC]       -> partial[.][0], d_cls [N][NC]
//   part bit 1: objectness BCE of obj_logit[n]      -> partial[.][1], d_obj [N]
//   part bit 2: SmoothL1 of reg_delta[n][4]         -> partial[.][2], d_reg [N][4]      (beta = 1)
// base: pred rows [N][ndet] with the base xyxy box in columns 0..3.
__global__ __launch_bounds__(256) void loss_rows_kernel(const int* __restrict__ row_b, const int* __restrict__ row_a, int N, int A, int NC,
                                                       int Lmax, const unsigned char* __restrict__ fg, const int* __restrict__ matched_gt,
                                                       const float* __restrict__ pred_iou, const float* __restrict__ labels,
                                                       const f16* __restrict__ cls_logit, const f16* __restrict__ obj_logit,
                                                       const f16* __restrict__ reg_delta, const f16* __restrict__ base, int ndet, int part,
                                                       float* __restrict__ partial, f16* __restrict__ d_cls, f16* __restrict__ d_obj,
                                                       f16* __restrict__ d_reg) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float s_cls = 0.f, s_obj = 0.f, s_reg = 0.f;
  for (int n = blockIdx.x * 4 + wv; n < N; n += gridDim.x * 4) {
    const long idx = (long)row_b[n] * A + row_a[n];
    const bool f = fg[idx] != 0;
    const float* L = labels + ((long)row_b[n] * Lmax + (f ? matched_gt[idx] : 0)) * 5;
    if (part & 1) {
      int gc = (int)L[0];
      gc = gc < 0 ? 0 : (gc >= NC ? NC - 1 : gc);
      const float tiou = f ? pred_iou[idx] : 0.f;
      for (int c = lane; c < NC; c += 64) {
        const float x = (float)cls_logit[(long)n * NC + c], t = c == gc ? tiou : 0.f;
        if (f) s_cls += bce_logits(x, t);
        if (d_cls) d_cls[(long)n * NC + c] = f ? (f16)(1.0f / (1.0f + expf(-x)) - t) : (f16)0.f;
      }
    }
    if ((part & 2) && lane == 0) {
      const float x = (float)obj_logit[n], t = f ? 1.f : 0.f;
      s_obj += bce_logits(x, t);
      if (d_obj) d_obj[n] = (f16)(1.0f / (1.0f + expf(-x)) - t);
    }
    if ((part & 4) && lane < 4) {
      float g = 0.f;
      if (f) {
        const f16* bb = base + (long)n * ndet;
        const float x1 = (float)bb[0], y1 = (float)bb[1], x2 = (float)bb[2], y2 = (float)bb[3];
        const float w = x2 - x1, h = y2 - y1, cx = x1 + 0.5f * w, cy = y1 + 0.5f * h;
        // encode_reg_preds: ((gx - cx) / w, (gy - cy) / h, log(gw / w + 1e-8), log(gh / h + 1e-8))
        float t;
        if (lane == 0) t = (L[1] - cx) / w;
        else if (lane == 1) t = (L[2] - cy) / h;
        else if (lane == 2) t = logf(L[3] / w + 1e-8f);
        else t = logf(L[4] / h + 1e-8f);
        const float dlt = (float)reg_delta[(long)n * 4 + lane] - t, ad = fabsf(dlt);
        s_reg += ad < 1.f ? 0.5f * dlt * dlt : ad - 0.5f;
        g = ad < 1.f ? dlt : (dlt > 0.f ? 1.f : -1.f);
      }
      if (d_reg) d_reg[(long)n * 4 + lane] = (f16)g;
    }
  }
  const float a = block_sum_256(s_cls, red), b2 = block_sum_256(s_obj, red), c2 = block_sum_256(s_reg, red);
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 3 + 0] = a;
    partial[blockIdx.x * 3 + 1] = b2;
    partial[blockIdx.x * 3 + 2] = c2;
  }
}

// sums[3] = sum over blocks of partial[block][3], fixed order (one wave per column, lanes stride the blocks, then a shuffle tree).
__global__ void loss_reduce_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ sums) {
  const int col = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (col >= 3) return;
  float s = 0.f;
  for (int i = lane; i < nblocks; i += 64) s += partial[i * 3 + col];
  s = wave_sum(s);
  if (lane == 0) sums[col] = s;
}

static int loss_blocks(long items, int per_block) {
  long nb = (items + per_block - 1) / per_block;
  return (int)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
}

extern "C" long tscd_loss_workspace_bytes(long B, long A, long N) {
  return 3L * 4 * 1024 * 2 + 256;
}

// sums[3] (fp32, device): iou / obj / cls sums; d_logit [B][A][1+NC] fp16 or NULL.
extern "C" int tscd_loss_dense_f16(const void* raw, const void* outputs, const unsigned char* fg_mask, const int* matched_gt,
                                   const float* pred_iou, const float* labels, int B, int A, int NC, int Lmax, void* workspace,
                                   long workspace_bytes, float* sums, void* d_logit, void* stream) {
  TSCD_CHECK_ARG(raw && outputs && fg_mask && matched_gt && pred_iou && labels && workspace && sums, "loss_dense: null pointer");
  TSCD_CHECK_ARG(B > 0 && A > 0 && NC > 0 && Lmax > 0, "loss_dense: bad sizes");
  TSCD_CHECK_ARG(workspace_bytes >= tscd_loss_workspace_bytes(B, A, 0), "loss_dense: workspace too small");
  const int nb = loss_blocks((long)B * A, 256);
  hipLaunchKernelGGL(loss_dense_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const f16*)raw, (const f16*)outputs, fg_mask,
                     matched_gt, pred_iou, labels, B, A, NC, Lmax, (float*)workspace, (f16*)d_logit, (f16*)nullptr, 0, 0, 0.f, 0.f, 0.f);
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(192), 0, (hipStream_t)stream, (const float*)workspace, nb, sums);
  TSCD_CHECK_LAUNCH("loss_dense");
  return TSCD_OK;
}

// The same with the gradient of the IoU-loss sum w.r.t. the raw box logits (d_reg [B][A][4] fp16): base-detector training
// (yolo_head.py:267-434, row N4), where the box branch trains.  hw / stride: the three pyramid levels' anchor counts and strides.
extern "C" int tscd_loss_dense_reg_f16(const void* raw, const void* outputs, const unsigned char* fg_mask, const int* matched_gt,
                                       const float* pred_iou, const float* labels, int B, int A, int NC, int Lmax, const int* hw,
                                       const int* stride, void* workspace, long workspace_bytes, float* sums, void* d_logit, void* d_reg,
                                       void* stream) {
  TSCD_CHECK_ARG(raw && outputs && fg_mask && matched_gt && pred_iou && labels && workspace && sums && hw && stride && d_reg,
                 "loss_dense_reg: null pointer");
  TSCD_CHECK_ARG(B > 0 && A > 0 && NC > 0 && Lmax > 0 && hw[0] + hw[1] + hw[2] == A, "loss_dense_reg: bad sizes");
  TSCD_CHECK_ARG(workspace_bytes >= tscd_loss_workspace_bytes(B, A, 0), "loss_dense_reg: workspace too small");
  const int nb = loss_blocks((long)B * A, 256);
  hipLaunchKernelGGL(loss_dense_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const f16*)raw, (const f16*)outputs, fg_mask,
                     matched_gt, pred_iou, labels, B, A, NC, Lmax, (float*)workspace, (f16*)d_logit, (f16*)d_reg, hw[0], hw[1],
                     (float)stride[0], (float)stride[1], (float)stride[2]);
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(192), 0, (hipStream_t)stream, (const float*)workspace, nb, sums);
  TSCD_CHECK_LAUNCH("loss_dense_reg");
  return TSCD_OK;
}

// sums[3]: refined class BCE / matched-objectness BCE / SmoothL1 sums over the N rows (parts not selected give 0).
extern "C" int tscd_loss_rows_f16(const int* row_b, const int* row_a, int N, int A, int NC, int Lmax, const unsigned char* fg_mask,
                                  const int* matched_gt, const float* pred_iou, const float* labels, const void* cls_logit,
                                  const void* obj_logit, const void* reg_delta, const void* base, int ndet, int part, void* workspace,
                                  long workspace_bytes, float* sums, void* d_cls, void* d_obj, void* d_reg, void* stream) {
  TSCD_CHECK_ARG(row_b && row_a && fg_mask && matched_gt && pred_iou && labels && workspace && sums, "loss_rows: null pointer");
  TSCD_CHECK_ARG(!(part & 1) || cls_logit, "loss_rows: class part without logits");
  TSCD_CHECK_ARG(!(part & 2) || obj_logit, "loss_rows: objectness part without logits");
  TSCD_CHECK_ARG(!(part & 4) || (reg_delta && base && ndet >= 4), "loss_rows: box part without deltas / base rows");
  TSCD_CHECK_ARG(workspace_bytes >= tscd_loss_workspace_bytes(0, 0, N), "loss_rows: workspace too small");
  if (N <= 0) {
    (void)hipMemsetAsync(sums, 0, 12, (hipStream_t)stream);
    return TSCD_OK;
  }
  const int nb = loss_blocks(N, 4);
  hipLaunchKernelGGL(loss_rows_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, row_b, row_a, N, A, NC, Lmax, fg_mask, matched_gt,
                     pred_iou, labels, (const f16*)cls_logit, (const f16*)obj_logit, (const f16*)reg_delta, (const f16*)base, ndet, part,
                     (float*)workspace, (f16*)d_cls, (f16*)d_obj, (f16*)d_reg);
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(192), 0, (hipStream_t)stream, (const float*)workspace, nb, sums);
  TSCD_CHECK_LAUNCH("loss_rows");
  return TSCD_OK;
}

// ------------------------------------------------------------------------------------------------------
// The scalar tail of get_matcher_refined_losses (tscd_head.py:1157-1203) in one launch: normalisation of the loss sums by the
// foreground counts, the clip of the matched-objectness loss at 15 (x / x.detach() * 15), the weighted total, the reported terms
// and the factors the backward multiplies the incoming gradient with.  As ~25 zero-dimensional torch ops (and as many in backward)
// this was 50 launches on the launch-bound stretch of the step.  Same fp32 operations in the same order (no contraction).
//   dense[3] = IoU / objectness / class sums, rows[3] = refined class (unused here) / matched objectness / matched SmoothL1 sums,
//   r_cls = the refined class loss (already normalised; produced by the eager class branch), num_fg / ngt int[B].
//   out[8]  = total, 3 iou, obj, cls, num_fg / max(num_gts, 1), r_cls, m_obj, 6 m_iou
//   w[6]    = d total / d dense[0..2], d total / d rows[0..2]
// ------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
__global__ void loss_combine_kernel(const float* __restrict__ dense, const float* __restrict__ rows, const float* __restrict__ r_cls,
                                    const int* __restrict__ num_fg_b, const int* __restrict__ ngt_b, int B, int lframe,
                                    float* __restrict__ out, float* __restrict__ w) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  long nfg = 0, nfl = 0, ngt = 0;
  for (int b = 0; b < B; ++b) {
    nfg += num_fg_b[b];
    if (b < lframe) nfl += num_fg_b[b];
    ngt += ngt_b[b];
  }
  const float num_fg = fmaxf((float)nfg, 1.f), num_fg_local = (float)nfl, num_gts = (float)ngt;
  const float li = dense[0] / num_fg, lo = dense[1] / num_fg, lc = dense[2] / num_fg;
  const float mo = rows[1] / num_fg_local, mi = rows[2] / num_fg_local;
  const bool clip = mo > 15.f;
  const float mo_c = clip ? mo / mo * 15.f : mo;
  const float rc = *r_cls;
  const float li3 = 3.0f * li, mi6 = 6.0f * mi;
  float total = li3 + lo;
  total = total + lc;
  total = total + rc;
  total = total + mo_c;
  total = total + mi6;
  out[0] = total; out[1] = li3; out[2] = lo; out[3] = lc; out[4] = num_fg / fmaxf(num_gts, 1.f); out[5] = rc; out[6] = mo_c; out[7] = mi6;
  w[0] = 3.0f / num_fg; w[1] = 1.0f / num_fg; w[2] = 1.0f / num_fg;
  w[3] = 0.f; w[4] = (clip ? 15.f / mo : 1.f) / num_fg_local; w[5] = 6.0f / num_fg_local;
}
extern "C" int tscd_loss_combine_f32(const float* dense, const float* rows, const float* r_cls, const int* num_fg, const int* ngt, int B,
                                     int lframe, float* out, float* weights, void* stream) {
  TSCD_CHECK_ARG(dense && rows && r_cls && num_fg && ngt && out && weights && B > 0 && lframe >= 0 && lframe <= B, "loss_combine: bad arguments");
  hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dense, rows, r_cls, num_fg, ngt, B, lframe, out, weights);
  TSCD_CHECK_LAUNCH("loss_combine");
  return TSCD_OK;
}

// Backward of the fused losses: out[r][pad + c] = fp16( float(d[r][c]) * g[c < split ? i0 : i1] ), columns [0, pad) zero.
// The upstream scalars g (fp32, on the device) are loss_scale * weight / num_fg: the PRODUCT is formed in fp32 and rounded once,
// as the reference's autograd does -- casting g to fp16 first overflows at GradScaler's initial 65536 once num_fg <= 6
// (tscd_head.py:1157-1203 normalises by the foreground counts).  One launch instead of cat / cast / multiply / pad.
__global__ __launch_bounds__(256) void scale_grad_kernel(const f16* __restrict__ d, const float* __restrict__ g, int i0, int i1, int split,
                                                        long rows, int C, int pad, f16* __restrict__ out) {
  const int CO = C + pad;
  const long total = rows * CO;
  const float g0 = g[i0], g1 = g[i1];
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long r = idx / CO;
    const int c = (int)(idx - r * CO) - pad;
    out[idx] = c < 0 ? (f16)0.f : (f16)((float)d[r * C + c] * (c < split ? g0 : g1));
  }
}

extern "C" int tscd_scale_grad_f16(const void* d, const float* g, int i0, int i1, int split, long rows, int C, int pad, void* out,
                                   void* stream) {
  TSCD_CHECK_ARG(d && g && out && rows >= 0 && C > 0 && pad >= 0 && i0 >= 0 && i1 >= 0, "scale_grad: bad arguments");
  if (rows == 0) return TSCD_OK;
  const long total = rows * (C + pad);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(scale_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)d, g, i0, i1, split, rows, C, pad,
                     (f16*)out);
  TSCD_CHECK_LAUNCH("scale_grad");
  return TSCD_OK;
}

// ---- class score of every selected proposal: sigmoid of its own class logit (tscd_head.py:1560-1600: pred_result[:, 5] with the
// gradient the reference keeps through it) and the scatter that is its backward.  raw [B][A][nch] fp16 (nch = 5 + NC), one
// proposal row r = (frame row_b[r], anchor row_a[r], class = pred_rows[r][cls_col] stored as a float16 class id).  Forward and
// backward round where torch's sigmoid / sigmoid_backward on fp16 tensors round (fp32 inside, one rounding at the store).
__global__ void cls_score_gather_kernel(const f16* __restrict__ raw, long A, int nch, const int* __restrict__ row_b,
                                        const int* __restrict__ row_a, const f16* __restrict__ pred_rows, int ld_rows, int cls_col,
                                        long N, f16* __restrict__ out) {
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < N; r += (long)gridDim.x * 256) {
    const int c = (int)(float)pred_rows[r * ld_rows + cls_col];
    const float x = (float)raw[((long)row_b[r] * A + row_a[r]) * nch + 5 + c];
    out[r] = (f16)(1.0f / (1.0f + expf(-x)));
  }
}
__global__ void cls_score_scatter_kernel(const f16* __restrict__ g, const f16* __restrict__ s, long A, int nch,
                                         const int* __restrict__ row_b, const int* __restrict__ row_a,
                                         const f16* __restrict__ pred_rows, int ld_rows, int cls_col, long N, f16* __restrict__ d_raw) {
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < N; r += (long)gridDim.x * 256) {
    const int c = (int)(float)pred_rows[r * ld_rows + cls_col];
    const float sv = (float)s[r];
    d_raw[((long)row_b[r] * A + row_a[r]) * nch + 5 + c] = (f16)((float)g[r] * (1.0f - sv) * sv);
  }
}
extern "C" int tscd_cls_score_gather_f16(const void* raw, int B, long A, int nch, const int* row_b, const int* row_a, const void* pred_rows,
                                         int ld_rows, int cls_col, long N, void* out, void* stream) {
  if (N == 0) return TSCD_OK;
  TSCD_CHECK_ARG(raw && row_b && row_a && pred_rows && out && B > 0 && A > 0 && nch > 5 && cls_col >= 0 && cls_col < ld_rows,
                 "cls_score_gather: bad arguments");
  const int blocks = (int)((N + 255) / 256 < 1024 ? (N + 255) / 256 : 1024);
  hipLaunchKernelGGL(cls_score_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)raw, A, nch, row_b, row_a,
                     (const f16*)pred_rows, ld_rows, cls_col, N, (f16*)out);
  TSCD_CHECK_LAUNCH("cls_score_gather");
  return TSCD_OK;
}
// d_raw [B][A][nch] is cleared here, then receives g * s * (1 - s) at each proposal's (frame, anchor, class) -- the rows are distinct
// (frame, anchor) pairs, so plain stores.
extern "C" int tscd_cls_score_scatter_f16(const void* g, const void* s, int B, long A, int nch, const int* row_b, const int* row_a,
                                          const void* pred_rows, int ld_rows, int cls_col, long N, void* d_raw, void* stream) {
  TSCD_CHECK_ARG(d_raw && B > 0 && A > 0 && nch > 5, "cls_score_scatter: bad arguments");
  (void)hipMemsetAsync(d_raw, 0, sizeof(f16) * B * A * nch, (hipStream_t)stream);
  if (N == 0) return TSCD_OK;
  TSCD_CHECK_ARG(g && s && row_b && row_a && pred_rows && cls_col >= 0 && cls_col < ld_rows, "cls_score_scatter: bad arguments");
  const int blocks = (int)((N + 255) / 256 < 1024 ? (N + 255) / 256 : 1024);
  hipLaunchKernelGGL(cls_score_scatter_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)g, (const f16*)s, A, nch,
                     row_b, row_a, (const f16*)pred_rows, ld_rows, cls_col, N, (f16*)d_raw);
  TSCD_CHECK_LAUNCH("cls_score_scatter");
  return TSCD_OK;
}

// ---- gradient of one pyramid level's raw prediction map [B][HW][nch] (batch stride given: the level is a slice of the clip's
// [B][A][nch] gradient) split into what the two trainable prediction convs need (tscd_head.py:213-239: obj_preds 1 channel,
// cls_preds NC channels): dY zero-padded to 8 / ncp channels for the weight-gradient GEMM, and the bias gradients = column sums
// over all pixels (fixed-order two-stage reduction: per-block partials, then one wave per column).  One pass over the slice instead
// of a pad, a copy, a column sum and a contiguous copy per conv.
template <int NV>
__global__ __launch_bounds__(256) void pred_grad_split_kernel(const f16* __restrict__ d_raw, long batch_stride, long HW, long M, int nch,
                                                             f16* __restrict__ dy_obj, f16* __restrict__ dy_cls,
                                                             float* __restrict__ partial) {
  constexpr int NCP = 8 * NV;
  __shared__ float colacc[4][NCP + 1];
  const int NC = nch - 5, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mo = 0.f, mc[NCP];                    // this thread's column sums (compile-time indices: registers)
#pragma unroll
  for (int c = 0; c < NCP; ++c) mc[c] = 0.f;
  for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
    const long b = m / HW;
    const f16* src = d_raw + b * batch_stride + (m - b * HW) * nch;
    const f16 vo = src[4];
    mo += (float)vo;
    f16x8 o8 = {vo, 0, 0, 0, 0, 0, 0, 0};
    *(f16x8*)(dy_obj + m * 8) = o8;
#pragma unroll
    for (int v8 = 0; v8 < NV; ++v8) {
      f16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = 8 * v8 + j;
        v[j] = c < NC ? src[5 + c] : (f16)0.f;
        mc[c] += (float)v[j];
      }
      *(f16x8*)(dy_cls + m * NCP + 8 * v8) = v;
    }
  }
  mo = wave_sum(mo);
  if (lane == 0) colacc[wave][NCP] = mo;
#pragma unroll
  for (int c = 0; c < NCP; ++c) {
    const float t = wave_sum(mc[c]);
    if (lane == 0) colacc[wave][c] = t;
  }
  __syncthreads();
  // partial row layout = the raw map's channels: [.. 4 = objectness, 5 + c = class c]
  if (threadIdx.x < NC) partial[(long)blockIdx.x * nch + 5 + threadIdx.x] = colacc[0][threadIdx.x] + colacc[1][threadIdx.x] + colacc[2][threadIdx.x] + colacc[3][threadIdx.x];
  if (threadIdx.x == 64) partial[(long)blockIdx.x * nch + 4] = colacc[0][NCP] + colacc[1][NCP] + colacc[2][NCP] + colacc[3][NCP];
}
__global__ void pred_grad_colsum_kernel(const float* __restrict__ partial, int nblocks, int nch, float* __restrict__ db) {
  const int col = 4 + (threadIdx.x >> 6) + 4 * blockIdx.x, lane = threadIdx.x & 63;
  if (col >= nch) return;
  float s = 0.f;
  for (int i = lane; i < nblocks; i += 64) s += partial[(long)i * nch + col];
  s = wave_sum(s);
  if (lane == 0) db[col] = s;
}
static int pred_grad_blocks(long M) {
  long nb = (M + 255) / 256;
  return (int)(nb < 1 ? 1 : (nb > 512 ? 512 : nb));
}
extern "C" long tscd_pred_grad_split_workspace_bytes(long M, int nch) { return (long)pred_grad_blocks(M) * nch * 4 + 256; }
// d_raw: [B][HW][nch] fp16 with batch stride `batch_stride` elements (rows of a frame contiguous); dy_obj [B*HW][8], dy_cls
// [B*HW][ncp] fp16 (ncp = NC rounded up to 8); db [nch] fp32: entries 4 (objectness) and 5.. (classes) are written.
extern "C" int tscd_pred_grad_split_f16(const void* d_raw, long batch_stride, int B, long HW, int nch, int ncp, void* dy_obj, void* dy_cls,
                                        float* db, void* workspace, long workspace_bytes, void* stream) {
  TSCD_CHECK_ARG(d_raw && dy_obj && dy_cls && db && workspace && B > 0 && HW > 0 && nch > 5 && nch <= 40, "pred_grad_split: bad arguments");
  TSCD_CHECK_ARG(ncp % 8 == 0 && ncp >= nch - 5 && ncp <= 40 && batch_stride >= HW * nch, "pred_grad_split: bad strides");
  const long M = (long)B * HW;
  TSCD_CHECK_ARG(workspace_bytes >= tscd_pred_grad_split_workspace_bytes(M, nch), "pred_grad_split: workspace too small");
  const int nb = pred_grad_blocks(M);
#define PGS(NVV)                                                                                                               \
  case NVV:                                                                                                                     \
    hipLaunchKernelGGL((pred_grad_split_kernel<NVV>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const f16*)d_raw, batch_stride, \
                       HW, M, nch, (f16*)dy_obj, (f16*)dy_cls, (float*)workspace);                                              \
    break
  switch (ncp / 8) {
    PGS(1); PGS(2); PGS(3); PGS(4); PGS(5);
  }
#undef PGS
  hipLaunchKernelGGL(pred_grad_colsum_kernel, dim3((nch - 4 + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nb, nch, db);
  TSCD_CHECK_LAUNCH("pred_grad_split");
  return TSCD_OK;
}
