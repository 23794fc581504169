// SimOTA label assignment on the device (reference: TSCDHead.get_fg_idx / get_assignments / get_in_boxes_info /
// dynamic_k_matching, yolox/models/tscd_head.py:1712-1851, 1330-1544; bboxes_iou yolox/utils/boxes.py:131-154).
// The reference loops on the host over images and ground-truth boxes with .item() syncs and per-image
// torch.cuda.empty_cache(); here three launches handle the whole clip and nothing returns to the host:
//   simota_cost     thread per (image, anchor): in-box / in-centre tests (radius 4.5 strides), pairwise IoU, the
//                   BCE(sqrt(sig(cls) sig(obj)), onehot) class cost, cost = cls + 3*(-log(iou+1e-8)) + 1e5*[not both]
//   simota_match    workgroup per (image, gt): dynamic k = clamp(int(sum of the 10 largest IoUs), 1), then the k
//                   smallest-cost anchors (ties -> lower anchor index)
//   simota_resolve  thread per (image, anchor): anchors claimed by several gts go to the gt of minimum cost (first
//                   minimum), emits fg mask, matched gt index and the matched IoU
// pred_iou is the IoU of the matched (gt, anchor) pair.  The reference forms it as (matching_matrix * pair_wise_ious).sum(0)
// (tscd_head.py:1540-1543): in fp16 the IoU of two boxes that are far apart in both axes is NaN (prod(br - tl) overflows to inf,
// times the zero overlap flag), and 0 * NaN = NaN then poisons the sum of every anchor that has such a pair with ANY other gt --
// NaN class targets, a NaN loss and a skipped optimizer step.  Deliberate deviation: finite, the matched pair.
// `emu16` reproduces the reference's fp16 arithmetic under autocast (IoU chain evaluated on fp16 tensors with per-op
// rounding, fp16 sigmoid of the objectness); with emu16 = 0 the same fp16 inputs are evaluated in fp32.
#include "common.h"

#define SIMOTA_INF 3.0e38f

struct SimotaParams {
  const f16* outputs;   // [B][A][5+NC]: cx,cy,w,h (decoded), obj logit, cls logits
  const float* labels;  // [B][Lmax][5]: cls, cx, cy, w, h  (values already fp16-representable, T1)
  int B, A, NC, Lmax, G;   // G = max gts handled (stride of the cost planes)
  int hw[3], w[3], stride[3], a0[3];
  float* cost;          // [B][G][A]
  float* iou;           // [B][G][A]   (-1 for anchors outside the candidate set)
  int* ngt;             // [B]
  int emu16;
  float radius;         // centre-prior radius in strides: 4.5 (TSCD head, tscd_head.py:1470) or 2.5 (YOLOX base head, yolo_head.py:594)
};

__device__ __forceinline__ float r16(float x, int emu) { return emu ? (float)(f16)x : x; }
__device__ __forceinline__ float clamp_log(float x) { return fmaxf(logf(x), -100.0f); }   // torch BCE clamps log at -100

__global__ void simota_count_kernel(const float* __restrict__ labels, int B, int Lmax, int* __restrict__ ngt) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int g = 0;
  for (int l = 0; l < Lmax; ++l) {
    const float* r = labels + ((long)b * Lmax + l) * 5;
    if (r[0] + r[1] + r[2] + r[3] + r[4] > 0.f) ++g;   // nlabel = (labels.sum(dim=2) > 0).sum(dim=1)
  }
  ngt[b] = g;
}

__global__ void simota_cost_kernel(SimotaParams p) {
  const int b = blockIdx.y, a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= p.A) return;
  const int emu = p.emu16;
  int G = p.ngt[b];
  if (G > p.G) G = p.G;
  const int k = a >= p.a0[2] ? 2 : (a >= p.a0[1] ? 1 : 0);
  const int la = a - p.a0[k];
  const float st = (float)p.stride[k];
  const float xc = r16(r16((float)(la % p.w[k]) * st, emu) + 0.5f * st, emu);
  const float yc = r16(r16((float)(la / p.w[k]) * st, emu) + 0.5f * st, emu);
  const f16* o = p.outputs + ((long)b * p.A + a) * (5 + p.NC);
  const float bx = (float)o[0], by = (float)o[1], bw = (float)o[2], bh = (float)o[3];
  // pass 1: is the anchor inside any gt box or any gt centre region?
  bool any = false;
  for (int g = 0; g < G; ++g) {
    const float* L = p.labels + ((long)b * p.Lmax + g) * 5;
    float gx = L[1], gy = L[2], gw = L[3], gh = L[4];
    float l = r16(gx - 0.5f * gw, emu), r = r16(gx + 0.5f * gw, emu), t = r16(gy - 0.5f * gh, emu), d = r16(gy + 0.5f * gh, emu);
    bool inb = fminf(fminf(r16(xc - l, emu), r16(yc - t, emu)), fminf(r16(r - xc, emu), r16(d - yc, emu))) > 0.f;
    float rad = r16(p.radius * st, emu);
    float cl = r16(gx - rad, emu), cr = r16(gx + rad, emu), ct = r16(gy - rad, emu), cb = r16(gy + rad, emu);
    bool inc = fminf(fminf(r16(xc - cl, emu), r16(yc - ct, emu)), fminf(r16(cr - xc, emu), r16(cb - yc, emu))) > 0.f;
    any = any || inb || inc;
  }
  // class probabilities p_c = sqrt(sig(cls) * sig(obj)) in fp32 (autocast disabled there, tscd_head.py:1390-1397)
  float so = 1.0f / (1.0f + expf(-(float)o[4]));
  so = r16(so, emu);   // obj_preds_ is not .float()'ed in the reference: its sigmoid runs on the fp16 tensor
  float base = 0.f;
  if (any)
    for (int c = 0; c < p.NC; ++c) {
      float pc = sqrtf((1.0f / (1.0f + expf(-(float)o[5 + c]))) * so);
      base += -clamp_log(1.0f - pc);
    }
  for (int g = 0; g < G; ++g) {
    long idx = ((long)b * p.G + g) * p.A + a;
    if (!any) {
      p.cost[idx] = SIMOTA_INF;
      p.iou[idx] = -1.0f;
      continue;
    }
    const float* L = p.labels + ((long)b * p.Lmax + g) * 5;
    float gx = L[1], gy = L[2], gw = L[3], gh = L[4];
    float l = r16(gx - 0.5f * gw, emu), r = r16(gx + 0.5f * gw, emu), t = r16(gy - 0.5f * gh, emu), d = r16(gy + 0.5f * gh, emu);
    bool inb = fminf(fminf(r16(xc - l, emu), r16(yc - t, emu)), fminf(r16(r - xc, emu), r16(d - yc, emu))) > 0.f;
    float rad = r16(p.radius * st, emu);
    float cl = r16(gx - rad, emu), cr = r16(gx + rad, emu), ct = r16(gy - rad, emu), cb = r16(gy + rad, emu);
    bool inc = fminf(fminf(r16(xc - cl, emu), r16(yc - ct, emu)), fminf(r16(cr - xc, emu), r16(cb - yc, emu))) > 0.f;
    // bboxes_iou(gt, pred, xyxy=False)
    float tlx = fmaxf(l, r16(bx - r16(bw * 0.5f, emu), emu)), tly = fmaxf(t, r16(by - r16(bh * 0.5f, emu), emu));
    float brx = fminf(r, r16(bx + r16(bw * 0.5f, emu), emu)), bry = fminf(d, r16(by + r16(bh * 0.5f, emu), emu));
    float area_a = r16(gw * gh, emu), area_b = r16(bw * bh, emu);
    float en = (tlx < brx && tly < bry) ? 1.f : 0.f;
    float area_i = r16(r16(r16(brx - tlx, emu) * r16(bry - tly, emu), emu) * en, emu);
    float iou = r16(area_i / r16(r16(area_a + area_b, emu) - area_i, emu), emu);
    float eps = emu ? 0.0f : 1e-8f;   // 1e-8 underflows to 0 when added to an fp16 tensor
    float iou_loss = r16(-logf(r16(iou + eps, emu)), emu);
    int gc = (int)L[0];
    float pg = sqrtf((1.0f / (1.0f + expf(-(float)o[5 + gc]))) * so);
    float cls_cost = base + clamp_log(1.0f - pg) - clamp_log(pg);
    p.cost[idx] = cls_cost + r16(3.0f * iou_loss, emu) + ((inb && inc) ? 0.f : 100000.0f);
    p.iou[idx] = iou;
  }
}

// block-wide arg-extreme of (value, index): max (largest=1) or min, ties -> lower index
__device__ void block_argbest(float v, int i, int largest, float* sv, int* si, float& ov, int& oi) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    float v2 = __shfl_xor(v, o, 64);
    int i2 = __shfl_xor(i, o, 64);
    bool take = largest ? (v2 > v || (v2 == v && i2 < i)) : (v2 < v || (v2 == v && i2 < i));
    if (take) { v = v2; i = i2; }
  }
  if (lane == 0) { sv[wv] = v; si[wv] = i; }
  __syncthreads();
  if (tid == 0) {
    float bv = sv[0];
    int bi = si[0];
    for (int w = 1; w < nw; ++w) {
      bool take = largest ? (sv[w] > bv || (sv[w] == bv && si[w] < bi)) : (sv[w] < bv || (sv[w] == bv && si[w] < bi));
      if (take) { bv = sv[w]; bi = si[w]; }
    }
    sv[0] = bv; si[0] = bi;
  }
  __syncthreads();
  ov = sv[0]; oi = si[0];
  __syncthreads();
}

// workgroup per (gt, image): matches[b][g][0..k) = chosen anchors, nmatch[b][g] = k
__global__ __launch_bounds__(1024) void simota_match_kernel(SimotaParams p, int topk, int* __restrict__ matches,
                                                           int* __restrict__ nmatch) {
  __shared__ float sv[16];
  __shared__ int si[16];
  __shared__ int picked[16];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  int G = p.ngt[b];
  if (G > p.G) G = p.G;
  if (g >= G) {
    if (tid == 0) nmatch[b * p.G + g] = 0;
    return;
  }
  const float* iou = p.iou + ((long)b * p.G + g) * p.A;
  const float* cost = p.cost + ((long)b * p.G + g) * p.A;
  // dynamic k: sum of the (up to) 10 largest IoUs among candidate anchors
  float sum = 0.f;
  int npick = 0;
  for (int round = 0; round < topk; ++round) {
    float bv = -2.f;
    int bi = 0x7fffffff;
    for (int a = tid; a < p.A; a += blockDim.x) {
      float v = iou[a];
      if (v < 0.f && !(v != v)) continue;   // not a candidate
      bool used = false;
      for (int q = 0; q < npick; ++q) used = used || picked[q] == a;
      if (used) continue;
      if (v != v) v = 3.0e38f;   // torch.topk ranks NaN above everything
      if (v > bv || (v == bv && a < bi)) { bv = v; bi = a; }
    }
    float ov; int oi;
    block_argbest(bv, bi, 1, sv, si, ov, oi);
    if (oi == 0x7fffffff) break;   // fewer candidates than topk
    if (tid == 0) picked[npick] = oi;
    __syncthreads();
    sum += iou[oi];
    ++npick;
  }
  if (p.emu16) sum = (float)(f16)sum;
  int k = (sum != sum) ? 1 : (int)sum;   // .int() truncates; NaN -> implementation garbage, clamp below
  if (k < 1) k = 1;
  if (k > topk) k = topk;
  // k smallest costs.  +inf is a legal cost (a gt whose fp16 area overflows has IoU 0 with everything: -log(0) = inf) and
  // torch.topk(largest=False) then still returns k entries -- the lowest indices among the ties -- so the running best
  // starts AT +inf with an out-of-range index instead of below it.
  int nm = 0;
  for (int round = 0; round < k; ++round) {
    float bv = __builtin_inff();
    int bi = 0x7fffffff;
    for (int a = tid; a < p.A; a += blockDim.x) {
      if (iou[a] < 0.f && !(iou[a] != iou[a])) continue;
      bool used = false;
      for (int q = 0; q < nm; ++q) used = used || picked[q] == a;
      if (used) continue;
      float v = cost[a];
      if (v != v) continue;   // NaN is never among the smallest
      if (v < bv || (v == bv && a < bi)) { bv = v; bi = a; }
    }
    float ov; int oi;
    block_argbest(bv, bi, 0, sv, si, ov, oi);
    if (oi == 0x7fffffff) break;
    if (tid == 0) { picked[nm] = oi; matches[((long)b * p.G + g) * topk + nm] = oi; }
    __syncthreads();
    ++nm;
  }
  if (tid == 0) nmatch[b * p.G + g] = nm;
}

__global__ void simota_resolve_kernel(SimotaParams p, int topk, const int* __restrict__ matches, const int* __restrict__ nmatch,
                                      unsigned char* __restrict__ fg_mask, int* __restrict__ matched_gt,
                                      float* __restrict__ pred_iou, int* __restrict__ num_fg) {
  const int b = blockIdx.y, a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= p.A) return;
  int G = p.ngt[b];
  if (G > p.G) G = p.G;
  int cnt = 0, first = -1;
  for (int g = 0; g < G; ++g) {
    int n = nmatch[b * p.G + g];
    for (int j = 0; j < n; ++j)
      if (matches[((long)b * p.G + g) * topk + j] == a) { ++cnt; if (first < 0) first = g; }
  }
  int mg = first;
  if (cnt > 1) {
    // torch.min(cost[:, a], dim=0) over ALL gts: the first NaN if there is one (NaN propagates), else the first minimum
    float bv = p.cost[((long)b * p.G) * p.A + a];
    mg = 0;
    for (int g = 1; g < G; ++g) {
      float v = p.cost[((long)b * p.G + g) * p.A + a];
      if (!(bv != bv) && (v != v || v < bv)) { bv = v; mg = g; }
    }
  }
  long o = (long)b * p.A + a;
  fg_mask[o] = cnt > 0;
  matched_gt[o] = cnt > 0 ? mg : -1;
  pred_iou[o] = cnt > 0 ? p.iou[((long)b * p.G + mg) * p.A + a] : 0.f;
  if (cnt > 0) atomicAdd(&num_fg[b], 1);
}

extern "C" long tscd_simota_workspace_bytes(int B, int A, int G, int topk) {
  return 2L * 4 * B * G * A + 4L * B * G * topk + 4L * B * G + 4L * B + 4096;
}

extern "C" int tscd_simota_assign_r(const void* outputs, const float* labels, int B, int A, int NC, int Lmax, int G,
                                    const int* hw3, const int* w3, const int* stride3, int topk, int emu16, float center_radius,
                                    void* workspace, long workspace_bytes, unsigned char* fg_mask, int* matched_gt, float* pred_iou,
                                    int* num_fg, int* ngt, void* stream);
extern "C" int tscd_simota_assign(const void* outputs, const float* labels, int B, int A, int NC, int Lmax, int G,
                                  const int* hw3, const int* w3, const int* stride3, int topk, int emu16, void* workspace,
                                  long workspace_bytes, unsigned char* fg_mask, int* matched_gt, float* pred_iou, int* num_fg,
                                  int* ngt, void* stream) {
  return tscd_simota_assign_r(outputs, labels, B, A, NC, Lmax, G, hw3, w3, stride3, topk, emu16, 4.5f, workspace, workspace_bytes, fg_mask,
                              matched_gt, pred_iou, num_fg, ngt, stream);
}
// The same with the centre-prior radius as an argument (in strides): the YOLOX base head uses 2.5 (yolo_head.py:594).
extern "C" int tscd_simota_assign_r(const void* outputs, const float* labels, int B, int A, int NC, int Lmax, int G,
                                    const int* hw3, const int* w3, const int* stride3, int topk, int emu16, float center_radius,
                                    void* workspace, long workspace_bytes, unsigned char* fg_mask, int* matched_gt, float* pred_iou,
                                    int* num_fg, int* ngt, void* stream) {
  TSCD_CHECK_ARG(outputs && labels && hw3 && w3 && stride3 && workspace && fg_mask && matched_gt && pred_iou && num_fg && ngt,
                 "simota_assign: null pointer");
  TSCD_CHECK_ARG(B > 0 && A > 0 && G > 0 && G <= Lmax && topk > 0 && topk <= 16, "simota_assign: bad sizes");
  TSCD_CHECK_ARG(workspace_bytes >= tscd_simota_workspace_bytes(B, A, G, topk), "simota_assign: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  SimotaParams p;
  p.outputs = (const f16*)outputs; p.labels = labels;
  p.B = B; p.A = A; p.NC = NC; p.Lmax = Lmax; p.G = G; p.emu16 = emu16; p.radius = center_radius;
  int acc = 0;
  for (int k = 0; k < 3; ++k) { p.hw[k] = hw3[k]; p.w[k] = w3[k]; p.stride[k] = stride3[k]; p.a0[k] = acc; acc += hw3[k]; }
  TSCD_CHECK_ARG(acc == A, "simota_assign: level sizes do not add up to A");
  char* ws = (char*)workspace;
  auto take = [&](long bytes) { char* r = ws; ws += (bytes + 255) / 256 * 256; return r; };
  p.cost = (float*)take(4L * B * G * A);
  p.iou = (float*)take(4L * B * G * A);
  int* matches = (int*)take(4L * B * G * topk);
  int* nmatch = (int*)take(4L * B * G);
  p.ngt = ngt;
  (void)hipMemsetAsync(num_fg, 0, 4L * B, st);
  hipLaunchKernelGGL(simota_count_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, labels, B, Lmax, ngt);
  hipLaunchKernelGGL(simota_cost_kernel, dim3(cdiv(A, 256), B), dim3(256), 0, st, p);
  hipLaunchKernelGGL(simota_match_kernel, dim3(G, B), dim3(1024), 0, st, p, topk, matches, nmatch);
  hipLaunchKernelGGL(simota_resolve_kernel, dim3(cdiv(A, 256), B), dim3(256), 0, st, p, topk, matches, nmatch, fg_mask,
                     matched_gt, pred_iou, num_fg);
  TSCD_CHECK_LAUNCH("simota_assign");
  return TSCD_OK;
}
