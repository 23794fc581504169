// Proposal stage, device side: box decode + scoring, per-frame confidence / top-k selection, feature-row gather
// and scatter.  Replaces the host-driven loops of the reference's TSCDHead (tscd_head.py:755-770 decode_outputs,
// :1546-1693 postprocess_widx with use_pre_nms=False / cat_ota_fg=False, :976-1006 find_feature_score), which sync
// the device once per kept proposal.  Here one launch handles every frame of the clip and nothing returns to the
// host except the per-frame counts.
//
// fp16 semantics: the reference runs these ops on fp16 tensors (autocast / model.half()), each elementwise op
// rounding to fp16.  The kernels reproduce that op-by-op rounding so scores / boxes are the same fp16 values.
#include "common.h"

struct LevelTable {
  const f16* raw[3];   // per level [B, h*w, n_ch] (reg 4 | obj 1 | cls NC), written by the three pred convs
  int hw[3], w[3], stride[3], a0[3];
};

__device__ __forceinline__ f16 h_sigmoid(f16 x) { return (f16)(1.0f / (1.0f + expf(-(float)x))); }

// One thread per (frame, anchor).
//   outputs [B,A,n_ch] : decoded cx,cy,w,h + RAW obj/cls logits      (training: SimOTA + losses)      (optional)
//   det     [B,A,7+NC] : x1,y1,x2,y2, sig(obj), max sig(cls), argmax, sig(cls)...  (pred_result rows) (always)
//   score   [B,A]      : sig(obj) * max sig(cls)   (fp16 product, as detections[:,4]*detections[:,5])
__global__ void decode_score_kernel(LevelTable t, int B, int A, int NC, f16* __restrict__ outputs, f16* __restrict__ det,
                                    f16* __restrict__ score) {
  const long total = (long)B * A;
  const int nch = 5 + NC, ndet = 7 + NC;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int b = idx / A, a = idx % A;
    int k = a >= t.a0[2] ? 2 : (a >= t.a0[1] ? 1 : 0);
    int la = a - t.a0[k];
    const f16* r = t.raw[k] + ((long)b * t.hw[k] + la) * nch;
    f16 gx = (f16)(float)(la % t.w[k]), gy = (f16)(float)(la / t.w[k]);
    f16 st = (f16)(float)t.stride[k];
    // decode_outputs: (xy + grid) * stride ; exp(wh) * stride   -- each op rounds to fp16
    f16 cx = (f16)((float)(f16)((float)r[0] + (float)gx) * (float)st);
    f16 cy = (f16)((float)(f16)((float)r[1] + (float)gy) * (float)st);
    f16 bw = (f16)((float)(f16)expf((float)r[2]) * (float)st);
    f16 bh = (f16)((float)(f16)expf((float)r[3]) * (float)st);
    if (outputs) {
      f16* o = outputs + idx * nch;
      o[0] = cx; o[1] = cy; o[2] = bw; o[3] = bh;
      for (int c = 4; c < nch; ++c) o[c] = r[c];
    }
    f16* d = det + idx * ndet;
    // postprocess_widx: cxcywh -> xyxy with w/2 (exact in fp16) then add/sub rounded
    f16 hw2 = (f16)((float)bw * 0.5f), hh2 = (f16)((float)bh * 0.5f);
    d[0] = (f16)((float)cx - (float)hw2);
    d[1] = (f16)((float)cy - (float)hh2);
    d[2] = (f16)((float)cx + (float)hw2);
    d[3] = (f16)((float)cy + (float)hh2);
    f16 so = h_sigmoid(r[4]);
    d[4] = so;
    f16 best = (f16)(-1.f);
    int bi = 0;
    for (int c = 0; c < NC; ++c) {
      f16 sc = h_sigmoid(r[5 + c]);
      d[7 + c] = sc;
      if (sc > best) { best = sc; bi = c; }   // first maximum, like torch.max
    }
    d[5] = best;
    d[6] = (f16)(float)bi;
    score[idx] = (f16)((float)so * (float)best);
  }
}

extern "C" int tscd_decode_score(const void* raw0, const void* raw1, const void* raw2, const int* hw3, const int* w3,
                                 const int* stride3, int B, int NC, void* outputs, void* det, void* score, void* stream) {
  TSCD_CHECK_ARG(raw0 && raw1 && raw2 && det && score && hw3 && w3 && stride3, "decode_score: null pointer");
  LevelTable t;
  t.raw[0] = (const f16*)raw0; t.raw[1] = (const f16*)raw1; t.raw[2] = (const f16*)raw2;
  int A = 0;
  for (int k = 0; k < 3; ++k) {
    TSCD_CHECK_ARG(hw3[k] > 0 && w3[k] > 0 && hw3[k] % w3[k] == 0, "decode_score: bad level %d", k);
    t.hw[k] = hw3[k]; t.w[k] = w3[k]; t.stride[k] = stride3[k]; t.a0[k] = A;
    A += hw3[k];
  }
  long total = (long)B * A;
  int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(decode_score_kernel, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, (hipStream_t)stream, t, B, A, NC,
                     (f16*)outputs, (f16*)det, (f16*)score);
  TSCD_CHECK_LAUNCH("decode_score");
  return TSCD_OK;
}

// ------------------------------------------------------------------------------------------------------
// Per-frame selection (postprocess_widx :1591-1607,1622): keep score >= thr; if fewer than `min_keep` keep the
// top `min_keep`; if more than `max_keep` (>0) keep the top `max_keep`.  Output: ascending anchor indices
// (torch.where order) + count.  One workgroup per frame.  Top-k = exact radix select on the fp16 bit
// pattern (scores are >= 0 so the pattern is monotone); ties at the k-th value are broken towards the LOWER anchor
// index (torch.topk leaves tie order unspecified; this choice is deterministic and documented in DESIGN.md).
// ------------------------------------------------------------------------------------------------------
#define SEL_THREADS 1024
__global__ __launch_bounds__(SEL_THREADS) void select_kernel(const f16* __restrict__ score, int A, f16 thr, int min_keep,
                                                             int max_keep, int cap, int* __restrict__ out_idx,
                                                             int* __restrict__ out_count) {
  __shared__ int hist[256];
  __shared__ int s_misc[8];
  __shared__ int wsum[SEL_THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const unsigned short* key = (const unsigned short*)(score + (long)b * A);
  const unsigned short thr_bits = __builtin_bit_cast(unsigned short, thr);

  // count(score >= thr)
  int cnt = 0;
  for (int a = tid; a < A; a += SEL_THREADS) cnt += (key[a] >= thr_bits && key[a] < 0x8000) ? 1 : 0;
  cnt = (int)wave_sum((float)cnt);
  if ((tid & 63) == 0) wsum[tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0) {
    int c = 0;
    for (int i = 0; i < SEL_THREADS / 64; ++i) c += wsum[i];
    int K = c;
    if (min_keep > 0 && c < min_keep) K = min_keep;
    if (max_keep > 0 && c > max_keep) K = max_keep;
    if (K > A) K = A;
    s_misc[0] = c;
    s_misc[1] = K;
  }
  __syncthreads();
  const int above = s_misc[0], K = s_misc[1];
  unsigned short T;        // k-th largest key
  int n_greater;           // #keys strictly greater than T
  if (K == above) {
    T = thr_bits; n_greater = -1;   // plain threshold selection: take every key >= thr
  } else {
    // radix select, high byte then low byte (negative scores cannot occur: sigmoid products)
    int remaining = K;
    unsigned prefix = 0;
    for (int pass = 0; pass < 2; ++pass) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      for (int a = tid; a < A; a += SEL_THREADS) {
        unsigned k = key[a];
        if (k >= 0x8000) k = 0;   // -0 / negatives sort lowest
        if (pass == 0) atomicAdd(&hist[k >> 8], 1);
        else if ((k >> 8) == prefix) atomicAdd(&hist[k & 255], 1);
      }
      __syncthreads();
      if (tid == 0) {
        int acc = 0, bin = 255;
        for (; bin > 0; --bin) {
          if (acc + hist[bin] >= remaining) break;
          acc += hist[bin];
        }
        s_misc[2] = bin;
        s_misc[3] = remaining - acc;   // how many to take inside this bin
        s_misc[4] = (pass == 0 ? 0 : s_misc[4]) + acc;
      }
      __syncthreads();
      if (pass == 0) { prefix = s_misc[2]; remaining = s_misc[3]; }
      else { prefix = (prefix << 8) | s_misc[2]; remaining = s_misc[3]; }
      __syncthreads();
    }
    T = (unsigned short)prefix;
    n_greater = K - remaining;   // keys > T all selected; `remaining` keys == T selected in index order
  }
  const int ties_to_take = (n_greater < 0) ? (1 << 30) : (K - n_greater);

  // ordered compaction: chunks of SEL_THREADS anchors, block-wide exclusive scan of the keep flags
  __shared__ int base, tie_base;
  if (tid == 0) { base = 0; tie_base = 0; }
  __syncthreads();
  for (int a0 = 0; a0 < A; a0 += SEL_THREADS) {
    int a = a0 + tid;
    unsigned k = a < A ? key[a] : 0;
    if (k >= 0x8000) k = 0;
    bool gt = a < A && k > T, eq = a < A && k == T;
    // rank of this tie among ties (index order)
    unsigned long long eqm = __ballot(eq);
    int lane = tid & 63, wv = tid >> 6;
    int eq_before = __popcll(eqm & ((1ull << lane) - 1));
    if (lane == 0) wsum[wv] = __popcll(eqm);
    __syncthreads();
    int eq_off = tie_base;
    for (int i = 0; i < wv; ++i) eq_off += wsum[i];
    int eq_total = 0;
    for (int i = 0; i < SEL_THREADS / 64; ++i) eq_total += wsum[i];
    __syncthreads();
    bool keep = gt || (eq && (eq_off + eq_before) < ties_to_take);
    unsigned long long km = __ballot(keep);
    int keep_before = __popcll(km & ((1ull << lane) - 1));
    if (lane == 0) wsum[wv] = __popcll(km);
    __syncthreads();
    int off = base;
    for (int i = 0; i < wv; ++i) off += wsum[i];
    int keep_total = 0;
    for (int i = 0; i < SEL_THREADS / 64; ++i) keep_total += wsum[i];
    if (keep && off + keep_before < cap) out_idx[(long)b * cap + off + keep_before] = a;
    __syncthreads();
    if (tid == 0) { base += keep_total; tie_base += eq_total; }
    __syncthreads();
  }
  if (tid == 0) out_count[b] = base < cap ? base : cap;
}

extern "C" int tscd_select_proposals(const void* score, int B, int A, float conf_thresh, int min_keep, int max_keep, int cap,
                                     int* out_idx, int* out_count, void* stream) {
  TSCD_CHECK_ARG(score && out_idx && out_count && B > 0 && A > 0 && cap > 0, "select_proposals: bad args");
  TSCD_CHECK_ARG(max_keep == 0 || max_keep <= cap, "select_proposals: cap %d < max_keep %d", cap, max_keep);
  TSCD_CHECK_ARG(min_keep <= cap, "select_proposals: cap %d < min_keep %d", cap, min_keep);
  f16 thr = (f16)conf_thresh;   // the reference compares an fp16 tensor with the python scalar => scalar rounds to fp16
  hipLaunchKernelGGL(select_kernel, dim3(B), dim3(SEL_THREADS), 0, (hipStream_t)stream, (const f16*)score, A, thr, min_keep,
                     max_keep, cap, out_idx, out_count);
  TSCD_CHECK_LAUNCH("select_proposals");
  return TSCD_OK;
}

// ------------------------------------------------------------------------------------------------------
// Row gather (find_feature_score): for proposal n of frame b at anchor a, copy W-channel rows of the three per-level
// NHWC feature maps (cls_convs2 / reg_convs2 / edge) and its detection row.  One wave per (proposal, map).
// rows[n] = (b, a) pairs are built on the device from the selection output (frame offsets from a host-side
// prefix sum of the counts).  scatter = exact adjoint (each anchor is selected at most once per frame).
// ------------------------------------------------------------------------------------------------------
struct MapTable {
  const f16* map[3][3];   // [which feature][level]  -> [B, hw_k, W]
  f16* gmap[3][3];
  int hw[3], a0[3];
};

__global__ void gather_rows_kernel(MapTable t, const int* __restrict__ sel_idx, int cap, const int* __restrict__ frame_off,
                                   int B, int N, int W, int ld, int nmaps, f16* __restrict__ out0, f16* __restrict__ out1,
                                   f16* __restrict__ out2) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  int n = wave / nmaps, f = wave % nmaps;
  if (n >= N) return;
  int b = 0;
  while (b + 1 < B && frame_off[b + 1] <= n) ++b;
  int a = sel_idx[(long)b * cap + (n - frame_off[b])];
  int k = a >= t.a0[2] ? 2 : (a >= t.a0[1] ? 1 : 0);
  const f16* src = t.map[f][k] + ((long)b * t.hw[k] + (a - t.a0[k])) * ld;
  f16* dst = (f == 0 ? out0 : (f == 1 ? out1 : out2)) + (long)n * W;
  for (int c = lane * 8; c < W; c += 512) *(f16x8*)(dst + c) = *(const f16x8*)(src + c);
}

// addmask bit f: feature f's rows are ADDED to what its maps hold (a gradient map another consumer of the same feature map has
// already written; proposals are distinct anchors of a frame, so no two rows meet) instead of written into a zeroed map
__global__ void scatter_rows_kernel(MapTable t, const int* __restrict__ sel_idx, int cap, const int* __restrict__ frame_off,
                                    int B, int N, int W, int ld, int nmaps, const f16* __restrict__ g0, const f16* __restrict__ g1,
                                    const f16* __restrict__ g2, int addmask) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  int n = wave / nmaps, f = wave % nmaps;
  if (n >= N) return;
  int b = 0;
  while (b + 1 < B && frame_off[b + 1] <= n) ++b;
  int a = sel_idx[(long)b * cap + (n - frame_off[b])];
  int k = a >= t.a0[2] ? 2 : (a >= t.a0[1] ? 1 : 0);
  f16* dst = t.gmap[f][k] + ((long)b * t.hw[k] + (a - t.a0[k])) * ld;
  const f16* src = (f == 0 ? g0 : (f == 1 ? g1 : g2)) + (long)n * W;
  if ((addmask >> f) & 1) {
    for (int c = lane * 8; c < W; c += 512) {
      f16x8 v = *(const f16x8*)(dst + c);
      const f16x8 g = *(const f16x8*)(src + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] + (float)g[e]);        // one rounding, as the tensor add it replaces
      *(f16x8*)(dst + c) = v;
    }
  } else {
    for (int c = lane * 8; c < W; c += 512) *(f16x8*)(dst + c) = *(const f16x8*)(src + c);
  }
}

// maps: 9 pointers [feature 0..2][level 0..2]; features beyond nmaps may be null.  map_ld: pixel stride of every map in elements (W
// for contiguous maps; the towers' outputs are W-channel slices of a wider buffer).
extern "C" int tscd_gather_rows(const void* const* maps, const int* hw3, const int* sel_idx, int cap, const int* frame_off,
                                int B, int N, int W, int map_ld, int nmaps, void* out0, void* out1, void* out2, int scatter,
                                void* stream) {
  // scatter: 0 gather; bit 0 set: scatter out0..2 into the maps, bits 1..3: feature 0..2 ADDS to its maps instead of overwriting
  TSCD_CHECK_ARG(maps && hw3 && sel_idx && frame_off && N >= 0 && W % 8 == 0 && map_ld % 8 == 0 && map_ld >= W && nmaps >= 1 && nmaps <= 3,
                 "gather_rows: bad args");
  if (N == 0) return TSCD_OK;
  MapTable t;
  int A = 0;
  for (int k = 0; k < 3; ++k) { t.hw[k] = hw3[k]; t.a0[k] = A; A += hw3[k]; }
  for (int f = 0; f < 3; ++f)
    for (int k = 0; k < 3; ++k) {
      t.map[f][k] = f < nmaps ? (const f16*)maps[f * 3 + k] : nullptr;
      t.gmap[f][k] = f < nmaps ? (f16*)maps[f * 3 + k] : nullptr;
      if (f < nmaps) TSCD_CHECK_ARG(maps[f * 3 + k] != nullptr && ((uintptr_t)maps[f * 3 + k] & 15) == 0, "gather_rows: bad map");
    }
  long waves = (long)N * nmaps;
  dim3 grid((unsigned)((waves * 64 + 255) / 256)), block(256);
  if (!(scatter & 1))
    hipLaunchKernelGGL(gather_rows_kernel, grid, block, 0, (hipStream_t)stream, t, sel_idx, cap, frame_off, B, N, W, map_ld, nmaps,
                       (f16*)out0, (f16*)out1, (f16*)out2);
  else
    hipLaunchKernelGGL(scatter_rows_kernel, grid, block, 0, (hipStream_t)stream, t, sel_idx, cap, frame_off, B, N, W, map_ld, nmaps,
                       (const f16*)out0, (const f16*)out1, (const f16*)out2, (scatter >> 1) & 7);
  TSCD_CHECK_LAUNCH("gather_rows");
  return TSCD_OK;
}

// Gather detection rows (pred_result) + per-proposal scores: det [B,A,ndet] -> out [N,ndet].
__global__ void gather_det_kernel(const f16* __restrict__ det, int A, int ndet, const int* __restrict__ sel_idx, int cap,
                                  const int* __restrict__ frame_off, int B, int N, f16* __restrict__ out, int* __restrict__ row_b,
                                  int* __restrict__ row_a) {
  long total = (long)N * ndet;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int n = idx / ndet, c = idx % ndet;
    int b = 0;
    while (b + 1 < B && frame_off[b + 1] <= n) ++b;
    int a = sel_idx[(long)b * cap + (n - frame_off[b])];
    out[idx] = det[((long)b * A + a) * ndet + c];
    if (c == 0 && row_b) { row_b[n] = b; row_a[n] = a; }
  }
}
// row_b / row_a (optional, both or neither): the (frame, anchor) of every gathered row -- what the reference rebuilds with
// torch.where / cat over the per-frame index lists (tscd_head.py:1647-1651).
extern "C" int tscd_gather_det_rows(const void* det, int A, int ndet, const int* sel_idx, int cap, const int* frame_off, int B, int N,
                                    void* out, int* row_b, int* row_a, void* stream) {
  TSCD_CHECK_ARG(det && sel_idx && frame_off && out && (!row_b == !row_a), "gather_det: null pointer");
  if (N == 0) return TSCD_OK;
  long total = (long)N * ndet;
  int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(gather_det_kernel, dim3(grid > 2048 ? 2048 : grid), dim3(256), 0, (hipStream_t)stream, (const f16*)det, A,
                     ndet, sel_idx, cap, frame_off, B, N, (f16*)out, row_b, row_a);
  TSCD_CHECK_LAUNCH("gather_det");
  return TSCD_OK;
}
extern "C" int tscd_gather_det(const void* det, int A, int ndet, const int* sel_idx, int cap, const int* frame_off, int B, int N,
                               void* out, void* stream) {
  return tscd_gather_det_rows(det, A, ndet, sel_idx, cap, frame_off, B, N, out, nullptr, nullptr, stream);
}
