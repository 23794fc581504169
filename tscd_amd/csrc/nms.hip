// Eval tail on the device (reference: decode_reg_preds5 tscd_head.py:914-949, post_process.postprocess
// yolox/models/post_process.py:9-85, torchvision.ops.batched_nms).
//
//   nms_expand   per local frame: refined boxes = base xyxy (+) deltas, obj = sig(matched obj), class scores =
//                sig(refined cls); every (proposal, class) with score >= thr and obj*score >= thr becomes a
//                candidate (row-major order, as torch.where).  Mode 1 builds the "original detections" set instead
//                (one candidate per proposal: class = argmax id, score = obj * cls_conf).
//   nms_sort     bitonic sort by (score desc, candidate index asc) -- torchvision visits boxes in decreasing score order.  Up to
//                65536 candidates per frame: one workgroup per frame, 32-bit keys, <= 16384 entirely in LDS, beyond that with
//                global-memory cross-chunk steps.  More (the ImageNet-VID exp sets no upper proposal clamp: up to A anchors x 30
//                classes = 204 120 candidates at 576^2, post_process.py:9-85 takes whatever arrives): 64-bit keys in global memory,
//                the same network spread over many workgroups -- chunk-local stages in LDS, cross-chunk strides one launch each.
//   nms_greedy   one workgroup per (frame, class): class-aware NMS == independent greedy NMS per class (what the
//                coordinate-offset trick of batched_nms implements).  The class's boxes are pulled from the sorted list into an
//                LDS tile of 2048 and suppressed greedily with IoU > thr (fp32 arithmetic); a class with more boxes runs tile by
//                tile (blocked greedy NMS: a tile is first suppressed by the boxes kept in earlier tiles -- a global list of at
//                most max_rows boxes per (frame, class), since a proposal gives a class at most one candidate -- then within
//                itself): the same keep set as the one-tile pass, no capacity limit.
//                coord_trick = n > 0: the reference-faithful arithmetic of torchvision.ops.batched_nms on fp16 boxes for frames
//                with at most n candidates -- every box is shifted by fp16(class id) * fp16(max coordinate + 1) and rounded to
//                fp16 before the IoUs (boxes of the high class ids quantise to 4-8 px; tests/test_deviation_bounds.py) --
//                instead of exact per-class suppression (torchvision itself switches to per-class NMS without offsets above
//                5000 boxes on a GPU: n = 5000 reproduces that rule, 0 = never, INT_MAX = always).
//   nms_compact  ordered compaction of the survivors -> [k,7] rows (x1,y1,x2,y2,obj,cls_score,cls_id) in decreasing
//                score order, plus the kept candidate indices (INT output) and the count.
// torchvision is absent from the reference tree, so NMS parity is pinned by oracle/nms_ref.py's restatement only.
#include "common.h"

#define NMS_MAXC32 65536  // candidates per frame the 32-bit sort key can index (16-bit index field); more: 64-bit keys
#define NMS_MAXPC 2048    // boxes per (frame, class) in one LDS tile of the greedy pass
#define NMS_FLUSH (NMS_MAXPC - 256)  // a tile is processed (and its survivors appended to the kept list) once more than this many boxes are pulled

// ---- candidate record arrays, all [L][capM] -------------------------------------------------------------
struct Cand {
  float* box;      // [L][capM][4]
  float* score;    // nms score = obj * cls_score (fp16-rounded product, as fp32)
  f16* aux;        // [L][capM][3]: obj, cls_score, cls_id  (fp16 values of the output row)
  int* cls;        // class id
  int* count;      // [L]
  float* maxc;     // [L] largest coordinate among the frame's candidates (the coordinate trick's offset unit)
};

__device__ __forceinline__ f16 hs(f16 x) { return (f16)(1.0f / (1.0f + expf(-(float)x))); }
__device__ __forceinline__ f16 hr(float x) { return (f16)x; }

// One workgroup per frame; rows processed in chunks with an ordered block scan.
__global__ __launch_bounds__(256) void nms_expand_kernel(const f16* __restrict__ det, int ndet, const int* __restrict__ frame_off,
                                                        const f16* __restrict__ cls_logit, const f16* __restrict__ obj_logit,
                                                        const f16* __restrict__ reg_delta, int NC, float thr, int capM,
                                                        int mode, float clip, Cand c) {
  __shared__ int wsum[4];
  __shared__ int base;
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r0 = frame_off[f], n = frame_off[f + 1] - r0;
  const f16 thr16 = (f16)thr;
  if (tid == 0) base = 0;
  __syncthreads();
  float cmax = -INFINITY;
  const int per_row = mode == 0 ? NC : 1;          // modes 1 / 2: one candidate per row (class = arg-max id)
  const long total = (long)n * per_row;
  for (long i0 = 0; i0 < total; i0 += 256) {
    long i = i0 + tid;
    bool keep = false;
    float bx[4] = {0, 0, 0, 0}, sc = 0.f;
    f16 a_obj = (f16)0.f, a_cs = (f16)0.f;
    int cid = 0;
    if (i < total) {
      int row = (int)(i / per_row), cc = (int)(i % per_row);
      const f16* d = det + (long)(r0 + row) * ndet;
      if (mode == 0) {
        f16 s = hs(cls_logit[(long)(r0 + row) * NC + cc]);
        f16 o = hs(obj_logit[r0 + row]);
        f16 prod = hr((float)o * (float)s);
        keep = s >= thr16 && prod >= thr16;
        if (keep) {
          // decode_reg_preds5, every op rounded to fp16 like the reference's fp16 tensors
          const f16* dl = reg_delta + (long)(r0 + row) * 4;
          f16 w = hr((float)d[2] - (float)d[0]), h = hr((float)d[3] - (float)d[1]);
          f16 cx = hr((float)d[0] + (float)hr(0.5f * (float)w)), cy = hr((float)d[1] + (float)hr(0.5f * (float)h));
          f16 dw = hr(fminf((float)dl[2], clip)), dh = hr(fminf((float)dl[3], clip));
          f16 pcx = hr((float)hr((float)dl[0] * (float)w) + (float)cx), pcy = hr((float)hr((float)dl[1] * (float)h) + (float)cy);
          f16 pw = hr((float)hr(expf((float)dw)) * (float)w), ph = hr((float)hr(expf((float)dh)) * (float)h);
          bx[0] = (float)hr((float)pcx - (float)hr(0.5f * (float)pw));
          bx[1] = (float)hr((float)pcy - (float)hr(0.5f * (float)ph));
          bx[2] = (float)hr((float)pcx + (float)hr(0.5f * (float)pw));
          bx[3] = (float)hr((float)pcy + (float)hr(0.5f * (float)ph));
          sc = (float)prod; a_obj = o; a_cs = s; cid = cc;
        }
      } else {
        f16 prod = hr((float)d[4] * (float)d[5]);
        keep = prod >= thr16;
        if (keep) {
          bx[0] = (float)d[0]; bx[1] = (float)d[1]; bx[2] = (float)d[2]; bx[3] = (float)d[3];
          sc = (float)prod; a_obj = d[4]; a_cs = d[5]; cid = (int)(float)d[6];
        }
      }
    }
    unsigned long long m = __ballot(keep);
    int before = __popcll(m & ((1ull << lane) - 1));
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wv; ++w) off += wsum[w];
    int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (keep) {
      int pos = off + before;
      if (pos < capM) {
        long o = (long)f * capM + pos;
        c.box[o * 4 + 0] = bx[0]; c.box[o * 4 + 1] = bx[1]; c.box[o * 4 + 2] = bx[2]; c.box[o * 4 + 3] = bx[3];
        c.score[o] = sc;
        cmax = fmaxf(cmax, fmaxf(fmaxf(bx[0], bx[1]), fmaxf(bx[2], bx[3])));
        c.aux[o * 3 + 0] = a_obj; c.aux[o * 3 + 1] = a_cs; c.aux[o * 3 + 2] = (f16)(float)cid;
        c.cls[o] = mode == 2 ? 0 : cid;            // mode 2: class-agnostic suppression (utils/boxes.py:59-64), the row keeps its class
      }
    }
    __syncthreads();
    if (tid == 0) base += tot;
    __syncthreads();
  }
  cmax = wave_max(cmax);
  __syncthreads();
  if (lane == 0) wsum[wv] = __float_as_int(cmax);
  __syncthreads();
  if (tid == 0) {
    c.count[f] = base < capM ? base : capM;
    c.maxc[f] = fmaxf(fmaxf(__int_as_float(wsum[0]), __int_as_float(wsum[1])), fmaxf(__int_as_float(wsum[2]), __int_as_float(wsum[3])));
  }
}

// Bitonic sort of (score desc, index asc) keys.  Scores are fp16-representable and non-negative, so their fp16 bit pattern
// orders them: key = fp16 bits << 16 | (0xFFFF - index), larger key first.  Up to 16384 surviving candidates of a frame are
// sorted entirely in LDS (64 KiB).  Frames with more (the ImageNet-VID exp has no upper proposal clamp: 2000 proposals x 30
// classes = 60000 candidates) keep their keys in a global workspace: compare-exchange steps whose stride reaches across
// 16384-key chunks run on global memory (L2-resident, <= 256 KiB per frame), every run of smaller strides is done chunk by
// chunk in LDS.  Same network, same keys => the same order as the single-chunk sort.  This 32-bit form indexes 65536 candidates per
// frame (the index field of the key); larger frames take the 64-bit form below.
#define NMS_CHUNK 16384
__device__ __forceinline__ void bitonic_lds_steps(unsigned int* keys, int base, int n, int size, int first_stride, int tid) {
  // strides first_stride .. 1 of the merge of width `size`, for the n keys (n a power of two <= NMS_CHUNK) at global offset `base`
  for (int stride = first_stride; stride > 0; stride >>= 1) {
    for (int i = tid; i < (n >> 1); i += 1024) {
      int lo = (i / stride) * (stride << 1) + (i % stride), hi = lo + stride;
      bool desc = (((base + lo) & size) == 0);   // descending runs first => overall descending
      unsigned int a = keys[lo], b = keys[hi];
      if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void nms_sort_kernel(const float* __restrict__ score, const int* __restrict__ count, int capM,
                                                       unsigned int* __restrict__ gkeys, int gstride, int* __restrict__ order) {
  extern __shared__ unsigned int keys[];
  const int f = blockIdx.x, tid = threadIdx.x;
  const int M = count[f];
  int P = 1;
  while (P < M) P <<= 1;
  auto make_key = [&](int i) -> unsigned int {
    if (i >= M) return 0u;   // padding sorts last (real keys have a non-zero low half ... or a non-zero score)
    f16 h = (f16)score[(long)f * capM + i];
    return ((unsigned int)__builtin_bit_cast(unsigned short, h) << 16) | (unsigned)(0xFFFFu - (unsigned)i);
  };
  if (P <= NMS_CHUNK) {
    for (int i = tid; i < P; i += 1024) keys[i] = make_key(i);
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) bitonic_lds_steps(keys, 0, P, size, size >> 1, tid);
    for (int i = tid; i < M; i += 1024) order[(long)f * capM + i] = (int)(0xFFFFu - (keys[i] & 0xFFFFu));
    return;
  }
  unsigned int* gk = gkeys + (long)f * gstride;
  // chunks: fill + all merges that stay inside a chunk
  for (int c0 = 0; c0 < P; c0 += NMS_CHUNK) {
    for (int i = tid; i < NMS_CHUNK; i += 1024) keys[i] = make_key(c0 + i);
    __syncthreads();
    for (int size = 2; size <= NMS_CHUNK; size <<= 1) bitonic_lds_steps(keys, c0, NMS_CHUNK, size, size >> 1, tid);
    for (int i = tid; i < NMS_CHUNK; i += 1024) gk[c0 + i] = keys[i];
    __syncthreads();
  }
  for (int size = 2 * NMS_CHUNK; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride >= NMS_CHUNK; stride >>= 1) {       // across chunks: global memory
      for (int i = tid; i < (P >> 1); i += 1024) {
        int lo = (i / stride) * (stride << 1) + (i % stride), hi = lo + stride;
        bool desc = ((lo & size) == 0);
        unsigned int a = gk[lo], b = gk[hi];
        if ((a < b) == desc) { gk[lo] = b; gk[hi] = a; }
      }
      __syncthreads();                                                        // same workgroup, same CU: stores visible after the barrier
    }
    for (int c0 = 0; c0 < P; c0 += NMS_CHUNK) {                               // strides < chunk: chunk by chunk in LDS
      for (int i = tid; i < NMS_CHUNK; i += 1024) keys[i] = gk[c0 + i];
      __syncthreads();
      bitonic_lds_steps(keys, c0, NMS_CHUNK, size, NMS_CHUNK >> 1, tid);
      for (int i = tid; i < NMS_CHUNK; i += 1024) gk[c0 + i] = keys[i];
      __syncthreads();
    }
  }
  for (int i = tid; i < M; i += 1024) order[(long)f * capM + i] = (int)(0xFFFFu - (gk[i] & 0xFFFFu));
}

// ---- 64-bit-key sort for frames with more than 65536 candidates: key = fp16 score bits << 32 | (0xFFFFFFFF - index), larger first.
// Keys live in global memory [L][P] (P = capM rounded up to a power of two); frame f only sorts its first Pf = pow2(count[f]) keys
// (stages of a width above Pf are skipped for it), so the work follows the survivors of the score filter, not the capacity.
#define NMS_CH64 8192       // keys per LDS chunk (64 KiB)
__device__ __forceinline__ int nms_pow2_dev(int m) {
  int P = 1;
  while (P < m) P <<= 1;
  return P;
}
__device__ __forceinline__ void bitonic64_lds_steps(unsigned long long* keys, int base, int size, int first_stride, int tid) {
  for (int stride = first_stride; stride > 0; stride >>= 1) {
    for (int i = tid; i < (NMS_CH64 >> 1); i += 1024) {
      int lo = (i / stride) * (stride << 1) + (i % stride), hi = lo + stride;
      bool desc = (((base + lo) & size) == 0);
      unsigned long long a = keys[lo], b = keys[hi];
      if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
    }
    __syncthreads();
  }
}
// stage 0: build the keys of one chunk and run every merge that stays inside it (widths 2 .. NMS_CH64)
__global__ __launch_bounds__(1024) void nms_sort64_local_kernel(const float* __restrict__ score, const int* __restrict__ count, int capM,
                                                               unsigned long long* __restrict__ gkeys, int P) {
  extern __shared__ unsigned long long keys64[];
  const int f = blockIdx.y, c0 = blockIdx.x * NMS_CH64, tid = threadIdx.x;
  const int M = count[f];
  if (c0 >= nms_pow2_dev(M) && c0 > 0) return;
  for (int i = tid; i < NMS_CH64; i += 1024) {
    const int g = c0 + i;
    unsigned long long k = 0ull;
    if (g < M) {
      f16 h = (f16)score[(long)f * capM + g];
      k = ((unsigned long long)__builtin_bit_cast(unsigned short, h) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)g);
    }
    keys64[i] = k;
  }
  __syncthreads();
  for (int size = 2; size <= NMS_CH64; size <<= 1) bitonic64_lds_steps(keys64, c0, size, size >> 1, tid);
  unsigned long long* gk = gkeys + (long)f * P;
  for (int i = tid; i < NMS_CH64; i += 1024) gk[c0 + i] = keys64[i];
}
// one cross-chunk compare-exchange stride of the merge of width `size`
__global__ __launch_bounds__(256) void nms_sort64_global_kernel(const int* __restrict__ count, unsigned long long* __restrict__ gkeys, int P,
                                                               int size, int stride) {
  const int f = blockIdx.y;
  const int Pf = nms_pow2_dev(count[f]);
  if (size > Pf) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (Pf >> 1)) return;
  unsigned long long* gk = gkeys + (long)f * P;
  const int lo = (i / stride) * (stride << 1) + (i % stride), hi = lo + stride;
  const bool desc = ((lo & size) == 0);
  unsigned long long a = gk[lo], b = gk[hi];
  if ((a < b) == desc) { gk[lo] = b; gk[hi] = a; }
}
// the strides below a chunk of the merge of width `size`, chunk by chunk in LDS
__global__ __launch_bounds__(1024) void nms_sort64_merge_kernel(const int* __restrict__ count, unsigned long long* __restrict__ gkeys, int P,
                                                               int size) {
  extern __shared__ unsigned long long keys64[];
  const int f = blockIdx.y, c0 = blockIdx.x * NMS_CH64, tid = threadIdx.x;
  const int Pf = nms_pow2_dev(count[f]);
  if (size > Pf || c0 >= Pf) return;
  unsigned long long* gk = gkeys + (long)f * P;
  for (int i = tid; i < NMS_CH64; i += 1024) keys64[i] = gk[c0 + i];
  __syncthreads();
  bitonic64_lds_steps(keys64, c0, size, NMS_CH64 >> 1, tid);
  for (int i = tid; i < NMS_CH64; i += 1024) gk[c0 + i] = keys64[i];
}
__global__ __launch_bounds__(256) void nms_sort64_order_kernel(const int* __restrict__ count, const unsigned long long* __restrict__ gkeys,
                                                              int P, int capM, int* __restrict__ order) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i < count[f]) order[(long)f * capM + i] = (int)(0xFFFFFFFFu - (unsigned)(gkeys[(long)f * P + i] & 0xFFFFFFFFull));
}

// IoU(a, b) > thr with a = the earlier (kept) box, every product and sum rounded on its own like the oracle's numpy float32
// arithmetic: with fused multiply-adds (hipcc's default contraction) an IoU within an ulp of the threshold flips -- 2 of 96 000
// candidates in tests/test_gpu_nms_unbounded.py, and differently in differently shaped loops.
__device__ __forceinline__ bool nms_suppresses(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2, float by2,
                                               float thr) {
#pragma clang fp contract(off)
  const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f);
  const float ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
  const float inter = iw * ih;
  const float aa = (ax2 - ax1) * (ay2 - ay1);
  const float ab = (bx2 - bx1) * (by2 - by1);
  const float iou = inter / (aa + ab - inter);
  return iou > thr;
}

// Greedy NMS for one (frame, class), tile by tile.  keep[f][sorted position] = 1 for survivors.
// agnostic != 0 (the coordinate-trick pass): ONE workgroup per frame takes every candidate, each shifted by its own class offset --
// torchvision's batched_nms is a class-agnostic NMS on the shifted boxes, and in fp16 boxes of neighbouring classes do meet (offsets
// are multiples of max + 1, boxes reach below 0 and the shifted coordinates round to 4-8 px).  Frames are split between the two
// passes by their candidate count: coord_trick = n sends frames with <= n candidates to the agnostic pass, the others to the
// per-class pass.
// kept_box [L][NCg][max_rows][4]: the boxes this (frame, class) has kept so far (only touched when the class overflows one tile).
__global__ __launch_bounds__(256) void nms_greedy_kernel(const float* __restrict__ box, const int* __restrict__ cls,
                                                        const int* __restrict__ order, const int* __restrict__ count, int capM,
                                                        float iou_thr, unsigned char* __restrict__ keep, float* __restrict__ kept_box,
                                                        int max_rows, const float* __restrict__ maxc, int coord_trick, int agnostic) {
  __shared__ float bx[NMS_MAXPC][4];
  __shared__ int pos[NMS_MAXPC];
  __shared__ unsigned char dead[NMS_MAXPC];
  __shared__ float kst[256][4];              // staging of the kept list (cross-tile suppression)
  __shared__ int wsum[4];
  __shared__ int base, nkept_s;
  const int f = blockIdx.x, cid = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int M = count[f];
  const bool trick = coord_trick > 0 && M <= coord_trick;
  if (trick != (agnostic != 0)) return;                  // this frame belongs to the other pass (uniform per workgroup)
  float* kb = kept_box ? kept_box + ((long)f * gridDim.y + cid) * max_rows * 4 : nullptr;
  // coordinate trick (torchvision.ops.batched_nms on the evaluator's fp16 tensors): offset = fp16(class id) * fp16(max + 1)
  const float unit = trick ? (float)(f16)((float)(f16)maxc[f] + 1.0f) : 0.f;
  if (tid == 0) { base = 0; nkept_s = 0; }
  __syncthreads();
  // one tile: suppress by the boxes kept in earlier tiles, greedy inside the tile, publish the survivors
  auto process_tile = [&](int n, bool more) {
    const int nk = nkept_s;
    // The kept list is written and re-read by this workgroup across tiles: it is read with agent-scope (sc1) loads, which are served by
    // L2 -- a plain load may hit a line the CU's L1 / scalar cache fetched while an earlier tile read its neighbours, before this
    // workgroup's own later appends to the same line (seen as 2-40 missed suppressions once a class needs three tiles) -- and staged
    // through LDS 256 boxes at a time so every thread compares against them from there.
    for (int k0 = 0; k0 < nk; k0 += 256) {
      const int kn = nk - k0 < 256 ? nk - k0 : 256;
      if (tid < kn) {
        const float* src = kb + (long)(k0 + tid) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) kst[tid][e] = __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      for (int j = tid; j < n; j += 256) {
        if (dead[j]) continue;
        const float x1 = bx[j][0], y1 = bx[j][1], x2 = bx[j][2], y2 = bx[j][3];
        bool d = false;
        for (int k = 0; k < kn && !d; ++k) d = nms_suppresses(kst[k][0], kst[k][1], kst[k][2], kst[k][3], x1, y1, x2, y2, iou_thr);
        if (d) dead[j] = 1;
      }
      __syncthreads();
    }
    for (int i = 0; i < n; ++i) {
      if (dead[i]) continue;   // uniform: read after the barrier of the previous iteration
      const float x1 = bx[i][0], y1 = bx[i][1], x2 = bx[i][2], y2 = bx[i][3];
      for (int j = i + 1 + tid; j < n; j += 256) {
        if (dead[j]) continue;
        if (nms_suppresses(x1, y1, x2, y2, bx[j][0], bx[j][1], bx[j][2], bx[j][3], iou_thr)) dead[j] = 1;
      }
      __syncthreads();
    }
    for (int i = tid; i < n; i += 256) keep[(long)f * capM + pos[i]] = dead[i] ? 0 : 1;
    if (more && kb) {
      // append the tile's survivors to the kept list (their order does not matter to later tiles: any kept box suppresses)
      for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        const bool k = i < n && !dead[i];
        unsigned long long m = __ballot(k);
        int before = __popcll(m & ((1ull << lane) - 1));
        if (lane == 0) wsum[wv] = __popcll(m);
        __syncthreads();
        int off = nkept_s;
        for (int w = 0; w < wv; ++w) off += wsum[w];
        const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (k && off + before < max_rows) {
          float* d = kb + (long)(off + before) * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) __hip_atomic_store(d + e, bx[i][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (tid == 0) nkept_s += tot;
        __syncthreads();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's appends have left the CU before the barrier releases the readers
    }
    __syncthreads();
  };
  // pull this class' boxes out of the score-sorted list, preserving order; a full tile is processed before the pull goes on
  for (int i0 = 0; i0 < M; i0 += 256) {
    if (base > NMS_FLUSH) {                  // uniform (shared): the next 256 candidates might not fit
      process_tile(base, true);
      if (tid == 0) base = 0;
      __syncthreads();
    }
    int i = i0 + tid;
    int src = i < M ? order[(long)f * capM + i] : 0;
    const int ccls = i < M ? cls[(long)f * capM + src] : 0;
    bool mine = i < M && (agnostic || ccls == cid);
    unsigned long long m = __ballot(mine);
    int before = __popcll(m & ((1ull << lane) - 1));
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wv; ++w) off += wsum[w];
    int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (mine) {
      int p = off + before;
      const float* b = box + ((long)f * capM + src) * 4;
      if (trick) {
        const float off_c = (float)(f16)((float)(f16)(float)ccls * unit);
        bx[p][0] = (float)(f16)(b[0] + off_c); bx[p][1] = (float)(f16)(b[1] + off_c);
        bx[p][2] = (float)(f16)(b[2] + off_c); bx[p][3] = (float)(f16)(b[3] + off_c);
      } else {
        bx[p][0] = b[0]; bx[p][1] = b[1]; bx[p][2] = b[2]; bx[p][3] = b[3];
      }
      pos[p] = i;
      dead[p] = 0;
    }
    __syncthreads();
    if (tid == 0) base += tot;
    __syncthreads();
  }
  process_tile(base, false);
}

__global__ __launch_bounds__(256) void nms_compact_kernel(const float* __restrict__ box, const f16* __restrict__ aux,
                                                         const int* __restrict__ order, const unsigned char* __restrict__ keep,
                                                         const int* __restrict__ count, int capM, f16* __restrict__ out,
                                                         int* __restrict__ out_idx, int* __restrict__ out_count) {
  __shared__ int wsum[4];
  __shared__ int base;
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int M = count[f];
  if (tid == 0) base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < M; i0 += 256) {
    int i = i0 + tid;
    bool k = i < M && keep[(long)f * capM + i];
    unsigned long long m = __ballot(k);
    int before = __popcll(m & ((1ull << lane) - 1));
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wv; ++w) off += wsum[w];
    int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (k) {
      int src = order[(long)f * capM + i];
      long o = ((long)f * capM + off + before) * 7, s = (long)f * capM + src;
      out[o + 0] = (f16)box[s * 4 + 0]; out[o + 1] = (f16)box[s * 4 + 1];
      out[o + 2] = (f16)box[s * 4 + 2]; out[o + 3] = (f16)box[s * 4 + 3];
      out[o + 4] = aux[s * 3 + 0]; out[o + 5] = aux[s * 3 + 1]; out[o + 6] = aux[s * 3 + 2];
      out_idx[(long)f * capM + off + before] = src;
    }
    __syncthreads();
    if (tid == 0) base += tot;
    __syncthreads();
  }
  if (tid == 0) out_count[f] = base;
}

static int nms_pow2(int capM) {
  int P = 1;
  while (P < capM) P <<= 1;
  return P;
}
static int nms_pow2_64(int capM) {
  int P = nms_pow2(capM);
  return P < NMS_CH64 ? NMS_CH64 : P;
}

// max_rows: the largest number of detection rows (proposals) in one frame -- a class of a frame has at most that many boxes; the
// kept-box lists of the tiled greedy pass are reserved whenever the pull loop can flush a tile (more than NMS_FLUSH boxes of one class).
extern "C" long tscd_nms_workspace_bytes(int L, int capM, int NC, int max_rows) {
  long per = (long)capM * (16 + 4 + 6 + 4 + 4 + 1) + 64 + 8 * 256;
  per += 16L * capM + 256;                                      // kept boxes of the class-agnostic coordinate-trick pass (worst case)
  if (capM > NMS_MAXC32) per += 8L * nms_pow2_64(capM) + 256;   // 64-bit sort keys
  else if (capM > NMS_CHUNK) per += 4L * nms_pow2(capM) + 256;  // global 32-bit sort keys of frames with more than one LDS chunk
  if (max_rows > NMS_FLUSH) per += 16L * (NC > 0 ? NC : 1) * max_rows + 256;
  return (long)L * per + 8192;
}

// det [N,ndet] rows of the local frames (pred_result), frame_off [L+1] row offsets.
// mode 0: refined set (needs cls_logit [N,NC], obj_logit [N], reg_delta [N,4]); mode 1: original detections (one candidate per
// row, class-aware); mode 2: the same with class-agnostic suppression (YOLOX demo postprocess).
// coord_trick = n > 0: frames with at most n candidates are suppressed on boxes shifted by fp16(class) * fp16(max + 1) and rounded
// to fp16 (torchvision.ops.batched_nms on the evaluator's fp16 tensors, post_process.py:58-63) instead of exactly per class.
// out [L][capM][7] fp16, out_idx [L][capM] (kept candidate index = row*NC + class in mode 0, row in mode 1),
// out_count [L].  No capacity limits: capM is whatever the caller sized the outputs for (rows x classes in mode 0).
extern "C" int tscd_batched_nms(const void* det, int ndet, const int* frame_off, int L, const void* cls_logit,
                                const void* obj_logit, const void* reg_delta, int NC, float conf_thresh, float iou_thresh,
                                int capM, int max_rows, int mode, int coord_trick, void* workspace, long workspace_bytes, void* out,
                                int* out_idx, int* out_count, void* stream) {
  TSCD_CHECK_ARG(det && frame_off && workspace && out && out_idx && out_count, "batched_nms: null pointer");
  TSCD_CHECK_ARG(mode == 1 || mode == 2 || (cls_logit && obj_logit && reg_delta), "batched_nms: mode 0 needs refined predictions");
  TSCD_CHECK_ARG(capM > 0 && max_rows > 0 && NC > 0, "batched_nms: capM=%d max_rows=%d NC=%d", capM, max_rows, NC);
  TSCD_CHECK_ARG(workspace_bytes >= tscd_nms_workspace_bytes(L, capM, NC, max_rows), "batched_nms: workspace too small");
  if (L == 0) return TSCD_OK;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  auto take = [&](long bytes) { char* r = ws; ws += (bytes + 255) / 256 * 256; return r; };
  Cand c;
  c.box = (float*)take(16L * L * capM);
  c.score = (float*)take(4L * L * capM);
  c.aux = (f16*)take(6L * L * capM);
  c.cls = (int*)take(4L * L * capM);
  c.count = (int*)take(4L * L);
  c.maxc = (float*)take(4L * L);
  int* order = (int*)take(4L * L * capM);
  unsigned char* keep = (unsigned char*)take((long)L * capM);
  (void)hipMemsetAsync(keep, 0, (long)L * capM, st);
  hipLaunchKernelGGL(nms_expand_kernel, dim3(L), dim3(256), 0, st, (const f16*)det, ndet, frame_off, (const f16*)cls_logit,
                     (const f16*)obj_logit, (const f16*)reg_delta, NC, conf_thresh, capM, mode, logf(736.0f / 32.0f), c);
  TSCD_CHECK_LAUNCH("nms_expand");
  if (capM <= NMS_MAXC32) {
    const int P = nms_pow2(capM);
    unsigned int* gkeys = capM > NMS_CHUNK ? (unsigned int*)take(4L * L * P) : nullptr;
    hipLaunchKernelGGL(nms_sort_kernel, dim3(L), dim3(1024), (size_t)(P < NMS_CHUNK ? P : NMS_CHUNK) * 4, st, c.score, c.count, capM,
                       gkeys, P, order);
    TSCD_CHECK_LAUNCH("nms_sort");
  } else {
    static bool attr_set = false;
    if (!attr_set) {       // 64 KiB of dynamic LDS per workgroup
      (void)hipFuncSetAttribute((const void*)nms_sort64_local_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NMS_CH64 * 8);
      (void)hipFuncSetAttribute((const void*)nms_sort64_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NMS_CH64 * 8);
      attr_set = true;
    }
    const int P = nms_pow2_64(capM);
    unsigned long long* gkeys = (unsigned long long*)take(8L * L * P);
    const int nch = P / NMS_CH64;
    hipLaunchKernelGGL(nms_sort64_local_kernel, dim3(nch, L), dim3(1024), (size_t)NMS_CH64 * 8, st, c.score, c.count, capM, gkeys, P);
    for (int size = 2 * NMS_CH64; size <= P; size <<= 1) {
      for (int stride = size >> 1; stride >= NMS_CH64; stride >>= 1)
        hipLaunchKernelGGL(nms_sort64_global_kernel, dim3(P / 2 / 256, L), dim3(256), 0, st, c.count, gkeys, P, size, stride);
      hipLaunchKernelGGL(nms_sort64_merge_kernel, dim3(nch, L), dim3(1024), (size_t)NMS_CH64 * 8, st, c.count, gkeys, P, size);
    }
    hipLaunchKernelGGL(nms_sort64_order_kernel, dim3(cdiv(capM, 256), L), dim3(256), 0, st, c.count, gkeys, P, capM, order);
    TSCD_CHECK_LAUNCH("nms_sort64");
  }
  const int ncg = mode == 2 ? 1 : NC;
  float* kept_box = max_rows > NMS_FLUSH ? (float*)take(16L * L * ncg * max_rows) : nullptr;
  hipLaunchKernelGGL(nms_greedy_kernel, dim3(L, ncg), dim3(256), 0, st, c.box, c.cls, order, c.count, capM, iou_thresh, keep, kept_box,
                     max_rows, c.maxc, coord_trick, 0);
  TSCD_CHECK_LAUNCH("nms_greedy");
  if (coord_trick > 0) {
    // the class-agnostic pass over the shifted boxes of the frames with at most coord_trick candidates
    const int rows = coord_trick < capM ? coord_trick : capM;
    float* kept_all = rows > NMS_FLUSH ? (float*)take(16L * L * rows) : nullptr;
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(L, 1), dim3(256), 0, st, c.box, c.cls, order, c.count, capM, iou_thresh, keep, kept_all,
                       rows, c.maxc, coord_trick, 1);
    TSCD_CHECK_LAUNCH("nms_greedy (coordinate trick)");
  }
  hipLaunchKernelGGL(nms_compact_kernel, dim3(L), dim3(256), 0, st, c.box, c.aux, order, keep, c.count, capM, (f16*)out,
                     out_idx, out_count);
  TSCD_CHECK_LAUNCH("nms_compact");
  return TSCD_OK;
}
