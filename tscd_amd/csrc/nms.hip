// Eval tail on the device (reference: decode_reg_preds5 tscd_head.py:914-949, post_process.postprocess
// yolox/models/post_process.py:9-85, torchvision.ops.batched_nms).
//
//   nms_expand   per local frame: refined boxes = base xyxy (+) deltas, obj = sig(matched obj), class scores =
//                sig(refined cls); every (proposal, class) with score >= thr and obj*score >= thr becomes a
//                candidate (row-major order, as torch.where).  Mode 1 builds the "original detections" set instead
//                (one candidate per proposal: class = argmax id, score = obj * cls_conf).
//   nms_sort     one workgroup per frame: bitonic sort by (score desc, candidate index asc) -- torchvision visits boxes in
//                decreasing score order; <= 16384 candidates entirely in LDS, up to 65536 with global-memory cross-chunk steps.
//   nms_greedy   one workgroup per (frame, class): class-aware NMS == independent greedy NMS per class (what the
//                coordinate-offset trick of batched_nms implements); <= 2048 boxes of the class are pulled from the
//                sorted list into LDS and suppressed greedily with IoU > thr (fp32 arithmetic).
//   nms_compact  ordered compaction of the survivors -> [k,7] rows (x1,y1,x2,y2,obj,cls_score,cls_id) in decreasing
//                score order, plus the kept candidate indices (INT output) and the count.
// torchvision is absent from the reference tree, so NMS parity is pinned by oracle/nms_ref.py's restatement only.
#include "common.h"

#define NMS_MAXC 65536    // candidates per frame (16-bit index field of the sort key)
#define NMS_MAXPC 2048    // boxes per (frame, class)

// ---- candidate record arrays, all [L][capM] -------------------------------------------------------------
struct Cand {
  float* box;      // [L][capM][4]
  float* score;    // nms score = obj * cls_score (fp16-rounded product, as fp32)
  f16* aux;        // [L][capM][3]: obj, cls_score, cls_id  (fp16 values of the output row)
  int* cls;        // class id
  int* count;      // [L]
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
        c.aux[o * 3 + 0] = a_obj; c.aux[o * 3 + 1] = a_cs; c.aux[o * 3 + 2] = (f16)(float)cid;
        c.cls[o] = mode == 2 ? 0 : cid;            // mode 2: class-agnostic suppression (utils/boxes.py:59-64), the row keeps its class
      }
    }
    __syncthreads();
    if (tid == 0) base += tot;
    __syncthreads();
  }
  if (tid == 0) c.count[f] = base < capM ? base : capM;
}

// Bitonic sort of (score desc, index asc) keys.  Scores are fp16-representable and non-negative, so their fp16 bit pattern
// orders them: key = fp16 bits << 16 | (0xFFFF - index), larger key first.  Up to 16384 surviving candidates of a frame are
// sorted entirely in LDS (64 KiB).  Frames with more (the ImageNet-VID exp has no upper proposal clamp: 2000 proposals x 30
// classes = 60000 candidates) keep their keys in a global workspace: compare-exchange steps whose stride reaches across
// 16384-key chunks run on global memory (L2-resident, <= 256 KiB per frame), every run of smaller strides is done chunk by
// chunk in LDS.  Same network, same keys => the same order as the single-chunk sort.  Limit: 65536 candidates per frame (the
// index field of the key).
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

// Greedy NMS for one (frame, class).  keep[f][sorted position] = 1 for survivors.
__global__ __launch_bounds__(256) void nms_greedy_kernel(const float* __restrict__ box, const int* __restrict__ cls,
                                                        const int* __restrict__ order, const int* __restrict__ count, int capM,
                                                        float iou_thr, unsigned char* __restrict__ keep, int* __restrict__ overflow) {
  __shared__ float bx[NMS_MAXPC][4];
  __shared__ int pos[NMS_MAXPC];
  __shared__ unsigned char dead[NMS_MAXPC];
  __shared__ int wsum[4];
  __shared__ int base;
  const int f = blockIdx.x, cid = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int M = count[f];
  if (tid == 0) base = 0;
  __syncthreads();
  // pull this class' boxes out of the score-sorted list, preserving order
  for (int i0 = 0; i0 < M; i0 += 256) {
    int i = i0 + tid;
    int src = i < M ? order[(long)f * capM + i] : 0;
    bool mine = i < M && cls[(long)f * capM + src] == cid;
    unsigned long long m = __ballot(mine);
    int before = __popcll(m & ((1ull << lane) - 1));
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wv; ++w) off += wsum[w];
    int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (mine && off + before < NMS_MAXPC) {
      int p = off + before;
      const float* b = box + ((long)f * capM + src) * 4;
      bx[p][0] = b[0]; bx[p][1] = b[1]; bx[p][2] = b[2]; bx[p][3] = b[3];
      pos[p] = i;
      dead[p] = 0;
    }
    __syncthreads();
    if (tid == 0) base += tot;
    __syncthreads();
  }
  int n = base;
  if (n > NMS_MAXPC) {
    if (tid == 0) atomicAdd(overflow, 1);
    n = NMS_MAXPC;
  }
  for (int i = 0; i < n; ++i) {
    if (dead[i]) continue;   // uniform: read after the barrier of the previous iteration
    const float x1 = bx[i][0], y1 = bx[i][1], x2 = bx[i][2], y2 = bx[i][3];
    const float ai = (x2 - x1) * (y2 - y1);
    for (int j = i + 1 + tid; j < n; j += 256) {
      if (dead[j]) continue;
      float iw = fmaxf(fminf(x2, bx[j][2]) - fmaxf(x1, bx[j][0]), 0.f);
      float ih = fmaxf(fminf(y2, bx[j][3]) - fmaxf(y1, bx[j][1]), 0.f);
      float inter = iw * ih;
      float aj = (bx[j][2] - bx[j][0]) * (bx[j][3] - bx[j][1]);
      float iou = inter / (ai + aj - inter);
      if (iou > iou_thr) dead[j] = 1;
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += 256) keep[(long)f * capM + pos[i]] = dead[i] ? 0 : 1;
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

extern "C" long tscd_nms_workspace_bytes(int L, int capM) {
  long per = (long)capM * (16 + 4 + 6 + 4 + 4 + 1) + 64 + 7 * 256;
  if (capM > NMS_CHUNK) per += 4L * nms_pow2(capM);        // global sort keys of frames with more than one LDS chunk
  return (long)L * per + 4096;
}

// det [N,ndet] rows of the local frames (pred_result), frame_off [L+1] row offsets.
// mode 0: refined set (needs cls_logit [N,NC], obj_logit [N], reg_delta [N,4]); mode 1: original detections (one candidate per
// row, class-aware); mode 2: the same with class-agnostic suppression (YOLOX demo postprocess).
// out [L][capM][7] fp16, out_idx [L][capM] (kept candidate index = row*NC + class in mode 0, row in mode 1),
// out_count [L].  Returns TSCD_ERR_ARG if a frame exceeds the LDS sort capacity.
extern "C" int tscd_batched_nms(const void* det, int ndet, const int* frame_off, int L, const void* cls_logit,
                                const void* obj_logit, const void* reg_delta, int NC, float conf_thresh, float iou_thresh,
                                int capM, int mode, void* workspace, long workspace_bytes, void* out, int* out_idx,
                                int* out_count, int* overflow_flag, void* stream) {
  TSCD_CHECK_ARG(det && frame_off && workspace && out && out_idx && out_count && overflow_flag, "batched_nms: null pointer");
  TSCD_CHECK_ARG(mode == 1 || mode == 2 || (cls_logit && obj_logit && reg_delta), "batched_nms: mode 0 needs refined predictions");
  TSCD_CHECK_ARG(capM > 0 && capM <= NMS_MAXC, "batched_nms: capM=%d exceeds the %d candidates per frame the sort key can index", capM, NMS_MAXC);
  TSCD_CHECK_ARG(workspace_bytes >= tscd_nms_workspace_bytes(L, capM), "batched_nms: workspace too small");
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
  int* order = (int*)take(4L * L * capM);
  unsigned char* keep = (unsigned char*)take((long)L * capM);
  (void)hipMemsetAsync(keep, 0, (long)L * capM, st);
  hipLaunchKernelGGL(nms_expand_kernel, dim3(L), dim3(256), 0, st, (const f16*)det, ndet, frame_off, (const f16*)cls_logit,
                     (const f16*)obj_logit, (const f16*)reg_delta, NC, conf_thresh, capM, mode, logf(736.0f / 32.0f), c);
  TSCD_CHECK_LAUNCH("nms_expand");
  const int P = nms_pow2(capM);
  unsigned int* gkeys = capM > NMS_CHUNK ? (unsigned int*)take(4L * L * P) : nullptr;
  hipLaunchKernelGGL(nms_sort_kernel, dim3(L), dim3(1024), (size_t)(P < NMS_CHUNK ? P : NMS_CHUNK) * 4, st, c.score, c.count, capM,
                     gkeys, P, order);
  TSCD_CHECK_LAUNCH("nms_sort");
  hipLaunchKernelGGL(nms_greedy_kernel, dim3(L, mode == 2 ? 1 : NC), dim3(256), 0, st, c.box, c.cls, order, c.count, capM, iou_thresh, keep,
                     overflow_flag);
  TSCD_CHECK_LAUNCH("nms_greedy");
  hipLaunchKernelGGL(nms_compact_kernel, dim3(L), dim3(256), 0, st, c.box, c.aux, order, keep, c.count, capM, (f16*)out,
                     out_idx, out_count);
  TSCD_CHECK_LAUNCH("nms_compact");
  return TSCD_OK;
}
