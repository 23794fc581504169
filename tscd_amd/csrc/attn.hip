// CAFM / TaskAligned device pieces (reference: yolox/models/tscd_matching.py):
//   * cos_attn      PositionMHAttention (boxes=None) :31-60 and MHAttention :140-181 -- per head, cosine q.k (no
//                   scale), row softmax, @v.  q^.k^ <= 1, so exp(s - 1) is a fixed-shift softmax: single pass, no
//                   running max.  Query/key rows come in segments (one per frame) so all frames of TaskAligned run in
//                   one launch; proposal-feature tiles are staged through LDS, row sums use wavefront shuffles.
//   * se_fuse       SEModule(2) :264-283 gate (Linear 2->32, ReLU, Linear 32->2, sigmoid) applied to every
//                   (proposal, channel) pair, fused with the "+ time embedding" of with_pos_embed.
//   * add_layernorm tgt = LayerNorm(identity + attn_out)  (forward_post :566-589, :394-458) and plain LayerNorm.
//   * row_normalize x / (||x|| + 1e-6) rows for the Hungarian cost of double_match_embds :912-937; the cost matrix
//                   1 - (cos_reg + cos_cls)/2 is then ONE MFMA GEMM over the concatenated [reg | cls] rows
//                   (conv_igemm.hip, epilogue EPI_COST).
#include "common.h"

struct AttnTile { int row0, nrows, key0, key1; };

template <int D>
__global__ __launch_bounds__(256) void cos_attn_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                      const f16* __restrict__ v, int C, const AttnTile* __restrict__ tiles,
                                                      f16* __restrict__ out) {
  constexpr int R = 16, KT = 64, DS = D + 1, NC = (D + 15) / 16;
  __shared__ float qs[R][DS];
  __shared__ float ks[KT][DS];
  __shared__ float ps[R][KT + 1];
  __shared__ float kinv[KT];
  const int tid = threadIdx.x, h = blockIdx.y;
  const AttnTile t = tiles[blockIdx.x];
  const int r = tid >> 4, c16 = tid & 15;

  // q tile -> LDS (fp32), normalised
  for (int i = tid; i < R * D; i += 256) {
    int rr = i / D, j = i % D;
    qs[rr][j] = rr < t.nrows ? (float)q[(long)(t.row0 + rr) * C + h * D + j] : 0.f;
  }
  __syncthreads();
  {
    // 16 threads per row compute the norm
    float ss = 0.f;
    for (int j = c16; j < D; j += 16) ss += qs[r][j] * qs[r][j];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    float inv = 1.0f / sqrtf(ss);
    __syncthreads();
    for (int j = c16; j < D; j += 16) qs[r][j] *= inv;
  }
  float acc[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) acc[i] = 0.f;
  float lsum = 0.f;
  __syncthreads();

  for (int kb = t.key0; kb < t.key1; kb += KT) {
    const int nk = min(KT, t.key1 - kb);
    for (int i = tid; i < KT * D; i += 256) {
      int kk = i / D, j = i % D;
      ks[kk][j] = kk < nk ? (float)k[(long)(kb + kk) * C + h * D + j] : 0.f;
    }
    __syncthreads();
    {
      // 4 threads per key: inverse norm
      int kk = tid >> 2, part = tid & 3;
      float ss = 0.f;
      for (int j = part; j < D; j += 4) ss += ks[kk][j] * ks[kk][j];
      ss += __shfl_xor(ss, 1, 64);
      ss += __shfl_xor(ss, 2, 64);
      if (part == 0) kinv[kk] = 1.0f / sqrtf(ss);
    }
    __syncthreads();
    // scores: thread (r, kk = c16 + 16 i)
#pragma unroll
    for (int i = 0; i < KT / 16; ++i) {
      int kk = c16 + 16 * i;
      float s = 0.f;
      for (int j = 0; j < D; ++j) s += qs[r][j] * ks[kk][j];
      float pv = kk < nk ? __expf(s * kinv[kk] - 1.0f) : 0.f;
      ps[r][kk] = pv;
      lsum += pv;
    }
    __syncthreads();
    // values -> LDS (reuse ks), then O[r][c] += sum_kk p[r][kk] v[kk][c]
    for (int i = tid; i < KT * D; i += 256) {
      int kk = i / D, j = i % D;
      ks[kk][j] = kk < nk ? (float)v[(long)(kb + kk) * C + h * D + j] : 0.f;
    }
    __syncthreads();
    for (int kk = 0; kk < nk; ++kk) {
      float pv = ps[r][kk];
#pragma unroll
      for (int i = 0; i < NC; ++i)
        if (c16 + 16 * i < D) acc[i] += pv * ks[kk][c16 + 16 * i];
    }
    __syncthreads();
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o, 64);   // the 16 lanes of a row
  if (r < t.nrows) {
    float inv = 1.0f / lsum;
#pragma unroll
    for (int i = 0; i < NC; ++i)
      if (c16 + 16 * i < D) out[(long)(t.row0 + r) * C + h * D + c16 + 16 * i] = (f16)(acc[i] * inv);
  }
}

extern "C" int tscd_cos_attn_f16(const void* q, const void* k, const void* v, int C, int heads, const int* tiles, int ntiles,
                                 void* out, void* stream) {
  TSCD_CHECK_ARG(q && k && v && tiles && out && heads > 0 && C % heads == 0, "cos_attn: bad args");
  if (ntiles == 0) return TSCD_OK;
  const int D = C / heads;
  dim3 grid(ntiles, heads), block(256);
  hipStream_t st = (hipStream_t)stream;
#define CA(DV)                                                                                                          \
  hipLaunchKernelGGL((cos_attn_kernel<DV>), grid, block, 0, st, (const f16*)q, (const f16*)k, (const f16*)v, C,          \
                     (const AttnTile*)tiles, (f16*)out)
  switch (D) {
    case 4: CA(4); break;
    case 8: CA(8); break;
    case 16: CA(16); break;
    case 32: CA(32); break;
    case 40: CA(40); break;
    case 48: CA(48); break;
    case 64: CA(64); break;
    case 80: CA(80); break;
    case 128: CA(128); break;
    case 160: CA(160); break;
    default:
      tscd_set_error("cos_attn: head dim %d not instantiated (4/8/16/32/40/48/64/80/128/160)", D);
      return TSCD_ERR_ARG;
  }
#undef CA
  TSCD_CHECK_LAUNCH("cos_attn");
  return TSCD_OK;
}

// out[n][c] = reg*g0 + edge*g1 + pos[c],  (g0,g1) = sigmoid(W2 relu(W0 [reg, edge])).  w0 [32][2], w2 [2][32] fp32.
__global__ void se_fuse_kernel(const f16* __restrict__ reg, int ldr, const f16* __restrict__ edge, int lde,
                               const float* __restrict__ w0, const float* __restrict__ w2, const float* __restrict__ pos, long n,
                               int C, f16* __restrict__ out) {
  __shared__ float sw0[64], sw2[64];
  if (threadIdx.x < 64) { sw0[threadIdx.x] = w0[threadIdx.x]; sw2[threadIdx.x] = w2[threadIdx.x]; }
  __syncthreads();
  long total = n * C;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long row = idx / C;
    int c = idx % C;
    float a = (float)reg[row * ldr + c], b = (float)edge[row * lde + c];
    float g0 = 0.f, g1 = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      float hdn = fmaxf((float)(f16)(sw0[2 * j] * a + sw0[2 * j + 1] * b), 0.f);   // Linear(2->32) output is fp16
      g0 += sw2[j] * hdn;
      g1 += sw2[32 + j] * hdn;
    }
    g0 = (float)(f16)sigmoidf_((float)(f16)g0);
    g1 = (float)(f16)sigmoidf_((float)(f16)g1);
    float o = (float)(f16)((float)(f16)(a * g0) + (float)(f16)(b * g1));
    out[idx] = (f16)(o + (pos ? pos[c] : 0.f));
  }
}
extern "C" int tscd_se_fuse_f16(const void* reg, int ldr, const void* edge, int lde, const float* w0, const float* w2,
                                const float* pos, long n, int C, void* out, void* stream) {
  TSCD_CHECK_ARG(reg && edge && w0 && w2 && out, "se_fuse: null pointer");
  if (n == 0) return TSCD_OK;
  long total = n * C;
  int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(se_fuse_kernel, dim3(grid > 2048 ? 2048 : grid), dim3(256), 0, (hipStream_t)stream, (const f16*)reg, ldr,
                     (const f16*)edge, lde, w0, w2, pos, n, C, (f16*)out);
  TSCD_CHECK_LAUNCH("se_fuse");
  return TSCD_OK;
}

// out = LayerNorm(a (+ b)) * gamma + beta, one wave per row, eps = 1e-5 (nn.LayerNorm default).
__global__ void add_layernorm_kernel(const f16* __restrict__ a, const f16* __restrict__ b, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, int n, int C, float eps, f16* __restrict__ out) {
  int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= n) return;
  float s = 0.f, s2 = 0.f;
  for (int c = lane; c < C; c += 64) {
    float x = (float)a[(long)row * C + c];
    if (b) x = (float)(f16)(x + (float)b[(long)row * C + c]);
    s += x;
    s2 += x * x;
  }
  s = wave_sum(s);
  s2 = wave_sum(s2);
  float mean = s / C, var = fmaxf(s2 / C - mean * mean, 0.f), inv = 1.0f / sqrtf(var + eps);
  for (int c = lane; c < C; c += 64) {
    float x = (float)a[(long)row * C + c];
    if (b) x = (float)(f16)(x + (float)b[(long)row * C + c]);
    out[(long)row * C + c] = (f16)((x - mean) * inv * gamma[c] + beta[c]);
  }
}
extern "C" int tscd_add_layernorm_f16(const void* a, const void* b, const float* gamma, const float* beta, int n, int C,
                                      float eps, void* out, void* stream) {
  TSCD_CHECK_ARG(a && gamma && beta && out && C > 0, "add_layernorm: bad args");
  if (n == 0) return TSCD_OK;
  hipLaunchKernelGGL(add_layernorm_kernel, dim3((n * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const f16*)a,
                     (const f16*)b, gamma, beta, n, C, eps, (f16*)out);
  TSCD_CHECK_LAUNCH("add_layernorm");
  return TSCD_OK;
}

// out[row][off : off+C] = x[row] / (||x[row]|| + eps); one wave per row; rows >= n (padding) are zero-filled.
__global__ void row_normalize_kernel(const f16* __restrict__ x, int ldx, int n, int n_pad, int C, float eps, f16* __restrict__ out,
                                     int ldo) {
  int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= n_pad) return;
  if (row >= n) {
    for (int c = lane; c < C; c += 64) out[(long)row * ldo + c] = (f16)0.f;
    return;
  }
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) {
    float v = (float)x[(long)row * ldx + c];
    ss += v * v;
  }
  ss = wave_sum(ss);
  float inv = 1.0f / (sqrtf(ss) + eps);
  for (int c = lane; c < C; c += 64) out[(long)row * ldo + c] = (f16)((float)x[(long)row * ldx + c] * inv);
}
extern "C" int tscd_row_normalize_f16(const void* x, int ldx, int n, int n_pad, int C, float eps, void* out, int ldo,
                                      void* stream) {
  TSCD_CHECK_ARG(x && out && n_pad >= n && C > 0, "row_normalize: bad args");
  if (n_pad == 0) return TSCD_OK;
  hipLaunchKernelGGL(row_normalize_kernel, dim3((n_pad * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const f16*)x,
                     ldx, n, n_pad, C, eps, (f16*)out, ldo);
  TSCD_CHECK_LAUNCH("row_normalize");
  return TSCD_OK;
}
