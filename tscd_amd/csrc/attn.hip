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

// ---- row addressing of the CAFM frame chain (yolox/models/tscd_matching.py::_CafmChain; reference tscd_matching.py:761-888: the rows of a
// frame are re-ordered by its assignment before they meet the next frame's).  The permuted reads / writes ride in the kernels that
// consume / produce the rows instead of separate gather launches.  Declared for callers in include/tscd_hip.h.
struct RowMap {
  const long* idx;       // null: identity
  int seg_rows;          // rows per segment; row r -> segment s = r / seg_rows, local l = r % seg_rows
  int off0, off_stride;  // mapped row = s * seg_rows + idx[off0 + s * off_stride + l]
};
__device__ __forceinline__ long map_row(const RowMap& m, long r) {
  if (!m.idx) return r;
  const long s = r / m.seg_rows, l = r - s * m.seg_rows;
  return s * m.seg_rows + m.idx[m.off0 + s * m.off_stride + l];
}
struct SeFuseEx {
  RowMap edge_map;       // edge row of output row r; backward: also the row d edge of row r is written to
  int pos_seg_rows;      // 0: one pos / d pos vector; else row r takes pos + (r / pos_seg_rows) * pos_stride
  int pos_stride;
  // backward only: d reg[r] = fp16(d reg + sum_k add_reg[k][r] (r < add_reg_rows[k])), same for d edge (addend rows in OUTPUT order, pitch C)
  const f16* add_reg[3];
  int add_reg_rows[3];
  const f16* add_edge[2];
  int add_edge_rows[2];
};
struct LnEx {
  RowMap a_map;          // row of `a` read for output row r
  const f16* dy_add;     // backward: dy row r = fp16(dy[r] + dy_add[r])
  f16* dx_mapped;        // backward: a copy of dx row r at row a_map(r) of this buffer
};

// out[n][c] = reg*g0 + edge*g1 + pos[c],  (g0,g1) = sigmoid(W2 relu(W0 [reg, edge])).  w0 [32][2], w2 [2][32] fp32.
// (ONE instantiation serves the plain and the row-mapped entry points -- a zeroed `ex` is the identity: the per-op path and the fused
// chain then run the same machine code and stay bit-identical, tests/test_gpu_head_stages.py; the same holds for the three kernels below.)
__global__ __launch_bounds__(256) void se_fuse_kernel(const f16* __restrict__ reg, int ldr, const f16* __restrict__ edge, int lde,
                               const float* __restrict__ w0, const float* __restrict__ w2, const float* __restrict__ pos, long n,
                               int C, f16* __restrict__ out, SeFuseEx ex) {
  __shared__ float sw0[64], sw2[64];
  if (threadIdx.x < 64) { sw0[threadIdx.x] = w0[threadIdx.x]; sw2[threadIdx.x] = w2[threadIdx.x]; }
  __syncthreads();
  long total = n * C;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long row = idx / C;
    int c = idx % C;
    const long erow = map_row(ex.edge_map, row);
    float a = (float)reg[row * ldr + c], b = (float)edge[erow * lde + c];
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
    const float* pv = pos;
    if (pos && ex.pos_seg_rows > 0) pv = pos + (row / ex.pos_seg_rows) * ex.pos_stride;
    out[idx] = (f16)(o + (pv ? pv[c] : 0.f));
  }
}
static int se_fuse_launch(const void* reg, int ldr, const void* edge, int lde, const float* w0, const float* w2, const float* pos, long n,
                          int C, void* out, const SeFuseEx* ex, void* stream) {
  TSCD_CHECK_ARG(reg && edge && w0 && w2 && out, "se_fuse: null pointer");
  if (n == 0) return TSCD_OK;
  long total = n * C;
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  if (ex) {
    TSCD_CHECK_ARG(!ex->edge_map.idx || ex->edge_map.seg_rows > 0, "se_fuse: row map without a segment length");
    hipLaunchKernelGGL(se_fuse_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f16*)reg, ldr, (const f16*)edge, lde, w0,
                       w2, pos, n, C, (f16*)out, *ex);
  } else {
    hipLaunchKernelGGL(se_fuse_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f16*)reg, ldr, (const f16*)edge, lde, w0,
                       w2, pos, n, C, (f16*)out, SeFuseEx{});
  }
  TSCD_CHECK_LAUNCH("se_fuse");
  return TSCD_OK;
}
extern "C" int tscd_se_fuse_f16(const void* reg, int ldr, const void* edge, int lde, const float* w0, const float* w2,
                                const float* pos, long n, int C, void* out, void* stream) {
  return se_fuse_launch(reg, ldr, edge, lde, w0, w2, pos, n, C, out, nullptr, stream);
}
// ... with the chain's row addressing (struct TscdSeFuseEx of include/tscd_hip.h: permuted edge rows, one pos vector per row segment)
extern "C" int tscd_se_fuse_ex_f16(const void* reg, int ldr, const void* edge, int lde, const float* w0, const float* w2,
                                   const float* pos, long n, int C, void* out, const void* ex, void* stream) {
  TSCD_CHECK_ARG(ex, "se_fuse_ex: null extras");
  return se_fuse_launch(reg, ldr, edge, lde, w0, w2, pos, n, C, out, (const SeFuseEx*)ex, stream);
}

// out = LayerNorm(a (+ b)) * gamma + beta, one wave per row, eps = 1e-5 (nn.LayerNorm default).
__global__ void add_layernorm_kernel(const f16* __restrict__ a, const f16* __restrict__ b, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, int n, int C, float eps, f16* __restrict__ out, LnEx ex) {
  int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= n) return;
  const long arow = map_row(ex.a_map, row);
  float s = 0.f, s2 = 0.f;
  for (int c = lane; c < C; c += 64) {
    float x = (float)a[arow * C + c];
    if (b) x = (float)(f16)(x + (float)b[(long)row * C + c]);
    s += x;
    s2 += x * x;
  }
  s = wave_sum(s);
  s2 = wave_sum(s2);
  float mean = s / C, var = fmaxf(s2 / C - mean * mean, 0.f), inv = 1.0f / sqrtf(var + eps);
  for (int c = lane; c < C; c += 64) {
    float x = (float)a[arow * C + c];
    if (b) x = (float)(f16)(x + (float)b[(long)row * C + c]);
    out[(long)row * C + c] = (f16)((x - mean) * inv * gamma[c] + beta[c]);
  }
}
static int add_layernorm_launch(const void* a, const void* b, const float* gamma, const float* beta, int n, int C, float eps, void* out,
                                const LnEx* ex, void* stream) {
  TSCD_CHECK_ARG(a && gamma && beta && out && C > 0, "add_layernorm: bad args");
  if (n == 0) return TSCD_OK;
  if (ex) {
    TSCD_CHECK_ARG(!ex->a_map.idx || ex->a_map.seg_rows > 0, "add_layernorm: row map without a segment length");
    hipLaunchKernelGGL(add_layernorm_kernel, dim3((n * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const f16*)a,
                       (const f16*)b, gamma, beta, n, C, eps, (f16*)out, *ex);
  } else {
    hipLaunchKernelGGL(add_layernorm_kernel, dim3((n * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const f16*)a,
                       (const f16*)b, gamma, beta, n, C, eps, (f16*)out, LnEx{});
  }
  TSCD_CHECK_LAUNCH("add_layernorm");
  return TSCD_OK;
}
extern "C" int tscd_add_layernorm_f16(const void* a, const void* b, const float* gamma, const float* beta, int n, int C,
                                      float eps, void* out, void* stream) {
  return add_layernorm_launch(a, b, gamma, beta, n, C, eps, out, nullptr, stream);
}
// ... reading the rows of `a` through a row map (struct TscdLnEx of include/tscd_hip.h)
extern "C" int tscd_add_layernorm_ex_f16(const void* a, const void* b, const float* gamma, const float* beta, int n, int C,
                                         float eps, void* out, const void* ex, void* stream) {
  TSCD_CHECK_ARG(ex, "add_layernorm_ex: null extras");
  return add_layernorm_launch(a, b, gamma, beta, n, C, eps, out, (const LnEx*)ex, stream);
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

// ---------------------------------------------------------------------------------------------------------------------------
// Every Hungarian cost matrix of a clip in ONE launch (double_match_embds, tscd_matching.py:912-937): matrix p = 1 - rows[r0_p ..]
// . rows[c0_p ..]^T / 2 over the normalised [reg | cls] rows (K = 2C), NaN -> 0, fp16.  One wave per 32 x 32 output tile; the same
// v_mfma_f32_32x32x16_f16 chain over ascending k as the conv kernel's EPI_COST launch it replaces (one launch per matrix: 4 x 23 us of 32
// workgroups each on the step's critical path) -- bit-identical costs.
// ---------------------------------------------------------------------------------------------------------------------------
#define COST_MAX_PAIRS 16
struct CostPair { int r0, nr, c0, nc, tile0, tn; long out_off; };
struct CostPairs { CostPair p[COST_MAX_PAIRS]; int n; };
__global__ __launch_bounds__(256) void match_cost_pairs_kernel(const f16* __restrict__ rows, int ld, int K, CostPairs t, f16* __restrict__ out,
                                                               int ntiles) {
  const int lane = threadIdx.x & 63;
  const int tile = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  if (tile >= ntiles) return;
  int pi = 0;
  for (int i = 1; i < t.n; ++i) pi = tile >= t.p[i].tile0 ? i : pi;          // wave-uniform
  const CostPair q = t.p[pi];
  const int lt = tile - q.tile0, mt = lt / q.tn, nt = lt - mt * q.tn;
  // fragment rows: lane l holds row (l & 31), k = 16 ks + 8 (l >> 5) .. + 7.  Rows past the matrix read the last valid row (masked at the store).
  const int ra = min(mt * 32 + (lane & 31), q.nr - 1), rb = min(nt * 32 + (lane & 31), q.nc - 1);
  const f16* pa = rows + (long)(q.r0 + ra) * ld + (lane >> 5) * 8;
  const f16* pb = rows + (long)(q.c0 + rb) * ld + (lane >> 5) * 8;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll 8
  for (int k = 0; k < K; k += 16)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(pa + k), *(const f16x8*)(pb + k), acc, 0, 0, 0);
  const int col = nt * 32 + (lane & 31);
  if (col < q.nc) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      if (row < q.nr) {
        const float c = 1.0f - 0.5f * (float)(f16)acc[e];
        out[q.out_off + (long)row * q.nc + col] = (c != c) ? (f16)0.f : (f16)c;
      }
    }
  }
}
// pairs: npairs x (r0, nr, c0, nc) row ranges of `rows` [*, ld] (fp16, K <= ld, K % 16 == 0, 16-byte aligned); matrix p is written row-major
// [nr][nc] at out + the sum of the earlier matrices' sizes.
extern "C" int tscd_match_cost_pairs_f16(const void* rows, int ld, int K, const int* pairs, int npairs, void* out, void* stream) {
  TSCD_CHECK_ARG(rows && pairs && out && npairs > 0 && npairs <= COST_MAX_PAIRS, "match_cost_pairs: 1..%d pairs", COST_MAX_PAIRS);
  TSCD_CHECK_ARG(K > 0 && K % 16 == 0 && ld >= K && ld % 8 == 0 && ((uintptr_t)rows & 15) == 0, "match_cost_pairs: K=%d ld=%d", K, ld);
  CostPairs t;
  t.n = npairs;
  int tiles = 0;
  long off = 0;
  for (int i = 0; i < npairs; ++i) {
    CostPair& q = t.p[i];
    q.r0 = pairs[4 * i]; q.nr = pairs[4 * i + 1]; q.c0 = pairs[4 * i + 2]; q.nc = pairs[4 * i + 3];
    TSCD_CHECK_ARG(q.r0 >= 0 && q.c0 >= 0 && q.nr >= 0 && q.nc >= 0, "match_cost_pairs: bad range in pair %d", i);
    q.tn = (q.nc + 31) / 32;
    q.tile0 = tiles;
    q.out_off = off;
    tiles += ((q.nr + 31) / 32) * q.tn;
    off += (long)q.nr * q.nc;
  }
  if (tiles == 0) return TSCD_OK;
  hipLaunchKernelGGL(match_cost_pairs_kernel, dim3((tiles + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const f16*)rows, ld, K, t, (f16*)out,
                     tiles);
  TSCD_CHECK_LAUNCH("match_cost_pairs");
  return TSCD_OK;
}


// ====================================================================================================================
// Backward kernels (autograd of the modules above under tscd_trainer.py:171-176).
// ====================================================================================================================

// cos_attn backward.  O = softmax(q^ k^T) v per head.  One workgroup = 16 query rows x one head, two passes over the
// segment's keys (64 per tile, staged through LDS):
//   pass 1: l = sum_k exp(s-1), delta = sum_k P (dO.v)
//   pass 2: dS = P (dO.v - delta);  dq^ += dS k^;  dk^ += dS^T q^;  dv += P^T dO   (dk^, dv: fp32 atomics over query tiles)
// then the q normalisation backward in place (dq = (dq^ - q^ (q^.dq^)) / ||q||).  k's normalisation backward runs in
// cos_attn_bwd_finish once all tiles have added their share.
template <int D>
__global__ __launch_bounds__(256) void cos_attn_bwd_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                          const f16* __restrict__ v, const f16* __restrict__ dO, int C,
                                                          const AttnTile* __restrict__ tiles, f16* __restrict__ dq,
                                                          float* __restrict__ dkn, float* __restrict__ dv) {
  constexpr int R = 16, KT = 64, DS = D + 1, NC = (D + 15) / 16, NJ = (D + 31) / 32;
  __shared__ float qs[R][DS], gs[R][DS], ks[KT][DS], vs[KT][DS];
  __shared__ float ps[R][KT + 1], dss[R][KT + 1];
  __shared__ float qinv[R], lrow[R], drow[R];
  const int tid = threadIdx.x, h = blockIdx.y;
  const AttnTile t = tiles[blockIdx.x];
  const int r = tid >> 4, c16 = tid & 15;
  for (int i = tid; i < R * D; i += 256) {
    int rr = i / D, j = i % D;
    bool ok = rr < t.nrows;
    qs[rr][j] = ok ? (float)q[(long)(t.row0 + rr) * C + h * D + j] : 0.f;
    gs[rr][j] = ok ? (float)dO[(long)(t.row0 + rr) * C + h * D + j] : 0.f;
  }
  __syncthreads();
  {
    float ss = 0.f;
    for (int j = c16; j < D; j += 16) ss += qs[r][j] * qs[r][j];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    float inv = ss > 0.f ? 1.0f / sqrtf(ss) : 0.f;
    if (c16 == 0) qinv[r] = inv;
    __syncthreads();
    for (int j = c16; j < D; j += 16) qs[r][j] *= inv;
  }
  float accq[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) accq[i] = 0.f;
  float lsum = 0.f, dsum = 0.f;
  __syncthreads();
  for (int pass = 0; pass < 2; ++pass) {
    for (int kb = t.key0; kb < t.key1; kb += KT) {
      const int nk = min(KT, t.key1 - kb);
      for (int i = tid; i < KT * D; i += 256) {
        int kk = i / D, j = i % D;
        ks[kk][j] = kk < nk ? (float)k[(long)(kb + kk) * C + h * D + j] : 0.f;
        vs[kk][j] = kk < nk ? (float)v[(long)(kb + kk) * C + h * D + j] : 0.f;
      }
      __syncthreads();
      {
        int kk = tid >> 2, part = tid & 3;
        float ss = 0.f;
        for (int j = part; j < D; j += 4) ss += ks[kk][j] * ks[kk][j];
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        float inv = ss > 0.f ? 1.0f / sqrtf(ss) : 0.f;
        __syncthreads();
        for (int j = part; j < D; j += 4) ks[kk][j] *= inv;   // k^ in place
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < KT / 16; ++i) {
        int kk = c16 + 16 * i;
        float s = 0.f, g = 0.f;
        for (int j = 0; j < D; ++j) { s += qs[r][j] * ks[kk][j]; g += gs[r][j] * vs[kk][j]; }
        float pv = kk < nk ? __expf(s - 1.0f) : 0.f;
        if (pass == 0) {
          lsum += pv;
          dsum += pv * g;
        } else {
          float pn = pv * lrow[r];
          ps[r][kk] = pn;
          dss[r][kk] = pn * (g - drow[r]);
        }
      }
      if (pass == 1) {
        __syncthreads();
        for (int kk = 0; kk < nk; ++kk) {
          float d = dss[r][kk];
#pragma unroll
          for (int i = 0; i < NC; ++i)
            if (c16 + 16 * i < D) accq[i] += d * ks[kk][c16 + 16 * i];
        }
        // dk^[kk][c] += sum_r dS[r][kk] q^[r][c];  dv[kk][c] += sum_r P[r][kk] dO[r][c]   (2 keys x 32 channels per wave)
        for (int i = 0; i < KT / 8; ++i) {
          int kk = (tid >> 5) + 8 * i;
          if (kk >= nk) continue;
#pragma unroll
          for (int jn = 0; jn < NJ; ++jn) {
            int c = (tid & 31) + 32 * jn;
            if (c >= D) continue;
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) { a += dss[rr][kk] * qs[rr][c]; b += ps[rr][kk] * gs[rr][c]; }
            long o = (long)(kb + kk) * C + h * D + c;
            atomicAdd(dkn + o, a);
            atomicAdd(dv + o, b);
          }
        }
      }
      __syncthreads();
    }
    if (pass == 0) {
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { lsum += __shfl_xor(lsum, o, 64); dsum += __shfl_xor(dsum, o, 64); }
      if (c16 == 0) {
        float il = lsum > 0.f ? 1.0f / lsum : 0.f;
        lrow[r] = il;
        drow[r] = dsum * il;
      }
      __syncthreads();
    }
  }
  // q normalisation backward: dq = (dq^ - q^ (q^.dq^)) * (1/||q||)
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i)
    if (c16 + 16 * i < D) dot += accq[i] * qs[r][c16 + 16 * i];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
  if (r < t.nrows) {
#pragma unroll
    for (int i = 0; i < NC; ++i)
      if (c16 + 16 * i < D)
        dq[(long)(t.row0 + r) * C + h * D + c16 + 16 * i] = (f16)((accq[i] - qs[r][c16 + 16 * i] * dot) * qinv[r]);
  }
}

// one wave per (key row, head): dk = (dk^ - k^ (k^.dk^)) / ||k||, dv passes through; fp32 accumulators -> fp16 grads
__global__ void cos_attn_bwd_finish_kernel(const f16* __restrict__ k, const float* __restrict__ dkn, const float* __restrict__ dvacc,
                                           int n, int C, int heads, f16* __restrict__ dk, f16* __restrict__ dv) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  int row = wave / heads, h = wave % heads;
  if (row >= n) return;
  const int D = C / heads;
  float ss = 0.f, dot = 0.f;
  for (int j = lane; j < D; j += 64) {
    float x = (float)k[(long)row * C + h * D + j];
    ss += x * x;
    dot += x * dkn[(long)row * C + h * D + j];
  }
  ss = wave_sum(ss); dot = wave_sum(dot);
  float inv = ss > 0.f ? 1.0f / sqrtf(ss) : 0.f;
  for (int j = lane; j < D; j += 64) {
    long o = (long)row * C + h * D + j;
    float x = (float)k[o];
    dk[o] = (f16)(dkn[o] * inv - x * dot * inv * inv * inv);
    dv[o] = (f16)dvacc[o];
  }
}

// q/dO/dq [n,C], k/v/dk/dv [nk,C]; workspace: 2 * nk * C floats (dk^ and dv accumulators, zeroed here)
extern "C" int tscd_cos_attn_bwd_f16(const void* q, const void* k, const void* v, const void* dO, int n, int nk, int C, int heads,
                                     const int* tiles, int ntiles, void* workspace, long workspace_bytes, void* dq, void* dk,
                                     void* dv, void* stream) {
  TSCD_CHECK_ARG(q && k && v && dO && tiles && workspace && dq && dk && dv && heads > 0 && C % heads == 0, "cos_attn_bwd: bad args");
  TSCD_CHECK_ARG(workspace_bytes >= 8L * nk * C, "cos_attn_bwd: workspace too small");
  if (nk == 0) return TSCD_OK;
  hipStream_t st = (hipStream_t)stream;
  float* dkn = (float*)workspace;
  float* dva = dkn + (long)nk * C;
  (void)hipMemsetAsync(workspace, 0, 8L * nk * C, st);
  const int D = C / heads;
  dim3 grid(ntiles, heads), block(256);
#define CAB(DV)                                                                                                           \
  hipLaunchKernelGGL((cos_attn_bwd_kernel<DV>), grid, block, 0, st, (const f16*)q, (const f16*)k, (const f16*)v, (const f16*)dO, C, \
                     (const AttnTile*)tiles, (f16*)dq, dkn, dva)
  if (ntiles > 0) switch (D) {
    case 4: CAB(4); break;
    case 8: CAB(8); break;
    case 16: CAB(16); break;
    case 32: CAB(32); break;
    case 40: CAB(40); break;
    case 48: CAB(48); break;
    case 64: CAB(64); break;
    case 80: CAB(80); break;
    case 128: CAB(128); break;
    case 160: CAB(160); break;
    default:
      tscd_set_error("cos_attn_bwd: head dim %d not instantiated", D);
      return TSCD_ERR_ARG;
  }
#undef CAB
  long waves = (long)nk * heads;
  hipLaunchKernelGGL(cos_attn_bwd_finish_kernel, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, st, (const f16*)k, dkn, dva,
                     nk, C, heads, (f16*)dk, (f16*)dv);
  TSCD_CHECK_LAUNCH("cos_attn_bwd");
  return TSCD_OK;
}

// dk / dv (fp16) from the fp32 accumulators a cos_attn backward kernel left in `workspace` ([dk^ | dv], nk*C floats each)
extern "C" int tscd_cos_attn_bwd_finish_f16(const void* k, const void* workspace, int nk, int C, int heads, void* dk, void* dv,
                                            void* stream) {
  TSCD_CHECK_ARG(k && workspace && dk && dv && heads > 0 && C % heads == 0, "cos_attn_bwd_finish: bad args");
  if (nk == 0) return TSCD_OK;
  const float* dkn = (const float*)workspace;
  long waves = (long)nk * heads;
  hipLaunchKernelGGL(cos_attn_bwd_finish_kernel, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const f16*)k, dkn, dkn + (long)nk * C, nk, C, heads, (f16*)dk, (f16*)dv);
  TSCD_CHECK_LAUNCH("cos_attn_bwd_finish");
  return TSCD_OK;
}

// se_fuse backward: d reg, d edge (fp16), d pos [C], d w0 [32][2], d w2 [2][32] (fp32 atomics; caller zeroes them).
__global__ __launch_bounds__(256) void se_fuse_bwd_kernel(const f16* __restrict__ reg, int ldr, const f16* __restrict__ edge, int lde,
                                   const float* __restrict__ w0, const float* __restrict__ w2, const f16* __restrict__ dout, long n,
                                   int C, f16* dreg, f16* dedge, float* __restrict__ dpos,
                                   float* __restrict__ dw0, float* __restrict__ dw2, SeFuseEx ex) {
  __shared__ float sw0[64], sw2[64], acc[128];
  if (threadIdx.x < 64) { sw0[threadIdx.x] = w0[threadIdx.x]; sw2[threadIdx.x] = w2[threadIdx.x]; }
  if (threadIdx.x < 128) acc[threadIdx.x] = 0.f;
  __syncthreads();
  // per-thread partial sums: v[j] = dw2[0][j], v[32+j] = dw2[1][j], v[64+2j] = dw0[j][0], v[64+2j+1] = dw0[j][1]
  float v[128];
#pragma unroll
  for (int j = 0; j < 128; ++j) v[j] = 0.f;
  long total = n * C;
  // when the grid stride is a multiple of C a thread stays on one column: d pos accumulates in a register and costs one atomic per
  // thread (and row segment) instead of one per element (n x C atomics on C addresses were most of this kernel's 18 us)
  const bool fixed_col = ((long)gridDim.x * blockDim.x) % C == 0;
  float gsum = 0.f;
  int gcol = -1;
  long gseg = 0;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long row = idx / C;
    int c = idx % C;
    const long erow = map_row(ex.edge_map, row);
    float a = (float)reg[row * ldr + c], b = (float)edge[erow * lde + c], go = (float)dout[idx];
    float hid[32], z0 = 0.f, z1 = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      hid[j] = fmaxf(sw0[2 * j] * a + sw0[2 * j + 1] * b, 0.f);
      z0 += sw2[j] * hid[j];
      z1 += sw2[32 + j] * hid[j];
    }
    float g0 = sigmoidf_(z0), g1 = sigmoidf_(z1);
    float dz0 = go * a * g0 * (1.f - g0), dz1 = go * b * g1 * (1.f - g1);
    float da = go * g0, db = go * g1;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      float dh = hid[j] > 0.f ? dz0 * sw2[j] + dz1 * sw2[32 + j] : 0.f;
      da += dh * sw0[2 * j];
      db += dh * sw0[2 * j + 1];
      v[j] += dz0 * hid[j];
      v[32 + j] += dz1 * hid[j];
      v[64 + 2 * j] += dh * a;
      v[64 + 2 * j + 1] += dh * b;
    }
    {
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (ex.add_reg[k] && row < ex.add_reg_rows[k]) da += (float)ex.add_reg[k][idx];
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (ex.add_edge[k] && row < ex.add_edge_rows[k]) db += (float)ex.add_edge[k][idx];
    }
    dreg[idx] = (f16)da;
    dedge[erow * C + c] = (f16)db;
    const long seg = ex.pos_seg_rows > 0 ? row / ex.pos_seg_rows : 0;
    if (fixed_col) {
      if (gcol >= 0 && seg != gseg) { atomicAdd(dpos + gseg * ex.pos_stride + gcol, gsum); gsum = 0.f; }
      gsum += go; gcol = c; gseg = seg;
    } else {
      atomicAdd(dpos + seg * ex.pos_stride + c, go);
    }
  }
  if (fixed_col && gcol >= 0) atomicAdd(dpos + gseg * ex.pos_stride + gcol, gsum);
  // wave reduce-scatter: at each step a lane keeps one half of its values and receives the partner's share of that half,
  // so the 128 column sums cost 126 shuffles instead of 128 x 6; afterwards lane l owns values base(l), base(l)+1
  const int lane = threadIdx.x & 63;
  int base = 0;
#define RS_STEP(HALF, BIT)                                              \
  {                                                                     \
    const bool up = (lane & BIT) != 0;                                  \
    _Pragma("unroll") for (int i = 0; i < HALF; ++i) {                  \
      float keep = up ? v[i + HALF] : v[i];                             \
      float send = up ? v[i] : v[i + HALF];                             \
      v[i] = keep + __shfl_xor(send, BIT, 64);                          \
    }                                                                   \
    base += up ? HALF : 0;                                              \
  }
  RS_STEP(64, 32) RS_STEP(32, 16) RS_STEP(16, 8) RS_STEP(8, 4) RS_STEP(4, 2) RS_STEP(2, 1)
#undef RS_STEP
  atomicAdd(&acc[base], v[0]);
  atomicAdd(&acc[base + 1], v[1]);
  __syncthreads();
  if (threadIdx.x < 64) atomicAdd(dw2 + threadIdx.x, acc[threadIdx.x]);
  else if (threadIdx.x < 128) atomicAdd(dw0 + threadIdx.x - 64, acc[threadIdx.x]);
}
static int se_fuse_bwd_launch(const void* reg, int ldr, const void* edge, int lde, const float* w0, const float* w2, const void* dout, long n,
                              int C, void* dreg, void* dedge, float* dpos, float* dw0, float* dw2, bool zero, const SeFuseEx* ex, void* stream);
extern "C" int tscd_se_fuse_bwd_f16(const void* reg, int ldr, const void* edge, int lde, const float* w0, const float* w2,
                                    const void* dout, long n, int C, void* dreg, void* dedge, float* dpos, float* dw0, float* dw2,
                                    void* stream) {
  return se_fuse_bwd_launch(reg, ldr, edge, lde, w0, w2, dout, n, C, dreg, dedge, dpos, dw0, dw2, true, nullptr, stream);
}
// The same ADDING into dpos / dw0 / dw2 (the caller cleared them, e.g. as parts of one arena zeroed once for a whole backward chain).
extern "C" int tscd_se_fuse_bwd_acc_f16(const void* reg, int ldr, const void* edge, int lde, const float* w0, const float* w2,
                                        const void* dout, long n, int C, void* dreg, void* dedge, float* dpos, float* dw0, float* dw2,
                                        void* stream) {
  return se_fuse_bwd_launch(reg, ldr, edge, lde, w0, w2, dout, n, C, dreg, dedge, dpos, dw0, dw2, false, nullptr, stream);
}
// The adding form with the chain's extras (struct TscdSeFuseEx): permuted edge rows (d edge is written to the row it was read from),
// one d pos vector per row segment, and up to 3 / 2 fp16 addends summed into d reg / d edge before the single rounding (dreg / dedge
// may BE one of their addends: every element is read and written by the same thread).
extern "C" int tscd_se_fuse_bwd_ex_f16(const void* reg, int ldr, const void* edge, int lde, const float* w0, const float* w2,
                                       const void* dout, long n, int C, void* dreg, void* dedge, float* dpos, float* dw0, float* dw2,
                                       const void* ex, void* stream) {
  TSCD_CHECK_ARG(ex, "se_fuse_bwd_ex: null extras");
  return se_fuse_bwd_launch(reg, ldr, edge, lde, w0, w2, dout, n, C, dreg, dedge, dpos, dw0, dw2, false, (const SeFuseEx*)ex, stream);
}
static int se_fuse_bwd_launch(const void* reg, int ldr, const void* edge, int lde, const float* w0, const float* w2, const void* dout, long n,
                              int C, void* dreg, void* dedge, float* dpos, float* dw0, float* dw2, bool zero, const SeFuseEx* ex, void* stream) {
  TSCD_CHECK_ARG(reg && edge && w0 && w2 && dout && dreg && dedge && dpos && dw0 && dw2, "se_fuse_bwd: null pointer");
  if (n == 0) return TSCD_OK;
  // dpos [C] | dw0 [64] | dw2 [64] are atomic accumulators: cleared here (callers pass them as one C+128 float block or apart)
  hipStream_t sst = (hipStream_t)stream;
  if (!zero) {
  } else if (dw0 == dpos + C && dw2 == dw0 + 64) {
    (void)hipMemsetAsync(dpos, 0, sizeof(float) * (C + 128), sst);
  } else {
    (void)hipMemsetAsync(dpos, 0, sizeof(float) * C, sst);
    (void)hipMemsetAsync(dw0, 0, sizeof(float) * 64, sst);
    (void)hipMemsetAsync(dw2, 0, sizeof(float) * 64, sst);
  }
  long total = n * C;
  int grid = (int)((total + 1023) / 1024);                 // ~4 elements per thread
  if (grid > 256) grid = 256;
  if (ex) {
    TSCD_CHECK_ARG(!ex->edge_map.idx || ex->edge_map.seg_rows > 0, "se_fuse_bwd: row map without a segment length");
    hipLaunchKernelGGL(se_fuse_bwd_kernel, dim3(grid), dim3(256), 0, sst, (const f16*)reg, ldr, (const f16*)edge, lde, w0, w2,
                       (const f16*)dout, n, C, (f16*)dreg, (f16*)dedge, dpos, dw0, dw2, *ex);
  } else {
    hipLaunchKernelGGL(se_fuse_bwd_kernel, dim3(grid), dim3(256), 0, sst, (const f16*)reg, ldr, (const f16*)edge, lde, w0, w2,
                       (const f16*)dout, n, C, (f16*)dreg, (f16*)dedge, dpos, dw0, dw2, SeFuseEx{});
  }
  TSCD_CHECK_LAUNCH("se_fuse_bwd");
  return TSCD_OK;
}

// add_layernorm backward: y = LN(a (+ b)) * gamma + beta.  dx (= da = db) fp16; dgamma / dbeta fp32 atomics (zeroed by caller).
// One wave per row, rows strided over the grid; every lane keeps the d gamma / d beta partial sums of its columns
// (c = lane + 64 k, k < 16) in registers for all its rows, the block's four waves are combined through LDS, and each block adds
// ONE fp32 atomic per column (a per-element atomic was 4M atomics for TaskAligned's 2000 x 1024).
#define LN_BWD_NC 16
__global__ __launch_bounds__(256) void add_layernorm_bwd_kernel(const f16* __restrict__ a, const f16* __restrict__ b,
                                                                const float* __restrict__ gamma, const f16* __restrict__ dy, int n, int C,
                                                                float eps, f16* __restrict__ dx, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, LnEx ex) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = gridDim.x * 4;
  float ag[LN_BWD_NC], ab[LN_BWD_NC];
#pragma unroll
  for (int k = 0; k < LN_BWD_NC; ++k) ag[k] = ab[k] = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < n; row += nwaves) {
    const long arow = map_row(ex.a_map, row);
    // dy of the row: plus the second contribution where the caller has one (rounded to fp16 like the tensor add it replaces)
    auto dyv = [&](int c) -> float {
      float d = (float)dy[(long)row * C + c];
      if (ex.dy_add) d = (float)(f16)(d + (float)ex.dy_add[(long)row * C + c]);
      return d;
    };
    auto put = [&](int c, float v) {
      if (dx) dx[(long)row * C + c] = (f16)v;
      if (ex.dx_mapped) ex.dx_mapped[arow * C + c] = (f16)v;
    };
    // the row's first 1024 columns live in registers: x and dy are fetched ONCE (the three passes below were three dependent trips to
    // L2 per row -- with one row per wave the whole launch was that latency chain)
    float xr[LN_BWD_NC], dr[LN_BWD_NC];
#pragma unroll
    for (int k = 0; k < LN_BWD_NC; ++k) {
      const int c = lane + 64 * k;
      xr[k] = dr[k] = 0.f;
      if (c < C) {
        float x = (float)a[arow * C + c];
        if (b) x = (float)(f16)(x + (float)b[(long)row * C + c]);
        xr[k] = x;
        dr[k] = dyv(c);
      }
    }
    float s = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < LN_BWD_NC; ++k) { s += xr[k]; s2 += xr[k] * xr[k]; }
    for (int c = lane + 64 * LN_BWD_NC; c < C; c += 64) {
      float x = (float)a[arow * C + c];
      if (b) x = (float)(f16)(x + (float)b[(long)row * C + c]);
      s += x; s2 += x * x;
    }
    s = wave_sum(s); s2 = wave_sum(s2);
    float mean = s / C, var = fmaxf(s2 / C - mean * mean, 0.f), inv = 1.0f / sqrtf(var + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < LN_BWD_NC; ++k) {
      const int c = lane + 64 * k;
      if (c < C) {
        const float g = dr[k] * gamma[c];
        m1 += g; m2 += g * ((xr[k] - mean) * inv);
      }
    }
    for (int c = lane + 64 * LN_BWD_NC; c < C; c += 64) {
      float x = (float)a[arow * C + c];
      if (b) x = (float)(f16)(x + (float)b[(long)row * C + c]);
      float xh = (x - mean) * inv, g = dyv(c) * gamma[c];
      m1 += g; m2 += g * xh;
    }
    m1 = wave_sum(m1) / C; m2 = wave_sum(m2) / C;
#pragma unroll
    for (int k = 0; k < LN_BWD_NC; ++k) {
      const int c = lane + 64 * k;
      if (c < C) {
        const float xh = (xr[k] - mean) * inv, d = dr[k];
        put(c, inv * (d * gamma[c] - m1 - xh * m2));
        ag[k] += d * xh;
        ab[k] += d;
      }
    }
    for (int c = lane + 64 * LN_BWD_NC; c < C; c += 64) {        // C > 1024: the tail falls back to per-element atomics
      float x = (float)a[arow * C + c];
      if (b) x = (float)(f16)(x + (float)b[(long)row * C + c]);
      float xh = (x - mean) * inv, d = dyv(c);
      put(c, inv * (d * gamma[c] - m1 - xh * m2));
      atomicAdd(dgamma + c, d * xh);
      atomicAdd(dbeta + c, d);
    }
  }
#pragma unroll
  for (int k = 0; k < LN_BWD_NC; ++k) {
    const int c = lane + 64 * k;
    if (64 * k >= C) break;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();
      red[wave][lane] = pass ? ab[k] : ag[k];
      __syncthreads();
      if (wave == 0 && c < C) atomicAdd((pass ? dbeta : dgamma) + c, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
    }
  }
}
static int add_layernorm_bwd_launch(const void* a, const void* b, const float* gamma, const void* dy, int n, int C, float eps, void* dx,
                                    float* dgamma, float* dbeta, bool zero, const LnEx* ex, void* stream);
extern "C" int tscd_add_layernorm_bwd_f16(const void* a, const void* b, const float* gamma, const void* dy, int n, int C, float eps,
                                          void* dx, float* dgamma, float* dbeta, void* stream) {
  return add_layernorm_bwd_launch(a, b, gamma, dy, n, C, eps, dx, dgamma, dbeta, true, nullptr, stream);
}
// The same ADDING into dgamma / dbeta (cleared by the caller).
extern "C" int tscd_add_layernorm_bwd_acc_f16(const void* a, const void* b, const float* gamma, const void* dy, int n, int C, float eps,
                                              void* dx, float* dgamma, float* dbeta, void* stream) {
  return add_layernorm_bwd_launch(a, b, gamma, dy, n, C, eps, dx, dgamma, dbeta, false, nullptr, stream);
}
// The adding form with the chain's extras (struct TscdLnEx): `a` read through a row map, a second dy contribution, and dx also (or only:
// dx may be null then) written to the mapped row of a second buffer.
extern "C" int tscd_add_layernorm_bwd_ex_f16(const void* a, const void* b, const float* gamma, const void* dy, int n, int C, float eps,
                                             void* dx, float* dgamma, float* dbeta, const void* ex, void* stream) {
  TSCD_CHECK_ARG(ex && (dx || ((const LnEx*)ex)->dx_mapped), "add_layernorm_bwd_ex: no extras / no destination");
  return add_layernorm_bwd_launch(a, b, gamma, dy, n, C, eps, dx, dgamma, dbeta, false, (const LnEx*)ex, stream);
}
static int add_layernorm_bwd_launch(const void* a, const void* b, const float* gamma, const void* dy, int n, int C, float eps, void* dx,
                                    float* dgamma, float* dbeta, bool zero, const LnEx* ex, void* stream) {
  TSCD_CHECK_ARG(a && gamma && dy && (dx || ex) && dgamma && dbeta && C > 0, "add_layernorm_bwd: bad args");
  if (!zero) {
  } else if (dbeta == dgamma + C) {                                            // atomic accumulators
    (void)hipMemsetAsync(dgamma, 0, sizeof(float) * 2 * C, (hipStream_t)stream);
  } else {
    (void)hipMemsetAsync(dgamma, 0, sizeof(float) * C, (hipStream_t)stream);
    (void)hipMemsetAsync(dbeta, 0, sizeof(float) * C, (hipStream_t)stream);
  }
  if (n == 0) return TSCD_OK;
  int grid = (n + 3) / 4;
  if (grid > 256) grid = 256;                      // >= 1 row per wave; at most one block per CU
  if (ex) {
    TSCD_CHECK_ARG(!ex->a_map.idx || ex->a_map.seg_rows > 0, "add_layernorm_bwd: row map without a segment length");
    hipLaunchKernelGGL(add_layernorm_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f16*)a, (const f16*)b, gamma,
                       (const f16*)dy, n, C, eps, (f16*)dx, dgamma, dbeta, *ex);
  } else {
    hipLaunchKernelGGL(add_layernorm_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f16*)a, (const f16*)b, gamma,
                       (const f16*)dy, n, C, eps, (f16*)dx, dgamma, dbeta, LnEx{});
  }
  TSCD_CHECK_LAUNCH("add_layernorm_bwd");
  return TSCD_OK;
}
