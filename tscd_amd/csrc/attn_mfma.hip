// Cosine multi-head attention of CAFM / TaskAligned on the matrix cores (reference: PositionMHAttention / MHAttention,
// yolox/models/tscd_matching.py:31-60, 140-181): out = softmax_k(q^.k^) v per head, q^ = q/||q||, k^ = k/||k||, rows grouped
// in segments (one frame's queries attend to that frame's keys).  Head dims that are multiples of 16 (TSCD-L: 32 in CAFM,
// 128 in TaskAligned) run here; the others keep the scalar kernels of attn.hip.
//
// One wave = one (32-query tile, head) task; a workgroup is a single wave, so the only synchronisation is the wave's own
// LDS wait.  Every product is arranged so that the QUERY index lives on the lanes (MFMA column = lane & 31):
//   S^T  [key  x query] = K^   Q^^T    A = k^ rows (each lane normalises and feeds its own key row), B = q^ rows (registers)
//   O^T  [d    x query] = V^T  P^T     A = V^T by ds_read_b64_tr_b16 from the LDS copy of the V tile, B = P^T straight from
//                                      the S^T accumulator registers (no transpose: the key order of the reduction is
//                                      permuted identically on both operands)
// so softmax sums, the 1/l scaling and (backward) delta = dO.O and the q-normalisation gradient are all per-lane scalars.
// cos <= 1 bounds the logits: exp(s - 1) needs no running maximum.
// Backward (same tiling, given dO, O and the saved row sums l):
//   dP^T = V dO^T,  dS^T = P^T o (dP^T - delta),  dQ^^T = K^^T dS^T (A by transposing reads of the k^ tile),
//   dK^ = dS^T Q^ and dV = P^T dO with both operands read transposed from LDS tiles (dS / P written once per key tile as
//   fp16 [query][key]); their results have the channel on the lanes, so the fp32 atomics that sum them over query tiles are
//   two full 128-byte segments per instruction.  k's normalisation gradient is applied by cos_attn_bwd_finish (attn.hip).
#include "common.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
struct AttnTile { int row0, nrows, key0, key1; };

// ds_read_b64_tr_b16 pair: the 16-lane group reads 4 rows x 16 columns twice (rows r..r+3 and r+ROW2..r+ROW2+3) and each
// lane receives its column's 8 values.  `p` points at (row r, column 0 of the group's 16-column block).
template <int ROW2>
__device__ __forceinline__ f16x8 tr_pair(const f16* p, int pitch, int lane) {
  const int i = lane & 15, q = i >> 2, c = i & 3;
  auto a0 = (__attribute__((address_space(3))) s16x4*)(p + q * pitch + 4 * c);
  auto a1 = (__attribute__((address_space(3))) s16x4*)(p + (q + ROW2) * pitch + 4 * c);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(a1);
  f16x4 lo4 = __builtin_bit_cast(f16x4, lo), hi4 = __builtin_bit_cast(f16x4, hi);   // whole-vector casts (see conv_wgrad.hip)
  return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
}

// LDS row pitch (halves) of a [32][D] tile: pitch/2 = 16 mod 64 dwords puts the 4 rows of a transposing read on disjoint
// bank quarters
__host__ __device__ constexpr int tile_pitch(int D) { return D <= 32 ? 32 : 160; }

// row of x (raw fp16) -> this lane's MFMA fragments (8 channels per 16-wide k-step: d = 16 s + 8 (lane>>5) + j) and the
// inverse norm of the whole head row (the two half-lanes of a row exchange their partial sums)
template <int KS>
__device__ __forceinline__ float load_row(const f16* __restrict__ row, bool ok, int half, f16x8 (&f)[KS]) {
  float ss = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    f[s] = ok ? *(const f16x8*)(row + 16 * s + 8 * half) : z;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += (float)f[s][j] * (float)f[s][j];
  }
  ss += __shfl_xor(ss, 32, 64);
  return ss > 0.f ? 1.0f / sqrtf(ss) : 0.f;
}
// The same in two steps, so that a loop can FETCH the next tile's raw rows before the matrix products of the current one (the
// one-wave workgroups of this file are latency-bound: a fetch that is waited for at once costs a full L2 round trip per tile)
template <int KS>
__device__ __forceinline__ void load_raw(const f16* __restrict__ row, bool ok, int half, f16x8 (&f)[KS]) {
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    f[s] = ok ? *(const f16x8*)(row + 16 * s + 8 * half) : z;
  }
}
template <int KS>
__device__ __forceinline__ float inv_norm(const f16x8 (&f)[KS]) {
  float ss = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += (float)f[s][j] * (float)f[s][j];
  ss += __shfl_xor(ss, 32, 64);
  return ss > 0.f ? 1.0f / sqrtf(ss) : 0.f;
}
template <int KS>
__device__ __forceinline__ void scale_row(f16x8 (&f)[KS], float inv) {
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) f[s][j] = (f16)((float)f[s][j] * inv);
}

// Waves per (query tile, head) task: the task's key tiles are dealt round-robin to NW waves (cos <= 1 needs no running maximum, so
// the waves' partial sums simply add) and wave 0 adds the partials in wave order -- the one-wave loop over 16 key tiles was a pure
// latency chain (a 32-row tile per microsecond).  Bounded by the 64 KB of static LDS: 4 waves up to head dim 32, 2 above.
__host__ __device__ constexpr int attn_waves(int KS) { return KS <= 2 ? 4 : 2; }
__host__ __device__ constexpr int attn_waves_kv(int KS) { return KS <= 2 ? 4 : 1; }      // the key-stationary kernel keeps two tiles per wave

template <int KS, int NW>
__global__ __launch_bounds__(64 * NW) void cos_attn_mfma_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                               const f16* __restrict__ v, int C, int heads,
                                                               const AttnTile* __restrict__ tiles, f16* __restrict__ out,
                                                               float* __restrict__ lsum) {
  constexpr int D = 16 * KS, MT = (KS + 1) / 2, P = tile_pitch(MT * 32), RS = MT * 16 + 1;
  __shared__ __attribute__((aligned(16))) f16 vt_all[NW][32 * P];
  __shared__ float red[NW > 1 ? (NW - 1) * 64 * RS : 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, col = lane & 31, half = lane >> 5, blk = (lane >> 4) & 1;
  f16* const vt = vt_all[wv];
  const AttnTile t = tiles[blockIdx.x / heads];
  const int h = blockIdx.x % heads;
  const bool qok = col < t.nrows;
  const long qrow = t.row0 + (qok ? col : 0);

  f16x8 qf[KS];
  scale_row<KS>(qf, load_row<KS>(q + qrow * C + h * D, qok, half, qf));
  // zero the pad columns of the V tile once (head dims that are not a multiple of 32)
  if (MT * 32 > D)
    for (int i = lane; i < 32 * (MT * 32 - D); i += 64) vt[(i / (MT * 32 - D)) * P + D + i % (MT * 32 - D)] = (f16)0.f;

  f32x16 oacc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[m][e] = 0.f;
  float psum = 0.f;

  f16x8 kn[KS], vn[KS];            // raw rows of this wave's NEXT key tile, in flight under the current tile's products
  const int iters = ((t.key1 - t.key0 + 31) / 32 + NW - 1) / NW;       // the same count for every wave (the loop holds barriers)
  {
    const int kb0 = t.key0 + 32 * wv;
    const bool ok0 = kb0 + col < t.key1;
    const long r0 = ok0 ? kb0 + col : t.key0;
    load_raw<KS>(k + r0 * C + h * D, ok0, half, kn);
    load_raw<KS>(v + r0 * C + h * D, ok0, half, vn);
  }
  for (int it = 0; it < iters; ++it) {
    const int kb = t.key0 + 32 * (it * NW + wv);
    f16x8 kf[KS], vf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { kf[s] = kn[s]; vf[s] = vn[s]; }
    if (it + 1 < iters) {
      const int kbn = kb + 32 * NW;
      const bool okn = kbn + col < t.key1;
      const long rn = okn ? kbn + col : t.key0;
      load_raw<KS>(k + rn * C + h * D, okn, half, kn);
      load_raw<KS>(v + rn * C + h * D, okn, half, vn);
    }
    scale_row<KS>(kf, inv_norm<KS>(kf));
    __syncthreads();   // orders the previous tile's transposing reads before the overwrite (each wave has its own tile)
#pragma unroll
    for (int s = 0; s < KS; ++s) *(f16x8*)(vt + col * P + 16 * s + 8 * half) = vf[s];
    f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[s], qf[s], st, 0, 0, 0);
    // P^T in the accumulator layout: register e <-> key (e&3) + 8 (e>>2) + 4 half
    f16x8 pb[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int kl = (e & 3) + 8 * (e >> 2) + 4 * half;
      float pv = (kb + kl < t.key1) ? __expf(st[e] - 1.0f) : 0.f;
      psum += pv;
      pb[e >> 3][e & 7] = (f16)pv;
    }
    __syncthreads();
    // O^T[d][query] += V^T P^T: k-step s2 <-> keys 16 s2 + 4 half + {0..3, 8..11} on both operands
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f16x8 a = tr_pair<8>(vt + (16 * s2 + 4 * half) * P + m * 32 + 16 * blk, P, lane);
        oacc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb[s2], oacc[m], 0, 0, 0);
      }
  }
  psum += __shfl_xor(psum, 32, 64);
  if (NW > 1) {                                  // partial sums of waves 1.. -> wave 0, added in wave order
    if (wv > 0) {
      float* r = red + ((wv - 1) * 64 + lane) * RS;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) r[m * 16 + e] = oacc[m][e];
      r[MT * 16] = psum;
    }
    __syncthreads();
    if (wv > 0) return;
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) {
      const float* r = red + ((ww - 1) * 64 + lane) * RS;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[m][e] += r[m * 16 + e];
      psum += r[MT * 16];
    }
  }
  const float il = psum > 0.f ? 1.0f / psum : 0.f;
  if (qok) {
    if (half == 0) lsum[qrow * heads + h] = psum;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = m * 32 + 8 * g + 4 * half;
        if (d0 < D) {
          f16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (f16)(oacc[m][4 * g + j] * il);
          *(f16x4*)(out + qrow * C + h * D + d0) = o;
        }
      }
  }
}

// KV = true: the one-launch form -- dq written directly, dk^ / dv summed over query tiles by fp32 atomics into the workspace
// (2 * nk * C floats, zeroed by the caller).  KV = false: dq only, and delta = dO.O per (query, head) to `delta_out` for the
// key-stationary kernel below (dkn doubles as delta_out).
template <int KS, bool KV, int NW>
__global__ __launch_bounds__(64 * NW) void cos_attn_mfma_bwd_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                                   const f16* __restrict__ v, const f16* __restrict__ dO,
                                                                   const f16* __restrict__ o, const float* __restrict__ lsum, int C,
                                                                   int heads, const AttnTile* __restrict__ tiles, f16* __restrict__ dq,
                                                                   float* __restrict__ dkn, float* __restrict__ dvacc) {
  static_assert(!KV || NW == 1, "the one-launch (atomic) form runs one wave per task");
  constexpr int D = 16 * KS, MT = (KS + 1) / 2, P = tile_pitch(MT * 32), TP = 32, RS = MT * 16 + 1;
  __shared__ __attribute__((aligned(16))) f16 qt[32 * P], gt[KV ? 32 * P : 8], kt_all[NW][32 * P];
  __shared__ __attribute__((aligned(16))) f16 ts[KV ? 32 * TP : 8], tp[KV ? 32 * TP : 8];    // dS and P, [query][key]
  __shared__ float red[NW > 1 ? (NW - 1) * 64 * RS : 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, col = lane & 31, half = lane >> 5, blk = (lane >> 4) & 1;
  f16* const kt = kt_all[wv];
  const AttnTile t = tiles[blockIdx.x / heads];
  const int h = blockIdx.x % heads;
  const bool qok = col < t.nrows;
  const long qrow = t.row0 + (qok ? col : 0);

  f16x8 qf[KS], gf[KS];
  const float qinv = load_row<KS>(q + qrow * C + h * D, qok, half, qf);
  scale_row<KS>(qf, qinv);
  float delta = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    gf[s] = qok ? *(const f16x8*)(dO + qrow * C + h * D + 16 * s + 8 * half) : z;
    f16x8 of = qok ? *(const f16x8*)(o + qrow * C + h * D + 16 * s + 8 * half) : z;
#pragma unroll
    for (int j = 0; j < 8; ++j) delta += (float)gf[s][j] * (float)of[j];
  }
  delta += __shfl_xor(delta, 32, 64);
  const float l = qok ? lsum[qrow * heads + h] : 0.f;
  const float il = l > 0.f ? 1.0f / l : 0.f;
  if (!KV && qok && half == 0 && wv == 0) dkn[qrow * heads + h] = delta;
  // LDS copies of q^ and dO ([query][d], zero pad columns) for the transposed operands of dK^ / dV (wave 0 writes the shared q^ copy)
  for (int i = lane; i < 32 * P; i += 64) {
    if (wv == 0) qt[i] = (f16)0.f;
    kt[i] = (f16)0.f;
    if (KV) gt[i] = (f16)0.f;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (wv == 0) *(f16x8*)(qt + col * P + 16 * s + 8 * half) = qf[s];
    if (KV) *(f16x8*)(gt + col * P + 16 * s + 8 * half) = gf[s];
  }

  f32x16 dqacc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) dqacc[m][e] = 0.f;

  f16x8 kn[KS], vn[KS];            // raw rows of this wave's NEXT key tile (see the forward kernel)
  const int iters = ((t.key1 - t.key0 + 31) / 32 + NW - 1) / NW;
  {
    const int kb0 = t.key0 + 32 * wv;
    const bool ok0 = kb0 + col < t.key1;
    const long r0 = ok0 ? kb0 + col : t.key0;
    load_raw<KS>(k + r0 * C + h * D, ok0, half, kn);
    load_raw<KS>(v + r0 * C + h * D, ok0, half, vn);
  }
  for (int it = 0; it < iters; ++it) {
    const int kb = t.key0 + 32 * (it * NW + wv);
    f16x8 kf[KS], vf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { kf[s] = kn[s]; vf[s] = vn[s]; }
    if (it + 1 < iters) {
      const int kbn = kb + 32 * NW;
      const bool okn = kbn + col < t.key1;
      const long rn = okn ? kbn + col : t.key0;
      load_raw<KS>(k + rn * C + h * D, okn, half, kn);
      load_raw<KS>(v + rn * C + h * D, okn, half, vn);
    }
    scale_row<KS>(kf, inv_norm<KS>(kf));
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KS; ++s) *(f16x8*)(kt + col * P + 16 * s + 8 * half) = kf[s];
    f32x16 st, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = dp[e] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[s], qf[s], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[s], gf[s], dp, 0, 0, 0);
    }
    f16x8 sb[2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f16x4 s4, p4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = 4 * g + j;
        const int kl = j + 8 * g + 4 * half;
        float pn = (kb + kl < t.key1) ? __expf(st[e] - 1.0f) * il : 0.f;
        float ds = pn * (dp[e] - delta);
        s4[j] = (f16)ds;
        p4[j] = (f16)pn;
        sb[e >> 3][e & 7] = (f16)ds;
      }
      if (KV) {
        *(f16x4*)(ts + col * TP + 8 * g + 4 * half) = s4;     // [query = this lane's column][4 consecutive keys]
        *(f16x4*)(tp + col * TP + 8 * g + 4 * half) = p4;
      }
    }
    __syncthreads();
    // dQ^^T[d][query] += K^^T dS^T (keys permuted like the forward's V^T P^T)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f16x8 a = tr_pair<8>(kt + (16 * s2 + 4 * half) * P + m * 32 + 16 * blk, P, lane);
        dqacc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, sb[s2], dqacc[m], 0, 0, 0);
      }
    // dK^[key][d] = sum_q dS[q][key] q^[q][d],  dV[key][d] = sum_q P[q][key] dO[q][d]: both operands transposed from LDS,
    // k-step s2 <-> queries 16 s2 + 8 half + 0..7
    if (KV)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      f32x16 akk, avv;
#pragma unroll
      for (int e = 0; e < 16; ++e) akk[e] = avv[e] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int r0 = 16 * s2 + 8 * half;
        f16x8 aS = tr_pair<4>(ts + r0 * TP + 16 * blk, TP, lane);
        f16x8 aP = tr_pair<4>(tp + r0 * TP + 16 * blk, TP, lane);
        f16x8 bQ = tr_pair<4>(qt + r0 * P + m * 32 + 16 * blk, P, lane);
        f16x8 bG = tr_pair<4>(gt + r0 * P + m * 32 + 16 * blk, P, lane);
        akk = __builtin_amdgcn_mfma_f32_32x32x16_f16(aS, bQ, akk, 0, 0, 0);
        avv = __builtin_amdgcn_mfma_f32_32x32x16_f16(aP, bG, avv, 0, 0, 0);
      }
      const int d = m * 32 + col;
      if (d < D) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int kl = (e & 3) + 8 * (e >> 2) + 4 * half;
          if (kb + kl < t.key1) {
            const long off = (long)(kb + kl) * C + h * D + d;
            atomicAdd(dkn + off, akk[e]);
            atomicAdd(dvacc + off, avv[e]);
          }
        }
      }
    }
  }
  if (NW > 1) {                                  // partial dq^ of waves 1.. -> wave 0, added in wave order
    if (wv > 0) {
      float* r = red + ((wv - 1) * 64 + lane) * RS;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) r[m * 16 + e] = dqacc[m][e];
    }
    __syncthreads();
    if (wv > 0) return;
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) {
      const float* r = red + ((ww - 1) * 64 + lane) * RS;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) dqacc[m][e] += r[m * 16 + e];
    }
  }
  // q normalisation backward, per lane: dq = (dq^ - q^ (q^.dq^)) / ||q||; this lane's d set is m*32 + 8 g + 4 half + j
  float dot = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f16x4 q4 = *(const f16x4*)(qt + col * P + m * 32 + 8 * g + 4 * half);
#pragma unroll
      for (int j = 0; j < 4; ++j) dot += dqacc[m][4 * g + j] * (float)q4[j];
    }
  dot += __shfl_xor(dot, 32, 64);
  if (qok) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = m * 32 + 8 * g + 4 * half;
        if (d0 < D) {
          f16x4 q4 = *(const f16x4*)(qt + col * P + d0), r;
#pragma unroll
          for (int j = 0; j < 4; ++j) r[j] = (f16)((dqacc[m][4 * g + j] - (float)q4[j] * dot) * qinv);
          *(f16x4*)(dq + qrow * C + h * D + d0) = r;
        }
      }
  }
}

// Key-stationary half of the two-launch backward: one wave = one (32-key tile, head) task looping over the query tiles that attend
// to it, so dK^ and dV accumulate in registers (no atomics, no fp32 workspace, no finish pass) and k's normalisation gradient is a
// per-lane scalar like q's above.  The products mirror the query-stationary kernel with the roles of the two sides exchanged:
//   S   [query x key] = Q^  K^^T     A = q^ rows of the tile, B = k^ rows (registers, loaded once)
//   dP  [query x key] = dO  V^T
//   dK^^T [d x key]  += Q^^T dS,  dV^T [d x key] += dO^T P     A by transposing reads of the q^ / dO tiles, B = dS / P straight
//                                                              from the accumulators (query order permuted alike on both)
// 1/l and delta = dO.O vary along the accumulator's rows here: the tile's 32 values are staged in LDS.
// ktiles: (key row0, nkeys <= 32, query0, query1).  delta: [n][heads] fp32 written by cos_attn_mfma_bwd_kernel<KS, false>.
template <int KS, int NW>
__global__ __launch_bounds__(64 * NW) void cos_attn_mfma_bwd_kv_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                                      const f16* __restrict__ v, const f16* __restrict__ dO,
                                                                      const float* __restrict__ lsum, const float* __restrict__ delta,
                                                                      int C, int heads, const AttnTile* __restrict__ ktiles,
                                                                      f16* __restrict__ dk, f16* __restrict__ dv) {
  constexpr int D = 16 * KS, MT = (KS + 1) / 2, P = tile_pitch(MT * 32), RS = 2 * MT * 16 + 1;
  __shared__ __attribute__((aligned(16))) f16 qt_all[NW][32 * P], gt_all[NW][32 * P], kt[32 * P];
  __shared__ float ils_all[NW][32], dls_all[NW][32];
  __shared__ float red[NW > 1 ? (NW - 1) * 64 * RS : 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, col = lane & 31, half = lane >> 5, blk = (lane >> 4) & 1;
  f16* const qt = qt_all[wv];
  f16* const gt = gt_all[wv];
  float* const ils = ils_all[wv];
  float* const dls = dls_all[wv];
  const AttnTile t = ktiles[blockIdx.x / heads];
  const int h = blockIdx.x % heads;
  const bool kok = col < t.nrows;
  const long krow = t.row0 + (kok ? col : 0);

  f16x8 kf[KS], vf[KS];
  const float kinv = load_row<KS>(k + krow * C + h * D, kok, half, kf);
  scale_row<KS>(kf, kinv);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    vf[s] = kok ? *(const f16x8*)(v + krow * C + h * D + 16 * s + 8 * half) : z;
  }
  for (int i = lane; i < 32 * P; i += 64) {
    qt[i] = (f16)0.f;
    gt[i] = (f16)0.f;
    if (wv == 0) kt[i] = (f16)0.f;
  }
  __syncthreads();
  if (wv == 0)
#pragma unroll
    for (int s = 0; s < KS; ++s) *(f16x8*)(kt + col * P + 16 * s + 8 * half) = kf[s];

  f32x16 dkacc[MT], dvacc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) dkacc[m][e] = dvacc[m][e] = 0.f;

  // raw rows (and row scalars) of this wave's NEXT query tile are fetched under the current tile's products -- up to head dim 64; above,
  // the two extra fragment sets cost more (accumulator-file moves) than the latency they hide: there a tile is fetched when it is due
  constexpr bool PF = KS <= 4;
  f16x8 qn[KS], gn[KS];
  float ln = 0.f, dln = 0.f;
  const int iters = ((t.key1 - t.key0 + 31) / 32 + NW - 1) / NW;       // query tiles dealt round-robin to the NW waves
  auto fetch = [&](int qb) {
    const bool ok = qb + col < t.key1;
    const long r = ok ? qb + col : t.key0;
    load_raw<KS>(q + r * C + h * D, ok, half, qn);
    load_raw<KS>(dO + r * C + h * D, ok, half, gn);
    ln = ok ? lsum[r * heads + h] : 0.f;
    dln = ok ? delta[r * heads + h] : 0.f;
  };
  if (PF) fetch(t.key0 + 32 * wv);
  for (int it = 0; it < iters; ++it) {
    const int qb = t.key0 + 32 * (it * NW + wv);
    if (!PF) fetch(qb);
    f16x8 qf[KS], gf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { qf[s] = qn[s]; gf[s] = gn[s]; }
    const float l = ln, dl = dln;
    if (PF && it + 1 < iters) fetch(qb + 32 * NW);
    scale_row<KS>(qf, inv_norm<KS>(qf));
    __syncthreads();   // single-wave workgroup: the previous tile's transposing reads are done
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      *(f16x8*)(qt + col * P + 16 * s + 8 * half) = qf[s];
      *(f16x8*)(gt + col * P + 16 * s + 8 * half) = gf[s];
    }
    if (half == 0) {
      ils[col] = l > 0.f ? 1.0f / l : 0.f;
      dls[col] = dl;
    }
    f32x16 st, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = dp[e] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_f16(qf[s], kf[s], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(gf[s], vf[s], dp, 0, 0, 0);
    }
    __syncthreads();
    // accumulator register e <-> query (e&3) + 8 (e>>2) + 4 half of the tile, this lane's key
    f16x8 sb[2], pb[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ql = (e & 3) + 8 * (e >> 2) + 4 * half;
      const float pn = (kok && qb + ql < t.key1) ? __expf(st[e] - 1.0f) * ils[ql] : 0.f;
      sb[e >> 3][e & 7] = (f16)(pn * (dp[e] - dls[ql]));
      pb[e >> 3][e & 7] = (f16)pn;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f16x8 aq = tr_pair<8>(qt + (16 * s2 + 4 * half) * P + m * 32 + 16 * blk, P, lane);
        f16x8 ag = tr_pair<8>(gt + (16 * s2 + 4 * half) * P + m * 32 + 16 * blk, P, lane);
        dkacc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq, sb[s2], dkacc[m], 0, 0, 0);
        dvacc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ag, pb[s2], dvacc[m], 0, 0, 0);
      }
  }
  if (NW > 1) {                                  // partial dk^ / dv of waves 1.. -> wave 0, added in wave order
    if (wv > 0) {
      float* r = red + ((wv - 1) * 64 + lane) * RS;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) { r[m * 32 + e] = dkacc[m][e]; r[m * 32 + 16 + e] = dvacc[m][e]; }
    }
    __syncthreads();
    if (wv > 0) return;
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) {
      const float* r = red + ((ww - 1) * 64 + lane) * RS;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dkacc[m][e] += r[m * 32 + e]; dvacc[m][e] += r[m * 32 + 16 + e]; }
    }
  }
  // k normalisation backward, per lane: dk = (dk^ - k^ (k^.dk^)) / ||k||; this lane's d set is m*32 + 8 g + 4 half + j
  float dot = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f16x4 k4 = *(const f16x4*)(kt + col * P + m * 32 + 8 * g + 4 * half);
#pragma unroll
      for (int j = 0; j < 4; ++j) dot += dkacc[m][4 * g + j] * (float)k4[j];
    }
  dot += __shfl_xor(dot, 32, 64);
  if (kok) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = m * 32 + 8 * g + 4 * half;
        if (d0 < D) {
          f16x4 k4 = *(const f16x4*)(kt + col * P + d0), rk, rv;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            rk[j] = (f16)((dkacc[m][4 * g + j] - (float)k4[j] * dot) * kinv);
            rv[j] = (f16)dvacc[m][4 * g + j];
          }
          *(f16x4*)(dk + krow * C + h * D + d0) = rk;
          *(f16x4*)(dv + krow * C + h * D + d0) = rv;
        }
      }
  }
}

extern "C" int tscd_cos_attn_mfma_supported(int head_dim) { return head_dim >= 16 && head_dim <= 160 && head_dim % 16 == 0; }

// tiles: (row0, nrows <= 32, key0, key1).  lsum [n][heads] fp32: softmax denominators (sum_k exp(s-1)), kept for backward.
extern "C" int tscd_cos_attn_mfma_f16(const void* q, const void* k, const void* v, int C, int heads, const int* tiles, int ntiles,
                                      void* out, float* lsum, void* stream) {
  TSCD_CHECK_ARG(q && k && v && tiles && out && lsum && heads > 0 && C % heads == 0, "cos_attn_mfma: bad args");
  const int D = C / heads;
  TSCD_CHECK_ARG(tscd_cos_attn_mfma_supported(D) && C % 8 == 0, "cos_attn_mfma: head dim %d not supported", D);
  if (ntiles == 0) return TSCD_OK;
  dim3 grid(ntiles * heads);
  hipStream_t st = (hipStream_t)stream;
#define CAM(KSV)                                                                                                              \
  case KSV:                                                                                                                    \
    hipLaunchKernelGGL((cos_attn_mfma_kernel<KSV, attn_waves(KSV)>), grid, dim3(64 * attn_waves(KSV)), 0, st, (const f16*)q,   \
                       (const f16*)k, (const f16*)v, C, heads, (const AttnTile*)tiles, (f16*)out, lsum);                       \
    break
  switch (D / 16) {
    CAM(1); CAM(2); CAM(3); CAM(4); CAM(5); CAM(6); CAM(7); CAM(8); CAM(9); CAM(10);
  }
#undef CAM
  TSCD_CHECK_LAUNCH("cos_attn_mfma");
  return TSCD_OK;
}

// gradients w.r.t. the raw q (written), and fp32 accumulators of d k^ and d v in `workspace` (>= 8*nk*C bytes, zeroed
// here) that tscd_cos_attn_bwd_finish_f16 turns into dk / dv.
extern "C" int tscd_cos_attn_mfma_bwd_f16(const void* q, const void* k, const void* v, const void* dO, const void* out,
                                          const float* lsum, int n, int nk, int C, int heads, const int* tiles, int ntiles,
                                          void* workspace, long workspace_bytes, void* dq, void* stream) {
  TSCD_CHECK_ARG(q && k && v && dO && out && lsum && tiles && workspace && dq && heads > 0 && C % heads == 0, "cos_attn_mfma_bwd: bad args");
  const int D = C / heads;
  TSCD_CHECK_ARG(tscd_cos_attn_mfma_supported(D) && C % 8 == 0, "cos_attn_mfma_bwd: head dim %d not supported", D);
  TSCD_CHECK_ARG(workspace_bytes >= 8L * nk * C, "cos_attn_mfma_bwd: workspace too small");
  (void)n;
  hipStream_t st = (hipStream_t)stream;
  if (nk > 0) (void)hipMemsetAsync(workspace, 0, 8L * nk * C, st);
  if (ntiles == 0) return TSCD_OK;
  float* dkn = (float*)workspace;
  float* dva = dkn + (long)nk * C;
  dim3 grid(ntiles * heads), block(64);
#define CAB(KSV)                                                                                                               \
  case KSV:                                                                                                                     \
    hipLaunchKernelGGL((cos_attn_mfma_bwd_kernel<KSV, true, 1>), grid, block, 0, st, (const f16*)q, (const f16*)k, (const f16*)v, \
                       (const f16*)dO, (const f16*)out, lsum, C, heads, (const AttnTile*)tiles, (f16*)dq, dkn, dva);             \
    break
  switch (D / 16) {
    CAB(1); CAB(2); CAB(3); CAB(4); CAB(5); CAB(6); CAB(7); CAB(8); CAB(9); CAB(10);
  }
#undef CAB
  TSCD_CHECK_LAUNCH("cos_attn_mfma_bwd");
  return TSCD_OK;
}

// The two-launch backward for segment lists whose KEY ranges are disjoint (every key row belongs to at most one segment: CAFM's and
// TaskAligned's per-frame attention): a query-stationary launch writes dq and delta = dO.O, a key-stationary launch writes dk and
// dv (fp16, final) -- no atomics, no fp32 accumulators, no finish pass.  ktiles: (key row0, nkeys <= 32, query0, query1), the
// queries that attend to that key tile; delta_ws: n * heads floats.  Key rows outside every ktile are not written.
extern "C" int tscd_cos_attn_mfma_bwd2_f16(const void* q, const void* k, const void* v, const void* dO, const void* out,
                                           const float* lsum, int n, int nk, int C, int heads, const int* tiles, int ntiles,
                                           const int* ktiles, int nktiles, float* delta_ws, void* dq, void* dk, void* dv,
                                           void* stream) {
  TSCD_CHECK_ARG(q && k && v && dO && out && lsum && tiles && ktiles && delta_ws && dq && dk && dv && heads > 0 && C % heads == 0,
                 "cos_attn_mfma_bwd2: bad args");
  const int D = C / heads;
  TSCD_CHECK_ARG(tscd_cos_attn_mfma_supported(D) && C % 8 == 0, "cos_attn_mfma_bwd2: head dim %d not supported", D);
  (void)n; (void)nk;
  hipStream_t st = (hipStream_t)stream;
  if (ntiles == 0 || nktiles == 0) return TSCD_OK;
#define CAB2(KSV)                                                                                                                \
  case KSV:                                                                                                                      \
    hipLaunchKernelGGL((cos_attn_mfma_bwd_kernel<KSV, false, attn_waves(KSV)>), dim3(ntiles * heads), dim3(64 * attn_waves(KSV)), 0, st, \
                       (const f16*)q, (const f16*)k, (const f16*)v, (const f16*)dO, (const f16*)out, lsum, C, heads,                  \
                       (const AttnTile*)tiles, (f16*)dq, delta_ws, (float*)nullptr);                                                 \
    hipLaunchKernelGGL((cos_attn_mfma_bwd_kv_kernel<KSV, attn_waves_kv(KSV)>), dim3(nktiles * heads), dim3(64 * attn_waves_kv(KSV)), 0, \
                       st, (const f16*)q, (const f16*)k, (const f16*)v, (const f16*)dO, lsum, (const float*)delta_ws, C, heads,       \
                       (const AttnTile*)ktiles, (f16*)dk, (f16*)dv);                                                                 \
    break
  switch (D / 16) {
    CAB2(1); CAB2(2); CAB2(3); CAB2(4); CAB2(5); CAB2(6); CAB2(7); CAB2(8); CAB2(9); CAB2(10);
  }
#undef CAB2
  TSCD_CHECK_LAUNCH("cos_attn_mfma_bwd2");
  return TSCD_OK;
}
