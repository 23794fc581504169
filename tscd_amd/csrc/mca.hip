// MCA: global-to-local multi-head cosine attention of the TSCD head, fused for gfx950.
//
// Reference: Attention_mca_g2l.forward / find_similar_round2 (yolox/models/post_trans.py:601-714, 581-599) driven per
// local frame by MCA_tscd_g2l_reg.forward (:1127-1162).  For a local frame with n1 proposals and key set
// {its own n1 proposals} U {all global-frame proposals} (N2 keys), per head h (4 heads, d = W/4):
//     S_c = 25 * score[key] * q^c.k^c      S_r = 25 * q^r.k^r            (q^,k^ L2-normalised per head)
//     P   = ( softmax_row(S_c) + softmax_row(S_r) ) / 2
//     out_c = P @ v_c   out_r = P @ v_r                                  (first aggregation)
//     R_c = mean_h v^c.v^c^T  R_r = mean_h v^r.v^r^T                     (raw cosine similarity of the values)
//     A_c = [R_c > .75] * softmax_row(mean_h P), renormalised by its row sum;  A_r likewise with [R_r > .99]
//     soft_c = A_c @ V_c(all heads)   soft_r = A_r @ V_r                 (second aggregation, "sim_round2")
// The unfused reference materialises ~14 [4,n1,N2] tensors per call.  Here nothing of size n1 x N2 touches HBM:
//   1. mca_prep      normalises q/k/v per head, zero-pads each head to DP = ceil32(d) and writes V transposed.
//   2. mca_stats     pass A: row sums l = sum_key exp(S - 25) for both branches.  |S| <= 25 because q^.k^ in [-1,1]
//                    and score in [0,1], so 25 is a valid fixed softmax shift (no running max needed).
//   3. mca_aggregate pass B: recomputes S with MFMA, forms normalised P, exchanges the 4 heads' tiles through LDS to
//                    build mean_h P / R masks, and runs all four aggregations as MFMA products against V^T.
//                    softmax_row(mean_h P) needs no second normaliser: its denominator cancels in the renormalisation,
//                    A = mask*exp(meanP) / sum(mask*exp(meanP)).
//   4. mca_finalize  sums the key-split partials and writes fp16 rows straight into the concat buffers of the two
//                    output linears (which are the MFMA GEMM of conv_igemm.hip).
// Workgroup = 4 waves = the 4 heads of one 32-query tile; key blocks (32 keys) are strided over gridDim.y splits so
// ~N_L/32 * splits workgroups fill the chip.  All local frames of a clip run in ONE launch: a q-tile belongs to one
// frame and its key blocks are [own frame rows] then [global rows]; the reference's per-frame recomputation of the
// global keys' K/V projections disappears (they are shared).
#include "mca_common.h"

struct McaParams {
  // prepared operands
  const f16 *qn_c, *qn_r;   // [N_L][WP]
  const f16 *kn_c, *kn_r;   // [Npad][WP]
  const f16 *vn_c, *vn_r;   // [Npad][WP]
  const f16 *vt_c, *vt_r;   // [WP][Npad]
  // fragment-major copies for this file's kernels: the 64 x 16 bytes one wave loads for one MFMA B-operand are 1 KB contiguous
  // (a row-major operand costs 32 cache lines per wave load: 16 bytes of 32 rows).  kF / vF: fragment (key block jb, head h, ks) at
  // (((jb HEADS + h) KS + ks) 64 + lane) x 16 B, lane = (key & 31) | ((dim >> 3 & 1) << 5); vtF: fragment (jb, h, 32-dim tile nt, 16-key
  // half ks) at ((((jb HEADS + h) NT + nt) 2 + ks) 64 + lane) x 16 B, lane = (dim & 31) | ((key >> 3 & 1) << 5).
  const f16 *kF_c, *kF_r, *vF_c, *vF_r, *vtF_c, *vtF_r;
  const f16* score;         // [N]
  const McaTile* tiles;
  float* lsum;              // [S][HEADS][2][N_L]
  float *part_oc, *part_or, *part_sc, *part_sr;   // [S][N_L][WP]
  float* part_den;          // [S][2][N_L]
  int N, N_L, Npad, WP, S;
  float scale, sim_thresh, conf_thresh;
};

// -------------------------------------------------------------------------------------------------- prep
// One wave per (row, branch).  q: [N_L][W]; kv: [N][2W] (k | v).  Pads each head to DP.
__global__ void mca_prep_kernel(const f16* __restrict__ q_c, const f16* __restrict__ kv_c, const f16* __restrict__ q_r,
                                const f16* __restrict__ kv_r, int N, int N_L, int Npad, int W, int d, int DP, f16* qn_c,
                                f16* kn_c, f16* vn_c, f16* vt_c, f16* qn_r, f16* kn_r, f16* vn_r, f16* vt_r, f16* xcat_c,
                                f16* xcat_r, f16* kF_c, f16* kF_r, f16* vF_c, f16* vF_r, f16* vtF_c, f16* vtF_r) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  int n = wave >> 1, br = wave & 1;
  if (n >= Npad) return;
  const int WP = MCA_HEADS * DP;
  const f16* q = br ? q_r : q_c;
  const f16* kv = br ? kv_r : kv_c;
  f16* qn = br ? qn_r : qn_c;
  f16* kn = br ? kn_r : kn_c;
  f16* vn = br ? vn_r : vn_c;
  f16* vt = br ? vt_r : vt_c;
  f16* xcat = br ? xcat_r : xcat_c;
  f16* kF = br ? kF_r : kF_c;          // fragment-major copies (nullptr: the caller, the backward, does not use them)
  f16* vF = br ? vF_r : vF_c;
  f16* vtF = br ? vtF_r : vtF_c;
  const int KSn = DP >> 4, NTn = DP >> 5, jb = n >> 5;
  for (int h = 0; h < MCA_HEADS; ++h) {
    float qv[2] = {0.f, 0.f}, kvv[2] = {0.f, 0.f}, vv[2] = {0.f, 0.f};
    float sq = 0.f, sk = 0.f, sv = 0.f;
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
      int j = lane + rnd * 64;
      if (j < d && n < N) {
        kvv[rnd] = (float)kv[(long)n * 2 * W + h * d + j];
        vv[rnd] = (float)kv[(long)n * 2 * W + W + h * d + j];
        if (n < N_L) qv[rnd] = (float)q[(long)n * W + h * d + j];
      }
      sq += qv[rnd] * qv[rnd];
      sk += kvv[rnd] * kvv[rnd];
      sv += vv[rnd] * vv[rnd];
    }
    sq = wave_sum(sq); sk = wave_sum(sk); sv = wave_sum(sv);
    // x / ||x|| with no epsilon, like the reference (post_trans.py:623-628)
    float iq = 1.0f / sqrtf(sq), ik = 1.0f / sqrtf(sk), iv = 1.0f / sqrtf(sv);
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
      int j = lane + rnd * 64;
      if (j < DP) {
        bool real = j < d && n < N;
        long o = (long)n * WP + h * DP + j;
        kn[o] = real ? (f16)(kvv[rnd] * ik) : (f16)0.f;
        vn[o] = real ? (f16)(vv[rnd] * iv) : (f16)0.f;
        if (vt) vt[(long)(h * DP + j) * Npad + n] = real ? (f16)vv[rnd] : (f16)0.f;      // row-major V^T: no reader left (the aggregation reads vtF)
        if (kF) {
          const long of = ((((long)jb * MCA_HEADS + h) * KSn + (j >> 4)) * 64 + ((n & 31) | (((j >> 3) & 1) << 5))) * 8 + (j & 7);
          kF[of] = real ? (f16)(kvv[rnd] * ik) : (f16)0.f;
          vF[of] = real ? (f16)(vv[rnd] * iv) : (f16)0.f;
          const long ot = (((((long)jb * MCA_HEADS + h) * NTn + (j >> 5)) * 2 + ((n >> 4) & 1)) * 64 + ((j & 31) | (((n >> 3) & 1) << 5))) * 8 + (n & 7);
          vtF[ot] = real ? (f16)vv[rnd] : (f16)0.f;
        }
        if (n < N_L) {
          qn[o] = real ? (f16)(qv[rnd] * iq) : (f16)0.f;
          if (j < d && xcat) xcat[(long)n * 2 * W + W + h * d + j] = (f16)vv[rnd];   // x_ori = own value rows
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------- pass A
template <int DP, bool F16L>
__global__ __launch_bounds__(256) void mca_stats_kernel(McaParams p, int ntiles) {
  constexpr int KS = DP / 16;
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  int s, tile;
  if (!xcd_remap(blockIdx.x, p.S, ntiles, s, tile)) return;      // class = key split: its key fragments stay in one XCD's L2
  const McaTile t = p.tiles[tile];
  const int n_local = ((t.loc_end + 31) >> 5) - (t.loc_start >> 5);
  const int n_glob = ((p.N + 31) >> 5) - (p.N_L >> 5);
  const int nblk = n_local + n_glob;

  int qrow = t.row0 + (lane & 31);
  if (qrow >= t.row0 + t.nrows) qrow = t.row0;   // clamp (rows beyond the tile are discarded at the end)
  f16x8 qc[KS], qr[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    long o = (long)qrow * p.WP + h * DP + ks * 16 + (lane >> 5) * 8;
    qc[ks] = *(const f16x8*)(p.qn_c + o);
    qr[ks] = *(const f16x8*)(p.qn_r + o);
  }
  float lc[16], lr[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) lc[e] = lr[e] = 0.f;

  for (int j = s; j < nblk; j += p.S) {
    int key0, lo, hi;
    block_range(t, p.N_L, p.N, j, key0, lo, hi, n_local);
    const int key = key0 + (lane & 31);
    f32x16 sc, sr;
#pragma unroll
    for (int e = 0; e < 16; ++e) sc[e] = sr[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const long of = ((((long)(key0 >> 5) * MCA_HEADS + h) * KS + ks) * 64 + lane) * 8;
      f16x8 kc = *(const f16x8*)(p.kF_c + of);
      f16x8 kr = *(const f16x8*)(p.kF_r + of);
      sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qc[ks], kc, sc, 0, 0, 0);
      sr = __builtin_amdgcn_mfma_f32_32x32x16_f16(qr[ks], kr, sr, 0, 0, 0);
    }
    const bool valid = key >= lo && key < hi;
    const float kscore = valid ? (float)p.score[key] : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      lc[e] += valid ? __expf(mca_logit_cls<F16L>(sc[e], kscore, p.scale) - p.scale) : 0.f;
      lr[e] += valid ? __expf(mca_logit_reg<F16L>(sr[e], p.scale) - p.scale) : 0.f;
    }
  }
  // row sums: reduce over the 32 key lanes of each half-wave
#pragma unroll
  for (int e = 0; e < 16; ++e) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      lc[e] += __shfl_xor(lc[e], o, 64);
      lr[e] += __shfl_xor(lr[e], o, 64);
    }
  }
  if ((lane & 31) == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      if (row < t.nrows) {
        long o = (((long)s * MCA_HEADS + h) * 2) * p.N_L + t.row0 + row;
        p.lsum[o] = lc[e];
        p.lsum[o + p.N_L] = lr[e];
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------- pass B
template <int DP, bool NEED_REG, bool F16L>
__global__ __launch_bounds__(256, 2) void mca_aggregate_kernel(McaParams p, int ntiles) {
  constexpr int KS = DP / 16, NT = DP / 32;
  constexpr int PST = 33;   // fp32 tile row stride
  constexpr int HST = 40;   // fp16 tile row stride (80 B: conflict-free b128 rows)
  __shared__ float Pf[MCA_HEADS][32][PST];
  __shared__ float Rc[MCA_HEADS][32][PST];
  __shared__ float Rr[MCA_HEADS][32][PST];
  __shared__ __attribute__((aligned(16))) f16 Ph[MCA_HEADS][32][HST];
  __shared__ __attribute__((aligned(16))) f16 Wc[32][HST];
  __shared__ __attribute__((aligned(16))) f16 Wr[32][HST];
  __shared__ float invl[MCA_HEADS][2][32];
  __shared__ float den[2][32];

  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  // class = key split (as the backward's passes, mca_common.h): the ~24 MB of key-side fragments of a 16 x 500 clip do not fit one
  // XCD's 4 MB L2, the 1/8 a split class touches does -- with (tile, split) ids dealt round-robin every XCD streamed all of them
  int s, tile;
  if (!xcd_remap(blockIdx.x, p.S, ntiles, s, tile)) return;
  const McaTile t = p.tiles[tile];
  const int n_local = ((t.loc_end + 31) >> 5) - (t.loc_start >> 5);
  const int n_glob = ((p.N + 31) >> 5) - (p.N_L >> 5);
  const int nblk = n_local + n_glob;

  // 1/l per (head, branch, row): sum the stats partials of all key splits
  {
    int hh = tid >> 6, rem = tid & 63, br = rem >> 5, row = rem & 31;
    float l = 0.f;
    if (row < t.nrows)
      for (int ss = 0; ss < p.S; ++ss) l += p.lsum[(((long)ss * MCA_HEADS + hh) * 2 + br) * p.N_L + t.row0 + row];
    invl[hh][br][row] = l > 0.f ? 1.0f / l : 0.f;
    if (tid < 64) den[tid >> 5][tid & 31] = 0.f;
  }
  int qrow = t.row0 + (lane & 31);
  if (qrow >= t.row0 + t.nrows) qrow = t.row0;
  // DP >= 64: the local-row v^ fragments are re-read (L1-hot, 2 KB per head) in every key block instead of living in registers --
  // 128 accumulator + 64 fragment + 32 score registers do not fit the 256 of an occupancy-2 wave
  // (DP >= 96, both branches: 192 accumulator registers -- the q fragments are re-read as well)
  constexpr bool VL_RELOAD = NEED_REG ? DP >= 64 : DP >= 96, Q_RELOAD = DP >= 96 && NEED_REG;
  f16x8 qc[Q_RELOAD ? 1 : KS], qr[Q_RELOAD ? 1 : KS], vlc[VL_RELOAD ? 1 : KS], vlr[VL_RELOAD ? 1 : KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    long o = (long)qrow * p.WP + h * DP + ks * 16 + (lane >> 5) * 8;
    if constexpr (!Q_RELOAD) {
      qc[ks] = *(const f16x8*)(p.qn_c + o);
      qr[ks] = *(const f16x8*)(p.qn_r + o);
    }
    if constexpr (!VL_RELOAD) {
      vlc[ks] = *(const f16x8*)(p.vn_c + o);   // local rows of v^ are rows [0, N_L) of vn
      vlr[ks] = *(const f16x8*)(p.vn_r + o);
    }
  }
  f32x16 oc[NT], orr[NT], sfc[NT], sfr[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) oc[n][e] = orr[n][e] = sfc[n][e] = sfr[n][e] = 0.f;
  __syncthreads();

  for (int j = s; j < nblk; j += p.S) {
    int key0, lo, hi;
    block_range(t, p.N_L, p.N, j, key0, lo, hi, n_local);
    const int key = key0 + (lane & 31);
    const bool valid = key >= lo && key < hi;
    const float kscore = valid ? (float)p.score[key] : 0.f;
    const int col = lane & 31;
    // two passes (scores, then raw similarities) so that only two 32x32 accumulators are live at a time: with all four the
    // DP = 64 instantiation spilled its q / v fragments to scratch and re-read them inside this loop
    {
      f32x16 sc, sr;
#pragma unroll
      for (int e = 0; e < 16; ++e) sc[e] = sr[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const long of = ((((long)(key0 >> 5) * MCA_HEADS + h) * KS + ks) * 64 + lane) * 8;
        f16x8 kc = *(const f16x8*)(p.kF_c + of);
        f16x8 kr = *(const f16x8*)(p.kF_r + of);
        if constexpr (Q_RELOAD) {
          const long oq = (long)qrow * p.WP + h * DP + ks * 16 + (lane >> 5) * 8;
          sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.qn_c + oq), kc, sc, 0, 0, 0);
          sr = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.qn_r + oq), kr, sr, 0, 0, 0);
        } else {
          sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qc[ks], kc, sc, 0, 0, 0);
          sr = __builtin_amdgcn_mfma_f32_32x32x16_f16(qr[ks], kr, sr, 0, 0, 0);
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        // (the LDS reads stand OUTSIDE the select: a load under the condition is not speculated, and the 32 selects of a block became
        // 32 exec-mask branch pairs -- 11.5 M VALU + 3 M SALU instructions per launch next to 0.85 M MFMAs, rocprofv3 SQ_INSTS_*)
        const float il_c = invl[h][0][row], il_r = invl[h][1][row];
        const float ec = __expf(mca_logit_cls<F16L>(sc[e], kscore, p.scale) - p.scale) * il_c;
        const float er = __expf(mca_logit_reg<F16L>(sr[e], p.scale) - p.scale) * il_r;
        float pc = valid ? ec : 0.f;
        float pr = valid ? er : 0.f;
        float pp = 0.5f * (pc + pr);
        Pf[h][row][col] = pp;
        Ph[h][row][col] = (f16)pp;
      }
    }
    {
      f32x16 rc, rr;
#pragma unroll
      for (int e = 0; e < 16; ++e) rc[e] = rr[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const long of = ((((long)(key0 >> 5) * MCA_HEADS + h) * KS + ks) * 64 + lane) * 8;
        f16x8 vc = *(const f16x8*)(p.vF_c + of);
        f16x8 vr = *(const f16x8*)(p.vF_r + of);
        if constexpr (VL_RELOAD) {
          const long ol = (long)qrow * p.WP + h * DP + ks * 16 + (lane >> 5) * 8;
          rc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.vn_c + ol), vc, rc, 0, 0, 0);
          rr = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.vn_r + ol), vr, rr, 0, 0, 0);
        } else {
          rc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlc[ks], vc, rc, 0, 0, 0);
          rr = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlr[ks], vr, rr, 0, 0, 0);
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        Rc[h][row][col] = valid ? rc[e] : -2.f;   // invalid keys can never pass the similarity masks
        Rr[h][row][col] = valid ? rr[e] : -2.f;
      }
    }
    __syncthreads();
    {
      // all 256 threads: head-mean of P and of the raw similarities -> aggregation weights for 4 (row, key) cells
      const int row = tid >> 3, c0 = (tid & 7) * 4;
      float dc = 0.f, dr = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float mp = 0.25f * (Pf[0][row][c0 + c] + Pf[1][row][c0 + c] + Pf[2][row][c0 + c] + Pf[3][row][c0 + c]);
        float mc = 0.25f * (Rc[0][row][c0 + c] + Rc[1][row][c0 + c] + Rc[2][row][c0 + c] + Rc[3][row][c0 + c]);
        float mr = 0.25f * (Rr[0][row][c0 + c] + Rr[1][row][c0 + c] + Rr[2][row][c0 + c] + Rr[3][row][c0 + c]);
        float e = __expf(mp);
        float wc = mc > p.sim_thresh ? e : 0.f;
        float wr = mr > p.conf_thresh ? e : 0.f;
        Wc[row][c0 + c] = (f16)wc;
        Wr[row][c0 + c] = (f16)wr;
        dc += wc;
        dr += wr;
      }
      // 8 consecutive lanes share a row
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) {
        dc += __shfl_xor(dc, o, 64);
        dr += __shfl_xor(dr, o, 64);
      }
      if ((tid & 7) == 0) {
        den[0][row] += dc;
        den[1][row] += dr;
      }
    }
    __syncthreads();
    // aggregations: [32 rows x 32 keys] @ V^T-stored values of this head
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ks * 16 + (lane >> 5) * 8;
      f16x8 ap = *(const f16x8*)(&Ph[h][lane & 31][ko]);
      f16x8 awc = *(const f16x8*)(&Wc[lane & 31][ko]);
      f16x8 awr = *(const f16x8*)(&Wr[lane & 31][ko]);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const long o = (((((long)(key0 >> 5) * MCA_HEADS + h) * NT + n) * 2 + ks) * 64 + lane) * 8;
        f16x8 bvc = *(const f16x8*)(p.vtF_c + o);
        oc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap, bvc, oc[n], 0, 0, 0);
        sfc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(awc, bvc, sfc[n], 0, 0, 0);
        if (NEED_REG) {
          f16x8 bvr = *(const f16x8*)(p.vtF_r + o);
          orr[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap, bvr, orr[n], 0, 0, 0);
          sfr[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(awr, bvr, sfr[n], 0, 0, 0);
        }
      }
    }
    // next iteration's Pf/Ph/R writes are ordered behind this iteration's reads by the two barriers above
  }
  __syncthreads();
  // write the split's partial accumulators
#pragma unroll
  for (int n = 0; n < NT; ++n) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      if (row < t.nrows) {
        long o = ((long)s * p.N_L + t.row0 + row) * p.WP + h * DP + n * 32 + (lane & 31);
        p.part_oc[o] = oc[n][e];
        p.part_sc[o] = sfc[n][e];
        if (NEED_REG) {
          p.part_or[o] = orr[n][e];
          p.part_sr[o] = sfr[n][e];
        }
      }
    }
  }
  if (tid < 64) {
    int br = tid >> 5, row = tid & 31;
    if (row < t.nrows) p.part_den[((long)s * 2 + br) * p.N_L + t.row0 + row] = den[br][row];
  }
}

// -------------------------------------------------------------------------------------------------- finalize
__global__ void mca_finalize_kernel(const float* __restrict__ part_oc, const float* __restrict__ part_or,
                                    const float* __restrict__ part_sc, const float* __restrict__ part_sr,
                                    const float* __restrict__ part_den, const float* __restrict__ lsum, float* __restrict__ stats_out,
                                    int S, int N_L, int W, int d, int DP, int need_reg, f16* xcat_c, f16* xcat_r, f16* cat3_c,
                                    f16* cat3_r) {
  // stats_out (optional, for backward): [HEADS][2][N_L] softmax row sums, then [2][N_L] sim_round2 denominators
  if (stats_out) {
    long nst = (long)MCA_HEADS * 2 * N_L + 2L * N_L;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nst; i += (long)gridDim.x * blockDim.x) {
      float a = 0.f;
      if (i < (long)MCA_HEADS * 2 * N_L) {
        for (int s = 0; s < S; ++s) a += lsum[(long)s * MCA_HEADS * 2 * N_L + i];
      } else {
        long j = i - (long)MCA_HEADS * 2 * N_L;
        for (int s = 0; s < S; ++s) a += part_den[(long)s * 2 * N_L + j];
      }
      stats_out[i] = a;
    }
  }
  const int WP = MCA_HEADS * DP;
  long total = (long)N_L * W;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int r = idx / W, c = idx % W;
    int pc = (c / d) * DP + (c % d);
    float o_c = 0.f, s_c = 0.f, o_r = 0.f, s_r = 0.f, d_c = 0.f, d_r = 0.f;
    for (int s = 0; s < S; ++s) {
      long o = ((long)s * N_L + r) * WP + pc;
      o_c += part_oc[o];
      s_c += part_sc[o];
      d_c += part_den[((long)s * 2 + 0) * N_L + r];
      if (need_reg) {
        o_r += part_or[o];
        s_r += part_sr[o];
        d_r += part_den[((long)s * 2 + 1) * N_L + r];
      }
    }
    xcat_c[(long)r * 2 * W + c] = (f16)o_c;
    cat3_c[(long)r * 3 * W + c] = (f16)(s_c / d_c);
    if (need_reg) {
      xcat_r[(long)r * 2 * W + c] = (f16)o_r;
      cat3_r[(long)r * 3 * W + c] = (f16)(s_r / d_r);
    }
  }
}

// host launcher of the prep kernel, shared with mca_bwd.hip (kernels cannot be launched across translation units)
void mca_launch_prep(hipStream_t st, const f16* q_c, const f16* kv_c, const f16* q_r, const f16* kv_r, int N, int N_L, int Npad, int W,
                     int d, int DP, f16* qn_c, f16* kn_c, f16* vn_c, f16* vt_c, f16* qn_r, f16* kn_r, f16* vn_r, f16* vt_r, f16* xcat_c,
                     f16* xcat_r) {
  // the backward keeps the row-major layouts only (its own layout kernel re-lays them)
  long waves = 2L * Npad;
  hipLaunchKernelGGL(mca_prep_kernel, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, st, q_c, kv_c, q_r, kv_r, N, N_L, Npad,
                     W, d, DP, qn_c, kn_c, vn_c, vt_c, qn_r, kn_r, vn_r, vt_r, xcat_c, xcat_r, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

// -------------------------------------------------------------------------------------------------- C-ABI
extern "C" int tscd_mca_dp(int d) { return (d + 31) / 32 * 32; }

// Workspace sizes (bytes) so the caller (torch) owns every allocation.
// The normalised rows (the part of the forward's state the backward reads: tscd_mca_backward_fws) may live in a buffer of their own
// (rows_workspace, tscd_mca_rows_bytes): the caller then keeps only that small buffer alive until the backward and frees the scratch
// (the key-split partial sums are ~10x larger).  Without one they are carved from the front of `workspace`.
extern "C" long tscd_mca_rows_bytes(int N, int N_L, int W) {
  const int WP = MCA_HEADS * tscd_mca_dp(W / MCA_HEADS);
  return mca_rows_bytes(N, N_L, WP);
}
extern "C" long tscd_mca_workspace_bytes(int N, int N_L, int W, int S) {
  int d = W / MCA_HEADS, DP = tscd_mca_dp(d), WP = MCA_HEADS * DP;
  long Npad = (N + 31) / 32 * 32 + 32;
  long halves = 2L * WP * Npad + 6L * Npad * WP;                                          // V^T (unused slot) + the fragment-major copies
  long floats = (long)S * MCA_HEADS * 2 * N_L + 4L * S * N_L * WP + 2L * S * N_L;
  return mca_rows_bytes(N, N_L, WP) + halves * 2 + floats * 4 + 8192;
}

extern "C" int tscd_mca_forward(const void* q_c, const void* kv_c, const void* q_r, const void* kv_r, const void* score,
                                const int* tiles, int ntiles, int N, int N_L, int W, int S, int need_reg, float scale,
                                float sim_thresh, float conf_thresh, void* workspace, long workspace_bytes, void* xcat_c,
                                void* xcat_r, void* cat3_c, void* cat3_r, float* stats_out, void* rows_workspace, int fp16_logits,
                                void* stream) {
  TSCD_CHECK_ARG(q_c && kv_c && q_r && kv_r && score && tiles && workspace && xcat_c && cat3_c, "mca_forward: null pointer");
  TSCD_CHECK_ARG(!need_reg || (xcat_r && cat3_r), "mca_forward: reg outputs requested without buffers");
  TSCD_CHECK_ARG(W % MCA_HEADS == 0 && W % 8 == 0, "mca_forward: W=%d must be a multiple of 8 (4 heads)", W);
  TSCD_CHECK_ARG(N_L > 0 && N >= N_L && ntiles > 0 && S > 0 && S <= 64, "mca_forward: bad sizes N=%d N_L=%d S=%d", N, N_L, S);
  const int d = W / MCA_HEADS, DP = tscd_mca_dp(d), WP = MCA_HEADS * DP;
  TSCD_CHECK_ARG(DP <= 128, "mca_forward: head dim %d not supported (max 128)", d);
  TSCD_CHECK_ARG(workspace_bytes >= tscd_mca_workspace_bytes(N, N_L, W, S), "mca_forward: workspace too small");
  const int Npad = (N + 31) / 32 * 32 + 32;
  hipStream_t st = (hipStream_t)stream;

  char* ws = (char*)workspace;
  auto take = [&](long bytes) { char* r = ws; ws += (bytes + 255) / 256 * 256; return r; };
  // (the 256-B rounding is covered by the +8192 slack for <= 20 carve-outs)
  const McaRows rows = mca_rows_carve(rows_workspace ? rows_workspace : take(mca_rows_bytes(N, N_L, WP)), N, N_L, WP);
  f16 *qn_c = rows.qn_c, *qn_r = rows.qn_r, *kn_c = rows.kn_c, *kn_r = rows.kn_r, *vn_c = rows.vn_c, *vn_r = rows.vn_r;
  f16* vt_c = (f16*)take(2L * WP * Npad);
  f16* vt_r = (f16*)take(2L * WP * Npad);
  f16* kF_c = (f16*)take(2L * Npad * WP);
  f16* kF_r = (f16*)take(2L * Npad * WP);
  f16* vF_c = (f16*)take(2L * Npad * WP);
  f16* vF_r = (f16*)take(2L * Npad * WP);
  f16* vtF_c = (f16*)take(2L * Npad * WP);
  f16* vtF_r = (f16*)take(2L * Npad * WP);
  float* lsum = (float*)take(4L * S * MCA_HEADS * 2 * N_L);
  float* part_oc = (float*)take(4L * S * N_L * WP);
  float* part_or = (float*)take(4L * S * N_L * WP);
  float* part_sc = (float*)take(4L * S * N_L * WP);
  float* part_sr = (float*)take(4L * S * N_L * WP);
  float* part_den = (float*)take(4L * S * 2 * N_L);
  TSCD_CHECK_ARG(ws - (char*)workspace <= workspace_bytes, "mca_forward: workspace carve-out overflow");

  {
    long waves = 2L * Npad;
    hipLaunchKernelGGL(mca_prep_kernel, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, st, (const f16*)q_c,
                       (const f16*)kv_c, (const f16*)q_r, (const f16*)kv_r, N, N_L, Npad, W, d, DP, qn_c, kn_c, vn_c, (f16*)nullptr,
                       qn_r, kn_r, vn_r, (f16*)nullptr, (f16*)xcat_c, (f16*)(need_reg ? xcat_r : nullptr), kF_c, kF_r, vF_c, vF_r, vtF_c, vtF_r);
    TSCD_CHECK_LAUNCH("mca_prep");
  }
  McaParams p;
  p.qn_c = qn_c; p.qn_r = qn_r; p.kn_c = kn_c; p.kn_r = kn_r; p.vn_c = vn_c; p.vn_r = vn_r; p.vt_c = vt_c; p.vt_r = vt_r;
  p.kF_c = kF_c; p.kF_r = kF_r; p.vF_c = vF_c; p.vF_r = vF_r; p.vtF_c = vtF_c; p.vtF_r = vtF_r;
  p.score = (const f16*)score; p.tiles = (const McaTile*)tiles; p.lsum = lsum;
  p.part_oc = part_oc; p.part_or = part_or; p.part_sc = part_sc; p.part_sr = part_sr; p.part_den = part_den;
  p.N = N; p.N_L = N_L; p.Npad = Npad; p.WP = WP; p.S = S;
  p.scale = scale; p.sim_thresh = sim_thresh; p.conf_thresh = conf_thresh;
  dim3 grid(xcd_grid(S, ntiles)), block(256);
#define MCA_LAUNCH_L(DPV, FL)                                                                        \
  do {                                                                                              \
    hipLaunchKernelGGL((mca_stats_kernel<DPV, FL>), grid, block, 0, st, p, ntiles);                         \
    if (need_reg)                                                                                   \
      hipLaunchKernelGGL((mca_aggregate_kernel<DPV, true, FL>), grid, block, 0, st, p, ntiles);             \
    else                                                                                            \
      hipLaunchKernelGGL((mca_aggregate_kernel<DPV, false, FL>), grid, block, 0, st, p, ntiles);            \
  } while (0)
#define MCA_LAUNCH(DPV)                                                                             \
  do {                                                                                              \
    if (fp16_logits) MCA_LAUNCH_L(DPV, true);                                                       \
    else MCA_LAUNCH_L(DPV, false);                                                                  \
  } while (0)
  if (DP == 32) MCA_LAUNCH(32);
  else if (DP == 64) MCA_LAUNCH(64);
  else if (DP == 96) MCA_LAUNCH(96);
  else MCA_LAUNCH(128);
#undef MCA_LAUNCH
#undef MCA_LAUNCH_L
  TSCD_CHECK_LAUNCH("mca_stats/aggregate");
  long total = (long)N_L * W;
  int fg = (int)((total + 255) / 256);
  hipLaunchKernelGGL(mca_finalize_kernel, dim3(fg > 2048 ? 2048 : fg), dim3(256), 0, st, part_oc, part_or, part_sc, part_sr,
                     part_den, lsum, stats_out, S, N_L, W, d, DP, need_reg, (f16*)xcat_c, (f16*)xcat_r, (f16*)cat3_c,
                     (f16*)cat3_r);
  TSCD_CHECK_LAUNCH("mca_finalize");
  return TSCD_OK;
}
