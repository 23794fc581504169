// MCA backward: gradient of the fused global-to-local cosine attention of mca.hip w.r.t. the projected q / k / v rows
// and the key confidence scores (reference: autograd of Attention_mca_g2l.forward, yolox/models/post_trans.py:601-714,
// inside MCA_tscd_g2l_reg.forward :1127-1162; trained through tscd_trainer.py:171-176).
//
// Forward (per head h, query row r, key k; see mca.hip):
//   Pc = softmax_k(25*score_k*qc^.kc^)   Pr = softmax_k(25*qr^.kr^)   P = (Pc+Pr)/2
//   Oc = P vc   Or = P vr             M = mean_h P   E = exp(M)   Ac = [Rc>.75] E / Dc   Ar = [Rr>.99] E / Dr
//   softc = Ac Vc(all heads)   softr = Ar Vr            (the R masks are constants: no gradient through v^)
// Backward, given dOc, dOr, dsoftc, dsoftr:
//   dA_c[r,k] = dsoftc_r . Vc_k (all heads)        dM = Ac (dA_c - dsoftc_r.softc_r) + Ar (dA_r - dsoftr_r.softr_r)
//   dP_h = dOc_h . vc_h + dOr_h . vr_h + dM/4
//   delta_c[h,r] = sum_k Pc dP/2, delta_r[h,r] = sum_k Pr dP/2
//   dSc = Pc (dP/2 - delta_c), dSr = Pr (dP/2 - delta_r)
//   dq^c = (25 score dSc) k^c        dk^c = (25 score dSc)^T q^c        dscore = sum_{h,r} 25 dSc (q^c.k^c)
//   dq^r = (25 dSr) k^r              dk^r = (25 dSr)^T q^r
//   dvc  = P^T dOc + Ac^T dsoftc     dvr  = P^T dOr + Ar^T dsoftr
//
// Three passes over the (32-query tile, 32-key block) pairs, none with atomics:
//   1. mca_bwd_delta_kernel  workgroup = query tile x 4 heads (one wave per head).  The only pass that couples the heads
//      (M, A and dA are sums over heads, exchanged through LDS): it accumulates delta and writes what the other passes need
//      of that coupling -- dM, Ac, Ar, 12 KB per pair (~100 MB per call at 16 x 500 proposals: far less than what HBM moves
//      in the time the pair's MFMAs take).
//   2. mca_bwd_dq_kernel     wave = (query tile, head, split of the key blocks).  S^T = K Q^T puts the query on the lane
//      axis, so dS^T leaves the MFMA already in A-operand order for dq = dS K: no transpose, no LDS, no barrier.
//   3. mca_bwd_dkv_kernel    wave = (key block, head, split of the query tiles).  S = Q K^T puts the key on the lane axis;
//      P, dS, Ac, Ar are A operands of dk = dS^T Q, dv = P^T dO + A^T dsoft as they leave the MFMA; dk / dv / dscore stay in
//      registers over all query tiles that see the block.
// The contraction index of the second products is the C-register order of the first (row 8i + 4*half + j <-> register 4i + j),
// and their B operands are prepared in that order.  Splits write separate partial sums which mca_bwd_finish_kernel adds in a
// fixed order: results are deterministic and nothing has to be zeroed.
//
// Operand layout: every MFMA operand is stored FRAGMENT-MAJOR -- the 64 x 16 bytes one wave loads for one MFMA are 1 KB
// contiguous -- by two layout kernels per call (keys: mca_bwd_layout_kernel, queries + upstream gradients per tile:
// mca_bwd_tiles_kernel).  Row-major operands cost 32 cache lines per wave load (a 16-byte piece of 32 different rows), the
// fragment-major ones 8; the passes are bound by exactly that (texture-addresser / L2 request rate), not by MFMA.
//   row fragment  F(rb, h, ks)[lane]   = X[32 rb + (lane & 31)][h DP + 16 ks + 8 (lane >> 5) .. + 7]
//   col fragment  T(rb, h, n, m)[lane] = X[32 rb + 16 m + 4 (lane >> 5) + {0,1,2,3,8,9,10,11}][h DP + 32 n + (lane & 31)]
// (rb = key block, or query tile with rows beyond the tile zero).
#include "mca_common.h"
#include <type_traits>

struct McaBwdParams {
  // fragment-major operands (see above); *F = row fragments, *T = col fragments
  const f16 *qF_c, *qF_r, *vlF_c, *vlF_r, *gOF_c, *gOF_r, *gSF_c, *gSF_r;   // per query tile
  const f16 *qT_c, *qT_r, *gOT_c, *gOT_r, *gST_c, *gST_r;                   // per query tile
  const f16 *kF_c, *kF_r, *vnF_c, *vnF_r, *vrF_c, *vrF_r;                   // per key block (k^, v^, raw v)
  const f16 *kT_c, *kT_r;                                                   // per key block
  const McaTile* tiles;
  const float* stats;        // [HEADS][2][N_L] softmax row sums, then [2][N_L] sim_round2 denominators
  const float* gdot;         // [2][N_L]: dsoftc.softc, dsoftr.softr
  const float* score_pad;    // [Npad] fp32 (0 beyond N)
  float* delta_part;         // [S][HEADS][2][N_L]
  float* rowstat;            // [tile][HEADS][256]: 1/l_c, 1/l_r, delta_c, delta_r x 32 rows (0 on padded rows), rest unused
  // per (query tile, key block) 32 x 32 blocks, register-major: [tile][maxblk][4 i][64 lanes][4 j]
  float *dMq, *dMk;          // dM / 8 with the query / the key on the lane axis
  f16 *Ack, *Ark;            // sim_round2 weights, key on the lane axis
  float *dqn_c, *dqn_r;      // [S2][N_L][WP]
  float *dkn_c, *dkn_r, *dv_c, *dv_r;   // [S3][Npad][WP]
  float* dscore_part;        // [S3 * HEADS][Npad]
  int N, N_L, Npad, WP, S, maxblk, S2, S3;
  long dq_stride, dk_stride; // floats between the partial sums of consecutive splits
  float scale, sim_thresh, conf_thresh;
};

template <int DP>
__device__ __forceinline__ long rowfrag(int rb, int h, int ks) { return ((((long)rb * MCA_HEADS + h) * (DP / 16) + ks) << 9); }
template <int DP>
__device__ __forceinline__ long colfrag(int rb, int h, int n, int m) { return ((((((long)rb * MCA_HEADS + h) * (DP / 32) + n) << 1) + m) << 9); }

// ------------------------------------------------------------------------------------------------ operand layouts
// Key side: one workgroup per (32-key block, array).  Sources: k^ / v^ rows of the forward prep (padded row-major
// [Npad][WP]) and the raw value rows kv[:, W:2W].  Arrays 0..5 = row fragments of k^c k^r v^c v^r vc vr, 6..7 = col
// fragments of k^c k^r; block (0, 0) also pads the scores.
__global__ __launch_bounds__(256) void mca_bwd_layout_kernel(const f16* __restrict__ kn_c, const f16* __restrict__ kn_r,
                                                             const f16* __restrict__ vn_c, const f16* __restrict__ vn_r,
                                                             const f16* __restrict__ kv_c, const f16* __restrict__ kv_r, int N,
                                                             int Npad, int W, int d, int DP, f16* kF_c, f16* kF_r, f16* vnF_c,
                                                             f16* vnF_r, f16* vrF_c, f16* vrF_r, f16* kT_c, f16* kT_r,
                                                             const f16* __restrict__ score, float* score_pad, float scale) {
  const int b = blockIdx.x, arr = blockIdx.y, tid = threadIdx.x;
  const int WP = MCA_HEADS * DP, KS = DP / 16, NT = DP / 32;
  if (arr == 0 && tid < 32) score_pad[b * 32 + tid] = b * 32 + tid < N ? (float)score[b * 32 + tid] : 0.f;
  const int nchunk = MCA_HEADS * KS * 64;     // = HEADS * NT * 2 * 64
  if (arr < 6) {
    const f16* src = arr == 0 ? kn_c : arr == 1 ? kn_r : arr == 2 ? vn_c : arr == 3 ? vn_r : nullptr;
    f16* dst = arr == 0 ? kF_c : arr == 1 ? kF_r : arr == 2 ? vnF_c : arr == 3 ? vnF_r : arr == 4 ? vrF_c : vrF_r;
    const f16* raw = arr == 4 ? kv_c : kv_r;
    for (int c = tid; c < nchunk; c += 256) {
      const int l = c & 63, ks = (c >> 6) % KS, h = (c >> 6) / KS;
      const int row = b * 32 + (l & 31), j0 = ks * 16 + (l >> 5) * 8;
      f16x8 v;
      if (src) {
        v = *(const f16x8*)(src + (long)row * WP + h * DP + j0);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (row < N && j0 + e < d) ? raw[(long)row * 2 * W + W + h * d + j0 + e] : (f16)0.f;
      }
      *(f16x8*)(dst + ((long)b * nchunk + c) * 8) = v;
    }
  } else {
    const f16* src = arr == 6 ? kn_c : kn_r;
    f16* dst = arr == 6 ? kT_c : kT_r;
    for (int c = tid; c < nchunk; c += 256) {
      const int l = c & 63, m = (c >> 6) & 1, n = ((c >> 7) % NT), h = (c >> 7) / NT;
      const int ch = h * DP + n * 32 + (l & 31), r0 = b * 32 + 16 * m + 4 * (l >> 5);
      f16x8 v;
      // the factor d(logit)/d(q^.k^) of the key row is folded in: 25 * score_k (cls), 25 (reg)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = r0 + (e & 3) + 8 * (e >> 2);
        const float f = arr == 6 ? (row < N ? scale * (float)score[row] : 0.f) : scale;
        v[e] = (f16)(f * (float)src[(long)row * WP + ch]);
      }
      *(f16x8*)(dst + ((long)b * nchunk + c) * 8) = v;
    }
  }
}

// Query side: one workgroup per (query tile, array).  Arrays 0..7 = row fragments of q^c q^r v^c v^r dOc dOr dsoftc dsoftr,
// 8..13 = col fragments of q^c q^r dOc dOr dsoftc dsoftr (rows beyond the tile are zero); array 14 = gdot = dsoft . soft.
// cls_grad = 0 (dO_c = dS_c = null: no gradient reaches the class branch's outputs): the arrays only the zeros would fill -- v^c,
// dOc, dsoftc in either form -- are not written (the CLS_G = false passes do not read them) and gdot_c = 0.
__global__ __launch_bounds__(256) void mca_bwd_tiles_kernel(const McaTile* __restrict__ tiles, const f16* __restrict__ qn_c,
                                                            const f16* __restrict__ qn_r, const f16* __restrict__ vn_c,
                                                            const f16* __restrict__ vn_r, const f16* __restrict__ dO_c, int ld_dO,
                                                            const f16* __restrict__ dO_r, const f16* __restrict__ dS_c, int ld_dS,
                                                            const f16* __restrict__ dS_r, const f16* __restrict__ soft_c,
                                                            const f16* __restrict__ soft_r, int ld_soft, int N_L, int W, int d,
                                                            int DP, int need_reg, int cls_grad, McaBwdParams p, float* gdot) {
  const int t = blockIdx.x, arr = blockIdx.y, tid = threadIdx.x;
  if (!cls_grad && (arr == 2 || arr == 4 || arr == 6 || arr == 10 || arr == 12)) return;
  const McaTile tl = tiles[t];
  const int WP = MCA_HEADS * DP, KS = DP / 16, NT = DP / 32;
  const int nchunk = MCA_HEADS * KS * 64;
  if (arr == 14) {          // one wave per row
    const int wv = tid >> 6, lane = tid & 63;
    for (int j = wv; j < tl.nrows; j += 4) {
      const long row = tl.row0 + j;
      float gc = 0.f, gr = 0.f;
      for (int c = lane; c < W; c += 64) {
        if (cls_grad) gc += (float)dS_c[row * ld_dS + c] * (float)soft_c[row * ld_soft + c];
        if (need_reg) gr += (float)dS_r[row * ld_dS + c] * (float)soft_r[row * ld_soft + c];
      }
      gc = wave_sum(gc); gr = wave_sum(gr);
      if (lane == 0) { gdot[row] = gc; gdot[N_L + row] = gr; }
    }
    return;
  }
  const bool col = arr >= 8;
  const int a = col ? arr - 8 + 2 * (arr >= 10) : arr;     // source 0 q^c 1 q^r 2 v^c 3 v^r 4 dOc 5 dOr 6 dsoftc 7 dsoftr
  const bool is_reg_grad = a == 5 || a == 7;
  const f16* padded = a == 0 ? qn_c : a == 1 ? qn_r : a == 2 ? vn_c : a == 3 ? vn_r : nullptr;     // [rows][WP]
  const f16* raw = a == 4 ? dO_c : a == 5 ? dO_r : a == 6 ? dS_c : dS_r;                             // [rows][ld], d per head
  const int ld = (a == 4 || a == 5) ? ld_dO : ld_dS;
  f16* dst = const_cast<f16*>(arr == 0 ? p.qF_c : arr == 1 ? p.qF_r : arr == 2 ? p.vlF_c : arr == 3 ? p.vlF_r : arr == 4 ? p.gOF_c
                              : arr == 5 ? p.gOF_r : arr == 6 ? p.gSF_c : arr == 7 ? p.gSF_r : arr == 8 ? p.qT_c : arr == 9 ? p.qT_r
                              : arr == 10 ? p.gOT_c : arr == 11 ? p.gOT_r : arr == 12 ? p.gST_c : p.gST_r);
  const bool zero_all = is_reg_grad && !need_reg;
  const bool halve = a == 4 || a == 5;     // dO enters only as dO/2 (dP/2 = dO.v/2 + dM/8; dv = (Pc + Pr)^T dO/2)
  auto fetch = [&](int j, int c) -> f16 {      // row j of the tile, padded column c
    if (j >= tl.nrows || zero_all) return (f16)0.f;
    const long row = tl.row0 + j;
    if (padded) return padded[row * WP + c];
    const int h = c / DP, jj = c % DP;
    const f16 v = jj < d ? raw[row * ld + h * d + jj] : (f16)0.f;
    return halve ? (f16)(0.5f * (float)v) : v;
  };
  for (int c = tid; c < nchunk; c += 256) {
    const int l = c & 63;
    f16x8 v;
    if (!col) {
      const int ks = (c >> 6) % KS, h = (c >> 6) / KS;
      const int j = l & 31, c0 = h * DP + ks * 16 + (l >> 5) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fetch(j, c0 + e);
    } else {
      const int m = (c >> 6) & 1, n = (c >> 7) % NT, h = (c >> 7) / NT;
      const int ch = h * DP + n * 32 + (l & 31), j0 = 16 * m + 4 * (l >> 5);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fetch(j0 + (e & 3) + 8 * (e >> 2), ch);
    }
    *(f16x8*)(dst + ((long)t * nchunk + c) * 8) = v;
  }
}

// ------------------------------------------------------------------------------------------------ pass 1: delta, dM, Ac, Ar
// CLS_G = false: no gradient reaches the class branch's attention outputs (dOc = dsoftc = 0 -- agg_iou, whose class output is only
// used detached as assignment cost).  Everything that only carries those zeros is left out: the v^c.v^c similarity
// (its mask only gates Ac), dOc.vc, dsoftc.Vc, their fragments and Ac -- 20 of 32 MFMAs per visit remain, the query-side fragments fit
// the register file again (the full form spills 48 VGPRs at DP 64), and the result is bit for bit what the zeros would have given.
// With ONE branch carrying sim_round2 weights (CLS_G != NEED_REG) its similarity and its dA travel in the exchange's 2nd and 3rd
// array at once: 3 barriers per visit instead of 5.
template <int DP, bool NEED_REG, bool CLS_G>
__global__ __launch_bounds__(256, 2) void mca_bwd_delta_kernel(McaBwdParams p, int ntiles) {
  static_assert(NEED_REG || CLS_G, "mca_bwd_delta_kernel: no upstream gradient at all");
  constexpr int KS = DP / 16;
  constexpr int PST = 33;
  constexpr bool BOTH = NEED_REG && CLS_G;            // both branches carry sim_round2 weights: two exchange rounds
  constexpr int BR = NEED_REG ? 1 : 0;                // the one branch that does, otherwise
  __shared__ float X[3][MCA_HEADS][32][PST];          // cross-head exchange (BOTH: round 1 P, Rc, Rr; round 2 dM, dAc, dAr.  else P, R, dA)
  __shared__ float A32[2][32][PST];                   // Ac, Ar
  __shared__ float invl[MCA_HEADS][2][32];
  __shared__ float dinv[2][32];
  __shared__ float gdt[2][32];

  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  int s, tile;
  if (!xcd_remap(blockIdx.x, p.S, ntiles, s, tile)) return;      // class = key split: its key fragments stay in one L2
  const McaTile t = p.tiles[tile];
  const int n_local = ((t.loc_end + 31) >> 5) - (t.loc_start >> 5);
  const int n_glob = ((p.N + 31) >> 5) - (p.N_L >> 5);
  const int nblk = n_local + n_glob;
  {
    const int rem = tid & 63, br = rem >> 5, row = rem & 31;
    const bool ok = row < t.nrows;
    const float l = ok ? p.stats[((long)h * 2 + br) * p.N_L + t.row0 + row] : 0.f;
    invl[h][br][row] = l > 0.f ? 1.0f / l : 0.f;
    if (tid < 64) {
      const float dn = ok ? p.stats[(long)MCA_HEADS * 2 * p.N_L + (long)br * p.N_L + t.row0 + row] : 0.f;
      dinv[br][row] = dn > 0.f ? 1.0f / dn : 0.f;
      gdt[br][row] = ok ? p.gdot[(long)br * p.N_L + t.row0 + row] : 0.f;
    }
  }
  f16x8 qc[KS], qr[KS], vlc[KS], vlr[KS], gOc[KS], gOr[KS], gSc[KS], gSr[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const long o = rowfrag<DP>(tile, h, ks) + lane * 8;
    qc[ks] = *(const f16x8*)(p.qF_c + o);
    qr[ks] = *(const f16x8*)(p.qF_r + o);
    if (CLS_G) {
      vlc[ks] = *(const f16x8*)(p.vlF_c + o);
      gOc[ks] = *(const f16x8*)(p.gOF_c + o);
      gSc[ks] = *(const f16x8*)(p.gSF_c + o);
    }
    if (NEED_REG) {
      vlr[ks] = *(const f16x8*)(p.vlF_r + o);
      gOr[ks] = *(const f16x8*)(p.gOF_r + o);
      gSr[ks] = *(const f16x8*)(p.gSF_r + o);
    }
  }
  float dl_c[16], dl_r[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) dl_c[e] = dl_r[e] = 0.f;
  __syncthreads();

  const int col = lane & 31, half = lane >> 5;
  const float sc2 = p.scale * 1.4426950408889634f;      // logits in log2 units: v_exp_f32 is 2^x
  for (int j = s; j < nblk; j += p.S) {
    int key0, lo, hi;
    block_range(t, p.N_L, p.N, j, key0, lo, hi, n_local);
    const int key = key0 + col, kb = key0 >> 5;
    const bool valid = key >= lo && key < hi;
    const float a_c = sc2 * p.score_pad[key], bias = valid ? sc2 : __builtin_inff();     // exp2(-inf) = 0 masks the key
    f32x16 sc, sr, rc, rr, gv, gac, gar;
#pragma unroll
    for (int e = 0; e < 16; ++e) sc[e] = sr[e] = rc[e] = rr[e] = gv[e] = gac[e] = gar[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const long o = rowfrag<DP>(kb, h, ks) + lane * 8;
      const f16x8 kc = *(const f16x8*)(p.kF_c + o), kr = *(const f16x8*)(p.kF_r + o);
      sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qc[ks], kc, sc, 0, 0, 0);
      sr = __builtin_amdgcn_mfma_f32_32x32x16_f16(qr[ks], kr, sr, 0, 0, 0);
      if (CLS_G) {
        const f16x8 vnc = *(const f16x8*)(p.vnF_c + o), vrc = *(const f16x8*)(p.vrF_c + o);
        rc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlc[ks], vnc, rc, 0, 0, 0);
        gv = __builtin_amdgcn_mfma_f32_32x32x16_f16(gOc[ks], vrc, gv, 0, 0, 0);       // (dOc.vc + dOr.vr) / 2 in one accumulator
        gac = __builtin_amdgcn_mfma_f32_32x32x16_f16(gSc[ks], vrc, gac, 0, 0, 0);
      }
      if (NEED_REG) {
        const f16x8 vnr = *(const f16x8*)(p.vnF_r + o), vrr = *(const f16x8*)(p.vrF_r + o);
        rr = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlr[ks], vnr, rr, 0, 0, 0);
        gv = __builtin_amdgcn_mfma_f32_32x32x16_f16(gOr[ks], vrr, gv, 0, 0, 0);
        gar = __builtin_amdgcn_mfma_f32_32x32x16_f16(gSr[ks], vrr, gar, 0, 0, 0);
      }
    }
    float pc[16], pr[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
      pc[e] = __builtin_amdgcn_exp2f(a_c * sc[e] - bias) * invl[h][0][row];
      pr[e] = __builtin_amdgcn_exp2f(sc2 * sr[e] - bias) * invl[h][1][row];
      X[0][h][row][col] = 0.5f * (pc[e] + pr[e]);
      if (BOTH) {
        X[1][h][row][col] = valid ? rc[e] : -2.f;
        X[2][h][row][col] = valid ? rr[e] : -2.f;
      } else {
        X[1][h][row][col] = valid ? (NEED_REG ? rr[e] : rc[e]) : -2.f;
        X[2][h][row][col] = NEED_REG ? gar[e] : gac[e];
      }
    }
    __syncthreads();
    const long blk = ((long)tile * p.maxblk + j) * 1024;
    if (BOTH) {
      {
        const int row = tid >> 3, c0 = (tid & 7) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int cc = c0 + c;
          const float mp = 0.25f * (X[0][0][row][cc] + X[0][1][row][cc] + X[0][2][row][cc] + X[0][3][row][cc]);
          const float mc = 0.25f * (X[1][0][row][cc] + X[1][1][row][cc] + X[1][2][row][cc] + X[1][3][row][cc]);
          const float mr = 0.25f * (X[2][0][row][cc] + X[2][1][row][cc] + X[2][2][row][cc] + X[2][3][row][cc]);
          const float ex = __expf(mp);
          A32[0][row][cc] = mc > p.sim_thresh ? ex * dinv[0][row] : 0.f;
          A32[1][row][cc] = mr > p.conf_thresh ? ex * dinv[1][row] : 0.f;
        }
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
        X[1][h][row][col] = gac[e];
        X[2][h][row][col] = gar[e];
      }
      __syncthreads();
      {
        const int row = tid >> 3, c0 = (tid & 7) * 4;
        f32x4 dm;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int cc = c0 + c;
          const float dac = X[1][0][row][cc] + X[1][1][row][cc] + X[1][2][row][cc] + X[1][3][row][cc];
          const float dar = X[2][0][row][cc] + X[2][1][row][cc] + X[2][2][row][cc] + X[2][3][row][cc];
          dm[c] = 0.125f * (A32[0][row][cc] * (dac - gdt[0][row]) + A32[1][row][cc] * (dar - gdt[1][row]));     // dM / 8
          X[0][0][row][cc] = dm[c];
        }
        // query on the lane axis: keys c0 .. c0+3 = 8 i + 4 half + j  ->  [i][lane = 32 half + row][j]
        *(f32x4*)(p.dMq + blk + (((tid & 7) >> 1) * 64 + (tid & 1) * 32 + row) * 4) = dm;
      }
    } else {      // one branch: mean probability, similarity and dA of a (row, key) are all there after the one exchange
      const int row = tid >> 3, c0 = (tid & 7) * 4;
      const float thr = NEED_REG ? p.conf_thresh : p.sim_thresh;
      f32x4 dm;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int cc = c0 + c;
        const float mp = 0.25f * (X[0][0][row][cc] + X[0][1][row][cc] + X[0][2][row][cc] + X[0][3][row][cc]);
        const float mb = 0.25f * (X[1][0][row][cc] + X[1][1][row][cc] + X[1][2][row][cc] + X[1][3][row][cc]);
        const float da = X[2][0][row][cc] + X[2][1][row][cc] + X[2][2][row][cc] + X[2][3][row][cc];
        const float a = mb > thr ? __expf(mp) * dinv[BR][row] : 0.f;
        A32[BR][row][cc] = a;
        dm[c] = 0.125f * (a * (da - gdt[BR][row]));                                                               // dM / 8
        X[0][0][row][cc] = dm[c];      // (this thread's own element: nobody else reads X[0][.][row][cc] in this phase)
      }
      *(f32x4*)(p.dMq + blk + (((tid & 7) >> 1) * 64 + (tid & 1) * 32 + row) * 4) = dm;
    }
    __syncthreads();
    {   // key on the lane axis: rows q0 .. q0+3 of key kk (X[0][0] and A32 stay untouched until the next visit's barriers)
      const int kk = tid >> 3, q0 = (tid & 7) * 4;
      const long o = blk + (((tid & 7) >> 1) * 64 + (tid & 1) * 32 + kk) * 4;
      *(f32x4*)(p.dMk + o) = f32x4{X[0][0][q0][kk], X[0][0][q0 + 1][kk], X[0][0][q0 + 2][kk], X[0][0][q0 + 3][kk]};
      if (CLS_G)
        *(f16x4*)(p.Ack + o) = f16x4{(f16)A32[0][q0][kk], (f16)A32[0][q0 + 1][kk], (f16)A32[0][q0 + 2][kk], (f16)A32[0][q0 + 3][kk]};
      if (NEED_REG)
        *(f16x4*)(p.Ark + o) = f16x4{(f16)A32[1][q0][kk], (f16)A32[1][q0 + 1][kk], (f16)A32[1][q0 + 2][kk], (f16)A32[1][q0 + 3][kk]};
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
      const float hdp = gv[e] + X[0][0][row][col];       // dP / 2
      dl_c[e] += pc[e] * hdp;
      dl_r[e] += pr[e] * hdp;
    }
    __syncthreads();
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      dl_c[e] += __shfl_xor(dl_c[e], o, 64);
      dl_r[e] += __shfl_xor(dl_r[e], o, 64);
    }
  }
  if (col == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
      if (row < t.nrows) {
        const long o = (((long)s * MCA_HEADS + h) * 2) * p.N_L + t.row0 + row;
        p.delta_part[o] = dl_c[e];
        p.delta_part[o + p.N_L] = dl_r[e];
      }
    }
  }
}

__global__ void mca_bwd_rowstat_kernel(const McaTile* __restrict__ tiles, const float* __restrict__ stats,
                                       const float* __restrict__ delta_part, int S, int N_L, float* __restrict__ rowstat) {
  const int tile = blockIdx.x, tid = threadIdx.x, h = tid >> 6, br = (tid >> 5) & 1, row = tid & 31;
  const McaTile t = tiles[tile];
  const bool ok = row < t.nrows;
  const float l = ok ? stats[((long)h * 2 + br) * N_L + t.row0 + row] : 0.f;
  float dsum = 0.f;
  if (ok)
    for (int ss = 0; ss < S; ++ss) dsum += delta_part[(((long)ss * MCA_HEADS + h) * 2 + br) * N_L + t.row0 + row];
  float* o = rowstat + ((long)tile * MCA_HEADS + h) * 256;     // 1 KB per (tile, head): one LDS-DMA instruction
  o[br * 32 + row] = l > 0.f ? 1.0f / l : 0.f;
  o[(2 + br) * 32 + row] = dsum;
}

// ------------------------------------------------------------------------------------------------ LDS staging helpers
// One wave instruction copies one 1 KB fragment (or 1 KB of a dM / A block) from memory into LDS without touching VGPRs;
// source and LDS image are both lane-linear, so the fragment is read back with one ds_read_b128 per lane, conflict-free.
__device__ __forceinline__ void dma_1k(const void* src_base, char* lds_dst, int lane) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)src_base + lane * 16),
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ void wait_all_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ f16x8 lds_frag(const char* base, int lane) { return *(const f16x8*)(base + lane * 16); }
// The compiler's own order for an MFMA whose operand comes from LDS is read, wait, multiply, one fragment at a time (~130 cycles
// of LDS latency per MFMA).  The passes below read a GROUP of fragments, then multiply the previous group: sched_barrier
// keeps the groups apart.
#define MCA_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// ------------------------------------------------------------------------------------------------ pass 2: dq
// Workgroup = 4 consecutive query tiles (one per wave) x one head x one split of the key blocks.  Per visited key block the
// key-side fragments (k^c k^r vc vr rows, k^c k^r columns: 24 KB at DP 64) are staged ONCE for the four waves, double
// buffered, one barrier per visit; each wave adds its own dM block (4 KB).  A wave whose tile does not see the block (local
// blocks of another frame) skips the arithmetic but keeps the barriers.  CLS_G = false (see pass 1): dOc = 0, so dOc.vc and the vc rows are left out.
template <int DP, bool NEED_REG, bool CLS_G>
__global__ __launch_bounds__(256) void mca_bwd_dq_kernel(McaBwdParams p, int ntiles) {
  constexpr int KS = DP / 16, NT = DP / 32;
  constexpr int A_BYTES = 4 * KS * 1024, C_BYTES = 4 * NT * 1024, W_OFF = A_BYTES + C_BYTES, STAGE = W_OFF + 4 * 4096;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int cls, h;
  if (!xcd_remap(blockIdx.x, (ntiles + 3) / 4 * p.S2, MCA_HEADS, cls, h)) return;   // class = (tile group, key split): 4 heads share dM
  const int s = cls % p.S2;
  const int T0 = 4 * (cls / p.S2), T1 = min(T0 + 3, ntiles - 1);
  const int tile = T0 + w;
  const bool active = tile < ntiles;
  const McaTile t = p.tiles[active ? tile : T1];
  const int lb0 = t.loc_start >> 5, lb1 = (t.loc_end + 31) >> 5, n_local = lb1 - lb0;
  const int gb0 = p.N_L >> 5, gb1 = (p.N + 31) >> 5;
  const int Lb0 = p.tiles[T0].loc_start >> 5, Lb1 = (p.tiles[T1].loc_end + 31) >> 5;     // union of the 4 tiles' local blocks
  const int nloc = Lb1 - Lb0, nv = nloc + gb1 - gb0;
  const int qi = lane & 31, half = lane >> 5;
  const int tsafe = active ? tile : T1;
  const float* rs = p.rowstat + ((long)tsafe * MCA_HEADS + h) * 256;
  const float il_c = rs[qi], il_r = rs[32 + qi], dl_c = rs[64 + qi], dl_r = rs[96 + qi];
  f16x8 qc[KS], qr[KS], gOc[KS], gOr[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const long o = rowfrag<DP>(tsafe, h, ks) + lane * 8;
    qc[ks] = *(const f16x8*)(p.qF_c + o);
    qr[ks] = *(const f16x8*)(p.qF_r + o);
    if (CLS_G) gOc[ks] = *(const f16x8*)(p.gOF_c + o);
    if (NEED_REG) gOr[ks] = *(const f16x8*)(p.gOF_r + o);
  }
  f32x16 dqc[NT], dqr[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) dqc[n][e] = dqr[n][e] = 0.f;
  const float sc2 = p.scale * 1.4426950408889634f;      // logits in log2 units: v_exp_f32 is 2^x

  auto decode = [&](int v, int& kb, int& j, bool& mine) {
    if (v < nloc) { kb = Lb0 + v; mine = active && kb >= lb0 && kb < lb1; j = kb - lb0; }
    else { kb = gb0 + v - nloc; mine = active; j = n_local + kb - gb0; }
  };
  auto issue = [&](int v, int buf) {
    int kb, j; bool mine;
    decode(v, kb, j, mine);
    char* sb = smem + buf * STAGE;
    const f16* rows[4] = {p.kF_c, p.kF_r, p.vrF_c, p.vrF_r};
#pragma unroll
    for (int a = 0; a < 4; ++a)
      if (a < 2 || (a == 2 && CLS_G) || (a == 3 && NEED_REG))
        for (int f = w; f < KS; f += 4) dma_1k(rows[a] + rowfrag<DP>(kb, h, f), sb + (a * KS + f) * 1024, lane);
    const f16* cols[2] = {p.kT_c, p.kT_r};
#pragma unroll
    for (int a = 0; a < 2; ++a)
      for (int f = w; f < 2 * NT; f += 4) dma_1k(cols[a] + colfrag<DP>(kb, h, 0, 0) + f * 512, sb + A_BYTES + (a * 2 * NT + f) * 1024, lane);
    // a wave that does not take part stages block 0 of its tile (finite numbers) and masks every key below
    const float* dm = p.dMq + ((long)tsafe * p.maxblk + (mine ? j : 0)) * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_1k(dm + i * 256, sb + W_OFF + w * 4096 + i * 1024, lane);
  };

  int v = s, buf = 0;
  if (v < nv) issue(v, 0);
  for (; v < nv; v += p.S2, buf ^= 1) {
    wait_all_loads();
    __syncthreads();
    if (v + p.S2 < nv) issue(v + p.S2, buf ^ 1);
    // (no branch around the arithmetic: the accumulators would be copied between register files at every merge)
    int kb, j; bool mine;
    decode(v, kb, j, mine);
    const int key0 = kb * 32;
    const int lo = !mine ? 0 : (v < nloc ? t.loc_start : p.N_L), hi = !mine ? 0 : (v < nloc ? t.loc_end : p.N);
    const char* sb = smem + buf * STAGE;
    const bool full = key0 >= lo && key0 + 32 <= hi;       // no key of the block is masked (all but frame-boundary blocks)
    f32x4 ks4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ks4[i] = *(const f32x4*)(p.score_pad + key0 + 8 * i + 4 * half);
    f32x16 sc, sr, gv;
#pragma unroll
    for (int e = 0; e < 16; ++e) sc[e] = sr[e] = gv[e] = 0.f;
    {      // rows = keys (A), columns = queries (B); fragments of k-step pair g + 1 are read while pair g multiplies
      f16x8 fa[2][2][4];
      auto rd = [&](int g, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int a = 0; a < (NEED_REG ? 4 : 3); ++a)
            if (a != 2 || CLS_G) fa[buf][u][a] = lds_frag(sb + (a * KS + 2 * g + u) * 1024, lane);
      };
      rd(0, 0);
      MCA_SCHED_FENCE();
#pragma unroll
      for (int g = 0; g < KS / 2; ++g) {
        if (g + 1 < KS / 2) rd(g + 1, (g + 1) & 1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int ks = 2 * g + u;
          sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][u][0], qc[ks], sc, 0, 0, 0);
          sr = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][u][1], qr[ks], sr, 0, 0, 0);
          if (CLS_G) gv = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][u][2], gOc[ks], gv, 0, 0, 0);
          if (NEED_REG) gv = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][u][3], gOr[ks], gv, 0, 0, 0);
        }
        MCA_SCHED_FENCE();
      }
    }
    f16x8 tc[2], tr[2];
    auto softmax_grad = [&](auto maskc) __attribute__((always_inline)) {
      constexpr bool MASK = decltype(maskc)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k0 = key0 + 8 * i + 4 * half;
        const f32x4 dm = *(const f32x4*)(sb + W_OFF + w * 4096 + (i * 64 + lane) * 16);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int e = 4 * i + jj;
          const float bias = (!MASK || (k0 + jj >= lo && k0 + jj < hi)) ? sc2 : __builtin_inff();
          const float pc = __builtin_amdgcn_exp2f(sc2 * ks4[i][jj] * sc[e] - bias) * il_c;
          const float pr = __builtin_amdgcn_exp2f(sc2 * sr[e] - bias) * il_r;
          const float hdp = gv[e] + dm[jj];                 // dP / 2
          tc[i >> 1][(i & 1) * 4 + jj] = (f16)(pc * (hdp - dl_c));      // d logit / 25 (score); the factor sits in k^T
          tr[i >> 1][(i & 1) * 4 + jj] = (f16)(pr * (hdp - dl_r));
        }
      }
    };
    if (full) softmax_grad(std::false_type{}); else softmax_grad(std::true_type{});
    {
      f16x8 fb[2][NT][2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          fb[m][n][0] = lds_frag(sb + A_BYTES + (n * 2 + m) * 1024, lane);
          fb[m][n][1] = lds_frag(sb + A_BYTES + (2 * NT + n * 2 + m) * 1024, lane);
        }
      MCA_SCHED_FENCE();
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          dqc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tc[m], fb[m][n][0], dqc[n], 0, 0, 0);
          dqr[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tr[m], fb[m][n][1], dqr[n], 0, 0, 0);
        }
    }
  }
  if (!active) return;
  float* oc = p.dqn_c + (long)s * p.dq_stride;
  float* orr = p.dqn_r + (long)s * p.dq_stride;
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
      if (row < t.nrows) {
        const long o = (long)(t.row0 + row) * p.WP + h * DP + n * 32 + qi;
        oc[o] = dqc[n][e];
        orr[o] = dqr[n][e];
      }
    }
}

// ------------------------------------------------------------------------------------------------ pass 3: dk, dv, dscore
// Workgroup = 4 consecutive key blocks (one per wave) x one head x one split of the query tiles.  A visit is (query tile,
// kind): kind 0 = the tile's own frame (local keys), kind 1 = the global keys; the tile's fragments (q^ / dO rows, q^ / dO /
// dsoft columns: 40 KB at DP 64) and its row statistics are staged once for the four waves, each wave adds its own dM / Ac /
// Ar block (8 KB).  Double buffered at DP <= 64 (146 KB of LDS), single buffered above.  CLS_G = false (see pass 1): dOc = dsoftc = 0,
// so dvc = P^T dOc + Ac^T dsoftc = 0 is neither computed nor written (the finish kernel is told), and dOc.vc, the vc rows, the dOc /
// dsoftc fragments and Ac are left out.
template <int DP, bool NEED_REG, bool CLS_G>
__global__ __launch_bounds__(256) void mca_bwd_dkv_kernel(McaBwdParams p, int ntiles) {
  constexpr int KS = DP / 16, NT = DP / 32;
  constexpr int NSTAGE = DP <= 64 ? 2 : 1;
  constexpr int A_BYTES = 4 * KS * 1024, C_BYTES = 12 * NT * 1024, RS_OFF = A_BYTES + C_BYTES, W_OFF = RS_OFF + 1024;
  constexpr int STAGE = W_OFF + 4 * 8192;
  __shared__ __attribute__((aligned(16))) char smem[NSTAGE * STAGE];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int gb0 = p.N_L >> 5, gb1 = (p.N + 31) >> 5;
  int cls, h;
  if (!xcd_remap(blockIdx.x, (gb1 + 3) / 4 * p.S3, MCA_HEADS, cls, h)) return;      // class = (key group, tile split): 4 heads share dM / A
  const int s = cls % p.S3;
  const int B0 = 4 * (cls / p.S3), B1 = B0 + 3;
  const int b = B0 + w;
  const bool active = b < gb1;
  const int bs = active ? b : 0;
  const int key0 = bs * 32, col = lane & 31, half = lane >> 5, key = key0 + col;
  const bool glob_any = B1 >= gb0 && B0 < gb1, glob_mine = active && b >= gb0;
  const float sc2 = p.scale * 1.4426950408889634f;      // logits in log2 units: v_exp_f32 is 2^x
  const float lg_c = sc2 * p.score_pad[key];
  f16x8 kc[KS], kr[KS], vc[KS], vr[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const long o = rowfrag<DP>(bs, h, ks) + lane * 8;
    kc[ks] = *(const f16x8*)(p.kF_c + o);
    kr[ks] = *(const f16x8*)(p.kF_r + o);
    if (CLS_G) vc[ks] = *(const f16x8*)(p.vrF_c + o);
    if (NEED_REG) vr[ks] = *(const f16x8*)(p.vrF_r + o);
  }
  f32x16 akc[NT], akr[NT], avc[NT], avr[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) akc[n][e] = akr[n][e] = avc[n][e] = avr[n][e] = 0.f;
  float dsco = 0.f;

  // next visit (tt, kind) after the given one that at least one of the four key blocks takes part in; tt >= ntiles = none
  auto advance = [&](int& tt, int& kind) {
    for (;;) {
      if (kind == 0) kind = 1; else { kind = 0; tt += p.S3; }
      if (tt >= ntiles) return;
      if (kind == 1) { if (glob_any) return; }
      else {
        const McaTile t = p.tiles[tt];
        if (B1 >= (t.loc_start >> 5) && B0 < ((t.loc_end + 31) >> 5)) return;
      }
    }
  };
  auto mine_of = [&](int tt, int kind, int& j, int& lo, int& hi) -> bool {
    const McaTile t = p.tiles[tt];
    const int lb0 = t.loc_start >> 5, lb1 = (t.loc_end + 31) >> 5;
    if (kind == 0) { j = b - lb0; lo = t.loc_start; hi = t.loc_end; return active && b >= lb0 && b < lb1; }
    j = lb1 - lb0 + b - gb0; lo = p.N_L; hi = p.N;
    return glob_mine;
  };
  auto issue = [&](int tt, int kind, int buf) {
    char* sb = smem + buf * STAGE;
    const f16* rows[4] = {p.qF_c, p.qF_r, p.gOF_c, p.gOF_r};
#pragma unroll
    for (int a = 0; a < 4; ++a)
      if (a < 2 || (a == 2 && CLS_G) || (a == 3 && NEED_REG))
        for (int f = w; f < KS; f += 4) dma_1k(rows[a] + rowfrag<DP>(tt, h, f), sb + (a * KS + f) * 1024, lane);
    const f16* cols[6] = {p.qT_c, p.qT_r, p.gOT_c, p.gST_c, p.gOT_r, p.gST_r};
#pragma unroll
    for (int a = 0; a < 6; ++a)
      if (a < 2 || (a < 4 && CLS_G) || (a >= 4 && NEED_REG))
        for (int f = w; f < 2 * NT; f += 4) dma_1k(cols[a] + colfrag<DP>(tt, h, 0, 0) + f * 512, sb + A_BYTES + (a * 2 * NT + f) * 1024, lane);
    if (w == 0) dma_1k(p.rowstat + ((long)tt * MCA_HEADS + h) * 256, sb + RS_OFF, lane);
    // a wave that does not take part stages block 0 of the tile (finite numbers) and masks its keys and A weights below
    int j, lo, hi;
    const bool mine = mine_of(tt, kind, j, lo, hi);
    const long blk = ((long)tt * p.maxblk + (mine ? j : 0)) * 1024;
    char* wb = sb + W_OFF + w * 8192;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_1k(p.dMk + blk + i * 256, wb + i * 1024, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (CLS_G) dma_1k(p.Ack + blk + i * 512, wb + 4096 + i * 1024, lane);
    if (NEED_REG)
#pragma unroll
      for (int i = 0; i < 2; ++i) dma_1k(p.Ark + blk + i * 512, wb + 6144 + i * 1024, lane);
  };

  int tt = s - p.S3, kind = 1, buf = 0;
  advance(tt, kind);
  if (tt < ntiles) issue(tt, kind, 0);
  while (tt < ntiles) {
    int ntt = tt, nkind = kind;
    advance(ntt, nkind);
    wait_all_loads();
    __syncthreads();
    if (NSTAGE == 2 && ntt < ntiles) issue(ntt, nkind, buf ^ 1);
    {   // (no branch around the arithmetic: the accumulators would be copied between register files at every merge)
      int j, lo, hi;
      const bool mine = mine_of(tt, kind, j, lo, hi);
      const float bias = (mine && key >= lo && key < hi) ? sc2 : __builtin_inff();     // exp2(-inf) = 0 masks the key
      const char* sb = smem + buf * STAGE;
      const char* wb = sb + W_OFF + w * 8192;
      f32x16 sc, sr, gv;
#pragma unroll
      for (int e = 0; e < 16; ++e) sc[e] = sr[e] = gv[e] = 0.f;
      {      // rows = queries (A), columns = keys (B); fragments of k-step pair g + 1 are read while pair g multiplies
        f16x8 fa[2][2][4];
        auto rd = [&](int g, int fbuf) __attribute__((always_inline)) {
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int a = 0; a < (NEED_REG ? 4 : 3); ++a)
              if (a != 2 || CLS_G) fa[fbuf][u][a] = lds_frag(sb + (a * KS + 2 * g + u) * 1024, lane);
        };
        rd(0, 0);
        MCA_SCHED_FENCE();
#pragma unroll
        for (int g = 0; g < KS / 2; ++g) {
          if (g + 1 < KS / 2) rd(g + 1, (g + 1) & 1);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int ks = 2 * g + u;
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][u][0], kc[ks], sc, 0, 0, 0);
            sr = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][u][1], kr[ks], sr, 0, 0, 0);
            if (CLS_G) gv = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][u][2], vc[ks], gv, 0, 0, 0);
            if (NEED_REG) gv = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g & 1][u][3], vr[ks], gv, 0, 0, 0);
          }
          MCA_SCHED_FENCE();
        }
      }
      const float* rs = (const float*)(sb + RS_OFF);
      f16x8 tc[2], tr[2], pp[2], ac[2], ar[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r0 = 8 * i + 4 * half;
        const f32x4 dm = *(const f32x4*)(wb + (i * 64 + lane) * 16);
        const f16x4 zero4 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        const f16x4 a_c = (CLS_G && mine) ? *(const f16x4*)(wb + 4096 + (i * 64 + lane) * 8) : zero4;
        const f16x4 a_r = (NEED_REG && mine) ? *(const f16x4*)(wb + 6144 + (i * 64 + lane) * 8) : zero4;
        const f32x4 il_c = *(const f32x4*)(rs + r0), il_r = *(const f32x4*)(rs + 32 + r0);
        const f32x4 dl_c = *(const f32x4*)(rs + 64 + r0), dl_r = *(const f32x4*)(rs + 96 + r0);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int e = 4 * i + jj, u = (i & 1) * 4 + jj;
          const float pc = __builtin_amdgcn_exp2f(lg_c * sc[e] - bias) * il_c[jj];     // il = 0 on padded rows
          const float pr = __builtin_amdgcn_exp2f(sc2 * sr[e] - bias) * il_r[jj];
          const float hdp = gv[e] + dm[jj];                  // dP / 2
          const float dsc = pc * (hdp - dl_c[jj]), dsr = pr * (hdp - dl_r[jj]);
          dsco += dsc * sc[e];
          tc[i >> 1][u] = (f16)dsc;          // the key row's factor (25 score_k, 25) is applied by mca_bwd_finish_kernel
          tr[i >> 1][u] = (f16)dsr;
          pp[i >> 1][u] = (f16)(pc + pr);    // 2 P against dO / 2
          ac[i >> 1][u] = a_c[jj];
          ar[i >> 1][u] = a_r[jj];
        }
      }
      const char* cb = sb + A_BYTES;
      {      // per (m, n): the 6 (4) column fragments of the next pair are read while this pair multiplies
        f16x8 fb[2][6];
        auto rd = [&](int q, int fbuf) __attribute__((always_inline)) {
          const int f = (q % NT) * 2 + q / NT;      // q = m * NT + n
#pragma unroll
          for (int a = 0; a < (NEED_REG ? 6 : 4); ++a)
            if (a < 2 || a >= 4 || CLS_G) fb[fbuf][a] = lds_frag(cb + (a * 2 * NT + f) * 1024, lane);
        };
        rd(0, 0);
        MCA_SCHED_FENCE();
#pragma unroll
        for (int q = 0; q < 2 * NT; ++q) {
          const int m = q / NT, n = q % NT;
          if (q + 1 < 2 * NT) rd(q + 1, (q + 1) & 1);
          akc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tc[m], fb[q & 1][0], akc[n], 0, 0, 0);
          akr[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tr[m], fb[q & 1][1], akr[n], 0, 0, 0);
          if (CLS_G) {
            avc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pp[m], fb[q & 1][2], avc[n], 0, 0, 0);
            avc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac[m], fb[q & 1][3], avc[n], 0, 0, 0);
          }
          if (NEED_REG) {
            avr[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pp[m], fb[q & 1][4], avr[n], 0, 0, 0);
            avr[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[m], fb[q & 1][5], avr[n], 0, 0, 0);
          }
          MCA_SCHED_FENCE();
        }
      }
    }
    if (NSTAGE == 1) {
      __syncthreads();
      if (ntt < ntiles) issue(ntt, nkind, 0);
    } else {
      buf ^= 1;
    }
    tt = ntt; kind = nkind;
  }
  if (!active) return;
  dsco += __shfl_xor(dsco, 32, 64);
  if (half == 0) p.dscore_part[((long)s * MCA_HEADS + h) * p.Npad + key] = dsco;
  float* okc = p.dkn_c + (long)s * p.dk_stride;
  float* okr = p.dkn_r + (long)s * p.dk_stride;
  float* ovc = p.dv_c + (long)s * p.dk_stride;
  float* ovr = p.dv_r + (long)s * p.dk_stride;
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const long o = (long)(key0 + (e & 3) + 8 * (e >> 2) + 4 * half) * p.WP + h * DP + n * 32 + col;
      okc[o] = akc[n][e];
      okr[o] = akr[n][e];
      if (CLS_G) ovc[o] = avc[n][e];
      if (NEED_REG) ovr[o] = avr[n][e];
    }
}

// ------------------------------------------------------------------------------------------------ normalisation backward
// One wave per (row, branch): x^ = x/||x||  =>  dx = (dx^ - x^ (x^.dx^)) / ||x||  per head, for q and k; v passes through
// (+ the gradient of the "x_ori" rows, i.e. d xcat[:, W:2W], for local rows).  Outputs fp16 dq [N_L,W], dkv [N,2W].
__global__ void mca_bwd_finish_kernel(const f16* __restrict__ q_c, const f16* __restrict__ kv_c, const f16* __restrict__ q_r,
                                      const f16* __restrict__ kv_r, const float* __restrict__ dqn_c, const float* __restrict__ dqn_r,
                                      const float* __restrict__ dkn_c, const float* __restrict__ dkn_r, const float* __restrict__ dv_c,
                                      const float* __restrict__ dv_r, const f16* __restrict__ dvloc_c, const f16* __restrict__ dvloc_r,
                                      int ld_dvloc, int N, int N_L, int W, int d, int DP, int need_reg, f16* dq_c, f16* dkv_c,
                                      f16* dq_r, f16* dkv_r, int nsq, long dq_stride, int nsk, long dk_stride,
                                      const float* __restrict__ dscore_part, int nds, int Npad, float* __restrict__ dscore,
                                      const f16* __restrict__ score, float scale, int cls_grad) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  int n = wave >> 1, br = wave & 1;
  if (n >= N) return;
  const int WP = MCA_HEADS * DP;
  if (dscore_part && br == 0) {     // split passes: dscore = fixed-order sum of the (split, head) partials
    float a = 0.f;
    for (int i = lane; i < nds; i += 64) a += dscore_part[(long)i * Npad + n];
    a = wave_sum(a);
    if (lane == 0) dscore[n] = scale * a;
  }
  const float kfac = br == 0 ? scale * (float)score[n] : scale;     // d(logit)/d(q^.k^) of this key row, left out by the dkv pass
  const f16* q = br ? q_r : q_c;
  const f16* kv = br ? kv_r : kv_c;
  const float* dqn = br ? dqn_r : dqn_c;
  const float* dkn = br ? dkn_r : dkn_c;
  const float* dv = br ? dv_r : dv_c;
  const f16* dvloc = br ? dvloc_r : dvloc_c;
  f16* dq = br ? dq_r : dq_c;
  f16* dkv = br ? dkv_r : dkv_c;
  const bool vgrad = br == 0 ? cls_grad != 0 : need_reg != 0;      // (no class-branch gradient: dvc = 0 and dv_c / dvloc_c are not read)
  for (int h = 0; h < MCA_HEADS; ++h) {
    float kx[2] = {0.f, 0.f}, qx[2] = {0.f, 0.f}, dk[2] = {0.f, 0.f}, dqq[2] = {0.f, 0.f};
    float sk = 0.f, sq = 0.f, kd = 0.f, qd = 0.f;
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
      int j = lane + rnd * 64;
      if (j < d) {
        kx[rnd] = (float)kv[(long)n * 2 * W + h * d + j];
        for (int ss = 0; ss < nsk; ++ss) dk[rnd] += dkn[ss * dk_stride + (long)n * WP + h * DP + j];
        dk[rnd] *= kfac;
        if (n < N_L) {
          qx[rnd] = (float)q[(long)n * W + h * d + j];
          for (int ss = 0; ss < nsq; ++ss) dqq[rnd] += dqn[ss * dq_stride + (long)n * WP + h * DP + j];
        }
      }
      sk += kx[rnd] * kx[rnd]; sq += qx[rnd] * qx[rnd];
      kd += kx[rnd] * dk[rnd]; qd += qx[rnd] * dqq[rnd];
    }
    sk = wave_sum(sk); sq = wave_sum(sq); kd = wave_sum(kd); qd = wave_sum(qd);
    float ik = 1.0f / sqrtf(sk), iq = 1.0f / sqrtf(sq);
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
      int j = lane + rnd * 64;
      if (j < d) {
        // x^.dx^ = (x.dx^)/||x||  =>  dx = dx^/||x|| - x (x.dx^)/||x||^3
        dkv[(long)n * 2 * W + h * d + j] = (f16)(dk[rnd] * ik - kx[rnd] * kd * ik * ik * ik);
        float g = 0.f;
        if (vgrad)
          for (int ss = 0; ss < nsk; ++ss) g += dv[ss * dk_stride + (long)n * WP + h * DP + j];
        if (n < N_L && vgrad) g += (float)dvloc[(long)n * ld_dvloc + h * d + j];
        dkv[(long)n * 2 * W + W + h * d + j] = (f16)g;
        if (n < N_L) dq[(long)n * W + h * d + j] = (f16)(dqq[rnd] * iq - qx[rnd] * qd * iq * iq * iq);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ C-ABI
namespace {
struct McaSplitPlan { int maxblk, nkb, S2, S3; long blocks; };
McaSplitPlan mca_split_plan(int N, int N_L, int ntiles) {
  McaSplitPlan pl;
  pl.maxblk = (N + 31) / 32 + 2;                     // >= blocks of one frame + global blocks, for any tile
  pl.nkb = (N + 31) / 32;
  pl.blocks = (long)ntiles * pl.maxblk;
  auto clampi = [](long v, long lo, long hi) { return (int)(v < lo ? lo : (v > hi ? hi : v)); };
  // one workgroup (4 tiles or 4 key blocks, one head) per CU is the target: the LDS stages allow no more
  const int gq = (ntiles + 3) / 4 * MCA_HEADS, gk = (pl.nkb + 3) / 4 * MCA_HEADS;
  pl.S2 = clampi((256 + gq / 2) / gq, 1, 16);
  pl.S3 = clampi((256 + gk / 2) / gk, 1, ntiles < 8 ? ntiles : 8);
  return pl;
}
}  // namespace

extern "C" long tscd_mca_backward_workspace_bytes(int N, int N_L, int W, int S, int ntiles) {
  int d = W / MCA_HEADS, DP = (d + 31) / 32 * 32, WP = MCA_HEADS * DP;
  long Npad = (N + 31) / 32 * 32 + 32;
  const McaSplitPlan pl = mca_split_plan(N, N_L, ntiles);
  long halves = 2L * N_L * WP + 4L * Npad * WP + 2L * WP * Npad          // forward prep layouts
                + 8L * Npad * WP                                        // key-side fragments
                + 14L * ntiles * 32 * WP                                // query-side fragments
                + 2L * pl.blocks * 1024;                                // Ac, Ar blocks
  long floats = 2L * N_L + (long)S * MCA_HEADS * 2 * N_L + pl.S2 * 2L * N_L * WP + pl.S3 * 4L * Npad * WP + Npad +
                (long)ntiles * MCA_HEADS * 256 + (long)pl.S3 * MCA_HEADS * Npad + 2L * pl.blocks * 1024;
  return halves * 2 + floats * 4 + 16384;
}

// defined in mca.hip
void mca_launch_prep(hipStream_t st, const f16* q_c, const f16* kv_c, const f16* q_r, const f16* kv_r, int N, int N_L, int Npad, int W,
                     int d, int DP, f16* qn_c, f16* kn_c, f16* vn_c, f16* vt_c, f16* qn_r, f16* kn_r, f16* vn_r, f16* vt_r, f16* xcat_c,
                     f16* xcat_r);

// dxcat_* : gradient w.r.t. the [N_L, 2W] concat (first W = attention output, second W = own value rows);
// dsoft_* : gradient w.r.t. the first W columns of the [N_L, 3W] concat (row stride ld_dsoft); soft_* the forward values.
// Outputs: dq_* [N_L, W], dkv_* [N, 2W] fp16; dscore [N] fp32.
extern "C" int tscd_mca_backward_fws(const void* q_c, const void* kv_c, const void* q_r, const void* kv_r, const void* score,
                                     const int* tiles, int ntiles, int N, int N_L, int W, int S, int need_reg, float scale,
                                     float sim_thresh, float conf_thresh, const float* stats, const void* dxcat_c,
                                     const void* dxcat_r, const void* dsoft_c, const void* dsoft_r, int ld_dsoft, const void* soft_c,
                                     const void* soft_r, int ld_soft, void* workspace, long workspace_bytes, void* dq_c, void* dkv_c,
                                     void* dq_r, void* dkv_r, float* dscore, const void* fwd_rows, long fwd_rows_bytes, void* stream);
extern "C" int tscd_mca_backward(const void* q_c, const void* kv_c, const void* q_r, const void* kv_r, const void* score,
                                 const int* tiles, int ntiles, int N, int N_L, int W, int S, int need_reg, float scale,
                                 float sim_thresh, float conf_thresh, const float* stats, const void* dxcat_c,
                                 const void* dxcat_r, const void* dsoft_c, const void* dsoft_r, int ld_dsoft, const void* soft_c,
                                 const void* soft_r, int ld_soft, void* workspace, long workspace_bytes, void* dq_c, void* dkv_c,
                                 void* dq_r, void* dkv_r, float* dscore, void* stream) {
  return tscd_mca_backward_fws(q_c, kv_c, q_r, kv_r, score, tiles, ntiles, N, N_L, W, S, need_reg, scale, sim_thresh, conf_thresh, stats,
                               dxcat_c, dxcat_r, dsoft_c, dsoft_r, ld_dsoft, soft_c, soft_r, ld_soft, workspace, workspace_bytes, dq_c,
                               dkv_c, dq_r, dkv_r, dscore, nullptr, 0, stream);
}
// fwd_rows (optional): the normalised q / k / v rows tscd_mca_forward wrote for the same inputs (its rows_workspace, or the front of its
// workspace), still intact: read in place, the prep launch is skipped.  The layout is mca_common.h's McaRows -- one definition for
// both files -- and fwd_rows_bytes must cover it (tscd_mca_rows_bytes).
extern "C" int tscd_mca_backward_fws(const void* q_c, const void* kv_c, const void* q_r, const void* kv_r, const void* score,
                                     const int* tiles, int ntiles, int N, int N_L, int W, int S, int need_reg, float scale,
                                     float sim_thresh, float conf_thresh, const float* stats, const void* dxcat_c,
                                     const void* dxcat_r, const void* dsoft_c, const void* dsoft_r, int ld_dsoft, const void* soft_c,
                                     const void* soft_r, int ld_soft, void* workspace, long workspace_bytes, void* dq_c, void* dkv_c,
                                     void* dq_r, void* dkv_r, float* dscore, const void* fwd_rows, long fwd_rows_bytes, void* stream) {
  TSCD_CHECK_ARG(q_c && kv_c && q_r && kv_r && score && tiles && stats && soft_c && workspace && dq_c && dkv_c && dq_r && dkv_r && dscore,
                 "mca_backward: null pointer");
  TSCD_CHECK_ARG(!need_reg || (dxcat_r && dsoft_r && soft_r), "mca_backward: reg gradients missing");
  // dxcat_c = dsoft_c = null: no gradient reaches the class branch's outputs (agg_iou: its class output is only used detached);
  // the passes then leave out what would only carry zeros -- same results as zero tensors, bit for bit
  const int cls_grad = (dxcat_c || dsoft_c) ? 1 : 0;
  TSCD_CHECK_ARG(cls_grad ? (dxcat_c && dsoft_c) : need_reg != 0,
                 "mca_backward: class-branch gradients must be both given or (with need_reg) both null");
  TSCD_CHECK_ARG(W % MCA_HEADS == 0 && W % 8 == 0 && N_L > 0 && N >= N_L && ntiles > 0 && S > 0 && S <= 64, "mca_backward: bad sizes");
  const int d = W / MCA_HEADS, DP = (d + 31) / 32 * 32, WP = MCA_HEADS * DP;
  TSCD_CHECK_ARG(DP <= 96, "mca_backward: head dim %d not supported (max 96)", d);
  TSCD_CHECK_ARG(workspace_bytes >= tscd_mca_backward_workspace_bytes(N, N_L, W, S, ntiles), "mca_backward: workspace too small");
  const int Npad = (N + 31) / 32 * 32 + 32;
  const McaSplitPlan pl = mca_split_plan(N, N_L, ntiles);
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  auto take = [&](long bytes) { char* r = ws; ws += (bytes + 255) / 256 * 256; return r; };
  auto takeh = [&](long halves) { return (f16*)take(2L * halves); };
  auto takef = [&](long floats) { return (float*)take(4L * floats); };
  const long kN = (long)Npad * WP, qN = (long)N_L * WP, tN = (long)ntiles * 32 * WP;
  f16* qn_c = takeh(qN); f16* qn_r = takeh(qN);
  f16* kn_c = takeh(kN); f16* kn_r = takeh(kN);
  f16* vn_c = takeh(kN); f16* vn_r = takeh(kN);
  f16* vt_c = takeh(kN); f16* vt_r = takeh(kN);
  McaBwdParams p;
  f16 *kF_c = takeh(kN), *kF_r = takeh(kN), *vnF_c = takeh(kN), *vnF_r = takeh(kN), *vrF_c = takeh(kN), *vrF_r = takeh(kN);
  f16 *kT_c = takeh(kN), *kT_r = takeh(kN);
  p.kF_c = kF_c; p.kF_r = kF_r; p.vnF_c = vnF_c; p.vnF_r = vnF_r; p.vrF_c = vrF_c; p.vrF_r = vrF_r; p.kT_c = kT_c; p.kT_r = kT_r;
  p.qF_c = takeh(tN); p.qF_r = takeh(tN); p.vlF_c = takeh(tN); p.vlF_r = takeh(tN);
  p.gOF_c = takeh(tN); p.gOF_r = takeh(tN); p.gSF_c = takeh(tN); p.gSF_r = takeh(tN);
  p.qT_c = takeh(tN); p.qT_r = takeh(tN); p.gOT_c = takeh(tN); p.gOT_r = takeh(tN); p.gST_c = takeh(tN); p.gST_r = takeh(tN);
  float* gdot = takef(2L * N_L);
  float* delta_part = takef((long)S * MCA_HEADS * 2 * N_L);
  const long dq_stride = qN, dk_stride = kN;
  float* dqn_c = takef(pl.S2 * dq_stride); float* dqn_r = takef(pl.S2 * dq_stride);
  float* dkn_c = takef(pl.S3 * dk_stride); float* dkn_r = takef(pl.S3 * dk_stride);
  float* dv_c = takef(pl.S3 * dk_stride); float* dv_r = takef(pl.S3 * dk_stride);
  float* score_pad = takef(Npad);
  float* rowstat = takef((long)ntiles * MCA_HEADS * 256);
  float* dscore_part = takef((long)pl.S3 * MCA_HEADS * Npad);
  float* dMq = takef(pl.blocks * 1024); float* dMk = takef(pl.blocks * 1024);
  f16* Ack = takeh(pl.blocks * 1024); f16* Ark = takeh(pl.blocks * 1024);
  TSCD_CHECK_ARG(ws - (char*)workspace <= workspace_bytes, "mca_backward: workspace carve-out overflow");
  p.tiles = (const McaTile*)tiles; p.stats = stats; p.gdot = gdot; p.score_pad = score_pad; p.delta_part = delta_part;
  p.rowstat = rowstat; p.dMq = dMq; p.dMk = dMk; p.Ack = Ack; p.Ark = Ark;
  p.dqn_c = dqn_c; p.dqn_r = dqn_r; p.dkn_c = dkn_c; p.dkn_r = dkn_r; p.dv_c = dv_c; p.dv_r = dv_r; p.dscore_part = dscore_part;
  p.N = N; p.N_L = N_L; p.Npad = Npad; p.WP = WP; p.S = S; p.maxblk = pl.maxblk; p.S2 = pl.S2; p.S3 = pl.S3;
  p.dq_stride = dq_stride; p.dk_stride = dk_stride;
  p.scale = scale; p.sim_thresh = sim_thresh; p.conf_thresh = conf_thresh;

  if (fwd_rows) {
    TSCD_CHECK_ARG(fwd_rows_bytes >= mca_rows_bytes(N, N_L, WP), "mca_backward: the forward's rows buffer is %ld bytes, %ld needed",
                   fwd_rows_bytes, mca_rows_bytes(N, N_L, WP));
    const McaRows rows = mca_rows_carve(const_cast<void*>(fwd_rows), N, N_L, WP);
    qn_c = rows.qn_c; qn_r = rows.qn_r; kn_c = rows.kn_c; kn_r = rows.kn_r; vn_c = rows.vn_c; vn_r = rows.vn_r;
  } else {
    mca_launch_prep(st, (const f16*)q_c, (const f16*)kv_c, (const f16*)q_r, (const f16*)kv_r, N, N_L, Npad, W, d, DP, qn_c, kn_c, vn_c,
                    nullptr, qn_r, kn_r, vn_r, nullptr, nullptr, nullptr);
  }
  hipLaunchKernelGGL(mca_bwd_layout_kernel, dim3(Npad / 32, 8), dim3(256), 0, st, kn_c, kn_r, vn_c, vn_r, (const f16*)kv_c,
                     (const f16*)kv_r, N, Npad, W, d, DP, kF_c, kF_r, vnF_c, vnF_r, vrF_c, vrF_r, kT_c, kT_r, (const f16*)score, score_pad, scale);
  hipLaunchKernelGGL(mca_bwd_tiles_kernel, dim3(ntiles, 15), dim3(256), 0, st, (const McaTile*)tiles, qn_c, qn_r, vn_c, vn_r,
                     (const f16*)dxcat_c, 2 * W, (const f16*)dxcat_r, (const f16*)dsoft_c, ld_dsoft, (const f16*)dsoft_r,
                     (const f16*)soft_c, (const f16*)soft_r, ld_soft, N_L, W, d, DP, need_reg, cls_grad, p, gdot);
  TSCD_CHECK_LAUNCH("mca_backward layouts");
  dim3 block(256);
#define MCAS(DPV, REG, CLS)                                                                                                   \
  do {                                                                                                                        \
    hipLaunchKernelGGL((mca_bwd_delta_kernel<DPV, REG, CLS>), dim3(xcd_grid(S, ntiles)), block, 0, st, p, ntiles);            \
    hipLaunchKernelGGL(mca_bwd_rowstat_kernel, dim3(ntiles), block, 0, st, p.tiles, stats, delta_part, S, N_L, rowstat);      \
    hipLaunchKernelGGL((mca_bwd_dq_kernel<DPV, REG, CLS>), dim3(xcd_grid((ntiles + 3) / 4 * pl.S2, MCA_HEADS)), block, 0, st, p, ntiles);  \
    hipLaunchKernelGGL((mca_bwd_dkv_kernel<DPV, REG, CLS>), dim3(xcd_grid((pl.nkb + 3) / 4 * pl.S3, MCA_HEADS)), block, 0, st, p, ntiles); \
  } while (0)
#define MCAB(DPV)                                              \
  do {                                                         \
    if (!need_reg) MCAS(DPV, false, true);                     \
    else if (cls_grad) MCAS(DPV, true, true);                  \
    else MCAS(DPV, true, false);                               \
  } while (0)
  if (DP == 32) MCAB(32);
  else if (DP == 64) MCAB(64);
  else MCAB(96);
#undef MCAB
#undef MCAS
  TSCD_CHECK_LAUNCH("mca_backward passes");
  {
    long waves = 2L * N;
    hipLaunchKernelGGL(mca_bwd_finish_kernel, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, st, (const f16*)q_c,
                       (const f16*)kv_c, (const f16*)q_r, (const f16*)kv_r, dqn_c, dqn_r, dkn_c, dkn_r, dv_c, dv_r,
                       cls_grad ? (const f16*)dxcat_c + W : (const f16*)dxcat_r + W,
                       need_reg ? (const f16*)dxcat_r + W : (const f16*)dxcat_c + W, 2 * W, N, N_L, W, d, DP,
                       need_reg, (f16*)dq_c, (f16*)dkv_c, (f16*)dq_r, (f16*)dkv_r, pl.S2, dq_stride, pl.S3, dk_stride, dscore_part,
                       pl.S3 * MCA_HEADS, Npad, dscore, (const f16*)score, scale, cls_grad);
    TSCD_CHECK_LAUNCH("mca_backward finish");
  }
  return TSCD_OK;
}
