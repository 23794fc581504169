// MCA backward: gradient of the fused global-to-local cosine attention of mca.hip w.r.t. the projected q / k / v rows
// and the key confidence scores (reference: autograd of Attention_mca_g2l.forward, yolox/models/post_trans.py:601-714,
// inside MCA_tscd_g2l_reg.forward :1127-1162; trained through tscd_trainer.py:171-176).
//
// Forward (per head h, query row r, key k; see mca.hip):
//   Pc = softmax_k(25*score_k*qc^.kc^)   Pr = softmax_k(25*qr^.kr^)   P = (Pc+Pr)/2
//   Oc = P vc   Or = P vr             M = mean_h P   E = exp(M)   Ac = [Rc>.75] E / Dc   Ar = [Rr>.99] E / Dr
//   softc = Ac Vc(all heads)   softr = Ar Vr            (the R masks are constants: no gradient through v^)
// Backward, given dOc, dOr, dsoftc, dsoftr (two passes over (query tile, key block), nothing of size N1 x N2 in HBM):
//   dA_c[r,k] = dsoftc_r . Vc_k (all heads)        dM = Ac (dA_c - dsoftc_r.softc_r) + Ar (dA_r - dsoftr_r.softr_r)
//   dP_h = dOc_h . vc_h + dOr_h . vr_h + dM/4
//   pass 1: delta_c[h,r] = sum_k Pc dP/2, delta_r[h,r] = sum_k Pr dP/2
//   pass 2: dSc = Pc (dP/2 - delta_c), dSr = Pr (dP/2 - delta_r)
//           dq^c += (25 score dSc) k^c        dk^c += (25 score dSc)^T q^c        dscore += sum_{h,r} 25 dSc (q^c.k^c)
//           dq^r += (25 dSr) k^r              dk^r += (25 dSr)^T q^r
//           dvc  += P^T dOc + Ac^T dsoftc     dvr  += P^T dOr + Ar^T dsoftr
// Workgroup = one 32-query tile x 4 heads (one wave per head) as in the forward; key blocks strided over gridDim.y.
// Products that reduce over keys use A-fragments read row-wise from LDS fp16 tiles; products that reduce over query rows
// (dk, dv) read the same tiles with ds_read_b64_tr_b16 (hardware transpose) and take their B operands from per-tile
// transposed copies prepared once (q^T, dO^T, dsoft^T: [tile][4*DP][32], 16-byte aligned rows).  dq accumulates in
// registers; dk / dv / dscore are summed over query tiles with fp32 atomics (two 128-byte segments per wave instruction).
#include "mca_common.h"
#include <type_traits>

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct McaBwdParams {
  const f16 *qn_c, *qn_r, *kn_c, *kn_r, *vn_c, *vn_r;   // normalised, padded [rows][WP]
  const f16 *vraw_c, *vraw_r;                           // raw v, padded [Npad][WP]
  const f16 *kt_c, *kt_r;                               // k^ transposed [WP][Npad]
  const f16 *qT_c, *qT_r;                               // per tile [ntiles][WP][32]
  const f16 *gO_c, *gO_r, *gS_c, *gS_r;                 // upstream grads, padded row-major [N_L][WP]
  const f16 *gOT_c, *gOT_r, *gST_c, *gST_r;             // per tile transposed [ntiles][WP][32]
  const f16* score;
  const McaTile* tiles;
  const float* stats;        // [HEADS][2][N_L] softmax row sums, then [2][N_L] sim_round2 denominators
  const float* gdot;         // [2][N_L]: dsoftc.softc, dsoftr.softr
  float* delta_part;         // [S][HEADS][2][N_L]
  float *dqn_c, *dqn_r;      // [N_L][WP]   (atomic accumulate)
  float *dkn_c, *dkn_r, *dv_c, *dv_r;   // [Npad][WP]
  float* dscore;             // [N]
  int N, N_L, Npad, WP, S;
  float scale, sim_thresh, conf_thresh;
  // split passes (mode 2): per (query tile, key block) 32x32 blocks written by pass 1, read by the dq / dkv passes
  float *dMq, *dMk;          // dM, [tile][maxblk][32 q][32 key] and [tile][maxblk][32 key][32 q]
  f16 *Ack, *Ark;            // sim_round2 weights, [tile][maxblk][32 key][32 q]
  const float* score_pad;    // [Npad] fp32 (0 beyond N)
  float* rowstat;            // [tile][HEADS][4][32]: 1/l_c, 1/l_r, delta_c, delta_r (0 on padded rows)
  float* dscore_part;        // [S3 * HEADS][Npad]
  int maxblk, S2, S3;
  long dq_stride, dk_stride; // floats between the partial sums of consecutive splits
};

__device__ __forceinline__ f16x8 tr_frag16(const f16* tile_row0_col0, int ldst, int lane) {
  const int i = lane & 15, q = i >> 2, p = i & 3;
  auto a0 = (__attribute__((address_space(3))) s16x4*)(tile_row0_col0 + q * ldst + 4 * p);
  auto a1 = (__attribute__((address_space(3))) s16x4*)(tile_row0_col0 + (q + 4) * ldst + 4 * p);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(a1);
  f16x4 lo4 = __builtin_bit_cast(f16x4, lo), hi4 = __builtin_bit_cast(f16x4, hi);   // whole-vector casts (see conv_wgrad.hip)
  return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ------------------------------------------------------------------------------------------------ layouts for backward
// One wave per (row, branch): extra operand layouts that the forward prep does not produce (raw v padded, k^ transposed,
// per-tile transposed q^) -- rows follow mca_prep_kernel's conventions.
__global__ void mca_bwd_layout_kernel(const f16* __restrict__ kn_c, const f16* __restrict__ kn_r, const f16* __restrict__ kv_c,
                                      const f16* __restrict__ kv_r, int N, int Npad, int W, int d, int DP, f16* vraw_c,
                                      f16* vraw_r, f16* kt_c, f16* kt_r, const f16* __restrict__ score, float* score_pad) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  int n = wave >> 1, br = wave & 1;
  if (n >= Npad) return;
  if (br == 0 && lane == 0) score_pad[n] = n < N ? (float)score[n] : 0.f;
  const int WP = MCA_HEADS * DP;
  const f16* kn = br ? kn_r : kn_c;
  const f16* kv = br ? kv_r : kv_c;
  f16* vraw = br ? vraw_r : vraw_c;
  f16* kt = br ? kt_r : kt_c;
  for (int c = lane; c < WP; c += 64) {
    int h = c / DP, j = c % DP;
    f16 v = (n < N && j < d) ? kv[(long)n * 2 * W + W + h * d + j] : (f16)0.f;
    vraw[(long)n * WP + c] = v;
    kt[(long)c * Npad + n] = kn[(long)n * WP + c];
  }
}

// Per q-tile: padded row-major + transposed copies of q^ and of the upstream gradients, and gdot = dsoft . soft.
__global__ void mca_bwd_tiles_kernel(const McaTile* __restrict__ tiles, int ntiles, const f16* __restrict__ qn_c,
                                     const f16* __restrict__ qn_r, const f16* __restrict__ dO_c, int ld_dO,
                                     const f16* __restrict__ dO_r, const f16* __restrict__ dS_c, int ld_dS,
                                     const f16* __restrict__ dS_r, const f16* __restrict__ soft_c, const f16* __restrict__ soft_r,
                                     int ld_soft, int N_L, int W, int d, int DP, int need_reg, f16* qT_c, f16* qT_r, f16* gO_c,
                                     f16* gO_r, f16* gS_c, f16* gS_r, f16* gOT_c, f16* gOT_r, f16* gST_c, f16* gST_r,
                                     float* gdot) {
  const int t = blockIdx.x, tid = threadIdx.x;
  const McaTile tl = tiles[t];
  const int WP = MCA_HEADS * DP;
  for (int i = tid; i < 32 * WP; i += blockDim.x) {
    int j = i / WP, c = i % WP;          // j = row inside the tile, c = padded column
    int h = c / DP, jj = c % DP;
    bool ok = j < tl.nrows && jj < d;
    long row = tl.row0 + j;
    f16 z = (f16)0.f;
    f16 a = ok ? dO_c[row * ld_dO + h * d + jj] : z;
    f16 b = ok ? dS_c[row * ld_dS + h * d + jj] : z;
    f16 ar = (ok && need_reg) ? dO_r[row * ld_dO + h * d + jj] : z;
    f16 br = (ok && need_reg) ? dS_r[row * ld_dS + h * d + jj] : z;
    if (j < tl.nrows) {
      gO_c[row * WP + c] = a; gS_c[row * WP + c] = b; gO_r[row * WP + c] = ar; gS_r[row * WP + c] = br;
    }
    long o = ((long)t * WP + c) * 32 + j;
    gOT_c[o] = a; gST_c[o] = b; gOT_r[o] = ar; gST_r[o] = br;
    qT_c[o] = j < tl.nrows ? qn_c[row * WP + c] : z;
    qT_r[o] = j < tl.nrows ? qn_r[row * WP + c] : z;
  }
  // gdot: one wave per row
  int wv = tid >> 6, lane = tid & 63;
  for (int j = wv; j < tl.nrows; j += (blockDim.x >> 6)) {
    long row = tl.row0 + j;
    float gc = 0.f, gr = 0.f;
    for (int c = lane; c < W; c += 64) {
      gc += (float)dS_c[row * ld_dS + c] * (float)soft_c[row * ld_soft + c];
      if (need_reg) gr += (float)dS_r[row * ld_dS + c] * (float)soft_r[row * ld_soft + c];
    }
    gc = wave_sum(gc); gr = wave_sum(gr);
    if (lane == 0) { gdot[row] = gc; gdot[N_L + row] = gr; }
  }
}

// ------------------------------------------------------------------------------------------------ core
template <int DP, bool NEED_REG, int PASS, bool STORE = false>
__global__ __launch_bounds__(256, PASS == 1 ? 2 : 1) void mca_bwd_core_kernel(McaBwdParams p) {
  constexpr int KS = DP / 16, NT = DP / 32;
  constexpr int PST = 33, HST = 40;
  __shared__ float X[3][MCA_HEADS][32][PST];          // cross-head exchange (round 1: P, Rc, Rr; round 2: -, dAc, dAr)
  __shared__ float A32[2][32][PST];                   // Ac, Ar (fp32)
  __shared__ __attribute__((aligned(16))) f16 A16[2][32][HST];              // Ac, Ar (fp16 MFMA operands)
  __shared__ __attribute__((aligned(16))) f16 T16[3][MCA_HEADS][32][HST];   // per head: Tc, Tr, P
  __shared__ float invl[MCA_HEADS][2][32];
  __shared__ float dlt[MCA_HEADS][2][32];
  __shared__ float dinv[2][32];
  __shared__ float gdt[2][32];

  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  const int tile = blockIdx.x;
  const McaTile t = p.tiles[tile];
  const int s = blockIdx.y;
  const int n_local = ((t.loc_end + 31) >> 5) - (t.loc_start >> 5);
  const int n_glob = ((p.N + 31) >> 5) - (p.N_L >> 5);
  const int nblk = n_local + n_glob;
  {
    int hh = tid >> 6, rem = tid & 63, br = rem >> 5, row = rem & 31;
    bool ok = row < t.nrows;
    float l = ok ? p.stats[((long)hh * 2 + br) * p.N_L + t.row0 + row] : 0.f;
    invl[hh][br][row] = l > 0.f ? 1.0f / l : 0.f;
    float dsum = 0.f;
    if (PASS == 2 && ok)
      for (int ss = 0; ss < p.S; ++ss) dsum += p.delta_part[(((long)ss * MCA_HEADS + hh) * 2 + br) * p.N_L + t.row0 + row];
    dlt[hh][br][row] = dsum;
    if (tid < 64) {
      float dn = ok ? p.stats[(long)MCA_HEADS * 2 * p.N_L + (long)br * p.N_L + t.row0 + row] : 0.f;
      dinv[br][row] = dn > 0.f ? 1.0f / dn : 0.f;
      gdt[br][row] = ok ? p.gdot[(long)br * p.N_L + t.row0 + row] : 0.f;
    }
  }
  int qrow = t.row0 + (lane & 31);
  if (qrow >= t.row0 + t.nrows) qrow = t.row0;
  f16x8 qc[KS], qr[KS], vlc[KS], vlr[KS], gOc[KS], gOr[KS], gSc[KS], gSr[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    long o = (long)qrow * p.WP + h * DP + ks * 16 + (lane >> 5) * 8;
    qc[ks] = *(const f16x8*)(p.qn_c + o);
    qr[ks] = *(const f16x8*)(p.qn_r + o);
    vlc[ks] = *(const f16x8*)(p.vn_c + o);
    gOc[ks] = *(const f16x8*)(p.gO_c + o);
    gSc[ks] = *(const f16x8*)(p.gS_c + o);
    if (NEED_REG) {
      vlr[ks] = *(const f16x8*)(p.vn_r + o);
      gOr[ks] = *(const f16x8*)(p.gO_r + o);
      gSr[ks] = *(const f16x8*)(p.gS_r + o);
    }
  }
  f32x16 dqc[NT], dqr[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) dqc[n][e] = dqr[n][e] = 0.f;
  float dl_c[16], dl_r[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) dl_c[e] = dl_r[e] = 0.f;
  __syncthreads();

  const int col = lane & 31, half = lane >> 5, g = lane >> 4;
  for (int j = s; j < nblk; j += p.S) {
    int key0, lo, hi;
    block_range(t, p.N_L, p.N, j, key0, lo, hi, n_local);
    const int key = key0 + col;
    const bool valid = key >= lo && key < hi;
    const float kscore = valid ? (float)p.score[key] : 0.f;
    f32x16 sc, sr, rc, rr, gvc, gvr, gac, gar;
#pragma unroll
    for (int e = 0; e < 16; ++e) sc[e] = sr[e] = rc[e] = rr[e] = gvc[e] = gvr[e] = gac[e] = gar[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      long o = (long)key * p.WP + h * DP + ks * 16 + half * 8;
      f16x8 kc = *(const f16x8*)(p.kn_c + o), kr = *(const f16x8*)(p.kn_r + o);
      f16x8 vnc = *(const f16x8*)(p.vn_c + o), vrc = *(const f16x8*)(p.vraw_c + o);
      sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qc[ks], kc, sc, 0, 0, 0);
      sr = __builtin_amdgcn_mfma_f32_32x32x16_f16(qr[ks], kr, sr, 0, 0, 0);
      rc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlc[ks], vnc, rc, 0, 0, 0);
      gvc = __builtin_amdgcn_mfma_f32_32x32x16_f16(gOc[ks], vrc, gvc, 0, 0, 0);
      gac = __builtin_amdgcn_mfma_f32_32x32x16_f16(gSc[ks], vrc, gac, 0, 0, 0);
      if (NEED_REG) {
        f16x8 vnr = *(const f16x8*)(p.vn_r + o), vrr = *(const f16x8*)(p.vraw_r + o);
        rr = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlr[ks], vnr, rr, 0, 0, 0);
        gvr = __builtin_amdgcn_mfma_f32_32x32x16_f16(gOr[ks], vrr, gvr, 0, 0, 0);
        gar = __builtin_amdgcn_mfma_f32_32x32x16_f16(gSr[ks], vrr, gar, 0, 0, 0);
      }
    }
    float pc[16], pr[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
      pc[e] = valid ? __expf(p.scale * kscore * sc[e] - p.scale) * invl[h][0][row] : 0.f;
      pr[e] = valid ? __expf(p.scale * sr[e] - p.scale) * invl[h][1][row] : 0.f;
      X[0][h][row][col] = 0.5f * (pc[e] + pr[e]);
      X[1][h][row][col] = valid ? rc[e] : -2.f;
      X[2][h][row][col] = (valid && NEED_REG) ? rr[e] : -2.f;
    }
    __syncthreads();
    {
      const int row = tid >> 3, c0 = (tid & 7) * 4;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        int cc = c0 + c;
        float mp = 0.25f * (X[0][0][row][cc] + X[0][1][row][cc] + X[0][2][row][cc] + X[0][3][row][cc]);
        float mc = 0.25f * (X[1][0][row][cc] + X[1][1][row][cc] + X[1][2][row][cc] + X[1][3][row][cc]);
        float mr = 0.25f * (X[2][0][row][cc] + X[2][1][row][cc] + X[2][2][row][cc] + X[2][3][row][cc]);
        float ex = __expf(mp);
        float ac = mc > p.sim_thresh ? ex * dinv[0][row] : 0.f;
        float ar = (NEED_REG && mr > p.conf_thresh) ? ex * dinv[1][row] : 0.f;
        A32[0][row][cc] = ac; A32[1][row][cc] = ar;
        A16[0][row][cc] = (f16)ac; A16[1][row][cc] = (f16)ar;
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
      X[1][h][row][col] = gac[e];
      X[2][h][row][col] = NEED_REG ? gar[e] : 0.f;
    }
    __syncthreads();
    {
      const int row = tid >> 3, c0 = (tid & 7) * 4;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        int cc = c0 + c;
        float dac = X[1][0][row][cc] + X[1][1][row][cc] + X[1][2][row][cc] + X[1][3][row][cc];
        float dar = X[2][0][row][cc] + X[2][1][row][cc] + X[2][2][row][cc] + X[2][3][row][cc];
        X[0][0][row][cc] = A32[0][row][cc] * (dac - gdt[0][row]) + A32[1][row][cc] * (dar - gdt[1][row]);   // dM
      }
      if (STORE) {
        const long blk = ((long)tile * p.maxblk + j) * 1024;
        *(f32x4*)(p.dMq + blk + row * 32 + c0) =
            f32x4{X[0][0][row][c0], X[0][0][row][c0 + 1], X[0][0][row][c0 + 2], X[0][0][row][c0 + 3]};
      }
    }
    __syncthreads();
    if (STORE) {   // key-major copies for the key-stationary pass (X[0][0] and A32 stay untouched until the next visit)
      const int kk = tid >> 3, q0 = (tid & 7) * 4;
      const long blk = ((long)tile * p.maxblk + j) * 1024 + kk * 32 + q0;
      *(f32x4*)(p.dMk + blk) = f32x4{X[0][0][q0][kk], X[0][0][q0 + 1][kk], X[0][0][q0 + 2][kk], X[0][0][q0 + 3][kk]};
      *(f16x4*)(p.Ack + blk) = f16x4{(f16)A32[0][q0][kk], (f16)A32[0][q0 + 1][kk], (f16)A32[0][q0 + 2][kk], (f16)A32[0][q0 + 3][kk]};
      if (NEED_REG)
        *(f16x4*)(p.Ark + blk) = f16x4{(f16)A32[1][q0][kk], (f16)A32[1][q0 + 1][kk], (f16)A32[1][q0 + 2][kk], (f16)A32[1][q0 + 3][kk]};
    }
    float dsco = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
      float hdp = 0.5f * (gvc[e] + gvr[e] + 0.25f * X[0][0][row][col]);
      if (PASS == 1) {
        dl_c[e] += pc[e] * hdp;
        dl_r[e] += pr[e] * hdp;
      } else {
        float dsc = pc[e] * (hdp - dlt[h][0][row]);
        float dsr = pr[e] * (hdp - dlt[h][1][row]);
        dsco += dsc * p.scale * sc[e];
        T16[0][h][row][col] = (f16)(dsc * p.scale * kscore);
        T16[1][h][row][col] = (f16)(dsr * p.scale);
        T16[2][h][row][col] = (f16)(0.5f * (pc[e] + pr[e]));
      }
    }
    if (PASS == 2) {
      dsco += __shfl_xor(dsco, 32, 64);
      if (half == 0 && valid) atomicAdd(p.dscore + key, dsco);
      __syncthreads();
      // ---- products that reduce over keys: dq^ += T K^  (A rows from LDS, B from the transposed k^) ----
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ko = ks * 16 + half * 8;
        f16x8 atc = *(const f16x8*)(&T16[0][h][col][ko]);
        f16x8 atr = *(const f16x8*)(&T16[1][h][col][ko]);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          long o = (long)(h * DP + n * 32 + col) * p.Npad + key0 + ko;
          dqc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(atc, *(const f16x8*)(p.kt_c + o), dqc[n], 0, 0, 0);
          dqr[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(atr, *(const f16x8*)(p.kt_r + o), dqr[n], 0, 0, 0);
        }
      }
      // ---- products that reduce over query rows: dk^ = T^T q^,  dv = P^T dO + A^T dsoft  (transposing LDS reads) ----
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        f32x16 akc, akr, avc, avr;
#pragma unroll
        for (int e = 0; e < 16; ++e) akc[e] = akr[e] = avc[e] = avr[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int kb = ks * 16 + 8 * (g >> 1), mb = 16 * (g & 1);
          long ob = ((long)tile * p.WP + h * DP + n * 32 + col) * 32 + ks * 16 + half * 8;
          f16x8 tcT = tr_frag16(&T16[0][h][kb][mb], HST, lane);
          f16x8 trT = tr_frag16(&T16[1][h][kb][mb], HST, lane);
          f16x8 pT = tr_frag16(&T16[2][h][kb][mb], HST, lane);
          f16x8 acT = tr_frag16(&A16[0][kb][mb], HST, lane);
          akc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tcT, *(const f16x8*)(p.qT_c + ob), akc, 0, 0, 0);
          akr = __builtin_amdgcn_mfma_f32_32x32x16_f16(trT, *(const f16x8*)(p.qT_r + ob), akr, 0, 0, 0);
          avc = __builtin_amdgcn_mfma_f32_32x32x16_f16(pT, *(const f16x8*)(p.gOT_c + ob), avc, 0, 0, 0);
          avc = __builtin_amdgcn_mfma_f32_32x32x16_f16(acT, *(const f16x8*)(p.gST_c + ob), avc, 0, 0, 0);
          if (NEED_REG) {
            f16x8 arT = tr_frag16(&A16[1][kb][mb], HST, lane);
            avr = __builtin_amdgcn_mfma_f32_32x32x16_f16(pT, *(const f16x8*)(p.gOT_r + ob), avr, 0, 0, 0);
            avr = __builtin_amdgcn_mfma_f32_32x32x16_f16(arT, *(const f16x8*)(p.gST_r + ob), avr, 0, 0, 0);
          }
        }
        // result rows = keys of this block, columns = channels of head h
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int krow = key0 + (e & 3) + 8 * (e >> 2) + 4 * half;
          if (krow >= lo && krow < hi) {
            long o = (long)krow * p.WP + h * DP + n * 32 + col;
            atomicAdd(p.dkn_c + o, akc[e]);
            atomicAdd(p.dkn_r + o, akr[e]);
            atomicAdd(p.dv_c + o, avc[e]);
            if (NEED_REG) atomicAdd(p.dv_r + o, avr[e]);
          }
        }
      }
    }
    __syncthreads();
  }
  if (PASS == 1) {
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
        int row = (e & 3) + 8 * (e >> 2) + 4 * half;
        if (row < t.nrows) {
          long o = (((long)s * MCA_HEADS + h) * 2) * p.N_L + t.row0 + row;
          p.delta_part[o] = dl_c[e];
          p.delta_part[o + p.N_L] = dl_r[e];
        }
      }
    }
  } else {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int row = (e & 3) + 8 * (e >> 2) + 4 * half;
        if (row < t.nrows) {
          long o = (long)(t.row0 + row) * p.WP + h * DP + n * 32 + col;
          atomicAdd(p.dqn_c + o, dqc[n][e]);
          atomicAdd(p.dqn_r + o, dqr[n][e]);
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------ pass 2, paired tiles
// Pass 2 is bound by its float atomics: every (query tile, key block) pair adds a 32 x WP block into each of dk^c, dk^r, dv^c
// (dv^r) -- ~0.6 G atomics per call at 16 x 500 proposals.  Two consecutive query tiles of the SAME frame see the same key
// blocks, so one workgroup walks the key blocks once for both: dk / dv / dscore contributions of the two tiles are summed in
// registers and flushed once (half the atomics), dq stays per tile in registers.  The q-side fragments are fetched per visit
// (L2-resident) instead of being held, which pays for the second tile's dq accumulators.  Tiles whose neighbour belongs to
// another frame (or the odd last tile) take the same loop with a group of one.
template <int DP, bool NEED_REG>
__global__ __launch_bounds__(256) void mca_bwd_pair_kernel(McaBwdParams p, int ntiles) {
  constexpr int KS = DP / 16, NT = DP / 32;
  constexpr int PST = 33, HST = 40;
  __shared__ float X[3][MCA_HEADS][32][PST];
  __shared__ float A32[2][32][PST];
  __shared__ __attribute__((aligned(16))) f16 A16[2][32][HST];
  __shared__ __attribute__((aligned(16))) f16 T16[3][MCA_HEADS][32][HST];
  __shared__ float invl[2][MCA_HEADS][2][32];
  __shared__ float dlt[2][MCA_HEADS][2][32];
  __shared__ float dinv[2][2][32];
  __shared__ float gdt[2][2][32];

  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  const int s = blockIdx.y;
  const int tile0 = 2 * blockIdx.x;
  const McaTile ta = p.tiles[tile0];
  const bool has_b = tile0 + 1 < ntiles;
  const McaTile tb = has_b ? p.tiles[tile0 + 1] : ta;
  const bool paired = has_b && tb.loc_start == ta.loc_start && tb.loc_end == ta.loc_end;
  {
    const int hh = tid >> 6, rem = tid & 63, br = rem >> 5, row = rem & 31;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const McaTile t = sub ? tb : ta;
      const bool ok = (sub == 0 || has_b) && row < t.nrows;
      const float l = ok ? p.stats[((long)hh * 2 + br) * p.N_L + t.row0 + row] : 0.f;
      invl[sub][hh][br][row] = l > 0.f ? 1.0f / l : 0.f;
      float dsum = 0.f;
      if (ok)
        for (int ss = 0; ss < p.S; ++ss) dsum += p.delta_part[(((long)ss * MCA_HEADS + hh) * 2 + br) * p.N_L + t.row0 + row];
      dlt[sub][hh][br][row] = dsum;
      if (tid < 64) {
        const float dn = ok ? p.stats[(long)MCA_HEADS * 2 * p.N_L + (long)br * p.N_L + t.row0 + row] : 0.f;
        dinv[sub][br][row] = dn > 0.f ? 1.0f / dn : 0.f;
        gdt[sub][br][row] = ok ? p.gdot[(long)br * p.N_L + t.row0 + row] : 0.f;
      }
    }
  }
  f32x16 dqc[2][NT], dqr[2][NT];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) dqc[sub][n][e] = dqr[sub][n][e] = 0.f;
  __syncthreads();

  const int col = lane & 31, half = lane >> 5, g = lane >> 4;
  // groups: {a, b} when paired, else {a} then {b}
  const int ngroups = paired ? 1 : (has_b ? 2 : 1);
  for (int grp = 0; grp < ngroups; ++grp) {
    const int first = paired ? 0 : grp;
    const McaTile tg = first ? tb : ta;
    const int n_local = ((tg.loc_end + 31) >> 5) - (tg.loc_start >> 5);
    const int n_glob = ((p.N + 31) >> 5) - (p.N_L >> 5);
    const int nblk = n_local + n_glob;
    for (int j = s; j < nblk; j += p.S) {
      int key0, lo, hi;
      block_range(tg, p.N_L, p.N, j, key0, lo, hi, n_local);
      const int key = key0 + col;
      const bool valid = key >= lo && key < hi;
      const float kscore = valid ? (float)p.score[key] : 0.f;
      f32x16 akc[NT], akr[NT], avc[NT], avr[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) akc[n][e] = akr[n][e] = avc[n][e] = avr[n][e] = 0.f;
      float dsco = 0.f;
      auto visit = [&](auto subc) __attribute__((always_inline)) {
        {
          constexpr int sub = decltype(subc)::value;   // compile-time: dq[sub] must stay in registers
          const McaTile t = sub ? tb : ta;
          const int tile = tile0 + sub;
          int qrow = t.row0 + (lane & 31);
          if (qrow >= t.row0 + t.nrows) qrow = t.row0;
          // scores and similarity products first (their accumulators die at the first exchange), the gradient products after
          // it: 64 fewer live registers than one fused loop
          f32x16 sc, sr, rc, rr;
#pragma unroll
          for (int e = 0; e < 16; ++e) sc[e] = sr[e] = rc[e] = rr[e] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const long oq = (long)qrow * p.WP + h * DP + ks * 16 + half * 8;
            const long o = (long)key * p.WP + h * DP + ks * 16 + half * 8;
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.qn_c + oq), *(const f16x8*)(p.kn_c + o), sc, 0, 0, 0);
            sr = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.qn_r + oq), *(const f16x8*)(p.kn_r + o), sr, 0, 0, 0);
            rc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.vn_c + oq), *(const f16x8*)(p.vn_c + o), rc, 0, 0, 0);
            if (NEED_REG) rr = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.vn_r + oq), *(const f16x8*)(p.vn_r + o), rr, 0, 0, 0);
          }
          float pc[16], pr[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            pc[e] = valid ? __expf(p.scale * kscore * sc[e] - p.scale) * invl[sub][h][0][row] : 0.f;
            pr[e] = valid ? __expf(p.scale * sr[e] - p.scale) * invl[sub][h][1][row] : 0.f;
            X[0][h][row][col] = 0.5f * (pc[e] + pr[e]);
            X[1][h][row][col] = valid ? rc[e] : -2.f;
            X[2][h][row][col] = (valid && NEED_REG) ? rr[e] : -2.f;
          }
          __syncthreads();
          f32x16 gvc, gvr, gac, gar;
#pragma unroll
          for (int e = 0; e < 16; ++e) gvc[e] = gvr[e] = gac[e] = gar[e] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const long oq = (long)qrow * p.WP + h * DP + ks * 16 + half * 8;
            const long o = (long)key * p.WP + h * DP + ks * 16 + half * 8;
            const f16x8 vrc = *(const f16x8*)(p.vraw_c + o);
            gvc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.gO_c + oq), vrc, gvc, 0, 0, 0);
            gac = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.gS_c + oq), vrc, gac, 0, 0, 0);
            if (NEED_REG) {
              const f16x8 vrr = *(const f16x8*)(p.vraw_r + o);
              gvr = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.gO_r + oq), vrr, gvr, 0, 0, 0);
              gar = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.gS_r + oq), vrr, gar, 0, 0, 0);
            }
          }
          {
            const int row = tid >> 3, c0 = (tid & 7) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              int cc = c0 + c;
              float mp = 0.25f * (X[0][0][row][cc] + X[0][1][row][cc] + X[0][2][row][cc] + X[0][3][row][cc]);
              float mc = 0.25f * (X[1][0][row][cc] + X[1][1][row][cc] + X[1][2][row][cc] + X[1][3][row][cc]);
              float mr = 0.25f * (X[2][0][row][cc] + X[2][1][row][cc] + X[2][2][row][cc] + X[2][3][row][cc]);
              float ex = __expf(mp);
              float ac = mc > p.sim_thresh ? ex * dinv[sub][0][row] : 0.f;
              float ar = (NEED_REG && mr > p.conf_thresh) ? ex * dinv[sub][1][row] : 0.f;
              A32[0][row][cc] = ac; A32[1][row][cc] = ar;
              A16[0][row][cc] = (f16)ac; A16[1][row][cc] = (f16)ar;
            }
          }
          __syncthreads();
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            X[1][h][row][col] = gac[e];
            X[2][h][row][col] = NEED_REG ? gar[e] : 0.f;
          }
          __syncthreads();
          {
            const int row = tid >> 3, c0 = (tid & 7) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              int cc = c0 + c;
              float dac = X[1][0][row][cc] + X[1][1][row][cc] + X[1][2][row][cc] + X[1][3][row][cc];
              float dar = X[2][0][row][cc] + X[2][1][row][cc] + X[2][2][row][cc] + X[2][3][row][cc];
              X[0][0][row][cc] = A32[0][row][cc] * (dac - gdt[sub][0][row]) + A32[1][row][cc] * (dar - gdt[sub][1][row]);   // dM
            }
          }
          __syncthreads();
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            float hdp = 0.5f * (gvc[e] + gvr[e] + 0.25f * X[0][0][row][col]);
            float dsc = pc[e] * (hdp - dlt[sub][h][0][row]);
            float dsr = pr[e] * (hdp - dlt[sub][h][1][row]);
            dsco += dsc * p.scale * sc[e];
            T16[0][h][row][col] = (f16)(dsc * p.scale * kscore);
            T16[1][h][row][col] = (f16)(dsr * p.scale);
            T16[2][h][row][col] = (f16)(0.5f * (pc[e] + pr[e]));
          }
          __syncthreads();
          // ---- products that reduce over keys: dq^ += T K^ ----
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int ko = ks * 16 + half * 8;
            f16x8 atc = *(const f16x8*)(&T16[0][h][col][ko]);
            f16x8 atr = *(const f16x8*)(&T16[1][h][col][ko]);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              long o = (long)(h * DP + n * 32 + col) * p.Npad + key0 + ko;
              dqc[sub][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(atc, *(const f16x8*)(p.kt_c + o), dqc[sub][n], 0, 0, 0);
              dqr[sub][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(atr, *(const f16x8*)(p.kt_r + o), dqr[sub][n], 0, 0, 0);
            }
          }
          // ---- products that reduce over query rows: dk^ += T^T q^,  dv += P^T dO + A^T dsoft ----
#pragma unroll
          for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              const int kb = ks * 16 + 8 * (g >> 1), mb = 16 * (g & 1);
              long ob = ((long)tile * p.WP + h * DP + n * 32 + col) * 32 + ks * 16 + half * 8;
              f16x8 tcT = tr_frag16(&T16[0][h][kb][mb], HST, lane);
              f16x8 trT = tr_frag16(&T16[1][h][kb][mb], HST, lane);
              f16x8 pT = tr_frag16(&T16[2][h][kb][mb], HST, lane);
              f16x8 acT = tr_frag16(&A16[0][kb][mb], HST, lane);
              akc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tcT, *(const f16x8*)(p.qT_c + ob), akc[n], 0, 0, 0);
              akr[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(trT, *(const f16x8*)(p.qT_r + ob), akr[n], 0, 0, 0);
              avc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pT, *(const f16x8*)(p.gOT_c + ob), avc[n], 0, 0, 0);
              avc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(acT, *(const f16x8*)(p.gST_c + ob), avc[n], 0, 0, 0);
              if (NEED_REG) {
                f16x8 arT = tr_frag16(&A16[1][kb][mb], HST, lane);
                avr[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pT, *(const f16x8*)(p.gOT_r + ob), avr[n], 0, 0, 0);
                avr[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(arT, *(const f16x8*)(p.gST_r + ob), avr[n], 0, 0, 0);
              }
            }
          }
          __syncthreads();
        }
      };
      if (paired) {
        visit(std::integral_constant<int, 0>{});
        visit(std::integral_constant<int, 1>{});
      } else if (first == 0) {
        visit(std::integral_constant<int, 0>{});
      } else {
        visit(std::integral_constant<int, 1>{});
      }
      // ---- one flush per key block for the whole group ----
      dsco += __shfl_xor(dsco, 32, 64);
      if (half == 0 && valid) atomicAdd(p.dscore + key, dsco);
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int krow = key0 + (e & 3) + 8 * (e >> 2) + 4 * half;
          if (krow >= lo && krow < hi) {
            long o = (long)krow * p.WP + h * DP + n * 32 + col;
            atomicAdd(p.dkn_c + o, akc[n][e]);
            atomicAdd(p.dkn_r + o, akr[n][e]);
            atomicAdd(p.dv_c + o, avc[n][e]);
            if (NEED_REG) atomicAdd(p.dv_r + o, avr[n][e]);
          }
        }
    }
  }
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    if (sub == 0 || has_b) {
      const McaTile t = sub ? tb : ta;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          int row = (e & 3) + 8 * (e >> 2) + 4 * half;
          if (row < t.nrows) {
            long o = (long)(t.row0 + row) * p.WP + h * DP + n * 32 + col;
            atomicAdd(p.dqn_c + o, dqc[sub][n][e]);
            atomicAdd(p.dqn_r + o, dqr[sub][n][e]);
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------ split passes (mode 2)
// Pass 2 without atomics and without workgroup barriers.  Everything that couples the four heads of a (query tile, key block)
// pair -- dM and the sim_round2 weights Ac / Ar -- is written once by pass 1 (mca_bwd_core_kernel<.., 1, true>: 12 KB per
// pair, ~100 MB per call at 16 x 500 proposals, far below what HBM moves in the time the pair's MFMAs take).  With those in
// memory a head is independent of the others, so a WAVE owns its output and keeps it in registers:
//   dq  pass (mca_bwd_dq_kernel):  wave = (query tile, head, split of the key blocks); S^T = K Q^T puts the query on the
//        lane axis, so dS^T leaves the MFMA already in A-operand order for dq = dS K (no LDS transpose);
//   dkv pass (mca_bwd_dkv_kernel): wave = (32-key block, head, split of the query tiles); S = Q K^T puts the key on the
//        lane axis, and P / dS / Ac / Ar are A operands of dk = dS^T Q, dv = P^T dO + A^T dsoft as they are.
// The contraction index of those second products is the C-register order of the first (rows 8i + 4*half + j), and the B
// operand is fetched in the same order from the transposed copies (k^T, per-tile q^T / dO^T / dsoft^T): two 8-byte loads.
// Splits write separate partial sums; mca_bwd_finish_kernel adds them in a fixed order (deterministic, nothing to zero).
__global__ void mca_bwd_rowstat_kernel(const McaTile* __restrict__ tiles, const float* __restrict__ stats,
                                       const float* __restrict__ delta_part, int S, int N_L, float* __restrict__ rowstat) {
  const int tile = blockIdx.x, tid = threadIdx.x, h = tid >> 6, br = (tid >> 5) & 1, row = tid & 31;
  const McaTile t = tiles[tile];
  const bool ok = row < t.nrows;
  const float l = ok ? stats[((long)h * 2 + br) * N_L + t.row0 + row] : 0.f;
  float dsum = 0.f;
  if (ok)
    for (int ss = 0; ss < S; ++ss) dsum += delta_part[(((long)ss * MCA_HEADS + h) * 2 + br) * N_L + t.row0 + row];
  float* o = rowstat + ((long)tile * MCA_HEADS + h) * 128;
  o[br * 32 + row] = l > 0.f ? 1.0f / l : 0.f;
  o[(2 + br) * 32 + row] = dsum;
}

__device__ __forceinline__ f16x8 ld_2x4(const f16* p) {   // 4 halves at p, 4 at p + 8 (the C-register row order of a 16-slab)
  f16x4 lo = *(const f16x4*)p, hi = *(const f16x4*)(p + 8);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int DP, bool NEED_REG>
__global__ __launch_bounds__(256) void mca_bwd_dq_kernel(McaBwdParams p) {
  constexpr int KS = DP / 16, NT = DP / 32;
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int tile = blockIdx.x, s = blockIdx.y;
  const McaTile t = p.tiles[tile];
  const int n_local = ((t.loc_end + 31) >> 5) - (t.loc_start >> 5);
  const int n_glob = ((p.N + 31) >> 5) - (p.N_L >> 5);
  const int nblk = n_local + n_glob;
  const int qi = lane & 31, half = lane >> 5;
  const int qrow = t.row0 + (qi < t.nrows ? qi : 0);
  const float* rs = p.rowstat + ((long)tile * MCA_HEADS + h) * 128;
  const float il_c = rs[qi], il_r = rs[32 + qi], dl_c = rs[64 + qi], dl_r = rs[96 + qi];
  f16x8 qc[KS], qr[KS], gOc[KS], gOr[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const long o = (long)qrow * p.WP + h * DP + ks * 16 + half * 8;
    qc[ks] = *(const f16x8*)(p.qn_c + o);
    qr[ks] = *(const f16x8*)(p.qn_r + o);
    gOc[ks] = *(const f16x8*)(p.gO_c + o);
    if (NEED_REG) gOr[ks] = *(const f16x8*)(p.gO_r + o);
  }
  f32x16 dqc[NT], dqr[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) dqc[n][e] = dqr[n][e] = 0.f;

  for (int j = s; j < nblk; j += p.S2) {
    int key0, lo, hi;
    block_range(t, p.N_L, p.N, j, key0, lo, hi, n_local);
    const long blk = ((long)tile * p.maxblk + j) * 1024;
    f32x16 sc, sr, gvc, gvr;
#pragma unroll
    for (int e = 0; e < 16; ++e) sc[e] = sr[e] = gvc[e] = gvr[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {      // rows = keys (A), columns = queries (B)
      const long o = (long)(key0 + qi) * p.WP + h * DP + ks * 16 + half * 8;
      sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.kn_c + o), qc[ks], sc, 0, 0, 0);
      sr = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.kn_r + o), qr[ks], sr, 0, 0, 0);
      gvc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.vraw_c + o), gOc[ks], gvc, 0, 0, 0);
      if (NEED_REG) gvr = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.vraw_r + o), gOr[ks], gvr, 0, 0, 0);
    }
    f16x8 tc[2], tr[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kb = key0 + 8 * i + 4 * half;
      const f32x4 dm = *(const f32x4*)(p.dMq + blk + qi * 32 + 8 * i + 4 * half);
      const f32x4 ks4 = *(const f32x4*)(p.score_pad + kb);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int e = 4 * i + jj;
        const bool valid = kb + jj >= lo && kb + jj < hi;
        const float pc = valid ? __expf(p.scale * ks4[jj] * sc[e] - p.scale) * il_c : 0.f;
        const float pr = valid ? __expf(p.scale * sr[e] - p.scale) * il_r : 0.f;
        const float hdp = 0.5f * (gvc[e] + gvr[e] + 0.25f * dm[jj]);
        tc[i >> 1][(i & 1) * 4 + jj] = (f16)(pc * (hdp - dl_c) * p.scale * ks4[jj]);
        tr[i >> 1][(i & 1) * 4 + jj] = (f16)(pr * (hdp - dl_r) * p.scale);
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const long o = (long)(h * DP + n * 32 + qi) * p.Npad + key0 + 16 * m + 4 * half;
        dqc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tc[m], ld_2x4(p.kt_c + o), dqc[n], 0, 0, 0);
        dqr[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tr[m], ld_2x4(p.kt_r + o), dqr[n], 0, 0, 0);
      }
  }
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

template <int DP, bool NEED_REG>
__global__ __launch_bounds__(256) void mca_bwd_dkv_kernel(McaBwdParams p, int ntiles) {
  constexpr int KS = DP / 16, NT = DP / 32;
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int b = blockIdx.x, s = blockIdx.y;
  const int key0 = b * 32, col = lane & 31, half = lane >> 5, key = key0 + col;
  const int gb0 = p.N_L >> 5, gb1 = (p.N + 31) >> 5;
  const bool is_glob = b >= gb0 && b < gb1;
  const float kscore = p.score_pad[key];
  f16x8 kc[KS], kr[KS], vc[KS], vr[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const long o = (long)key * p.WP + h * DP + ks * 16 + half * 8;
    kc[ks] = *(const f16x8*)(p.kn_c + o);
    kr[ks] = *(const f16x8*)(p.kn_r + o);
    vc[ks] = *(const f16x8*)(p.vraw_c + o);
    if (NEED_REG) vr[ks] = *(const f16x8*)(p.vraw_r + o);
  }
  f32x16 akc[NT], akr[NT], avc[NT], avr[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) akc[n][e] = akr[n][e] = avc[n][e] = avr[n][e] = 0.f;
  float dsco = 0.f;

  auto visit = [&](const int tt, const McaTile& t, const int j, const int lo, const int hi) __attribute__((always_inline)) {
    const bool valid = key >= lo && key < hi;
    const int qrow = t.row0 + (col < t.nrows ? col : 0);
    f32x16 sc, sr, gvc, gvr;
#pragma unroll
    for (int e = 0; e < 16; ++e) sc[e] = sr[e] = gvc[e] = gvr[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {      // rows = queries (A), columns = keys (B)
      const long o = (long)qrow * p.WP + h * DP + ks * 16 + half * 8;
      sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.qn_c + o), kc[ks], sc, 0, 0, 0);
      sr = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.qn_r + o), kr[ks], sr, 0, 0, 0);
      gvc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.gO_c + o), vc[ks], gvc, 0, 0, 0);
      if (NEED_REG) gvr = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8*)(p.gO_r + o), vr[ks], gvr, 0, 0, 0);
    }
    const long blk = ((long)tt * p.maxblk + j) * 1024 + col * 32;
    const float* rs = p.rowstat + ((long)tt * MCA_HEADS + h) * 128;
    f16x8 tc[2], tr[2], pp[2], ac[2], ar[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r0 = 8 * i + 4 * half;
      const f32x4 dm = *(const f32x4*)(p.dMk + blk + r0);
      const f16x4 a_c = *(const f16x4*)(p.Ack + blk + r0);
      f16x4 a_r = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
      if (NEED_REG) a_r = *(const f16x4*)(p.Ark + blk + r0);
      const f32x4 il_c = *(const f32x4*)(rs + r0), il_r = *(const f32x4*)(rs + 32 + r0);
      const f32x4 dl_c = *(const f32x4*)(rs + 64 + r0), dl_r = *(const f32x4*)(rs + 96 + r0);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int e = 4 * i + jj, u = (i & 1) * 4 + jj;
        const float pc = valid ? __expf(p.scale * kscore * sc[e] - p.scale) * il_c[jj] : 0.f;   // il = 0 on padded rows
        const float pr = valid ? __expf(p.scale * sr[e] - p.scale) * il_r[jj] : 0.f;
        const float hdp = 0.5f * (gvc[e] + gvr[e] + 0.25f * dm[jj]);
        const float dsc = pc * (hdp - dl_c[jj]), dsr = pr * (hdp - dl_r[jj]);
        dsco += dsc * p.scale * sc[e];
        tc[i >> 1][u] = (f16)(dsc * p.scale * kscore);
        tr[i >> 1][u] = (f16)(dsr * p.scale);
        pp[i >> 1][u] = (f16)(0.5f * (pc + pr));
        ac[i >> 1][u] = a_c[jj];
        ar[i >> 1][u] = a_r[jj];
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const long o = ((long)tt * p.WP + h * DP + n * 32 + col) * 32 + 16 * m + 4 * half;
        akc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tc[m], ld_2x4(p.qT_c + o), akc[n], 0, 0, 0);
        akr[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tr[m], ld_2x4(p.qT_r + o), akr[n], 0, 0, 0);
        avc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pp[m], ld_2x4(p.gOT_c + o), avc[n], 0, 0, 0);
        avc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac[m], ld_2x4(p.gST_c + o), avc[n], 0, 0, 0);
        if (NEED_REG) {
          avr[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pp[m], ld_2x4(p.gOT_r + o), avr[n], 0, 0, 0);
          avr[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[m], ld_2x4(p.gST_r + o), avr[n], 0, 0, 0);
        }
      }
  };

  for (int tt = s; tt < ntiles; tt += p.S3) {
    const McaTile t = p.tiles[tt];
    const int lb0 = t.loc_start >> 5, lb1 = (t.loc_end + 31) >> 5;
    if (b >= lb0 && b < lb1) visit(tt, t, b - lb0, t.loc_start, t.loc_end);
    if (is_glob) visit(tt, t, lb1 - lb0 + b - gb0, p.N_L, p.N);
  }
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
      ovc[o] = avc[n][e];
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
                                      const float* __restrict__ dscore_part, int nds, int Npad, float* __restrict__ dscore) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  int n = wave >> 1, br = wave & 1;
  if (n >= N) return;
  const int WP = MCA_HEADS * DP;
  if (dscore_part && br == 0) {     // split passes: dscore = fixed-order sum of the (split, head) partials
    float a = 0.f;
    for (int i = lane; i < nds; i += 64) a += dscore_part[(long)i * Npad + n];
    a = wave_sum(a);
    if (lane == 0) dscore[n] = a;
  }
  const f16* q = br ? q_r : q_c;
  const f16* kv = br ? kv_r : kv_c;
  const float* dqn = br ? dqn_r : dqn_c;
  const float* dkn = br ? dkn_r : dkn_c;
  const float* dv = br ? dv_r : dv_c;
  const f16* dvloc = br ? dvloc_r : dvloc_c;
  f16* dq = br ? dq_r : dq_c;
  f16* dkv = br ? dkv_r : dkv_c;
  const bool vgrad = (br == 0) || need_reg;
  for (int h = 0; h < MCA_HEADS; ++h) {
    float kx[2] = {0.f, 0.f}, qx[2] = {0.f, 0.f}, dk[2] = {0.f, 0.f}, dqq[2] = {0.f, 0.f};
    float sk = 0.f, sq = 0.f, kd = 0.f, qd = 0.f;
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
      int j = lane + rnd * 64;
      if (j < d) {
        kx[rnd] = (float)kv[(long)n * 2 * W + h * d + j];
        for (int ss = 0; ss < nsk; ++ss) dk[rnd] += dkn[ss * dk_stride + (long)n * WP + h * DP + j];
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
static int g_mca_bwd_pair = 2;
// Pass 2 of the backward.  2 (default): split passes -- dM / Ac / Ar blocks through memory, a wave-private dq pass and a
// key-stationary dk / dv pass, no atomics (falls back to 1 when the blocks would need more than 24 GB); 1: query-stationary,
// two query tiles of a frame per workgroup, dk / dv by float atomics; 0: the same with one tile per workgroup (the
// cross-checks).  Returns the previous setting.
extern "C" int tscd_mca_backward_set_paired(int on) { int o = g_mca_bwd_pair; g_mca_bwd_pair = on; return o; }

namespace {
struct McaSplitPlan { bool on; int maxblk, nkb, S2, S3; long blocks; };
McaSplitPlan mca_split_plan(int N, int N_L, int ntiles) {
  McaSplitPlan pl;
  pl.maxblk = (N + 31) / 32 + 2;                     // >= blocks of one frame + global blocks of any tile
  pl.nkb = (N + 31) / 32;
  pl.blocks = (long)ntiles * pl.maxblk;
  pl.on = pl.blocks * 1024 * 12 <= (24L << 30);
  auto clampi = [](long v, long lo, long hi) { return (int)(v < lo ? lo : (v > hi ? hi : v)); };
  pl.S2 = clampi((2048 + ntiles * 4 - 1) / (ntiles * 4), 1, 16);
  pl.S3 = clampi((2048 + pl.nkb * 4 - 1) / (pl.nkb * 4), 1, ntiles < 8 ? ntiles : 8);
  return pl;
}
}  // namespace

extern "C" long tscd_mca_backward_workspace_bytes(int N, int N_L, int W, int S, int ntiles) {
  int d = W / MCA_HEADS, DP = (d + 31) / 32 * 32, WP = MCA_HEADS * DP;
  long Npad = (N + 31) / 32 * 32 + 32;
  long halves = 2L * N_L * WP + 4L * Npad * WP + 2L * WP * Npad          // forward prep layouts
                + 2L * Npad * WP + 2L * WP * Npad                       // vraw, kt
                + 4L * N_L * WP + 6L * ntiles * WP * 32;                // gO/gS padded, qT/gOT/gST per tile
  const McaSplitPlan pl = mca_split_plan(N, N_L, ntiles);
  const long nsq = pl.on ? pl.S2 : 1, nsk = pl.on ? pl.S3 : 1;
  long floats = 2L * N_L + (long)S * MCA_HEADS * 2 * N_L + nsq * 2L * N_L * WP + nsk * 4L * Npad * WP;
  if (pl.on) {
    floats += Npad + (long)ntiles * MCA_HEADS * 128 + (long)pl.S3 * MCA_HEADS * Npad + 2L * pl.blocks * 1024;
    halves += 2L * pl.blocks * 1024;
  } else {
    floats += Npad;
  }
  return halves * 2 + floats * 4 + 16384;
}

// defined in mca.hip
void mca_launch_prep(hipStream_t st, const f16* q_c, const f16* kv_c, const f16* q_r, const f16* kv_r, int N, int N_L, int Npad, int W,
                     int d, int DP, f16* qn_c, f16* kn_c, f16* vn_c, f16* vt_c, f16* qn_r, f16* kn_r, f16* vn_r, f16* vt_r, f16* xcat_c,
                     f16* xcat_r);

// dxcat_* : gradient w.r.t. the [N_L, 2W] concat (first W = attention output, second W = own value rows);
// dsoft_* : gradient w.r.t. the first W columns of the [N_L, 3W] concat (row stride ld_dsoft); soft_* the forward values.
// Outputs: dq_* [N_L, W], dkv_* [N, 2W] fp16; dscore [N] fp32 (zeroed here).
extern "C" int tscd_mca_backward(const void* q_c, const void* kv_c, const void* q_r, const void* kv_r, const void* score,
                                 const int* tiles, int ntiles, int N, int N_L, int W, int S, int need_reg, float scale,
                                 float sim_thresh, float conf_thresh, const float* stats, const void* dxcat_c,
                                 const void* dxcat_r, const void* dsoft_c, const void* dsoft_r, int ld_dsoft, const void* soft_c,
                                 const void* soft_r, int ld_soft, void* workspace, long workspace_bytes, void* dq_c, void* dkv_c,
                                 void* dq_r, void* dkv_r, float* dscore, void* stream) {
  TSCD_CHECK_ARG(q_c && kv_c && q_r && kv_r && score && tiles && stats && dxcat_c && dsoft_c && soft_c && workspace && dq_c &&
                     dkv_c && dq_r && dkv_r && dscore, "mca_backward: null pointer");
  TSCD_CHECK_ARG(!need_reg || (dxcat_r && dsoft_r && soft_r), "mca_backward: reg gradients missing");
  TSCD_CHECK_ARG(W % MCA_HEADS == 0 && W % 8 == 0 && N_L > 0 && N >= N_L && ntiles > 0 && S > 0 && S <= 64, "mca_backward: bad sizes");
  const int d = W / MCA_HEADS, DP = (d + 31) / 32 * 32, WP = MCA_HEADS * DP;
  TSCD_CHECK_ARG(DP <= 96, "mca_backward: head dim %d not supported (max 96)", d);
  TSCD_CHECK_ARG(workspace_bytes >= tscd_mca_backward_workspace_bytes(N, N_L, W, S, ntiles), "mca_backward: workspace too small");
  const int Npad = (N + 31) / 32 * 32 + 32;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  auto take = [&](long bytes) { char* r = ws; ws += (bytes + 255) / 256 * 256; return r; };
  f16* qn_c = (f16*)take(2L * N_L * WP); f16* qn_r = (f16*)take(2L * N_L * WP);
  f16* kn_c = (f16*)take(2L * Npad * WP); f16* kn_r = (f16*)take(2L * Npad * WP);
  f16* vn_c = (f16*)take(2L * Npad * WP); f16* vn_r = (f16*)take(2L * Npad * WP);
  f16* vt_c = (f16*)take(2L * WP * Npad); f16* vt_r = (f16*)take(2L * WP * Npad);
  f16* vraw_c = (f16*)take(2L * Npad * WP); f16* vraw_r = (f16*)take(2L * Npad * WP);
  f16* kt_c = (f16*)take(2L * WP * Npad); f16* kt_r = (f16*)take(2L * WP * Npad);
  f16* gO_c = (f16*)take(2L * N_L * WP); f16* gO_r = (f16*)take(2L * N_L * WP);
  f16* gS_c = (f16*)take(2L * N_L * WP); f16* gS_r = (f16*)take(2L * N_L * WP);
  long tb = 2L * ntiles * WP * 32;
  f16* qT_c = (f16*)take(tb); f16* qT_r = (f16*)take(tb);
  f16* gOT_c = (f16*)take(tb); f16* gOT_r = (f16*)take(tb); f16* gST_c = (f16*)take(tb); f16* gST_r = (f16*)take(tb);
  float* gdot = (float*)take(4L * 2 * N_L);
  float* delta_part = (float*)take(4L * S * MCA_HEADS * 2 * N_L);
  const McaSplitPlan pl = mca_split_plan(N, N_L, ntiles);
  const bool split = g_mca_bwd_pair == 2 && pl.on;
  const int nsq = split ? pl.S2 : 1, nsk = split ? pl.S3 : 1;
  const long dq_stride = (long)N_L * WP, dk_stride = (long)Npad * WP;
  // accumulators: [split][rows][WP] per array; atomic targets (zeroed together with dscore) unless the passes are split
  long acc_floats = nsq * 2L * dq_stride + nsk * 4L * dk_stride;
  float* acc = (float*)take(4L * acc_floats);
  float* score_pad = (float*)take(4L * Npad);
  float *rowstat = nullptr, *dscore_part = nullptr, *dMq = nullptr, *dMk = nullptr;
  f16 *Ack = nullptr, *Ark = nullptr;
  if (split) {
    rowstat = (float*)take(4L * ntiles * MCA_HEADS * 128);
    dscore_part = (float*)take(4L * pl.S3 * MCA_HEADS * Npad);
    dMq = (float*)take(4L * pl.blocks * 1024); dMk = (float*)take(4L * pl.blocks * 1024);
    Ack = (f16*)take(2L * pl.blocks * 1024); Ark = (f16*)take(2L * pl.blocks * 1024);
  }
  TSCD_CHECK_ARG(ws - (char*)workspace <= workspace_bytes, "mca_backward: workspace carve-out overflow");
  float* dqn_c = acc; float* dqn_r = dqn_c + nsq * dq_stride;
  float* dkn_c = dqn_r + nsq * dq_stride; float* dkn_r = dkn_c + nsk * dk_stride;
  float* dv_c = dkn_r + nsk * dk_stride; float* dv_r = dv_c + nsk * dk_stride;
  if (!split) {
    (void)hipMemsetAsync(acc, 0, 4L * acc_floats, st);
    (void)hipMemsetAsync(dscore, 0, 4L * N, st);
  }

  {
    long waves = 2L * Npad;
    unsigned grid = (unsigned)((waves * 64 + 255) / 256);
    mca_launch_prep(st, (const f16*)q_c, (const f16*)kv_c, (const f16*)q_r, (const f16*)kv_r, N, N_L, Npad, W, d, DP, qn_c, kn_c, vn_c,
                    vt_c, qn_r, kn_r, vn_r, vt_r, nullptr, nullptr);
    hipLaunchKernelGGL(mca_bwd_layout_kernel, dim3(grid), dim3(256), 0, st, kn_c, kn_r, (const f16*)kv_c, (const f16*)kv_r, N, Npad,
                       W, d, DP, vraw_c, vraw_r, kt_c, kt_r, (const f16*)score, score_pad);
    hipLaunchKernelGGL(mca_bwd_tiles_kernel, dim3(ntiles), dim3(256), 0, st, (const McaTile*)tiles, ntiles, qn_c, qn_r,
                       (const f16*)dxcat_c, 2 * W, (const f16*)dxcat_r, (const f16*)dsoft_c, ld_dsoft, (const f16*)dsoft_r,
                       (const f16*)soft_c, (const f16*)soft_r, ld_soft, N_L, W, d, DP, need_reg, qT_c, qT_r, gO_c, gO_r, gS_c, gS_r,
                       gOT_c, gOT_r, gST_c, gST_r, gdot);
    TSCD_CHECK_LAUNCH("mca_backward prep");
  }
  McaBwdParams p;
  p.qn_c = qn_c; p.qn_r = qn_r; p.kn_c = kn_c; p.kn_r = kn_r; p.vn_c = vn_c; p.vn_r = vn_r;
  p.vraw_c = vraw_c; p.vraw_r = vraw_r; p.kt_c = kt_c; p.kt_r = kt_r; p.qT_c = qT_c; p.qT_r = qT_r;
  p.gO_c = gO_c; p.gO_r = gO_r; p.gS_c = gS_c; p.gS_r = gS_r; p.gOT_c = gOT_c; p.gOT_r = gOT_r; p.gST_c = gST_c; p.gST_r = gST_r;
  p.score = (const f16*)score; p.tiles = (const McaTile*)tiles; p.stats = stats; p.gdot = gdot; p.delta_part = delta_part;
  p.dqn_c = dqn_c; p.dqn_r = dqn_r; p.dkn_c = dkn_c; p.dkn_r = dkn_r; p.dv_c = dv_c; p.dv_r = dv_r; p.dscore = dscore;
  p.N = N; p.N_L = N_L; p.Npad = Npad; p.WP = WP; p.S = S;
  p.scale = scale; p.sim_thresh = sim_thresh; p.conf_thresh = conf_thresh;
  p.dMq = dMq; p.dMk = dMk; p.Ack = Ack; p.Ark = Ark; p.score_pad = score_pad; p.rowstat = rowstat; p.dscore_part = dscore_part;
  p.maxblk = pl.maxblk; p.S2 = pl.S2; p.S3 = pl.S3; p.dq_stride = dq_stride; p.dk_stride = dk_stride;
  dim3 grid(ntiles, S), block(256);
  dim3 pgrid((ntiles + 1) / 2, S);      // pass 2: two same-frame query tiles per workgroup (mca_bwd_pair_kernel)
#define MCAS(DPV, REG)                                                                                             \
  do {                                                                                                             \
    hipLaunchKernelGGL((mca_bwd_core_kernel<DPV, REG, 1, true>), grid, block, 0, st, p);                           \
    hipLaunchKernelGGL(mca_bwd_rowstat_kernel, dim3(ntiles), block, 0, st, p.tiles, stats, delta_part, S, N_L, rowstat); \
    hipLaunchKernelGGL((mca_bwd_dq_kernel<DPV, REG>), dim3(ntiles, pl.S2), block, 0, st, p);                       \
    hipLaunchKernelGGL((mca_bwd_dkv_kernel<DPV, REG>), dim3(pl.nkb, pl.S3), block, 0, st, p, ntiles);              \
  } while (0)
#define MCAB(DPV)                                                                            \
  do {                                                                                       \
    if (split) {                                                                             \
      if (need_reg) MCAS(DPV, true); else MCAS(DPV, false);                                  \
    } else if (need_reg) {                                                                   \
      hipLaunchKernelGGL((mca_bwd_core_kernel<DPV, true, 1>), grid, block, 0, st, p);        \
      if (g_mca_bwd_pair) hipLaunchKernelGGL((mca_bwd_pair_kernel<DPV, true>), pgrid, block, 0, st, p, ntiles); \
      else hipLaunchKernelGGL((mca_bwd_core_kernel<DPV, true, 2>), grid, block, 0, st, p);   \
    } else {                                                                                 \
      hipLaunchKernelGGL((mca_bwd_core_kernel<DPV, false, 1>), grid, block, 0, st, p);       \
      if (g_mca_bwd_pair) hipLaunchKernelGGL((mca_bwd_pair_kernel<DPV, false>), pgrid, block, 0, st, p, ntiles); \
      else hipLaunchKernelGGL((mca_bwd_core_kernel<DPV, false, 2>), grid, block, 0, st, p);  \
    }                                                                                        \
  } while (0)
  if (DP == 32) MCAB(32);
  else if (DP == 64) MCAB(64);
  else MCAB(96);
#undef MCAB
#undef MCAS
  TSCD_CHECK_LAUNCH("mca_backward core");
  {
    long waves = 2L * N;
    hipLaunchKernelGGL(mca_bwd_finish_kernel, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, st, (const f16*)q_c,
                       (const f16*)kv_c, (const f16*)q_r, (const f16*)kv_r, dqn_c, dqn_r, dkn_c, dkn_r, dv_c, dv_r,
                       (const f16*)dxcat_c + W, need_reg ? (const f16*)dxcat_r + W : (const f16*)dxcat_c + W, 2 * W, N, N_L, W, d, DP,
                       need_reg, (f16*)dq_c, (f16*)dkv_c, (f16*)dq_r, (f16*)dkv_r, nsq, dq_stride, nsk, dk_stride, dscore_part,
                       pl.S3 * MCA_HEADS, Npad, dscore);
    TSCD_CHECK_LAUNCH("mca_backward finish");
  }
  return TSCD_OK;
}
