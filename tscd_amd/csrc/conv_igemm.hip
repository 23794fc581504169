// Implicit-GEMM convolution for gfx950 (MI355X): NHWC fp16 activations, fp32 MFMA accumulation.
//
// Replaces the conv of BaseConv / nn.Conv2d on the TSCD path (reference: yolox/models/network_blocks.py:36-61,
// tscd_head.py:153-281, surrounding_extraction.py:223-232) and, with KH=KW=1 / H=1, the nn.Linear GEMMs of the
// proposal stage.  GEMM view:  M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin.
//   A[m][k] = x[b, ho*s+kh-p, wo*s+kw-p, cin]   (gathered on the fly, zero outside the image)
//   B[k][n] = wpk[n][k]                         (weights pre-packed [CoutPad][Kpad], K-contiguous, zero padded)
// Workgroup = 256 threads = 4 waves; block tile BM x BN x 32, each wave owns (BM/WAVES_M) x (BN/WAVES_N) as
// 32x32 v_mfma_f32_32x32x16_f16 tiles.  A/B k-tiles are staged global -> VGPR -> LDS (double buffered, one
// barrier per k-tile); LDS rows are 32 halves + 8 pad (80 B) which makes both the ds_write_b128 staging and the
// ds_read_b128 fragment reads conflict-free (MI355X LDS: 64 banks, 16-lane groups for b128).
// Block ids are remapped so that the 8 XCDs each get a contiguous range of M-tiles (all N-tiles of an M-tile
// share one L2, the packed weights stay L2-resident everywhere).
//
// Epilogues: raw fp16 store (+ per-channel partial sum / sum-of-squares for train-mode BatchNorm, written
// per M-tile so the reduction order is fixed => deterministic), bias, bias+ReLU, folded-BN(+SiLU) for eval.
#include "common.h"
#include <type_traits>

struct ConvParams {
  const f16* x;
  const f16* w;
  f16* y;
  const float* bias;
  const float* scale;
  const float* shift;
  float* bn_partial;
  int B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
  int ldx, ldy;
  int K, Kpad, M;
  int epilogue;
  int mtiles, ntiles, cstride;
  int korder;        // 0: k runs (tap, cin) as packed; 1: (32-channel chunk, tap) -- variant 5's order (Cin % 32 == 0 only)
  // Grouped launch (G > 1: the second convs of the head's towers, which read different channel slices of one activation buffer and
  // write different slices of one output buffer): Cin / Cout / cstride are PER GROUP, ntiles = G * ntiles_g N-tiles in all.  Group g
  // reads channels [g*Cin, (g+1)*Cin) of x, the packed weight block g (wrows_g rows of Kpad), and writes channels [g*Cout, ...) of y;
  // bias / scale / shift are indexed by the concatenated channel g*Cout + n, BatchNorm partial rows by g*cstride + n.
  int G, ntiles_g, wrows_g;
};

// what a workgroup needs to know about its group: pointer bases and channel bases
struct GroupView {
  const f16* x;
  const f16* w;
  int cbase;     // first output channel of the group in y / bias / scale / shift
  int sbase;     // first row of the group in the BatchNorm partial table
};
__device__ __forceinline__ GroupView group_view(const ConvParams& p, int nta, int& nt) {
  const int g = p.G > 1 ? nta / p.ntiles_g : 0;
  nt = nta - g * p.ntiles_g;
  GroupView v;
  v.x = p.x + g * p.Cin;
  v.w = p.w + (long)g * p.wrows_g * p.Kpad;
  v.cbase = g * p.Cout;
  v.sbase = g * p.cstride;
  return v;
}

enum { EPI_RAW = 0, EPI_BIAS = 1, EPI_BIAS_RELU = 2, EPI_BN_SILU = 3, EPI_BN = 4, EPI_COST = 5 };

// ---- epilogue shared by both main-loop variants.  C layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// The finished fp16 tile is staged in LDS (row pitch BN+8 halves: the two half-waves of a ds_write land on disjoint bank
// quarters, the 16-B read-back of a row is conflict-free) and leaves as 16-byte stores, 16 lanes per 256-B row segment,
// instead of 64 two-byte stores per lane; the per-channel BatchNorm sums stay in the MFMA layout (in-lane over 16 rows).
// MF = 32: f32x16 accumulators of v_mfma_f32_32x32x16_f16 (layout above); MF = 16: f32x4 accumulators of v_mfma_f32_16x16x32_f16
// (col = lane&15, row = 4*(lane>>4) + reg) -- TM / TN count MF x MF tiles.
template <int BM, int BN, int WAVES_M, int WAVES_N, int TM, int TN, int MF = 32, typename ACC = f32x16>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, ACC (&acc)[TM][TN], f16* smem, int m0, int n0, int mt, int tid,
                                              int lane, int wm, int wn, int cbase = 0, int sbase = 0) {
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, CP = BN + 8, NT = 64 * WAVES_M * WAVES_N;
  constexpr int NE = MF == 32 ? 16 : 4;            // accumulator registers per tile
  static_assert(WM == TM * MF && WN == TN * MF, "wave tile");
  f16* ctile = smem;                               // [BM][CP]
  float* red = (float*)(smem + BM * CP);           // [WAVES_M][BN][2]
  __syncthreads();                                 // every wave is done with the staging buffers of the main loop
  float s1[TN], s2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) s1[j] = s2[j] = 0.f;
  // The epilogue kind is a launch constant: dispatch ONCE to a body specialised on it.  (With the comparisons left inside the
  // 16 x TM x TN element loop the compiler kept them there: ~1300 branches and 80 % of the halo kernel's code, 9 us per workgroup.)
  auto stage = [&](auto epic) __attribute__((always_inline)) {
    constexpr int EPI = decltype(epic)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int cl = wn * WN + j * MF + (lane & (MF - 1));
      const int n = n0 + cl;
      const bool nok = n < p.Cout;
      float bias = 0.f, sc = 1.f, sh = 0.f;
      if (nok) {
        if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) bias = p.bias[cbase + n];
        if constexpr (EPI == EPI_BN_SILU || EPI == EPI_BN) { sc = p.scale[cbase + n]; sh = p.shift[cbase + n]; }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int row = MF == 32 ? wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) : wm * WM + i * 16 + 4 * (lane >> 4) + e;
          float v = acc[i][j][e];
          f16 o;
          if constexpr (EPI == EPI_RAW) {
            o = (f16)v;
            float vf = (float)o;  // statistics of the fp16-rounded conv output, as the reference's BN sees it
            s1[j] += vf;
            s2[j] += vf * vf;
          } else if constexpr (EPI == EPI_BIAS) {
            o = (f16)(v + bias);
          } else if constexpr (EPI == EPI_BIAS_RELU) {
            o = (f16)fmaxf(v + bias, 0.f);
          } else if constexpr (EPI == EPI_COST) {
            // Hungarian cost of double_match_embds (tscd_matching.py:912-937): 1 - (cos_reg + cos_cls)/2, NaN -> 0
            float c = 1.0f - 0.5f * (float)(f16)v;
            o = (c != c) ? (f16)0.f : (f16)c;
          } else {
            // eval: conv (fp16) -> BN with running stats (fp16 out) -> SiLU (fp16 out)
            float t = (float)(f16)((float)(f16)v * sc + sh);
            o = (EPI == EPI_BN_SILU) ? (f16)(t * sigmoidf_(t)) : (f16)t;
          }
          ctile[row * CP + cl] = o;
        }
      }
    }
  };
  switch (p.epilogue) {
    case EPI_RAW: stage(std::integral_constant<int, EPI_RAW>{}); break;
    case EPI_BIAS: stage(std::integral_constant<int, EPI_BIAS>{}); break;
    case EPI_BIAS_RELU: stage(std::integral_constant<int, EPI_BIAS_RELU>{}); break;
    case EPI_COST: stage(std::integral_constant<int, EPI_COST>{}); break;
    case EPI_BN_SILU: stage(std::integral_constant<int, EPI_BN_SILU>{}); break;
    default: stage(std::integral_constant<int, EPI_BN>{}); break;
  }
  const bool stats = p.epilogue == EPI_RAW && p.bn_partial != nullptr;
  if (stats) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float a = s1[j] + __shfl_xor(s1[j], 32, 64);
      float b = s2[j] + __shfl_xor(s2[j], 32, 64);
      if constexpr (MF == 16) {
        a += __shfl_xor(a, 16, 64);
        b += __shfl_xor(b, 16, 64);
      }
      if (lane < MF) {
        int col = wn * WN + j * MF + lane;
        red[(wm * BN + col) * 2 + 0] = a;
        red[(wm * BN + col) * 2 + 1] = b;
      }
    }
  }
  __syncthreads();
  f16* const yg = p.y + cbase;
  const bool vec_ok = (p.Cout & 7) == 0 && (p.ldy & 7) == 0 && ((uintptr_t)yg & 15) == 0;
  if (vec_ok) {
    constexpr int CPR = BN / 8;                    // 16-byte chunks per tile row
#pragma unroll
    for (int it = 0; it < BM * CPR / NT; ++it) {
      const int idx = tid + it * NT;
      const int row = idx / CPR, ch = idx % CPR;
      const int m = m0 + row, n = n0 + ch * 8;
      if (m < p.M && n < p.Cout) *(f16x8*)(yg + (long)m * p.ldy + n) = *(const f16x8*)(ctile + row * CP + ch * 8);
    }
  } else {                                         // ragged Cout (prediction heads: 1 / 4 / NC channels, cost matrices)
    for (int idx = tid; idx < BM * BN; idx += NT) {
      const int row = idx / BN, cl = idx % BN;
      const int m = m0 + row, n = n0 + cl;
      if (m < p.M && n < p.Cout) yg[(long)m * p.ldy + n] = ctile[row * CP + cl];
    }
  }
  if (stats && tid < BN) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES_M; ++w) {
      a += red[(w * BN + tid) * 2 + 0];
      b += red[(w * BN + tid) * 2 + 1];
    }
    // channel-major [2][G * cstride][mtiles]: the reduction over M-tiles then reads contiguous floats
    p.bn_partial[(long)(sbase + n0 + tid) * p.mtiles + mt] = a;
    p.bn_partial[((long)p.G * p.cstride + sbase + n0 + tid) * p.mtiles + mt] = b;
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p) {
  constexpr int LDSS = 40;  // halves per LDS row (32 + 8 pad)
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
  constexpr int A_CH = BM * 4 / 256;
  constexpr int B_CH = (BN * 4 + 255) / 256;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) f16 smem[2 * (BM + BN) * LDSS];
  f16* As = smem;
  f16* Bs = smem + 2 * BM * LDSS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // XCD-aware, bijective block -> tile map (blocks b, b+8, ... share an XCD / L2).
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int nta = lid % p.ntiles, mt = lid / p.ntiles;
  int nt;
  const GroupView gv = group_view(p, nta, nt);
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread gather state for the A (im2col) tile: A_CH rows, one 8-half chunk each ----
  const int a_chunk = tid & 3;
  const f16* a_base[A_CH];
  int a_hi0[A_CH], a_wi0[A_CH];
  bool a_ok[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    int m = m0 + (tid >> 2) + i * 64;
    a_ok[i] = m < p.M;
    int mm = a_ok[i] ? m : 0;
    int b = mm / (p.Ho * p.Wo);
    int rem = mm - b * (p.Ho * p.Wo);
    int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    a_hi0[i] = ho * p.stride - p.pad;
    a_wi0[i] = wo * p.stride - p.pad;
    a_base[i] = gv.x + (long)b * p.H * p.W * p.ldx;
  }
  int cin = a_chunk * 8, kh = 0, kw = 0;
  while (cin >= p.Cin) {
    cin -= p.Cin;
    if (++kw == p.KW) { kw = 0; ++kh; }
  }
  // korder 1 (Cin % 32 == 0): k-tile kt = (32-channel chunk kt / ntaps, tap kt % ntaps); the weight columns of that tile
  // sit at tap * Cin + chunk * 32 of the packed row
  const bool chunk_major = p.korder == 1;
  const int ntaps = p.KH * p.KW;
  int cm_chunk = 0, cm_tap = 0;

  f16x8 a_reg[A_CH], b_reg[B_CH];
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  auto load_tiles = [&](int kt) {
    if (chunk_major) {
      kh = cm_tap / p.KW;
      kw = cm_tap - kh * p.KW;
      cin = cm_chunk * 32 + a_chunk * 8;
    }
    const int kcol = chunk_major ? cm_tap * p.Cin + cm_chunk * 32 : kt * 32;
    const bool kvalid = kh < p.KH;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      int hi = a_hi0[i] + kh, wi = a_wi0[i] + kw;
      bool ok = a_ok[i] && kvalid && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      a_reg[i] = ok ? *(const f16x8*)(a_base[i] + ((long)hi * p.W + wi) * p.ldx + cin) : zero8;
    }
#pragma unroll
    for (int j = 0; j < B_CH; ++j) {
      int idx = tid + j * 256;
      if (idx < BN * 4) b_reg[j] = *(const f16x8*)(gv.w + (long)(n0 + (idx >> 2)) * p.Kpad + kcol + (idx & 3) * 8);
    }
    if (chunk_major) {
      if (++cm_tap == ntaps) { cm_tap = 0; ++cm_chunk; }
    } else {
      cin += 32;
      while (cin >= p.Cin) {
        cin -= p.Cin;
        if (++kw == p.KW) { kw = 0; ++kh; }
      }
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i)
      *(f16x8*)(As + (buf * BM + (tid >> 2) + i * 64) * LDSS + a_chunk * 8) = a_reg[i];
#pragma unroll
    for (int j = 0; j < B_CH; ++j) {
      int idx = tid + j * 256;
      if (idx < BN * 4) *(f16x8*)(Bs + (buf * BN + (idx >> 2)) * LDSS + (idx & 3) * 8) = b_reg[j];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = p.Kpad >> 5;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tiles(kt + 1);  // global loads in flight under the MFMAs
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *(const f16x8*)(As + (buf * BM + wm * WM + i * 32 + (lane & 31)) * LDSS + ks * 16 + (lane >> 5) * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *(const f16x8*)(Bs + (buf * BN + wn * WN + j * 32 + (lane & 31)) * LDSS + ks * 16 + (lane >> 5) * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  conv_epilogue<BM, BN, WAVES_M, WAVES_N, TM, TN>(p, acc, smem, m0, n0, mt, tid, lane, wm, wn, gv.cbase, gv.sbase);
}

// ====================================================================================================================
// Main-loop variant 2 (Cin % 32 == 0): LDS-DMA staging with a 4-deep ring.
//   * every k-tile (128 x 32 A rows + BN x 32 B rows, 64 B per row) is fetched with global_load_lds_dwordx4: one
//     wave-instruction lands 16 rows x 64 B contiguously in LDS, the per-lane SOURCE address does the im2col gather
//     (out-of-image taps and rows >= M read a 16-byte zero block instead) -- no staging VGPRs, no ds_write pass;
//   * the LDS image is lane-linear, so bank conflicts are removed by an XOR swizzle applied on both sides: the lane
//     that fills 16-B slot s of row r fetches k-chunk s ^ ((r>>2)&3), and the fragment read of chunk c goes to slot
//     c ^ ((r>>2)&3) (16 consecutive rows x one chunk then cover all 64 banks exactly once);
//   * three k-tiles stay in flight across the (single) barrier of a k-step: counted s_waitcnt vmcnt(8) + raw s_barrier,
//     so a workgroup alone on a CU still hides the ~1 us L2/MALL latency (the 18x18 and 36x36 levels launch only
//     80-330 workgroups for 256 CUs).
// With Cin % 32 == 0 a k-tile never straddles a filter tap: tap / channel offset are wave-uniform scalars and each lane
// only tests one bit of its precomputed 9-bit tap-validity mask.
// ====================================================================================================================
__device__ __attribute__((aligned(16))) f16 g_zero_chunk[8];

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else static_assert(N < 0, "unsupported vmcnt");
}

// Tiles: 128 x 256 with 8 waves (Cout > 128: the A gather is fetched once for 256 output channels -- the per-CU vector
// memory path, not the MFMA, bounds the 128 x 128 tile at ~64 FLOP per loaded byte), 128 x 128 and 128 x 64 with 4 waves.
// Tile height BM: 128 by default; 192 x 128 when that puts one workgroup on most CUs in a single round (36x36 level: 216
// workgroups instead of 324 half-filling a second round), 64 x 128 when 128-row tiles leave most CUs empty (18x18 level).
template <int BM, int BN, int WAVES_N>
__device__ __forceinline__ void conv_igemm_glds_tile(const ConvParams& p, const int bid, const int nblk) {
  constexpr int NS = 4, WAVES_M = 2, NW = WAVES_M * WAVES_N, NT = 64 * NW;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
  constexpr int A_LD = BM / 16 / NW, B_LD = BN / 16 / NW;       // glds wave-instructions (16 rows each) per wave and k-tile
  static_assert(BM % (16 * NW) == 0 && BM % 64 == 0, "tile height");
  constexpr int LD = A_LD + B_LD;
  static_assert(A_LD >= 1 && B_LD >= 1, "tile / wave split");
  __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int nta = lid % p.ntiles, mt = lid / p.ntiles;
  int nt;
  const GroupView gv = group_view(p, nta, nt);
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-lane source state: slot s = lane&3 of row (lane>>2) in each 16-row piece; chunk = s ^ ((row>>2)&3) ----
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
  const f16* a_src[A_LD];
  unsigned a_mask[A_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int m = m0 + wave * (16 * A_LD) + i * 16 + (lane >> 2);
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int b = mm / (p.Ho * p.Wo);
    const int rem = mm - b * (p.Ho * p.Wo);
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    a_src[i] = gv.x + ((long)(b * p.H + hi0) * p.W + wi0) * p.ldx + chunk * 8;
    unsigned mask = 0;
    for (int kh = 0; kh < p.KH; ++kh)
      for (int kw = 0; kw < p.KW; ++kw)
        if (ok && (unsigned)(hi0 + kh) < (unsigned)p.H && (unsigned)(wi0 + kw) < (unsigned)p.W) mask |= 1u << (kh * p.KW + kw);
    a_mask[i] = mask;
  }
  const f16* b_src[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j)
    b_src[j] = gv.w + (long)(n0 + wave * (16 * B_LD) + j * 16 + (lane >> 2)) * p.Kpad + chunk * 8;

  const int nk = p.Kpad >> 5;
  int ld_tap = 0, ld_cin = 0, ld_kw = 0;     // wave-uniform position of the NEXT k-tile to issue
  long ld_tapoff = 0;
  auto issue = [&](int stage) {
    char* sbase = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const f16* src = ((a_mask[i] >> ld_tap) & 1u) ? a_src[i] + ld_tapoff + ld_cin : g_zero_chunk;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sbase + (wave * (16 * A_LD) + i * 16) * 64), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b_src[j],
                                       (__attribute__((address_space(3))) void*)(sbase + A_BYTES + (wave * (16 * B_LD) + j * 16) * 64), 16,
                                       0, 0);
      b_src[j] += 32;
    }
    ld_cin += 32;
    if (ld_cin == p.Cin) {
      ld_cin = 0;
      ++ld_tap;
      if (++ld_kw == p.KW) { ld_kw = 0; ld_tapoff += (long)(p.W - p.KW + 1) * p.ldx; } else ld_tapoff += p.ldx;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment read offsets (bytes inside a stage): row r, chunk c -> r*64 + ((c ^ ((r>>2)&3)) * 16)
  const int frow = lane & 31, fsw = (frow >> 2) & 3, fk = lane >> 5;
  int a_off[TM][2], b_off[TN][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < TM; ++i) a_off[i][ks] = (wm * WM + i * 32 + frow) * 64 + (((ks * 2 + fk) ^ fsw) * 16);
#pragma unroll
    for (int j = 0; j < TN; ++j) b_off[j][ks] = A_BYTES + (wn * WN + j * 32 + frow) * 64 + (((ks * 2 + fk) ^ fsw) * 16);
  }

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) issue(s);

  // Software pipeline: the fragments of tile kt+1 are fetched into the second register set while the MFMAs of tile kt run,
  // so the only exposed waits per k-step are the (2-tiles-early) vmcnt and the barrier.
  f16x8 fa[2][2][TM], fb[2][2][TN];     // [register set][ks][tile]
  auto wait_tile = [&](int t) {          // tile t landed (this wave's share); up to NS-2 younger tiles stay in flight
    const int younger = nk - 1 - t;
    if (younger >= NS - 2) wait_vmcnt<(NS - 2) * LD>();
    else if (younger == 1) wait_vmcnt<LD>();
    else wait_vmcnt<0>();
  };
  auto read_frags = [&](auto setc, int t) {
    constexpr int S = decltype(setc)::value;
    const char* sb = smem + (t % NS) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[S][ks][i] = *(const f16x8*)(sb + a_off[i][ks]);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[S][ks][j] = *(const f16x8*)(sb + b_off[j][ks]);
    }
  };
  auto mfmas = [&](auto setc) {
    constexpr int S = decltype(setc)::value;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][ks][i], fb[S][ks][j], acc[i][j], 0, 0, 0);
  };
  // one k-step that has a successor: fetch tile kt+1 into the other register set, then the MFMAs of tile kt (no control
  // flow between the reads and the MFMAs, or hipcc falls back to lgkmcnt(0) at the join and the overlap is lost)
  auto phase = [&](auto setc, int kt) {
    constexpr int S = decltype(setc)::value;
    __builtin_amdgcn_sched_barrier(0);                       // keep the previous step's MFMAs ahead of the wait
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): this wave's reads of tile kt are in registers ...
    wait_tile(kt + 1);
    __builtin_amdgcn_s_barrier();                            // ... everyone's are: tile kt's buffer may be refilled
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NS < nk) issue((kt + NS) % NS);
    read_frags(std::integral_constant<int, 1 - S>{}, kt + 1);
    mfmas(setc);
  };
  wait_tile(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (NS - 1 < nk) issue(NS - 1);
  read_frags(std::integral_constant<int, 0>{}, 0);
  int kt = 0;
  for (; kt + 2 < nk; kt += 2) {
    phase(std::integral_constant<int, 0>{}, kt);
    phase(std::integral_constant<int, 1>{}, kt + 1);
  }
  if (nk - kt == 2) {
    phase(std::integral_constant<int, 0>{}, kt);
    mfmas(std::integral_constant<int, 1>{});
  } else {
    mfmas(std::integral_constant<int, 0>{});
  }
  conv_epilogue<BM, BN, WAVES_M, WAVES_N, TM, TN>(p, acc, (f16*)smem, m0, n0, mt, tid, lane, wm, wn, gv.cbase, gv.sbase);
}
template <int BM, int BN, int WAVES_N>
__global__ __launch_bounds__(128 * WAVES_N, (BM == 64 ? 3 : 2)) void conv_igemm_glds_kernel(ConvParams p) {
  conv_igemm_glds_tile<BM, BN, WAVES_N>(p, blockIdx.x, gridDim.x);
}
// Job-table form: SEVERAL independent GEMMs of this variant (the nn.Linear layers of the proposal stage that do not depend on each other:
// MCA's four input projections, its two output pairs, TaskAligned's / CAFM's q / k / v) in ONE launch -- each alone is a few dozen
// workgroups and a launch of its own on a latency-bound stretch.  A workgroup finds its job by its index (uniform scalar compares).
#define CONV_MAX_JOBS 8
struct ConvJobTable { ConvParams job[CONV_MAX_JOBS]; int wg_begin[CONV_MAX_JOBS + 1]; int njobs; };
template <int BM, int BN, int WAVES_N>
__global__ __launch_bounds__(128 * WAVES_N, (BM == 64 ? 3 : 2)) void conv_igemm_glds_jobs_kernel(ConvJobTable t) {
  const int bid = blockIdx.x;
  int j = 0;
  for (int i = 1; i < t.njobs; ++i) j = bid >= t.wg_begin[i] ? i : j;
  const ConvParams p = t.job[j];
  conv_igemm_glds_tile<BM, BN, WAVES_N>(p, bid - t.wg_begin[j], t.wg_begin[j + 1] - t.wg_begin[j]);
}

// ====================================================================================================================
// Main-loop variant 5: 3x3 / stride 1 / 'same' convolutions with Cin % 32 == 0, Cout a multiple of 256 -- the head towers
// (33 % of all FLOPs at 72x72) and the 256 / 512-channel 3x3 convs of the backbone.  256 x 256 tile, 8 waves, INPUT HALO staged
// once per 32-channel chunk and re-used by all 9 taps.
//
// Why: variants 1-4 all measure 650-700 TFLOP/s on the dominant shape whatever their loop structure (lockstep, staggered,
// BK 32 / 64, 128- or 256-wide tiles): the limit is the volume that has to come in through the per-CU LDS-DMA path, ~42 GB/s
// per CU as used here (24 KB per 128x256x32 step: A 8 KB gathered again for every tap + B 16 KB).  An implicit GEMM fetches
// each input pixel 9 times (once per tap) and each weight row once per 128 output pixels.  Here
//   * the tile is 192 / 224 / 256 consecutive output pixels x 256 channels: weights are fetched once per 256 pixels (half the B volume);
//   * per 32-channel chunk the 256 + 2W + 2 input pixels the 9 taps touch (flattened pixel range [m0 - W - 1, m0 + 256 + W])
//     are staged ONCE (26 KB at W = 72, double-buffered) and tap (kh, kw) reads its A fragments at a row shift of kh*W + kw:
//     A volume / 4.5.  Taps that fall outside the image (left / right / top / bottom edge, other frame) are handled at
//     fragment-read time: the lane's LDS address is redirected to a zero block (one v_cndmask per read);
//   => 170 KB of LDS-DMA per 9 k-steps of a 256 x 256 tile instead of 432 KB for the same work with 128 x 256 tiles.
// LDS image (A halo and B ring): [16-row piece][4 chunks of 16 B][16 rows] -- a ds_read_b128 of 16 consecutive rows of one chunk
// is 256 contiguous bytes at any row offset (two runs that wrap by 1 KiB = 0 mod 64 banks), so every fragment read is
// conflict-free at every tap shift without a swizzle; the LDS-DMA destination stays lane-linear (lane l fills slot l of its
// piece: row l & 15, chunk l >> 4).
// K order is chunk-major ((cin chunk, tap) instead of (tap, cin chunk)): the same products summed in a different order, so
// results agree with variants 0-4 to fp32 rounding, not bit for bit; variant 0 takes the same order with
// tscd_conv_set_korder(1), and then the two ARE bit-identical (tests/test_gpu_conv.py).
// Schedule: the staggered wave groups of variant 4 (L / M segments of 12 fragment reads + 2-6 DMA pieces / 16 MFMAs); B ring of 6
// k-steps; the A halo of chunk c+1 is issued during the first two steps of chunk c, into the buffer chunk c-1 has left.
// ====================================================================================================================
template <int N> __device__ __forceinline__ void wait_vmcnt_v5() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}

// TM16 = 16-row MFMA tiles per wave: 6 -> 192-pixel tiles, 7 -> 224, 8 -> 256 (the host picks whichever needs the fewest rounds x rows
// for the launch's M: 224 for the 72x72 level's 512-channel launches -- 742 workgroups in 3 rounds where 192-row tiles take 4).  The tap loop is unrolled (tap is a compile-time constant in every step: shifts, validity
// bits and the halo issue slots fold) and every channel chunk but the last runs without tail logic (fixed DMA distance,
// fixed vmcnt): the L segment's instruction count, not the MFMAs, bounded the first version of this loop.
template <int TM16>
__global__ __launch_bounds__(512, 1) void conv_igemm_halo_kernel(ConvParams p, int HRP) {
  constexpr int MAXP = 4;                                // A pieces per wave: halo of <= 512 rows
  constexpr int BM = 32 * TM16, BN = 256, NS = 6, WAVES_M = 2, WAVES_N = 4, WM = 16 * TM16, WN = 64;
  constexpr int TN16 = 4;                                // TM16 x 4 tiles of 16 x 16 per wave: v_mfma_f32_16x16x32_f16, one MFMA per (tile, 32-channel step)
  constexpr int B_BYTES = BN * 64;                       // 16 KB per k-step
  constexpr int SMEM_BYTES = 2 * (1024 + 480 * 64) + NS * B_BYTES;         // HRP <= 480; also >= the epilogue's staging tile
  __shared__ __attribute__((aligned(1024))) char smem[SMEM_BYTES];
  // layout: [A buffer 0][A buffer 1][B ring NS x 16 KB]; an A buffer = [zero block 1 KB][halo rows]: the per-lane fragment offsets
  // (halo row or zero block) are then relative to the buffer and the same for both buffers
  const int A_BYTES = 1024 + HRP * 64;
  char* const b_ring = smem + 2 * A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int nta = lid % p.ntiles, mt = lid / p.ntiles;
  int nt;
  const GroupView gv = group_view(p, nta, nt);
  const int m0 = mt * BM, n0 = nt * BN;
  const int W = p.W, HW = p.H * p.W;

  if (tid < 128) ((f16x8*)(smem + (tid >> 6) * A_BYTES))[tid & 63] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};

  // ---- DMA sources.  A: halo piece pc covers flattened input pixels m0 - W - 1 + 16 pc + (lane & 15), chunk lane >> 4 ----
  const int NP = HRP >> 4;
  const f16* a_src[MAXP];
  bool a_ok[MAXP];
  int na_wave = 0;                                       // halo pieces this wave issues per chunk
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int pc = wave + 8 * i;
    const long pix = (long)m0 - W - 1 + pc * 16 + (lane & 15);
    a_ok[i] = pc < NP && pix >= 0 && pix < p.M;
    a_src[i] = gv.x + (a_ok[i] ? pix : 0) * p.ldx + (lane >> 4) * 8;
    na_wave += pc < NP ? 1 : 0;
  }
  const bool burst_big = na_wave >= 3;
  // B: piece = 16 weight rows; wave w fills rows 32 w .. 32 w + 31 (2 pieces); lane: row lane & 15, chunk lane >> 4
  const f16* b_src[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) b_src[j] = gv.w + (long)(n0 + wave * 32 + j * 16 + (lane & 15)) * p.Kpad + (lane >> 4) * 8;

  const int nchunks = p.Cin >> 5, nk = nchunks * 9;
  int ld_tile = 0, ld_tap = 0, ld_cin = 0, ld_soff = 0;      // next B k-step this wave issues: (chunk ld_cin/32, tap ld_tap)
  auto issue_b = [&]() __attribute__((always_inline)) {
    char* sbase = b_ring + ld_soff;
    const int koff = ld_tap * p.Cin + ld_cin;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[j] + koff),
                                       (__attribute__((address_space(3))) void*)(sbase + (wave * 2 + j) * 1024), 16, 0, 0);
    ++ld_tile;
    ld_soff += B_BYTES;
    if (ld_soff == NS * B_BYTES) ld_soff = 0;
    if (++ld_tap == 9) { ld_tap = 0; ld_cin += 32; }
  };
  auto issue_a = [&](int chunk_idx) __attribute__((always_inline)) {      // the whole halo of channel chunk chunk_idx
    char* sbase = smem + (chunk_idx & 1) * A_BYTES + 1024;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      if (wave + 8 * i < NP) {                                             // wave-uniform
        const f16* src = a_ok[i] ? a_src[i] + chunk_idx * 32 : g_zero_chunk;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sbase + (wave + 8 * i) * 1024), 16, 0, 0);
      }
    }
  };

  f32x4 acc[TM16][TN16];
#pragma unroll
  for (int i = 0; i < TM16; ++i)
#pragma unroll
    for (int j = 0; j < TN16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addressing (16 x 16 x 32 MFMA: lane l holds row / column l & 15 and k = 8 (l >> 4) .. + 7, i.e. 16-byte chunk l >> 4:
  // a 16-row piece of the LDS image IS one fragment).  Same-box A/B against v_mfma_f32_32x32x16_f16 with the same tile, reads and
  // DMA: 110.9 -> 107.0 us on the dominant shape (tools/conv_build_ab.py); outputs bit-identical (the matrix core sums k in the same
  // order either way), so the equality with the register-staged loop still holds bit for bit.
  // row-tile i of this wave: output pixel mloc = wm*WM + 16 i + (lane & 15); tap (kh, kw) reads halo
  // row mloc + kh W + kw; byte offset of (halo row hr, chunk c) = (hr >> 4) * 1024 + c * 256 + (hr & 15) * 16
  //                                                            = 16 hr + 3 * ((16 hr) & ~255) + c * 256
  const int frow = lane & 15, fk = lane >> 4;
  // a_pre[tap][i]: byte offset inside an A buffer of this lane's fragment for row-tile i at tap `tap` -- the halo row, or the
  // zero block when the tap falls outside the image for this lane's output pixel (9 x TM registers, computed once)
  // (TM = 4, 256-pixel tiles: two 16-bit offsets per register -- 72 offsets next to 128 accumulator registers would spill)
  constexpr bool PACKED = TM16 >= 7;
  constexpr int NTAB = PACKED ? (TM16 + 1) / 2 : TM16;
  int a_pre[9][NTAB];
#pragma unroll
  for (int i = 0; i < TM16; ++i) {
    const int mloc = wm * WM + i * 16 + frow;
    const int m = m0 + mloc;
    const bool live = m < p.M;
    const int rem = live ? m % HW : 0;
    const int h = rem / W, w = rem - h * W;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const bool ok = live && (unsigned)(h + kh - 1) < (unsigned)p.H && (unsigned)(w + kw - 1) < (unsigned)W;
        const int t16 = (mloc + kh * W + kw) * 16;
        const int off = ok ? 1024 + t16 + 3 * (t16 & ~255) + fk * 256 : (lane & 15) * 16;      // < 32 KB
        if constexpr (!PACKED) a_pre[kh * 3 + kw][i] = off;
        else if (i & 1) a_pre[kh * 3 + kw][i >> 1] |= off << 16;
        else a_pre[kh * 3 + kw][i >> 1] = off;
      }
  }
  auto a_offset = [&](auto tapc, int i) __attribute__((always_inline)) {
    constexpr int TAP = decltype(tapc)::value;
    if constexpr (!PACKED) return a_pre[TAP][i];
    else {
      int v = a_pre[TAP][i >> 1];
      asm volatile("" : "+v"(v));                      // unpack at the read (hoisted out of the loop, the halves are 72 registers again)
      return (i & 1) ? (int)((unsigned)v >> 16) : (v & 0xffff);
    }
  };
  int b_off[TN16];
#pragma unroll
  for (int j = 0; j < TN16; ++j) b_off[j] = (wn * (WN / 16) + j) * 1024 + lane * 16;

  // ---- prologue ----
  issue_a(0);
  const int ahead = 4 + grp;
  for (int t = 0; t < ahead; ++t)
    if (ld_tile < nk) issue_b();
  // tail form of the wait: this wave's B pieces of k-steps <= t (and every older DMA) have landed
  auto wait_tail = [&](int t) __attribute__((always_inline)) {
    if (t >= nk) t = nk - 1;
    const int younger = ld_tile - 1 - t;
    if (younger >= 3) wait_vmcnt_v5<6>();
    else if (younger == 2) wait_vmcnt_v5<4>();
    else if (younger == 1) wait_vmcnt_v5<2>();
    else wait_vmcnt_v5<0>();
  };
  wait_tail(grp);                                                    // also retires the halo of chunk 0 (issued first)
  __builtin_amdgcn_s_waitcnt(0xC07F);                                // lgkmcnt(0): the zero block is written
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();                        // stagger

  f16x8 fa[TM16], fb[TN16];
  int rd_soff = 0;
  // one k-step = one (chunk, tap).  STEADY: not in the last chunk -- the DMA distance is full and a next chunk exists.
  auto step = [&](auto tapc, auto steadyc, int chunk_idx, int u) __attribute__((always_inline)) {
    constexpr int TAP = decltype(tapc)::value;
    constexpr bool STEADY = decltype(steadyc)::value;
    // ---------------- L(u) ----------------
    __builtin_amdgcn_sched_barrier(0);
    {
      const char* sa = smem + (chunk_idx & 1) * A_BYTES;
      const char* sb = b_ring + rd_soff;
#pragma unroll
      for (int i = 0; i < TM16; ++i) fa[i] = *(const f16x8*)(sa + a_offset(tapc, i));
#pragma unroll
      for (int j = 0; j < TN16; ++j) fb[j] = *(const f16x8*)(sb + b_off[j]);
      rd_soff += B_BYTES;
      if (rd_soff == NS * B_BYTES) rd_soff = 0;
    }
    if (STEADY || ld_tile < nk) issue_b();
    // halo of the next chunk: group 1 in the step of tap 0, group 0 in the step of tap 1 (WAR on the buffer chunk c-1 has left)
    if constexpr (STEADY && (TAP == 0 || TAP == 1)) {
      if (grp == 1 - TAP) issue_a(chunk_idx + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- M(u) ----------------
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TM16; ++i)
#pragma unroll
      for (int j = 0; j < TN16; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STEADY) {
      // 3 k-steps of B (6 pieces) stay in flight; for the 4 waits that follow this wave's halo burst (issued at tap 1 - grp)
      // the burst is younger than the pieces waited for and may stay in flight too (3 of its pieces are accounted)
      if constexpr (TAP >= 1 && TAP <= 3) {
        if (burst_big) wait_vmcnt_v5<9>(); else wait_vmcnt_v5<6>();
      } else if constexpr (TAP == 0 || TAP == 4) {
        if (burst_big && grp == (TAP == 0 ? 1 : 0)) wait_vmcnt_v5<9>(); else wait_vmcnt_v5<6>();
      } else {
        wait_vmcnt_v5<6>();
      }
    } else {
      wait_tail(u + 1 + grp);
    }
    __builtin_amdgcn_s_barrier();
  };
#define HALO_CHUNK(STEADYT, CI)                                                                         \
  step(std::integral_constant<int, 0>{}, STEADYT{}, CI, (CI) * 9 + 0); step(std::integral_constant<int, 1>{}, STEADYT{}, CI, (CI) * 9 + 1); \
  step(std::integral_constant<int, 2>{}, STEADYT{}, CI, (CI) * 9 + 2); step(std::integral_constant<int, 3>{}, STEADYT{}, CI, (CI) * 9 + 3); \
  step(std::integral_constant<int, 4>{}, STEADYT{}, CI, (CI) * 9 + 4); step(std::integral_constant<int, 5>{}, STEADYT{}, CI, (CI) * 9 + 5); \
  step(std::integral_constant<int, 6>{}, STEADYT{}, CI, (CI) * 9 + 6); step(std::integral_constant<int, 7>{}, STEADYT{}, CI, (CI) * 9 + 7); \
  step(std::integral_constant<int, 8>{}, STEADYT{}, CI, (CI) * 9 + 8);
  int ci = 0;
  for (; ci + 1 < nchunks; ++ci) { HALO_CHUNK(std::true_type, ci) }
  { HALO_CHUNK(std::false_type, ci) }
#undef HALO_CHUNK
  if (grp == 0) __builtin_amdgcn_s_barrier();
  conv_epilogue<BM, BN, WAVES_M, WAVES_N, TM16, TN16, 16, f32x4>(p, acc, (f16*)smem, m0, n0, mt, tid, lane, wm, wn, gv.cbase, gv.sbase);
}

// Tile choice: BN follows Cout (128 / 64 / 32); BM = 128.  Packed weights and BatchNorm partials are padded to a multiple of
// this BN.  (A 128 x 256 tile with 8 waves was measured: no faster on the 72x72 level -- the limit is bytes in flight per CU
// against the ~2 us loaded MALL latency, not bytes per FLOP -- and slower on the small levels, which get half the workgroups.)
static int pick_bn(int cout) { return cout > 64 ? 128 : (cout > 32 ? 64 : 32); }

extern "C" int tscd_conv_tile_n(int cout) { return pick_bn(cout); }
extern "C" int tscd_conv_tile_m(void) { return 128; }

// Per-call plan overrides (tests and A/B timing only; the product passes the automatic values): there is no process-global state.
//   variant  -1 automatic; 0 register-staged loop; 1 LDS-DMA 128-wide tiles only; 5 / 6 / 7 halo-staged 256- / 192- / 224-pixel tiles wherever
//            their shape allows
//   korder   1 = chunk-major k order in the register-staged loop (the halo variant's order; Cin % 32 == 0)
//   tile_m   64 / 128 / 192: tile height of the LDS-DMA 128-wide variant; 0 automatic
struct ConvPlan { int variant, korder, tile_m; };

// Tile height (output pixels) of the halo-staged variant for this launch, 0 = the launch does not take it.
// Shape: 3x3, stride 1, Cin % 32 == 0, Cout (per group) a multiple of 256, halo (tile + 2W + 2 pixels) <= 480 rows.  Automatic
// selection: only when the launch has enough tiles to give every CU one (its 1-workgroup-per-CU tiles cannot share a CU with
// anything), and then the height (192 / 256) whose last round leaves fewer CUs idle.
static int halo_tile_m(const ConvPlan& pl, long M, int W, int Cin, int Cout, int ksize, int stride, int G) {
  if (pl.variant != -1 && pl.variant != 5 && pl.variant != 6 && pl.variant != 7) return 0;
  if (ksize != 3 || stride != 1 || Cin % 32 != 0 || Cout % 256 != 0) return 0;
  auto fits = [&](int bm) { return (bm + 2 * W + 2 + 15) / 16 * 16 <= 480; };
  if (pl.variant == 5) return fits(256) ? 256 : 0;
  if (pl.variant == 6) return fits(192) ? 192 : 0;
  if (pl.variant == 7) return fits(224) ? 224 : 0;
  const long nt = (long)(Cout / 256) * G;
  int best = 0;
  double best_cost = 1e30;
  for (int bm : {192, 224, 256}) {
    if (!fits(bm)) continue;
    const long tiles = (M + bm - 1) / bm * nt;
    if (tiles < 200) continue;                               // clearly fewer tiles than CUs: the 128-row variants' finer grain wins
    const double cost = (double)((tiles + 255) / 256) * bm;  // rounds x rows per tile
    if (cost < best_cost) { best_cost = cost; best = bm; }
  }
  return best;
}

// Tile height of the LDS-DMA 128-channel-wide variant for this launch, from same-box A/B timing of the bench clip's shapes
// (tools/conv_tile_ab.py, profiles/r02_conv_tile_ab.txt; t128 = workgroups with 128-row tiles, 2 of which share a CU):
//   * t128 <= 200 (18x18 level: 82 / 164 workgroups): 64-row tiles, 3 per CU -- 31.4 -> 22.5 us (256 ch 3x3), 54 -> 48 us (512 ch);
//   * short K loops (<= 32 k-tiles: the 1x1 convs) up to 700 workgroups: 64-row tiles (prologue and epilogue of co-resident
//     workgroups overlap) -- 14.1 -> 12.2, 18.0 -> 16.6, 20.9 -> 16.5 us;
//   * long K loops whose 128-row grid needs a second, partly filled round while 192-row tiles fit 2 per CU in one
//     (72x72 128 -> 128 3x3: 648 -> 432 workgroups): 192-row tiles -- 49.4 -> 42.7 us;
//   * otherwise 128 (36x36 256 ch 3x3: 45 us with 128, 46 with 192, 52 with 64).
static int pick_bm(const ConvPlan& pl, long M, int Cin, int Cout, int ksize, int G) {
  if (Cin % 32 != 0 || pick_bn(Cout) != 128 || pl.variant == 0) return 128;      // only the <*, 128, 2> instantiations vary BM
  if (pl.tile_m == 64 || pl.tile_m == 128 || pl.tile_m == 192) return pl.tile_m;
  const long nt = (long)cdiv(Cout, 128) * G, nk = (long)ksize * ksize * Cin / 32;
  const long t128 = cdiv(M, 128) * nt, t192 = cdiv(M, 192) * nt;
  if (t128 <= 200) return 64;
  if (nk <= 32 && t128 <= 700) return 64;
  if (nk > 32 && t128 > 512 && t192 <= 512) return 192;
  return 128;
}

static int stat_tiles(const ConvPlan& pl, long M, int W, int Cin, int Cout, int ksize, int stride, int G) {
  const int bm = halo_tile_m(pl, M, W, Cin, Cout, ksize, stride, G);
  if (bm > 0) return cdiv(M, bm);
  return cdiv(M, pick_bm(pl, M, Cin, Cout, ksize, G));
}

// Number of per-channel partial-sum entries a raw-epilogue launch writes (one per M-tile of the variant it takes): the BatchNorm
// reduction must be told this count.  Cout / Cin per group.
extern "C" int tscd_conv_stat_tiles(long M, int W, int Cin, int Cout, int ksize, int stride) {
  return stat_tiles(ConvPlan{-1, 0, 0}, M, W, Cin, Cout, ksize, stride, 1);
}
extern "C" int tscd_conv_stat_tiles_ex(long M, int W, int Cin, int Cout, int ksize, int stride, int groups, int variant, int tile_m) {
  return stat_tiles(ConvPlan{variant, 0, tile_m}, M, W, Cin, Cout, ksize, stride, groups < 1 ? 1 : groups);
}

static int conv_launch(const void* x, const void* wpk, void* y, const float* bias, const float* scale, const float* shift,
                       float* bn_partial, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int ldx, int ldy,
                       int epilogue, int G, const ConvPlan& pl, void* stream) {
  TSCD_CHECK_ARG(x && wpk && y, "conv2d: null pointer");
  TSCD_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && G >= 1, "conv2d: bad shape B=%d H=%d W=%d Cin=%d Cout=%d groups=%d", B, H,
                 W, Cin, Cout, G);
  TSCD_CHECK_ARG(Cin % 8 == 0 && ldx % 8 == 0 && ldx >= G * Cin, "conv2d: Cin=%d / ldx=%d must be multiples of 8 (ldx >= groups * Cin)", Cin, ldx);
  TSCD_CHECK_ARG(ldy >= G * Cout, "conv2d: ldy=%d < groups * Cout=%d", ldy, G * Cout);
  TSCD_CHECK_ARG(G == 1 || Cout % 8 == 0, "conv2d: grouped launches need Cout %% 8 == 0 (Cout=%d)", Cout);
  TSCD_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)wpk & 15) == 0, "conv2d: x / w must be 16-byte aligned");
  TSCD_CHECK_ARG(ksize == 1 || ksize == 3, "conv2d: ksize %d not supported", ksize);
  TSCD_CHECK_ARG(stride == 1 || stride == 2, "conv2d: stride %d not supported", stride);
  TSCD_CHECK_ARG(epilogue >= 0 && epilogue <= 5, "conv2d: bad epilogue %d", epilogue);
  TSCD_CHECK_ARG(pl.variant == -1 || pl.variant == 0 || pl.variant == 1 || (pl.variant >= 5 && pl.variant <= 7), "conv2d: variant %d", pl.variant);
  if (epilogue == EPI_BIAS || epilogue == EPI_BIAS_RELU) TSCD_CHECK_ARG(bias, "conv2d: bias epilogue without bias");
  if (epilogue == EPI_BN_SILU || epilogue == EPI_BN) TSCD_CHECK_ARG(scale && shift, "conv2d: BN epilogue without scale/shift");
  ConvParams p;
  p.x = (const f16*)x; p.w = (const f16*)wpk; p.y = (f16*)y;
  p.bias = bias; p.scale = scale; p.shift = shift; p.bn_partial = bn_partial;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = ksize; p.KW = ksize; p.stride = stride;
  p.pad = (ksize - 1) / 2;
  p.Ho = (H + 2 * p.pad - ksize) / stride + 1;
  p.Wo = (W + 2 * p.pad - ksize) / stride + 1;
  p.ldx = ldx; p.ldy = ldy;
  p.K = ksize * ksize * Cin;
  p.Kpad = (p.K + 31) / 32 * 32;
  long M = (long)B * p.Ho * p.Wo;
  TSCD_CHECK_ARG(M < (1L << 31), "conv2d: M too large");
  p.M = (int)M;
  p.epilogue = epilogue;
  const int BNP = pick_bn(Cout);
  const bool dma_ok = Cin % 32 == 0 && BNP >= 64 && pl.variant != 0;
  // variant 5 (input halo staged once per channel chunk, 192- or 256-pixel x 256-channel tiles): 3x3 stride-1 convs
  const int v5_tm = halo_tile_m(pl, M, W, Cin, Cout, ksize, stride, G);
  const bool v5_ok = v5_tm > 0;
  const int BN = v5_ok ? 256 : BNP;
  const int BM = !dma_ok ? 128 : pick_bm(pl, M, Cin, Cout, ksize, G);
  p.korder = (pl.korder == 1 && Cin % 32 == 0) ? 1 : 0;
  p.G = G;
  p.cstride = cdiv(Cout, BNP) * BNP;
  p.wrows_g = p.cstride;
  hipStream_t s = (hipStream_t)stream;
  if (v5_ok) {
    const int hrp = (v5_tm + 2 * W + 2 + 15) / 16 * 16;
    p.mtiles = cdiv(M, v5_tm);
    p.ntiles_g = cdiv(Cout, 256);
    p.ntiles = p.ntiles_g * G;
    if (v5_tm == 192)
      hipLaunchKernelGGL((conv_igemm_halo_kernel<6>), dim3(p.mtiles * p.ntiles), dim3(512), 0, s, p, hrp);
    else if (v5_tm == 224)
      hipLaunchKernelGGL((conv_igemm_halo_kernel<7>), dim3(p.mtiles * p.ntiles), dim3(512), 0, s, p, hrp);
    else
      hipLaunchKernelGGL((conv_igemm_halo_kernel<8>), dim3(p.mtiles * p.ntiles), dim3(512), 0, s, p, hrp);
    TSCD_CHECK_LAUNCH("conv2d (halo)");
    return TSCD_OK;
  }
  p.mtiles = cdiv(M, BM);
  p.ntiles_g = cdiv(Cout, BN);
  p.ntiles = p.ntiles_g * G;
  dim3 grid(p.mtiles * p.ntiles), block(256);
  if (dma_ok && BN == 128)
    if (BM == 64) hipLaunchKernelGGL((conv_igemm_glds_kernel<64, 128, 2>), grid, block, 0, s, p);
    else if (BM == 192) hipLaunchKernelGGL((conv_igemm_glds_kernel<192, 128, 2>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv_igemm_glds_kernel<128, 128, 2>), grid, block, 0, s, p);
  else if (dma_ok && BN == 64)
    hipLaunchKernelGGL((conv_igemm_glds_kernel<128, 64, 2>), grid, block, 0, s, p);
  else if (BN == 128)
    hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2>), grid, block, 0, s, p);
  else if (BN == 64)
    hipLaunchKernelGGL((conv_igemm_kernel<128, 64, 2, 2>), grid, block, 0, s, p);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<128, 32, 4, 1>), grid, block, 0, s, p);
  TSCD_CHECK_LAUNCH("conv2d");
  return TSCD_OK;
}

extern "C" int tscd_conv2d_nhwc_f16(const void* x, const void* wpk, void* y, const float* bias, const float* scale,
                                    const float* shift, float* bn_partial, int B, int H, int W, int Cin, int Cout,
                                    int ksize, int stride, int ldx, int ldy, int epilogue, void* stream) {
  return conv_launch(x, wpk, y, bias, scale, shift, bn_partial, B, H, W, Cin, Cout, ksize, stride, ldx, ldy, epilogue, 1,
                     ConvPlan{-1, 0, 0}, stream);
}

// The general form: `groups` convolutions of Cin -> Cout channels each in one launch (group g reads channels [g*Cin, (g+1)*Cin) of
// x and writes channels [g*Cout, (g+1)*Cout) of y; wpk = the groups' packed matrices one after the other, each padded to
// tscd_conv_tile_n(Cout) rows; bias / scale / shift over the groups * Cout concatenated channels; bn_partial
// [2][groups * rows_pad][tscd_conv_stat_tiles_ex]) and the per-call plan overrides (variant / korder / tile_m: tests and A/B timing).
extern "C" int tscd_conv2d_nhwc_f16_ex(const void* x, const void* wpk, void* y, const float* bias, const float* scale,
                                       const float* shift, float* bn_partial, int B, int H, int W, int Cin, int Cout, int ksize,
                                       int stride, int ldx, int ldy, int epilogue, int groups, int variant, int korder, int tile_m,
                                       void* stream) {
  return conv_launch(x, wpk, y, bias, scale, shift, bn_partial, B, H, W, Cin, Cout, ksize, stride, ldx, ldy, epilogue, groups,
                     ConvPlan{variant, korder, tile_m}, stream);
}

// Several independent stride-1 GEMM-shaped launches (1x1 convs / nn.Linear as 1 x n images) in one launch of the 64 x 128 LDS-DMA tile:
// jobs[i] = {x, wpk, y, bias (fp32 or null), B, H, W, Cin, Cout, ldx, ldy, epilogue (EPI_RAW / EPI_BIAS / EPI_BIAS_RELU)}; every job needs
// Cin % 32 == 0 and Cout > 64 (the 128-wide tile).  Results are the single launches' bits (the tile height does not enter an element's sum).
struct ConvJobHost { const void* x; const void* wpk; void* y; const float* bias; int B, H, W, Cin, Cout, ldx, ldy, epilogue; };
extern "C" int tscd_conv2d_jobs_max(void) { return CONV_MAX_JOBS; }
extern "C" int tscd_conv2d_jobs_supported(int Cin, int Cout) { return (Cin % 32 == 0 && Cin >= 32 && pick_bn(Cout) == 128) ? 1 : 0; }
extern "C" int tscd_conv2d_jobs_nhwc_f16(const void* jobs_host, int njobs, void* stream) {
  TSCD_CHECK_ARG(jobs_host && njobs > 0 && njobs <= CONV_MAX_JOBS, "conv2d_jobs: 1..%d jobs", CONV_MAX_JOBS);
  const ConvJobHost* h = (const ConvJobHost*)jobs_host;
  ConvJobTable t;
  t.njobs = njobs;
  int wg = 0;
  for (int i = 0; i < njobs; ++i) {
    const ConvJobHost& a = h[i];
    TSCD_CHECK_ARG(a.x && a.wpk && a.y && a.B > 0 && a.H > 0 && a.W > 0, "conv2d_jobs: job %d: bad pointers / shape", i);
    TSCD_CHECK_ARG(tscd_conv2d_jobs_supported(a.Cin, a.Cout), "conv2d_jobs: job %d: Cin=%d Cout=%d not of the 128-wide LDS-DMA form", i, a.Cin, a.Cout);
    TSCD_CHECK_ARG(a.ldx % 8 == 0 && a.ldx >= a.Cin && a.ldy >= a.Cout && ((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.wpk & 15) == 0,
                   "conv2d_jobs: job %d: strides / alignment", i);
    TSCD_CHECK_ARG(a.epilogue == EPI_RAW || ((a.epilogue == EPI_BIAS || a.epilogue == EPI_BIAS_RELU) && a.bias), "conv2d_jobs: job %d: epilogue", i);
    ConvParams& p = t.job[i];
    p.x = (const f16*)a.x; p.w = (const f16*)a.wpk; p.y = (f16*)a.y;
    p.bias = a.bias; p.scale = nullptr; p.shift = nullptr; p.bn_partial = nullptr;
    p.B = a.B; p.H = a.H; p.W = a.W; p.Cin = a.Cin; p.Cout = a.Cout; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0;
    p.Ho = a.H; p.Wo = a.W; p.ldx = a.ldx; p.ldy = a.ldy;
    p.K = a.Cin; p.Kpad = (a.Cin + 31) / 32 * 32;
    const long M = (long)a.B * a.H * a.W;
    TSCD_CHECK_ARG(M < (1L << 31), "conv2d_jobs: M too large");
    p.M = (int)M; p.epilogue = a.epilogue;
    p.korder = 0; p.G = 1;
    p.cstride = cdiv(a.Cout, 128) * 128; p.wrows_g = p.cstride;
    p.mtiles = cdiv(M, 64); p.ntiles_g = cdiv(a.Cout, 128); p.ntiles = p.ntiles_g;
    t.wg_begin[i] = wg;
    wg += p.mtiles * p.ntiles;
  }
  t.wg_begin[njobs] = wg;
  hipLaunchKernelGGL((conv_igemm_glds_jobs_kernel<64, 128, 2>), dim3(wg), dim3(256), 0, (hipStream_t)stream, t);
  TSCD_CHECK_LAUNCH("conv2d_jobs");
  return TSCD_OK;
}
