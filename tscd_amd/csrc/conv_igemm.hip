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
};

enum { EPI_RAW = 0, EPI_BIAS = 1, EPI_BIAS_RELU = 2, EPI_BN_SILU = 3, EPI_BN = 4, EPI_COST = 5 };

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
  const int nt = lid % p.ntiles, mt = lid / p.ntiles;
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
    a_base[i] = p.x + (long)b * p.H * p.W * p.ldx;
  }
  int cin = a_chunk * 8, kh = 0, kw = 0;
  while (cin >= p.Cin) {
    cin -= p.Cin;
    if (++kw == p.KW) { kw = 0; ++kh; }
  }

  f16x8 a_reg[A_CH], b_reg[B_CH];
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  auto load_tiles = [&](int kt) {
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
      if (idx < BN * 4) b_reg[j] = *(const f16x8*)(p.w + (long)(n0 + (idx >> 2)) * p.Kpad + kt * 32 + (idx & 3) * 8);
    }
    cin += 32;
    while (cin >= p.Cin) {
      cin -= p.Cin;
      if (++kw == p.KW) { kw = 0; ++kh; }
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

  // ---- epilogue.  C layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
  float s1[TN], s2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WN + j * 32 + (lane & 31);
    const bool nok = n < p.Cout;
    float bias = 0.f, sc = 1.f, sh = 0.f;
    if (nok) {
      if (p.epilogue == EPI_BIAS || p.epilogue == EPI_BIAS_RELU) bias = p.bias[n];
      if (p.epilogue == EPI_BN_SILU || p.epilogue == EPI_BN) { sc = p.scale[n]; sh = p.shift[n]; }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int m = m0 + wm * WM + i * 32 + row;
        float v = acc[i][j][e];
        f16 o;
        if (p.epilogue == EPI_RAW) {
          o = (f16)v;
          float vf = (float)o;  // statistics of the fp16-rounded conv output, as the reference's BN sees it
          s1[j] += vf;
          s2[j] += vf * vf;
        } else if (p.epilogue == EPI_BIAS) {
          o = (f16)(v + bias);
        } else if (p.epilogue == EPI_BIAS_RELU) {
          o = (f16)fmaxf(v + bias, 0.f);
        } else if (p.epilogue == EPI_COST) {
          // Hungarian cost of double_match_embds (tscd_matching.py:912-937): 1 - (cos_reg + cos_cls)/2, NaN -> 0
          float c = 1.0f - 0.5f * (float)(f16)v;
          o = (c != c) ? (f16)0.f : (f16)c;
        } else {
          // eval: conv (fp16) -> BN with running stats (fp16 out) -> SiLU (fp16 out)
          float t = (float)(f16)((float)(f16)v * sc + sh);
          o = (p.epilogue == EPI_BN_SILU) ? (f16)(t * sigmoidf_(t)) : (f16)t;
        }
        if (nok && m < p.M) p.y[(long)m * p.ldy + n] = o;
      }
    }
  }
  if (p.epilogue == EPI_RAW && p.bn_partial != nullptr) {
    float* red = (float*)smem;  // [WAVES_M][BN][2]
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float a = s1[j] + __shfl_xor(s1[j], 32, 64);
      float b = s2[j] + __shfl_xor(s2[j], 32, 64);
      if (lane < 32) {
        int col = wn * WN + j * 32 + lane;
        red[(wm * BN + col) * 2 + 0] = a;
        red[(wm * BN + col) * 2 + 1] = b;
      }
    }
    __syncthreads();
    if (tid < BN) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES_M; ++w) {
        a += red[(w * BN + tid) * 2 + 0];
        b += red[(w * BN + tid) * 2 + 1];
      }
      p.bn_partial[((long)mt * 2 + 0) * p.cstride + n0 + tid] = a;
      p.bn_partial[((long)mt * 2 + 1) * p.cstride + n0 + tid] = b;
    }
  }
}

// Tile choice: BN follows Cout (128 / 64 / 32); BM = 128.
static int pick_bn(int cout) { return cout > 64 ? 128 : (cout > 32 ? 64 : 32); }

extern "C" int tscd_conv_tile_n(int cout) { return pick_bn(cout); }
extern "C" int tscd_conv_tile_m(void) { return 128; }

extern "C" int tscd_conv2d_nhwc_f16(const void* x, const void* wpk, void* y, const float* bias, const float* scale,
                                    const float* shift, float* bn_partial, int B, int H, int W, int Cin, int Cout,
                                    int ksize, int stride, int ldx, int ldy, int epilogue, void* stream) {
  TSCD_CHECK_ARG(x && wpk && y, "conv2d: null pointer");
  TSCD_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv2d: bad shape B=%d H=%d W=%d Cin=%d Cout=%d", B, H,
                 W, Cin, Cout);
  TSCD_CHECK_ARG(Cin % 8 == 0 && ldx % 8 == 0 && ldx >= Cin, "conv2d: Cin=%d / ldx=%d must be multiples of 8", Cin, ldx);
  TSCD_CHECK_ARG(ldy >= Cout, "conv2d: ldy=%d < Cout=%d", ldy, Cout);
  TSCD_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)wpk & 15) == 0, "conv2d: x / w must be 16-byte aligned");
  TSCD_CHECK_ARG(ksize == 1 || ksize == 3, "conv2d: ksize %d not supported", ksize);
  TSCD_CHECK_ARG(stride == 1 || stride == 2, "conv2d: stride %d not supported", stride);
  TSCD_CHECK_ARG(epilogue >= 0 && epilogue <= 5, "conv2d: bad epilogue %d", epilogue);
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
  const int BM = 128, BN = pick_bn(Cout);
  p.mtiles = cdiv(M, BM);
  p.ntiles = cdiv(Cout, BN);
  p.cstride = p.ntiles * BN;
  dim3 grid(p.mtiles * p.ntiles), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (BN == 128)
    hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2>), grid, block, 0, s, p);
  else if (BN == 64)
    hipLaunchKernelGGL((conv_igemm_kernel<128, 64, 2, 2>), grid, block, 0, s, p);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<128, 32, 4, 1>), grid, block, 0, s, p);
  TSCD_CHECK_LAUNCH("conv2d");
  return TSCD_OK;
}
