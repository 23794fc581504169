// HBM-bound NHWC fp16 helpers around the conv kernel: weight packing, train-mode BatchNorm finalize/apply (+SiLU,
// +residual), Focus space-to-depth, nearest 2x upsample / channel-slice copies (concat-free CSP/PAFPN buffers),
// SPP max-pools and the Haar wavelet high-frequency block.  Every kernel moves 16 B (8 halves) per lane so a
// wave reads/writes 1 KiB contiguous runs of channels.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";
void tscd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* tscd_last_error(void) { return g_err; }
extern "C" int tscd_abi_version(void) { return 1; }
// Queues a host function on the stream (hipLaunchHostFunc): work queued behind it starts when fn(arg) has returned.  The train
// step parks the stream on the assignment chain's latch this way (include/tscd_host.h: tscd_latch_wait_cb) and keeps queueing.
extern "C" int tscd_stream_host_func(void* stream, void* fn, void* arg) {
  TSCD_CHECK_ARG(fn, "stream_host_func: null function");
  hipError_t e = hipLaunchHostFunc((hipStream_t)stream, (hipHostFn_t)fn, arg);
  if (e != hipSuccess) {
    tscd_set_error("stream_host_func: %s", hipGetErrorString(e));
    return 2;
  }
  return 0;
}

#define EW_BLOCK 256
static inline int ew_grid(long n) {
  long g = (n + EW_BLOCK - 1) / EW_BLOCK;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// ------------------------------------------------------------------------------------------------------
// Weight packing: torch [Cout][Cin][KH][KW] (f32 or f16) -> [CoutPad][Kpad] f16, k = (kh*KW+kw)*CinPad + cin.
// mode 1 ("dgrad"): the packed matrix is the one that turns the SAME conv kernel into the data-gradient of a
// stride-1 conv: rows = Cin, k = ((KH-1-kh)*KW + (KW-1-kw))*CoutAsCin + cout.
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_weight_kernel(const T* __restrict__ w, f16* __restrict__ out, int Cout, int Cin, int CinPad, int KH,
                                   int KW, int rows_pad, int Kpad, int mode) {
  long total = (long)rows_pad * Kpad;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int row = idx / Kpad, k = idx % Kpad;
    float v = 0.f;
    if (mode == 0) {
      int tap = k / CinPad, ci = k % CinPad;
      if (row < Cout && tap < KH * KW && ci < Cin) {
        int kh = tap / KW, kw = tap % KW;
        v = (float)w[(((long)row * Cin + ci) * KH + kh) * KW + kw];
      }
    } else {
      // row = cin of the forward conv, inner channel = cout of the forward conv (CinPad carries Cout here)
      int tap = k / CinPad, co = k % CinPad;
      if (row < Cin && tap < KH * KW && co < Cout) {
        int kh = KH - 1 - tap / KW, kw = KW - 1 - tap % KW;
        v = (float)w[(((long)co * Cin + row) * KH + kh) * KW + kw];
      }
    }
    out[idx] = (f16)v;
  }
}

extern "C" int tscd_pack_conv_weight(const void* w, int src_is_f32, void* out, int Cout, int Cin, int CinPad, int KH, int KW,
                                     int rows_pad, int Kpad, int mode, void* stream) {
  TSCD_CHECK_ARG(w && out, "pack_conv_weight: null pointer");
  TSCD_CHECK_ARG(mode == 0 || mode == 1, "pack_conv_weight: bad mode");
  TSCD_CHECK_ARG(Kpad % 32 == 0 && Kpad >= KH * KW * CinPad, "pack_conv_weight: Kpad=%d too small", Kpad);
  long total = (long)rows_pad * Kpad;
  if (src_is_f32)
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                       (const float*)w, (f16*)out, Cout, Cin, CinPad, KH, KW, rows_pad, Kpad, mode);
  else
    hipLaunchKernelGGL(pack_weight_kernel<f16>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                       (const f16*)w, (f16*)out, Cout, Cin, CinPad, KH, KW, rows_pad, Kpad, mode);
  TSCD_CHECK_LAUNCH("pack_conv_weight");
  return TSCD_OK;
}

// All trainable weights re-packed in ONE launch after the optimizer step (the per-weight launches above cost ~90 launches and
// 0.4 ms per TSCD-L step: forward and data-gradient packing of ~45 convs / linears).  jobs: device array of PackJob; blockmap:
// one {job, tile} pair per workgroup.  A workgroup moves the PACK_TILE x PACK_TILE (cout, cin) sub-block of its weight through LDS:
// the source rows w[cout][ci0 .. ci0+32)[KH][KW] are contiguous runs (coalesced fp32 reads), the packed rows leave as 64-byte
// segments of consecutive cin (forward order) or consecutive cout (data-gradient order, taps flipped).  Element-wise gathering with
// run-time divisions took 200 us for the 79 MB of TSCD-L's trainable weights (9216-byte stride between the lanes of the
// data-gradient order).  Padding rows / columns of the packed copies are zero from the first (full) pack and are not rewritten.
#define PACK_TILE 32
#define PACK_TAPS 9          // taps staged per pass (3x3); larger filters take several passes
struct PackJob { const void* w; f16* out; int Cout, Cin, CinPad, KH, KW, rows_pad, Kpad, mode, src_f32, pad_; };
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const PackJob* __restrict__ jobs, const int2* __restrict__ blockmap) {
  __shared__ f16 tile[PACK_TILE][PACK_TILE * PACK_TAPS + 2];            // [cout][cin * taps + tap], pitch 145 dwords (odd)
  const int2 bm = blockmap[blockIdx.x];
  const PackJob j = jobs[bm.x];
  const int tiles_i = (j.Cin + PACK_TILE - 1) / PACK_TILE;
  const int co0 = (bm.y / tiles_i) * PACK_TILE, ci0 = (bm.y % tiles_i) * PACK_TILE;
  const int nco = min(PACK_TILE, j.Cout - co0), nci = min(PACK_TILE, j.Cin - ci0);
  const int KHW = j.KH * j.KW;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                // 8 rows of 32 lanes
  for (int t0 = 0; t0 < KHW; t0 += PACK_TAPS) {
    const int nt = min(PACK_TAPS, KHW - t0);
    if (t0) __syncthreads();
    // load: row r = source run w[co0 + r][ci0 .. ci0 + nci)[t0 .. t0 + nt): contiguous when the pass covers all taps
    for (int r = ty; r < nco; r += 8) {
      const long base = ((long)(co0 + r) * j.Cin + ci0) * KHW;
      if (nt == KHW) {
        for (int q = tx; q < nci * KHW; q += 32)
          tile[r][q] = j.src_f32 ? (f16)((const float*)j.w)[base + q] : ((const f16*)j.w)[base + q];
      } else {
        for (int q = tx; q < nci * nt; q += 32) {
          const int c = q / nt, t = q - c * nt;
          const long src = base + (long)c * KHW + t0 + t;
          tile[r][c * nt + t] = j.src_f32 ? (f16)((const float*)j.w)[src] : ((const f16*)j.w)[src];
        }
      }
    }
    __syncthreads();
    if (j.mode == 0) {
      // out[cout][tap * CinPad + cin]: lanes along cin
      for (int r = ty; r < nco; r += 8)
        for (int t = 0; t < nt; ++t)
          if (tx < nci) j.out[(long)(co0 + r) * j.Kpad + (t0 + t) * j.CinPad + ci0 + tx] = tile[r][tx * nt + t];
    } else {
      // out[cin][(KHW - 1 - tap) * CoutPad + cout]  (CinPad carries the padded cout count): lanes along cout
      for (int c = ty; c < nci; c += 8)
        for (int t = 0; t < nt; ++t)
          if (tx < nco) j.out[(long)(ci0 + c) * j.Kpad + (KHW - 1 - (t0 + t)) * j.CinPad + co0 + tx] = tile[tx][c * nt + t];
    }
  }
}
extern "C" int tscd_pack_block_elems(void) { return PACK_TILE; }
extern "C" int tscd_pack_conv_weights_batched(const void* jobs, const void* blockmap, long nblocks, void* stream) {
  TSCD_CHECK_ARG((jobs && blockmap) || nblocks == 0, "pack_conv_weights_batched: null table");
  if (nblocks == 0) return TSCD_OK;
  TSCD_CHECK_ARG(nblocks < (1L << 31), "pack_conv_weights_batched: too many blocks");
  hipLaunchKernelGGL(pack_weights_batched_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const PackJob*)jobs,
                     (const int2*)blockmap);
  TSCD_CHECK_LAUNCH("pack_conv_weights_batched");
  return TSCD_OK;
}

// ------------------------------------------------------------------------------------------------------
// Train-mode BatchNorm: reduce the conv kernel's per-M-tile partials ([2][cstride][mtiles], channel-major: a wave reads
// contiguous floats) in a fixed order (one wave per channel,
// fp64 accumulation), emit scale/shift for the apply pass, the saved mean/invstd for backward, and update the
// running statistics exactly like torch (momentum, unbiased variance) -- reference: network_blocks.py:52-58 with
// eps=1e-3, momentum=0.03 (exps/TSCD_OVIS/ovis_tscd_large.py:109-113).
// ------------------------------------------------------------------------------------------------------
// One launch may cover several BatchNorm modules (the fused first convs / grouped second convs of the head's towers write Ctot =
// nseg * seg_c channels): segment s = c / seg_c has its own gamma / beta / running statistics.  For a grouped conv launch the partial
// table keeps group_cstride rows per group of group_c channels (rows padded per group); dense launches pass group_c = Ctot.
struct BnSegs {
  const float* gamma[4];
  const float* beta[4];
  float* rm[4];
  float* rv[4];
  int seg_c, group_c, group_cstride, rows;     // rows = row count of one half of the partial table
};
__global__ void bn_finalize_kernel(const float* __restrict__ partial, int mtiles, int C, double count, BnSegs sg, float eps,
                                   float momentum, float* scale_shift, float* save_mean, float* save_invstd) {
  int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (c >= C) return;
  const int seg = c / sg.seg_c, cc = c - seg * sg.seg_c;
  const int grp = c / sg.group_c;
  const long row = (long)grp * sg.group_cstride + (c - grp * sg.group_c);
  double s1 = 0.0, s2 = 0.0;
  // issued ahead of the reduction so their latency overlaps the partial loads (the kernel is one dependent chain otherwise)
  const float* gamma = sg.gamma[seg];
  const float* beta = sg.beta[seg];
  float* running_mean = sg.rm[seg];
  float* running_var = sg.rv[seg];
  const float g = gamma ? gamma[cc] : 1.f, b = beta ? beta[cc] : 0.f;
  const float rm0 = running_mean ? running_mean[cc] : 0.f, rv0 = running_mean ? running_var[cc] : 0.f;
#pragma unroll 8
  for (int t = lane; t < mtiles; t += 64) {
    s1 += (double)partial[row * mtiles + t];
    s2 += (double)partial[((long)sg.rows + row) * mtiles + t];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (lane == 0) {
    double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0) var = 0;
    float invstd = (float)(1.0 / sqrt(var + (double)eps));
    scale_shift[c] = g * invstd;
    scale_shift[C + c] = b - (float)mean * g * invstd;
    if (save_mean) save_mean[c] = (float)mean;
    if (save_invstd) save_invstd[c] = invstd;
    if (running_mean) {
      double unbiased = count > 1 ? var * count / (count - 1.0) : var;
      running_mean[cc] = (1.f - momentum) * rm0 + momentum * (float)mean;
      running_var[cc] = (1.f - momentum) * rv0 + momentum * (float)unbiased;
    }
  }
}

extern "C" int tscd_bn_finalize(const float* partial, int mtiles, int cstride, int C, long count, const float* gamma,
                                const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                float* scale_shift, float* save_mean, float* save_invstd, void* stream) {
  TSCD_CHECK_ARG(partial && scale_shift && C > 0 && mtiles > 0 && count > 0, "bn_finalize: bad args");
  TSCD_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats must come in pairs");
  BnSegs sg;
  memset(&sg, 0, sizeof(sg));
  sg.gamma[0] = gamma; sg.beta[0] = beta; sg.rm[0] = running_mean; sg.rv[0] = running_var;
  sg.seg_c = C; sg.group_c = C; sg.group_cstride = cstride; sg.rows = cstride;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, partial, mtiles, C,
                     (double)count, sg, eps, momentum, scale_shift, save_mean, save_invstd);
  TSCD_CHECK_LAUNCH("bn_finalize");
  return TSCD_OK;
}

// Eval-mode fold: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale.
__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                               float* scale_shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = gamma[c] / sqrtf(rv[c] + eps);
  scale_shift[c] = s;
  scale_shift[C + c] = beta[c] - rm[c] * s;
}
extern "C" int tscd_bn_fold(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                            float* scale_shift, void* stream) {
  TSCD_CHECK_ARG(gamma && beta && rm && rv && scale_shift && C > 0, "bn_fold: bad args");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, rm, rv, eps, C,
                     scale_shift);
  TSCD_CHECK_LAUNCH("bn_fold");
  return TSCD_OK;
}

// y = act(half(x*scale+shift)) (+ residual), 8 channels per lane.  act: 0 none, 1 SiLU, 2 ReLU.
// A thread keeps ONE 8-channel chunk for its whole life (block = rows_per_block x C/8 threads), so scale/shift sit in
// registers and there is no index division in the loop; consecutive lanes still cover consecutive 16-B chunks of a row.
template <int ACT, bool RES>
__global__ __launch_bounds__(256) void bn_act_kernel(const f16* __restrict__ x, int ldx, f16* __restrict__ y, int ldy,
                                                     const float* __restrict__ ss, const f16* __restrict__ res, int ldr, long M, int C,
                                                     int cv, int rpb) {
  const int tid = threadIdx.x;
  if (tid >= cv * rpb) return;
  const int c8 = tid % cv, r = tid / cv;
  float sc[8], sh[8];
  for (int cc = c8; cc < (C >> 3); cc += cv) {      // cv < C/8 only for C > 2048
    const int c = cc << 3;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = ss[c + e]; sh[e] = ss[C + c + e]; }
    for (long m = (long)blockIdx.x * rpb + r; m < M; m += (long)gridDim.x * rpb) {
      f16x8 v = *(const f16x8*)(x + m * ldx + c);
      f16x8 rr;
      if (RES) rr = *(const f16x8*)(res + m * ldr + c);
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = (float)(f16)((float)v[e] * sc[e] + sh[e]);
        f16 h = ACT == 1 ? (f16)(t * sigmoid_fast_(t)) : (ACT == 2 ? (f16)fmaxf(t, 0.f) : (f16)t);
        if (RES) h = (f16)((float)h + (float)rr[e]);
        o[e] = h;
      }
      *(f16x8*)(y + m * ldy + c) = o;
    }
  }
}

extern "C" int tscd_bn_act_nhwc_f16(const void* x, int ldx, void* y, int ldy, const float* scale_shift, const void* residual,
                                    int ldr, long M, int C, int act, void* stream) {
  TSCD_CHECK_ARG(x && y && scale_shift, "bn_act: null pointer");
  TSCD_CHECK_ARG(C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && (residual == nullptr || ldr % 8 == 0),
                 "bn_act: C / strides must be multiples of 8 (C=%d ldx=%d ldy=%d)", C, ldx, ldy);
  TSCD_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)residual & 15) == 0,
                 "bn_act: pointers must be 16-byte aligned");
  TSCD_CHECK_ARG(act >= 0 && act <= 2, "bn_act: bad act %d", act);
  int cv = C / 8;
  if (cv > 256) cv = 256;
  const int rpb = 256 / cv;                       // rows per block pass
  long nblk = (M + rpb - 1) / rpb;
  if (nblk > 8192) nblk = 8192;
  dim3 grid((unsigned)nblk), block(256);
  hipStream_t st = (hipStream_t)stream;
#define BNACT(A, R)                                                                                                             \
  hipLaunchKernelGGL((bn_act_kernel<A, R>), grid, block, 0, st, (const f16*)x, ldx, (f16*)y, ldy, scale_shift, (const f16*)residual, \
                     ldr, M, C, cv, rpb)
  if (residual) {
    if (act == 1) BNACT(1, true); else if (act == 2) BNACT(2, true); else BNACT(0, true);
  } else {
    if (act == 1) BNACT(1, false); else if (act == 2) BNACT(2, false); else BNACT(0, false);
  }
#undef BNACT
  TSCD_CHECK_LAUNCH("bn_act");
  return TSCD_OK;
}

// ------------------------------------------------------------------------------------------------------
// Backbone / neck backward helpers (the base detector trains its whole backbone).
//   zero_insert2:   dY [B,Ho,Wo,C] -> [B,2Ho,2Wo,C] with dY at the even positions, zeros elsewhere.  A 3x3 / stride-2 / pad-1 conv
//                   y[o] = sum_k x[2o+k-1] w[k] has dX = the stride-1 data gradient of the zero-inserted dY, and dW = the stride-1
//                   weight gradient between x and the zero-inserted dY: both existing kernels, at 4x the FLOPs of the seven
//                   stride-2 convs of CSPDarknet + PAFPN only.
//   maxpool5_bwd:   gradient of the stride-1 5x5 max pool of SPP (network_blocks.py:183-205, pads act as -inf): dx[i] = sum of dy[o]
//                   over the windows o whose FIRST maximum in scan order is i (torch's choice on ties); gather form, deterministic.
//   upsample2_bwd:  gradient of the nearest-neighbour x2 upsample of PAFPN: dx[i] = sum of the 2x2 block of dy (fp32 sum).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zero_insert2_kernel(const f16* __restrict__ dy, int lddy, f16* __restrict__ out, int ldo, int B,
                                                           int Ho, int Wo, int C) {
  const int cv = C >> 3;
  const long total = (long)B * 2 * Ho * 2 * Wo * cv;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c8 = idx % cv;
    long t = idx / cv;
    const int x = t % (2 * Wo);
    t /= 2 * Wo;
    const int y = t % (2 * Ho), b = t / (2 * Ho);
    f16x8 v = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (!(x & 1) && !(y & 1)) v = *(const f16x8*)(dy + (((long)b * Ho + (y >> 1)) * Wo + (x >> 1)) * lddy + c8 * 8);
    *(f16x8*)(out + (((long)b * 2 * Ho + y) * 2 * Wo + x) * ldo + c8 * 8) = v;
  }
}
extern "C" int tscd_zero_insert2_nhwc_f16(const void* dy, int lddy, void* out, int ldo, int B, int Ho, int Wo, int C, void* stream) {
  TSCD_CHECK_ARG(dy && out && B > 0 && Ho > 0 && Wo > 0 && C % 8 == 0 && lddy % 8 == 0 && ldo % 8 == 0, "zero_insert2: bad arguments");
  const long total = (long)B * 4 * Ho * Wo * (C >> 3);
  hipLaunchKernelGGL(zero_insert2_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const f16*)dy, lddy, (f16*)out, ldo, B,
                     Ho, Wo, C);
  TSCD_CHECK_LAUNCH("zero_insert2");
  return TSCD_OK;
}

__global__ __launch_bounds__(256) void maxpool5_bwd_kernel(const f16* __restrict__ x, int ldx, const f16* __restrict__ dy, int lddy,
                                                           f16* __restrict__ dx, int lddx, int B, int H, int W, int C) {
  const long total = (long)B * H * W * C;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = idx % C;
    long t = idx / C;
    const int xx = t % W;
    t /= W;
    const int yy = t % H, b = t / H;
    const f16* xb = x + (long)b * H * W * ldx + c;
    const float me = (float)xb[((long)yy * W + xx) * ldx];
    float acc = 0.f;
    for (int oy = max(0, yy - 2); oy <= min(H - 1, yy + 2); ++oy)
      for (int ox = max(0, xx - 2); ox <= min(W - 1, xx + 2); ++ox) {
        // is (yy, xx) the first maximum of the window centred at (oy, ox)?
        bool win = true;
        for (int wy = max(0, oy - 2); wy <= min(H - 1, oy + 2) && win; ++wy)
          for (int wx = max(0, ox - 2); wx <= min(W - 1, ox + 2); ++wx) {
            const float v = (float)xb[((long)wy * W + wx) * ldx];
            const bool before = wy < yy || (wy == yy && wx < xx);
            if (v > me || (before && v == me)) { win = false; break; }
          }
        if (win) acc += (float)dy[(((long)b * H + oy) * W + ox) * lddy + c];
      }
    dx[(((long)b * H + yy) * W + xx) * lddx + c] = (f16)acc;
  }
}
extern "C" int tscd_maxpool5_bwd_nhwc_f16(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, int B, int H, int W, int C,
                                          void* stream) {
  TSCD_CHECK_ARG(x && dy && dx && B > 0 && H > 0 && W > 0 && C > 0, "maxpool5_bwd: bad arguments");
  const long total = (long)B * H * W * C;
  hipLaunchKernelGGL(maxpool5_bwd_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const f16*)x, ldx, (const f16*)dy, lddy,
                     (f16*)dx, lddx, B, H, W, C);
  TSCD_CHECK_LAUNCH("maxpool5_bwd");
  return TSCD_OK;
}

__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const f16* __restrict__ dy, int lddy, f16* __restrict__ dx, int lddx, int B, int H,
                                                            int W, int C) {
  const int cv = C >> 3;
  const long total = (long)B * H * W * cv;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c8 = idx % cv;
    long t = idx / cv;
    const int x = t % W;
    t /= W;
    const int y = t % H, b = t / H;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f16x8 v = *(const f16x8*)(dy + (((long)b * 2 * H + 2 * y + (q >> 1)) * 2 * W + 2 * x + (q & 1)) * lddy + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += (float)v[e];
    }
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)a[e];
    *(f16x8*)(dx + (((long)b * H + y) * W + x) * lddx + c8 * 8) = o;
  }
}
extern "C" int tscd_upsample2_bwd_nhwc_f16(const void* dy, int lddy, void* dx, int lddx, int B, int H, int W, int C, void* stream) {
  TSCD_CHECK_ARG(dy && dx && B > 0 && H > 0 && W > 0 && C % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0, "upsample2_bwd: bad arguments");
  const long total = (long)B * H * W * (C >> 3);
  hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const f16*)dy, lddy, (f16*)dx, lddx, B,
                     H, W, C);
  TSCD_CHECK_LAUNCH("upsample2_bwd");
  return TSCD_OK;
}

// ------------------------------------------------------------------------------------------------------
// Clip pre-processing on the device (reference: vid.py pull_item resize + data_augment.py:21-31 augment_hsv,
// :196-203 _mirror, :248-264 preproc): decoded uint8 BGR frames [B][hmax][wmax][3] -> letterboxed planar [B][3][H][W] (0..255,
// fp16 or fp32).  Per frame: bilinear resize to (nh, nw) with OpenCV's INTER_LINEAR geometry (half-pixel centres, edge clamp),
// rounded to uint8; optional HSV jitter on the resized pixel (8-bit HSV: H 0..179); optional horizontal flip of the resized image;
// paste top-left on a 114 canvas.  The same float32 arithmetic, in the same order, as yolox/data/data_augment.py's numpy
// transforms, one thread per output pixel.  meta[b] = {h, w, nh, nw, flip, hsv_on, hgain, sgain, vgain, bits(fp32 h/nh), bits(fp32 w/nw), 0}.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void hsv_jitter_u8(float& b, float& g, float& r, int hg, int sg, int vg) {
#pragma clang fp contract(off)      // numpy rounds every product and sum: no fused multiply-add here
  // _bgr2hsv
  const float v = fmaxf(fmaxf(r, g), b), mn = fminf(fminf(r, g), b), d = v - mn;
  float s = v > 0.f ? d / fmaxf(v, 1e-12f) * 255.0f : 0.f;
  const float dd = fmaxf(d, 1e-12f);
  float h = (v == r ? (g - b) / dd : (v == g ? 2.0f + (b - r) / dd : 4.0f + (r - g) / dd)) * 60.0f;
  if (d == 0.f) h = 0.f;
  if (h < 0.f) h += 360.0f;
  h *= 0.5f;
  int hi = ((int)rintf(h)) % 180, si = (int)rintf(s), vi = (int)v;
  // gains
  hi = (hi + hg) % 180;
  if (hi < 0) hi += 180;
  si = min(255, max(0, si + sg));
  vi = min(255, max(0, vi + vg));
  // _hsv2bgr
  const float hh = (float)hi * 2.0f, ss = (float)si / 255.0f, vv = (float)vi;
  const float c = vv * ss, hp = hh / 60.0f;
  const float x = c * (1.f - fabsf(fmodf(hp, 2.f) - 1.f));
  const int sel = ((int)floorf(hp)) % 6;
  float rr, gg, bb;
  switch (sel) {
    case 0: rr = c; gg = x; bb = 0.f; break;
    case 1: rr = x; gg = c; bb = 0.f; break;
    case 2: rr = 0.f; gg = c; bb = x; break;
    case 3: rr = 0.f; gg = x; bb = c; break;
    case 4: rr = x; gg = 0.f; bb = c; break;
    default: rr = c; gg = 0.f; bb = x; break;
  }
  const float m = vv - c;
  b = fminf(255.f, fmaxf(0.f, rintf(bb + m)));
  g = fminf(255.f, fmaxf(0.f, rintf(gg + m)));
  r = fminf(255.f, fmaxf(0.f, rintf(rr + m)));
}

template <typename T>
__global__ __launch_bounds__(256) void clip_preproc_kernel(const unsigned char* __restrict__ src, long frame_stride, int wmax,
                                                           const int* __restrict__ meta, T* __restrict__ out, int H, int W) {
#pragma clang fp contract(off)      // numpy rounds every product and sum: no fused multiply-add here
  const int b = blockIdx.z;
  const int* m = meta + b * 12;
  const int h = m[0], w = m[1], nh = m[2], nw = m[3], flip = m[4], hsv_on = m[5];
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  float vb = 114.f, vg = 114.f, vr = 114.f;
  if (y < nh && x < nw) {
    const int xs = flip ? nw - 1 - x : x;                    // the mirror acts on the RESIZED image
    // scales h / nh, w / nw rounded to fp32 on the host exactly as numpy rounds them (meta[9], meta[10] carry the bits)
    const float sy = ((float)y + 0.5f) * __int_as_float(m[9]) - 0.5f, sx = ((float)xs + 0.5f) * __int_as_float(m[10]) - 0.5f;
    const float fy0 = floorf(sy), fx0 = floorf(sx);
    const float fy = sy - fy0, fx = sx - fx0;
    const int y0 = min(h - 1, max(0, (int)fy0)), y1 = min(h - 1, max(0, (int)fy0 + 1));
    const int x0 = min(w - 1, max(0, (int)fx0)), x1 = min(w - 1, max(0, (int)fx0 + 1));
    const unsigned char* f = src + b * frame_stride;
    const unsigned char *p00 = f + ((long)y0 * wmax + x0) * 3, *p01 = f + ((long)y0 * wmax + x1) * 3;
    const unsigned char *p10 = f + ((long)y1 * wmax + x0) * 3, *p11 = f + ((long)y1 * wmax + x1) * 3;
    float v[3];
    const bool same = h == nh && w == nw;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (same) {
        v[c] = (float)p00[c];
      } else {
        const float top = (float)p00[c] * (1.f - fx) + (float)p01[c] * fx;
        const float bot = (float)p10[c] * (1.f - fx) + (float)p11[c] * fx;
        v[c] = fminf(255.f, fmaxf(0.f, rintf(top * (1.f - fy) + bot * fy)));
      }
    }
    vb = v[0]; vg = v[1]; vr = v[2];
    if (hsv_on) hsv_jitter_u8(vb, vg, vr, m[6], m[7], m[8]);
  }
  const long plane = (long)H * W, o = (long)b * 3 * plane + (long)y * W + x;
  out[o] = (T)vb;
  out[o + plane] = (T)vg;
  out[o + 2 * plane] = (T)vr;
}

extern "C" int tscd_clip_preproc_u8(const void* src, int B, int hmax, int wmax, const int* meta, void* out, int H, int W, int out_is_f16,
                                    void* stream) {
  TSCD_CHECK_ARG(src && meta && out && B > 0 && hmax > 0 && wmax > 0 && H > 0 && W > 0, "clip_preproc: bad arguments");
  dim3 grid(cdiv(W, 64), cdiv(H, 4), B), block(256);
  const long fs = (long)hmax * wmax * 3;
  if (out_is_f16)
    hipLaunchKernelGGL(clip_preproc_kernel<f16>, grid, block, 0, (hipStream_t)stream, (const unsigned char*)src, fs, wmax, meta, (f16*)out, H, W);
  else
    hipLaunchKernelGGL(clip_preproc_kernel<float>, grid, block, 0, (hipStream_t)stream, (const unsigned char*)src, fs, wmax, meta, (float*)out, H,
                       W);
  TSCD_CHECK_LAUNCH("clip_preproc");
  return TSCD_OK;
}

// ------------------------------------------------------------------------------------------------------
// Focus (network_blocks.py:267-289): NCHW image -> NHWC [B, H/2, W/2, 16] with channels
// (tl c0..2, bl c0..2, tr c0..2, br c0..2, 0, 0, 0, 0); the 4 zero channels keep Cin a multiple of 8.
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void focus_kernel(const T* __restrict__ x, f16* __restrict__ y, int B, int H, int W) {
  const int Ho = H >> 1, Wo = W >> 1;
  const long total = (long)B * Ho * Wo;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int wo = idx % Wo;
    long t = idx / Wo;
    int ho = t % Ho, b = t / Ho;
    f16x8 o0, o1;
    f16 v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // patch order tl, bl, tr, br  ->  (dy,dx) = (0,0),(1,0),(0,1),(1,1)
      int dy = q & 1, dx = q >> 1;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        v[q * 3 + c] = (f16)(float)x[(((long)b * 3 + c) * H + (2 * ho + dy)) * W + (2 * wo + dx)];
    }
    v[12] = v[13] = v[14] = v[15] = (f16)0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { o0[e] = v[e]; o1[e] = v[8 + e]; }
    *(f16x8*)(y + idx * 16) = o0;
    *(f16x8*)(y + idx * 16 + 8) = o1;
  }
}
extern "C" int tscd_focus_nchw_to_nhwc16(const void* x, int src_is_f32, void* y, int B, int H, int W, void* stream) {
  TSCD_CHECK_ARG(x && y && B > 0 && H % 2 == 0 && W % 2 == 0, "focus: bad args");
  long total = (long)B * (H / 2) * (W / 2);
  if (src_is_f32)
    hipLaunchKernelGGL(focus_kernel<float>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const float*)x,
                       (f16*)y, B, H, W);
  else
    hipLaunchKernelGGL(focus_kernel<f16>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const f16*)x,
                       (f16*)y, B, H, W);
  TSCD_CHECK_LAUNCH("focus");
  return TSCD_OK;
}

// Channel-slice copy with optional nearest 2x upsample: dst[b, h, w, :C] = src[b, h>>up, w>>up, :C].
__global__ void copy_slice_kernel(const f16* __restrict__ src, int lds_, f16* __restrict__ dst, int ldd, int B, int H, int W,
                                  int C, int up) {
  const int cv = C >> 3;
  const long total = (long)B * H * W * cv;
  const int Hs = H >> up, Ws = W >> up;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int c = (int)(idx % cv) << 3;
    long pix = idx / cv;
    int w = pix % W;
    long t = pix / W;
    int h = t % H, b = t / H;
    long spix = ((long)b * Hs + (h >> up)) * Ws + (w >> up);
    *(f16x8*)(dst + pix * ldd + c) = *(const f16x8*)(src + spix * lds_ + c);
  }
}
extern "C" int tscd_copy_slice_nhwc_f16(const void* src, int lds_, void* dst, int ldd, int B, int H, int W, int C,
                                        int upsample, void* stream) {
  TSCD_CHECK_ARG(src && dst && C % 8 == 0 && lds_ % 8 == 0 && ldd % 8 == 0, "copy_slice: bad args C=%d", C);
  TSCD_CHECK_ARG(upsample == 0 || (upsample == 1 && H % 2 == 0 && W % 2 == 0), "copy_slice: bad upsample");
  hipLaunchKernelGGL(copy_slice_kernel, dim3(ew_grid((long)B * H * W * (C / 8))), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                     (const f16*)src, lds_, (f16*)dst, ldd, B, H, W, C, upsample);
  TSCD_CHECK_LAUNCH("copy_slice");
  return TSCD_OK;
}

// Stride-1 max-pool (kernel ks, pad ks/2, -inf padding) on a channel slice.  SPP's 5/9/13 pools are the 5-pool
// applied 1x/2x/3x (max over nested windows), so the product calls this three times with ks=5.
__global__ void maxpool_kernel(const f16* __restrict__ src, int lds_, f16* __restrict__ dst, int ldd, int B, int H, int W,
                               int C, int ks) {
  const int cv = C >> 3, r = ks >> 1;
  const long total = (long)B * H * W * cv;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int c = (int)(idx % cv) << 3;
    long pix = idx / cv;
    int w = pix % W;
    long t = pix / W;
    int h = t % H, b = t / H;
    f16x8 m;
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = (f16)(-65504.f);
    for (int y = max(h - r, 0); y <= min(h + r, H - 1); ++y)
      for (int x = max(w - r, 0); x <= min(w + r, W - 1); ++x) {
        f16x8 v = *(const f16x8*)(src + (((long)b * H + y) * W + x) * lds_ + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
      }
    *(f16x8*)(dst + pix * ldd + c) = m;
  }
}
extern "C" int tscd_maxpool_nhwc_f16(const void* src, int lds_, void* dst, int ldd, int B, int H, int W, int C, int ks,
                                     void* stream) {
  TSCD_CHECK_ARG(src && dst && C % 8 == 0 && lds_ % 8 == 0 && ldd % 8 == 0 && (ks & 1), "maxpool: bad args");
  hipLaunchKernelGGL(maxpool_kernel, dim3(ew_grid((long)B * H * W * (C / 8))), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                     (const f16*)src, lds_, (f16*)dst, ldd, B, H, W, C, ks);
  TSCD_CHECK_LAUNCH("maxpool");
  return TSCD_OK;
}

// ------------------------------------------------------------------------------------------------------
// Haar wavelet high-frequency block (surrounding_extraction.py:12-127,257-267).
// With a=x[2y,2x], b=x[2y,2x+1], c=x[2y+1,2x], d=x[2y+1,2x+1]:  lh=(a+b-c-d)/2, hl=(a-b+c-d)/2, hh=(a-b-c+d)/2
// (the LL band is dropped by the block).  haar_hf writes [B,H/2,W/2,3C] (lh | hl | hh channel blocks), the
// 1x1 conv + ReLU on it is the MFMA conv kernel, haar_idwt_mul reconstructs with LL = 0 and multiplies with the
// 3x3 "content" branch:  out[2y,2x] = c*(lh+hl+hh)/2, out[2y,2x+1] = c*(lh-hl-hh)/2, out[2y+1,2x] = c*(-lh+hl-hh)/2,
// out[2y+1,2x+1] = c*(-lh-hl+hh)/2.  Intermediates are rounded to fp16 where the reference's fp16 ops round.
// ------------------------------------------------------------------------------------------------------
__global__ void haar_hf_kernel(const f16* __restrict__ x, int ldx, f16* __restrict__ hf, int B, int H, int W, int C) {
  const int cv = C >> 3, Ho = H >> 1, Wo = W >> 1;
  const long total = (long)B * Ho * Wo * cv;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int c = (int)(idx % cv) << 3;
    long pix = idx / cv;
    int wo = pix % Wo;
    long t = pix / Wo;
    int ho = t % Ho, b = t / Ho;
    const f16* p0 = x + (((long)b * H + 2 * ho) * W + 2 * wo) * ldx + c;
    f16x8 a = *(const f16x8*)p0, bb = *(const f16x8*)(p0 + ldx);
    f16x8 cc = *(const f16x8*)(p0 + (long)W * ldx), d = *(const f16x8*)(p0 + (long)W * ldx + ldx);
    f16x8 lh, hl, hh;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float fa = a[e], fb = bb[e], fc = cc[e], fd = d[e];
      lh[e] = (f16)(0.5f * (fa + fb - fc - fd));
      hl[e] = (f16)(0.5f * (fa - fb + fc - fd));
      hh[e] = (f16)(0.5f * (fa - fb - fc + fd));
    }
    f16* o = hf + pix * (3L * C) + c;
    *(f16x8*)(o) = lh;
    *(f16x8*)(o + C) = hl;
    *(f16x8*)(o + 2 * C) = hh;
  }
}
extern "C" int tscd_haar_hf_nhwc_f16(const void* x, int ldx, void* hf, int B, int H, int W, int C, void* stream) {
  TSCD_CHECK_ARG(x && hf && C % 8 == 0 && ldx % 8 == 0 && H % 2 == 0 && W % 2 == 0, "haar_hf: bad args (H=%d W=%d C=%d)", H, W,
                 C);
  hipLaunchKernelGGL(haar_hf_kernel, dim3(ew_grid((long)B * (H / 2) * (W / 2) * (C / 8))), dim3(EW_BLOCK), 0,
                     (hipStream_t)stream, (const f16*)x, ldx, (f16*)hf, B, H, W, C);
  TSCD_CHECK_LAUNCH("haar_hf");
  return TSCD_OK;
}

__global__ void haar_idwt_mul_kernel(const f16* __restrict__ hf, const f16* __restrict__ content, int ldc, f16* __restrict__ out,
                                     int ldo, int B, int H, int W, int C) {
  const int cv = C >> 3, Ho = H >> 1, Wo = W >> 1;
  const long total = (long)B * Ho * Wo * cv;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int c = (int)(idx % cv) << 3;
    long pix = idx / cv;
    int wo = pix % Wo;
    long t = pix / Wo;
    int ho = t % Ho, b = t / Ho;
    const f16* h = hf + pix * (3L * C) + c;
    f16x8 lh = *(const f16x8*)h, hl = *(const f16x8*)(h + C), hh = *(const f16x8*)(h + 2 * C);
    long p00 = ((long)b * H + 2 * ho) * W + 2 * wo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int dy = q >> 1, dx = q & 1;
      long pp = p00 + (long)dy * W + dx;
      f16x8 cc;
      if (content) cc = *(const f16x8*)(content + pp * ldc + c);
      f16x8 o;
      float s_lh = dy ? -0.5f : 0.5f, s_hl = dx ? -0.5f : 0.5f, s_hh = (dx ^ dy) ? -0.5f : 0.5f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        f16 rec = (f16)(s_lh * (float)lh[e] + s_hl * (float)hl[e] + s_hh * (float)hh[e]);
        o[e] = content ? (f16)((float)cc[e] * (float)rec) : rec;
      }
      *(f16x8*)(out + pp * ldo + c) = o;
    }
  }
}
// backward of out = content * idwt(hf): dcontent = d * idwt(hf), dhf = dwt_hf(d * content) (IDWT_Function.backward,
// surrounding_extraction.py:60-76); one pass over the 2x2 blocks
__global__ void haar_idwt_mul_bwd_kernel(const f16* __restrict__ d, const f16* __restrict__ hf, const f16* __restrict__ content, int ldc,
                                         f16* __restrict__ dcontent, f16* __restrict__ dhf, int B, int H, int W, int C) {
  const int cv = C >> 3, Ho = H >> 1, Wo = W >> 1;
  const long total = (long)B * Ho * Wo * cv;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int c = (int)(idx % cv) << 3;
    long pix = idx / cv;
    int wo = pix % Wo;
    long t = pix / Wo;
    int ho = t % Ho, b = t / Ho;
    const f16* h = hf + pix * (3L * C) + c;
    f16x8 lh = *(const f16x8*)h, hl = *(const f16x8*)(h + C), hh = *(const f16x8*)(h + 2 * C);
    float glh[8], ghl[8], ghh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) glh[e] = ghl[e] = ghh[e] = 0.f;
    long p00 = ((long)b * H + 2 * ho) * W + 2 * wo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int dy = q >> 1, dx = q & 1;
      long pp = p00 + (long)dy * W + dx;
      f16x8 cc = *(const f16x8*)(content + pp * ldc + c), dd = *(const f16x8*)(d + pp * (long)C + c), o;
      float s_lh = dy ? -0.5f : 0.5f, s_hl = dx ? -0.5f : 0.5f, s_hh = (dx ^ dy) ? -0.5f : 0.5f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        f16 rec = (f16)(s_lh * (float)lh[e] + s_hl * (float)hl[e] + s_hh * (float)hh[e]);
        o[e] = (f16)((float)dd[e] * (float)rec);
        float g = (float)dd[e] * (float)cc[e];
        glh[e] += s_lh * g;
        ghl[e] += s_hl * g;
        ghh[e] += s_hh * g;
      }
      *(f16x8*)(dcontent + pp * (long)C + c) = o;
    }
    f16x8 o1, o2, o3;
#pragma unroll
    for (int e = 0; e < 8; ++e) { o1[e] = (f16)glh[e]; o2[e] = (f16)ghl[e]; o3[e] = (f16)ghh[e]; }
    f16* oh = dhf + pix * (3L * C) + c;
    *(f16x8*)(oh) = o1;
    *(f16x8*)(oh + C) = o2;
    *(f16x8*)(oh + 2 * C) = o3;
  }
}
// d, dcontent: contiguous [B,H,W,C]; hf, dhf: contiguous [B,H/2,W/2,3C]; content: pixel stride ldc
extern "C" int tscd_haar_idwt_mul_bwd_nhwc_f16(const void* d, const void* hf, const void* content, int ldc, void* dcontent, void* dhf,
                                               int B, int H, int W, int C, void* stream) {
  TSCD_CHECK_ARG(d && hf && content && dcontent && dhf && C % 8 == 0 && ldc % 8 == 0 && H % 2 == 0 && W % 2 == 0,
                 "haar_idwt_mul_bwd: bad args");
  hipLaunchKernelGGL(haar_idwt_mul_bwd_kernel, dim3(ew_grid((long)B * (H / 2) * (W / 2) * (C / 8))), dim3(EW_BLOCK), 0,
                     (hipStream_t)stream, (const f16*)d, (const f16*)hf, (const f16*)content, ldc, (f16*)dcontent, (f16*)dhf, B, H, W,
                     C);
  TSCD_CHECK_LAUNCH("haar_idwt_mul_bwd");
  return TSCD_OK;
}

extern "C" int tscd_haar_idwt_mul_nhwc_f16(const void* hf, const void* content, int ldc, void* out, int ldo, int B, int H, int W,
                                           int C, void* stream) {
  // content may be NULL: plain inverse transform with LL = 0 -- which is also the adjoint of haar_hf (DWT_Function.backward,
  // surrounding_extraction.py:30-41), i.e. its backward
  TSCD_CHECK_ARG(hf && out && C % 8 == 0 && ldc % 8 == 0 && ldo % 8 == 0 && H % 2 == 0 && W % 2 == 0, "haar_idwt_mul: bad args");
  hipLaunchKernelGGL(haar_idwt_mul_kernel, dim3(ew_grid((long)B * (H / 2) * (W / 2) * (C / 8))), dim3(EW_BLOCK), 0,
                     (hipStream_t)stream, (const f16*)hf, (const f16*)content, ldc, (f16*)out, ldo, B, H, W, C);
  TSCD_CHECK_LAUNCH("haar_idwt_mul");
  return TSCD_OK;
}


// ------------------------------------------------------------------------------------------------------
// ModelEMA.update (reference: yolox/utils/ema.py:48-60): v = d * v + (1 - d) * model_v over every floating state tensor.
// One launch for all of them: the host passes a chunk table {dst, src, count} (<= EMA_CHUNK elements per chunk, one workgroup
// each); each element is read and written once (the two multi-tensor passes of the torch formulation move 5 words per element,
// this moves 3).  Same arithmetic as `_foreach_mul_(v, d); _foreach_add_(v, src, alpha=1-d)`: fma(1-d, src, fl(d*v)).
// ------------------------------------------------------------------------------------------------------
#define EMA_CHUNK 16384
struct EmaChunk { float* dst; const float* src; long n; };
__global__ __launch_bounds__(256) void ema_update_kernel(const EmaChunk* __restrict__ table, float d, float omd) {
  const EmaChunk c = table[blockIdx.x];
  if ((((uintptr_t)c.dst | (uintptr_t)c.src) & 15) == 0) {
    const long n4 = c.n >> 2;
    float4* dv = (float4*)c.dst;
    const float4* sv = (const float4*)c.src;
    for (long i = threadIdx.x; i < n4; i += 256) {
      float4 a = dv[i];
      const float4 b = sv[i];
      a.x = __builtin_fmaf(omd, b.x, d * a.x); a.y = __builtin_fmaf(omd, b.y, d * a.y);
      a.z = __builtin_fmaf(omd, b.z, d * a.z); a.w = __builtin_fmaf(omd, b.w, d * a.w);
      dv[i] = a;
    }
    for (long i = (n4 << 2) + threadIdx.x; i < c.n; i += 256) c.dst[i] = __builtin_fmaf(omd, c.src[i], d * c.dst[i]);
  } else {
    for (long i = threadIdx.x; i < c.n; i += 256) c.dst[i] = __builtin_fmaf(omd, c.src[i], d * c.dst[i]);
  }
}
extern "C" int tscd_ema_chunk_elems(void) { return EMA_CHUNK; }
// table: device array of nchunks {float* dst, const float* src, long n} records (24 bytes each)
extern "C" int tscd_ema_update_f32(const void* table, long nchunks, float d, float one_minus_d, void* stream) {
  TSCD_CHECK_ARG(table || nchunks == 0, "ema_update: null table");
  if (nchunks == 0) return TSCD_OK;
  TSCD_CHECK_ARG(nchunks < (1L << 31), "ema_update: too many chunks");
  hipLaunchKernelGGL(ema_update_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, (const EmaChunk*)table, d, one_minus_d);
  TSCD_CHECK_LAUNCH("ema_update");
  return TSCD_OK;
}


// ------------------------------------------------------------------------------------------------------
// Step tail (reference: ovis_tscd_large.py:164-197 SGD groups, tscd_trainer.py:171-178, ema.py:48-60):
// Nesterov-momentum SGD with the GradScaler's unscale and skip-on-overflow, AND the EMA update of the same parameter, in one
// pass over the trainable parameters -- 4 reads + 3 writes per element instead of the 6 + 3 (fused SGD) plus 2 + 1 (EMA) of
// the separate launches, and one launch instead of one per parameter group plus one.
//   g = grad / grad_scale (+ wd * p);  buf = momentum * buf + g;  p -= lr * (g + momentum * buf)   [nesterov; buf = 0 before
//   the first step, which makes the first step's buf = g as in torch.optim.SGD];  skipped when *found_inf != 0;
//   ema = d * ema + (1 - d) * p   (always: the reference's ema.update runs after a skipped step too).
// Chunk table like the EMA kernel's; each chunk names its parameter group (lr, weight decay per group as kernel arguments).
// ------------------------------------------------------------------------------------------------------
#define SGD_MAX_GROUPS 8
struct SgdChunk { float* p; const float* g; float* buf; float* ema; long n; long group; };
struct SgdGroups { float lr[SGD_MAX_GROUPS]; float wd[SGD_MAX_GROUPS]; };
// grad_base (optional): device array of per-parameter gradient pointers.  A chunk then carries the BYTE OFFSET of its slice in `g` and
// its parameter's index in bits 8.. of `group`: gradients are re-allocated by zero_grad(set_to_none=True) every step, and with this
// indirection the host uploads ~150 pointers instead of rebuilding the chunk table whenever one moved (1.2 ms of host time per step).
__global__ __launch_bounds__(256) void sgd_ema_kernel(const SgdChunk* __restrict__ table, SgdGroups grp, float momentum, int nesterov,
                                                      const float* __restrict__ grad_scale, const float* __restrict__ found_inf,
                                                      float d, float omd, const float* const* __restrict__ grad_base) {
  SgdChunk c = table[blockIdx.x];
  if (grad_base) {
    c.g = (const float*)((const char*)grad_base[c.group >> 8] + (uintptr_t)c.g);
    c.group &= 255;
  }
  const float lr = grp.lr[c.group], wd = grp.wd[c.group];
  const float inv = grad_scale ? 1.0f / *grad_scale : 1.0f;
  const bool skip = found_inf && *found_inf != 0.f;
  auto upd = [&](float& p, float g, float& buf, float& ema) {
    if (!skip) {
      g *= inv;
      if (wd != 0.f) g = __builtin_fmaf(wd, p, g);
      buf = __builtin_fmaf(momentum, buf, g);
      g = nesterov ? __builtin_fmaf(momentum, buf, g) : buf;
      p = __builtin_fmaf(-lr, g, p);
    }
    ema = __builtin_fmaf(omd, p, d * ema);
  };
  const bool has_ema = c.ema != nullptr;
  if ((((uintptr_t)c.p | (uintptr_t)c.g | (uintptr_t)c.buf | (uintptr_t)c.ema) & 15) == 0) {
    const long n4 = c.n >> 2;
    for (long i = threadIdx.x; i < n4; i += 256) {
      float4 p = ((float4*)c.p)[i], b = ((float4*)c.buf)[i];
      const float4 g = ((const float4*)c.g)[i];
      float4 e = has_ema ? ((float4*)c.ema)[i] : float4{0.f, 0.f, 0.f, 0.f};
      upd(p.x, g.x, b.x, e.x); upd(p.y, g.y, b.y, e.y); upd(p.z, g.z, b.z, e.z); upd(p.w, g.w, b.w, e.w);
      if (!skip) { ((float4*)c.p)[i] = p; ((float4*)c.buf)[i] = b; }
      if (has_ema) ((float4*)c.ema)[i] = e;
    }
    for (long i = (n4 << 2) + threadIdx.x; i < c.n; i += 256) {
      float p = c.p[i], b = c.buf[i], e = has_ema ? c.ema[i] : 0.f;
      upd(p, c.g[i], b, e);
      if (!skip) { c.p[i] = p; c.buf[i] = b; }
      if (has_ema) c.ema[i] = e;
    }
  } else {
    for (long i = threadIdx.x; i < c.n; i += 256) {
      float p = c.p[i], b = c.buf[i], e = has_ema ? c.ema[i] : 0.f;
      upd(p, c.g[i], b, e);
      if (!skip) { c.p[i] = p; c.buf[i] = b; }
      if (has_ema) c.ema[i] = e;
    }
  }
}
// table: device array of nchunks {float* p, const float* grad, float* momentum_buffer, float* ema (or null), long n, long group}
// records (48 bytes); lr / wd: ngroups (<= 8) floats on the host; grad_scale / found_inf: device scalars or null.
extern "C" int tscd_sgd_ema_step_indirect_f32(const void* table, long nchunks, const void* grad_base, const float* lr, const float* wd,
                                              int ngroups, float momentum, int nesterov, const float* grad_scale, const float* found_inf,
                                              float d, float one_minus_d, void* stream);
extern "C" int tscd_sgd_ema_step_f32(const void* table, long nchunks, const float* lr, const float* wd, int ngroups, float momentum,
                                     int nesterov, const float* grad_scale, const float* found_inf, float d, float one_minus_d,
                                     void* stream) {
  return tscd_sgd_ema_step_indirect_f32(table, nchunks, nullptr, lr, wd, ngroups, momentum, nesterov, grad_scale, found_inf, d, one_minus_d,
                                        stream);
}
// grad_base: null, or a device array of per-parameter gradient pointers; the chunks then hold {p, byte offset into the parameter's
// gradient, buf, ema, n, group | parameter index << 8}.
extern "C" int tscd_sgd_ema_step_indirect_f32(const void* table, long nchunks, const void* grad_base, const float* lr, const float* wd,
                                              int ngroups, float momentum, int nesterov, const float* grad_scale, const float* found_inf,
                                              float d, float one_minus_d, void* stream) {
  TSCD_CHECK_ARG(table || nchunks == 0, "sgd_ema_step: null table");
  TSCD_CHECK_ARG(lr && wd && ngroups > 0 && ngroups <= SGD_MAX_GROUPS, "sgd_ema_step: 1..%d parameter groups", SGD_MAX_GROUPS);
  if (nchunks == 0) return TSCD_OK;
  TSCD_CHECK_ARG(nchunks < (1L << 31), "sgd_ema_step: too many chunks");
  SgdGroups g;
  for (int i = 0; i < SGD_MAX_GROUPS; ++i) { g.lr[i] = i < ngroups ? lr[i] : 0.f; g.wd[i] = i < ngroups ? wd[i] : 0.f; }
  hipLaunchKernelGGL(sgd_ema_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, (const SgdChunk*)table, g, momentum,
                     nesterov, grad_scale, found_inf, d, one_minus_d, (const float* const*)grad_base);
  TSCD_CHECK_LAUNCH("sgd_ema_step");
  return TSCD_OK;
}


// ------------------------------------------------------------------------------------------------------
// Gradients of a step into one flat arena and / or their non-finite check, one launch (reference: the GradScaler's inf check inside
// scaler.step, tscd_trainer.py:171-176, and DDP's flattening of the gradients into buckets, :230-231).
//   row = {float* dst (or null: check only), long src_off (bytes into the parameter's gradient), long n, long pi}; src = src_base[pi] + src_off
//   dst[i] = src[i] * scale      (scale = 1, or 1 / world for backends without an averaging all-reduce)
//   *found_inf = 1 if any element read is Inf / NaN (never cleared here)
// torch's _amp_foreach_non_finite_check_and_unscale_ needs 6 launches for the ~150 gradient tensors of a TSCD-L step; this is one.
// ------------------------------------------------------------------------------------------------------
struct GradRow { float* dst; long src_off; long n; long pi; };
__global__ __launch_bounds__(256) void grad_gather_check_kernel(const GradRow* __restrict__ table, const float* const* __restrict__ src_base,
                                                                float scale, float* __restrict__ found_inf) {
  const GradRow r = table[blockIdx.x];
  const float* src = (const float*)((const char*)src_base[r.pi] + r.src_off);
  bool bad = false;
  auto fin = [](float v) { return fabsf(v) <= 3.402823466e+38f; };      // false for Inf and NaN
  if ((((uintptr_t)src | (uintptr_t)r.dst) & 15) == 0) {
    const long n4 = r.n >> 2;
    for (long i = threadIdx.x; i < n4; i += 256) {
      float4 v = ((const float4*)src)[i];
      bad |= !(fin(v.x) && fin(v.y) && fin(v.z) && fin(v.w));
      if (r.dst) { v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; ((float4*)r.dst)[i] = v; }
    }
    for (long i = (n4 << 2) + threadIdx.x; i < r.n; i += 256) {
      const float v = src[i];
      bad |= !fin(v);
      if (r.dst) r.dst[i] = v * scale;
    }
  } else {
    for (long i = threadIdx.x; i < r.n; i += 256) {
      const float v = src[i];
      bad |= !fin(v);
      if (r.dst) r.dst[i] = v * scale;
    }
  }
  if (found_inf && __ballot(bad) != 0ull && (threadIdx.x & 63) == 0) *found_inf = 1.0f;      // every writer stores the same value
}
// table: device array of nrows {float* dst or NULL, long src_off, long n (<= tscd_ema_chunk_elems()), long pi}; src_base: device array of
// per-parameter gradient pointers; found_inf: device scalar or NULL.
extern "C" int tscd_grad_gather_check_f32(const void* table, long nrows, const void* src_base, float scale, float* found_inf, void* stream) {
  TSCD_CHECK_ARG((table && src_base) || nrows == 0, "grad_gather_check: null table");
  if (nrows == 0) return TSCD_OK;
  TSCD_CHECK_ARG(nrows < (1L << 31), "grad_gather_check: too many rows");
  hipLaunchKernelGGL(grad_gather_check_kernel, dim3((unsigned)nrows), dim3(256), 0, (hipStream_t)stream, (const GradRow*)table,
                     (const float* const*)src_base, scale, found_inf);
  TSCD_CHECK_LAUNCH("grad_gather_check");
  return TSCD_OK;
}

// ------------------------------------------------------------------------------------------------------
// BaseConv in training mode as ONE host call: conv (raw output + per-tile BN partial sums) -> batch statistics
// (scale/shift, saved mean / invstd, running statistics) -> apply + activation (+ residual).  Three launches, the same three
// kernels as the separate entry points -- what it saves is the host side: one foreign call and one set of argument
// marshalling per BaseConv instead of three (134 BaseConvs per TSCD-L step).
// small: [4][C] floats = scale | shift | save_mean | save_invstd.  raw may equal out (in-place apply) when nothing keeps raw.
// ------------------------------------------------------------------------------------------------------
extern "C" int tscd_conv2d_nhwc_f16(const void* x, const void* wpk, void* y, const float* bias, const float* scale, const float* shift,
                                    float* bn_partial, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int ldx, int ldy,
                                    int epilogue, void* stream);
extern "C" int tscd_conv_tile_m(void);
extern "C" int tscd_conv_stat_tiles(long M, int W, int Cin, int Cout, int ksize, int stride);
extern "C" int tscd_conv_bn_act_train_f16(const void* x, const void* wpk, void* raw, int ldraw, float* partial, int cstride, int B, int H,
                                          int W, int Cin, int Cout, int ksize, int stride, int ldx, const float* gamma, const float* beta,
                                          float eps, float momentum, float* running_mean, float* running_var, float* small, void* out,
                                          int ldo, const void* residual, int ldr, int act, void* stream) {
  TSCD_CHECK_ARG(x && wpk && raw && partial && small && out && gamma && beta, "conv_bn_act_train: null pointer");
  const int pad = (ksize - 1) / 2;
  const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
  const long M = (long)B * Ho * Wo;
  int st = tscd_conv2d_nhwc_f16(x, wpk, raw, nullptr, nullptr, nullptr, partial, B, H, W, Cin, Cout, ksize, stride, ldx, ldraw, 0, stream);
  if (st != TSCD_OK) return st;
  const int mtiles = tscd_conv_stat_tiles(M, W, Cin, Cout, ksize, stride);     // partial-sum entries the conv launch wrote
  st = tscd_bn_finalize(partial, mtiles, cstride, Cout, M, gamma, beta, eps, momentum, running_mean, running_var, small, small + 2 * (long)Cout,
                        small + 3 * (long)Cout, stream);
  if (st != TSCD_OK) return st;
  return tscd_bn_act_nhwc_f16(raw, ldraw, out, ldo, small, residual, ldr, M, Cout, act, stream);
}

// ------------------------------------------------------------------------------------------------------
// The same for SEVERAL BaseConvs in one conv launch (reference: the four towers of TSCDHead.forward, tscd_head.py:315-324, all read the
// stem output):
//   groups == 1: convs that share their INPUT run as one dense conv over the concatenated output channels (Ctot = nseg * seg_c rows
//                of the packed matrix, the stem output staged once for all of them);
//   groups == nseg > 1: convs with separate inputs (channel slices g * Cin of x) and outputs (slices g * seg_c of raw / out): one
//                grouped launch (tscd_conv2d_nhwc_f16_ex).
// Each of the nseg BatchNorms keeps its own gamma / beta / running statistics (arrays of nseg device pointers); small [4][Ctot] =
// scale | shift | save_mean | save_invstd over the concatenated channels; ONE statistics launch and ONE apply launch cover them all.
// ------------------------------------------------------------------------------------------------------
extern "C" int tscd_conv2d_nhwc_f16_ex(const void* x, const void* wpk, void* y, const float* bias, const float* scale, const float* shift,
                                       float* bn_partial, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int ldx, int ldy,
                                       int epilogue, int groups, int variant, int korder, int tile_m, void* stream);
extern "C" int tscd_conv_stat_tiles_ex(long M, int W, int Cin, int Cout, int ksize, int stride, int groups, int variant, int tile_m);
extern "C" int tscd_conv_tile_n(int cout);
extern "C" int tscd_conv_bn_act_train_multi_f16(const void* x, const void* wpk, void* raw, int ldraw, float* partial, int B, int H, int W,
                                                int Cin, int seg_c, int nseg, int groups, int ksize, int ldx, const void* const* gamma,
                                                const void* const* beta, float eps, float momentum, void* const* running_mean,
                                                void* const* running_var, float* small, void* out, int ldo, int act, void* stream) {
  TSCD_CHECK_ARG(x && wpk && raw && partial && small && out && gamma && beta, "conv_bn_act_train_multi: null pointer");
  TSCD_CHECK_ARG(nseg >= 1 && nseg <= 4 && (groups == 1 || groups == nseg), "conv_bn_act_train_multi: nseg=%d groups=%d", nseg, groups);
  const long M = (long)B * H * W;                     // stride 1, 'same' padding
  const int Ctot = nseg * seg_c;
  const int cout_launch = groups == 1 ? Ctot : seg_c; // output channels per group of the conv launch
  int st = tscd_conv2d_nhwc_f16_ex(x, wpk, raw, nullptr, nullptr, nullptr, partial, B, H, W, Cin, cout_launch, ksize, 1, ldx, ldraw, 0, groups,
                                   -1, 0, 0, stream);
  if (st != TSCD_OK) return st;
  const int mtiles = tscd_conv_stat_tiles_ex(M, W, Cin, cout_launch, ksize, 1, groups, -1, 0);
  const int bn = tscd_conv_tile_n(cout_launch);
  BnSegs sg;
  memset(&sg, 0, sizeof(sg));
  for (int i = 0; i < nseg; ++i) {
    sg.gamma[i] = (const float*)gamma[i]; sg.beta[i] = (const float*)beta[i];
    sg.rm[i] = running_mean ? (float*)running_mean[i] : nullptr; sg.rv[i] = running_var ? (float*)running_var[i] : nullptr;
    TSCD_CHECK_ARG(sg.gamma[i] && sg.beta[i] && ((sg.rm[i] == nullptr) == (sg.rv[i] == nullptr)), "conv_bn_act_train_multi: segment %d", i);
  }
  sg.seg_c = seg_c;
  sg.group_c = cout_launch;
  sg.group_cstride = (cout_launch + bn - 1) / bn * bn;
  sg.rows = groups * sg.group_cstride;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(Ctot, 4)), dim3(256), 0, (hipStream_t)stream, partial, mtiles, Ctot, (double)M, sg, eps,
                     momentum, small, small + 2 * (long)Ctot, small + 3 * (long)Ctot);
  TSCD_CHECK_LAUNCH("bn_finalize (multi)");
  return tscd_bn_act_nhwc_f16(raw, ldraw, out, ldo, small, nullptr, 0, M, Ctot, act, stream);
}
