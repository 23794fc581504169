// Backward kernels of the trainable head convs (cls_convs2 / reg_convs2 towers, WaveletsHFBlock filters; reference:
// autograd of network_blocks.py:36-61 and surrounding_extraction.py:223-232 under tscd_trainer.py:171-176).
//
// conv_wgrad  dW[cout][tap,cin] = sum_pixels dY[pix][cout] * X[pix shifted by tap][cin]  (stride-1, "same" padding)
//   GEMM with the PIXEL axis as K: both operands are stored pixel-major ([pix][channel], NHWC), i.e. K is the slow
//   index of both.  Tiles are staged into LDS as they lie in memory (coalesced 16-B rows) and the MFMA fragments are
//   fetched with gfx950's hardware-transposing LDS read ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group,
//   returned column-major): two of them give a lane its 8 consecutive k (pixels) for its channel.  LDS rows are
//   128 + 32 halves (320 B = 80 dwords = 16 mod 64 banks) so the 4 rows of a transposed read hit disjoint bank
//   quarters.  Split-K over pixel ranges (gridDim.y) writes fp32 slabs that a second kernel reduces in a fixed order
//   (deterministic, no float atomics) into the torch weight layout [Cout][Cin][KH][KW].
// The data gradient of these stride-1 convs needs no kernel of its own: it is conv_igemm.hip run on dY with the
// weights packed in "dgrad" order (flipped taps, in/out swapped).
//
// bn_silu_bwd_*  gradient of SiLU(BatchNorm_train(raw)) w.r.t. raw, gamma, beta in two passes over [M,C]:
//   pass 1: dt = dy * silu'(t), per-channel partial sums of dt and dt*xhat (fixed-order reduction via bn_bwd_reduce);
//   pass 2: draw = gamma*invstd * (dt - mean(dt) - xhat*mean(dt*xhat)).
#include "common.h"
#include <type_traits>

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgradParams {
  const f16* x;
  const f16* dy;
  float* part;   // [S][Cout][Ncols]
  int B, H, W, Cin, Cout, KH, KW, pad;
  int ldx, lddy;
  int M, Ncols, S, ktiles_per_split, mtiles, ntiles;
  // grouped launch (the towers' second convs: separate channel slices of one x / dY buffer): Cin / Cout per group; group g reads
  // channels [g*Cin, ...) of x and [g*Cout, ...) of dy and owns the slabs part[g][S][Cout][Ncols]
  int G;
};

__device__ __forceinline__ f16x8 tr_frag(const f16* tile_row0_col0, int ldst, int lane) {
  // 16-lane group: lane i = 4q+p supplies &tile[q][4p]; receives column i of rows 0..3
  const int i = lane & 15, q = i >> 2, p = i & 3;
  auto a0 = (__attribute__((address_space(3))) s16x4*)(tile_row0_col0 + q * ldst + 4 * p);
  auto a1 = (__attribute__((address_space(3))) s16x4*)(tile_row0_col0 + (q + 4) * ldst + 4 * p);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(a1);
  // reinterpret the 64-bit results as whole vectors (per-element bit_cast of the builtin's result is miscompiled
  // by hipcc 7.2: it broadcasts element 0)
  f16x4 lo4 = __builtin_bit_cast(f16x4, lo), hi4 = __builtin_bit_cast(f16x4, hi);
  f16x8 r = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
  return r;
}

__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
  constexpr int BM = 128, BN = 128, BK = 32, LDST = 160;
  __shared__ __attribute__((aligned(16))) f16 Ys[2][BK][LDST];
  __shared__ __attribute__((aligned(16))) f16 Xs[2][BK][LDST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int mt = blockIdx.x / p.ntiles, nt = blockIdx.x % p.ntiles;
  const int c0 = mt * BM, n0 = nt * BN;
  const int s = blockIdx.y;
  p.x += blockIdx.z * p.Cin;
  p.dy += blockIdx.z * p.Cout;
  p.part += (long)blockIdx.z * p.S * p.Cout * p.Ncols;
  const int kt0 = s * p.ktiles_per_split;
  int kt1 = kt0 + p.ktiles_per_split;
  const int ktotal = (p.M + BK - 1) / BK;
  if (kt1 > ktotal) kt1 = ktotal;

  // staging assignment: 2 (pixel row, 8-channel chunk) pairs per thread and operand
  const int chunk = tid & 15, prow = tid >> 4;   // prow 0..15 (+16)
  const int ych = c0 + chunk * 8;                // dY channels
  const bool y_ok = ych < p.Cout;
  const int ncol = n0 + chunk * 8;               // (tap, cin) column of X
  const bool n_ok = ncol < p.Ncols;
  const int tap = n_ok ? ncol / p.Cin : 0, cin = n_ok ? ncol - tap * p.Cin : 0;
  const int dh = tap / p.KW - p.pad, dw = tap % p.KW - p.pad;
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  f16x8 yreg[2], xreg[2];

  auto load_tiles = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int m = kt * BK + prow + 16 * i;
      bool mok = m < p.M;
      int mm = mok ? m : 0;
      yreg[i] = (mok && y_ok) ? *(const f16x8*)(p.dy + (long)mm * p.lddy + ych) : zero8;
      int b = mm / (p.H * p.W);
      int rem = mm - b * (p.H * p.W);
      int ho = rem / p.W, wo = rem - ho * p.W;
      int hi = ho + dh, wi = wo + dw;
      bool ok = mok && n_ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      xreg[i] = ok ? *(const f16x8*)(p.x + (((long)b * p.H + hi) * p.W + wi) * p.ldx + cin) : zero8;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *(f16x8*)(&Ys[buf][prow + 16 * i][chunk * 8]) = yreg[i];
      *(f16x8*)(&Xs[buf][prow + 16 * i][chunk * 8]) = xreg[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (kt0 < kt1) {
    load_tiles(kt0);
    store_tiles(0);
  }
  __syncthreads();
  int buf = 0;
  // fragment geometry (see tr_frag): 16-lane group g -> channel block 16*(g&1), k block 8*(g>>1)
  const int g = lane >> 4;
  for (int kt = kt0; kt < kt1; ++kt) {
    if (kt + 1 < kt1) load_tiles(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kb = ks * 16 + 8 * (g >> 1);
      f16x8 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = tr_frag(&Ys[buf][kb][wm * 64 + i * 32 + 16 * (g & 1)], LDST, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = tr_frag(&Xs[buf][kb][wn * 64 + j * 32 + 16 * (g & 1)], LDST, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < kt1) store_tiles(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // partial tile -> part[s][row][col]: one base pointer per lane, row offsets as multiples of the (wave-uniform) row pitch; tiles that
  // lie inside the matrix (all but the last row / column of tiles) skip the per-element bounds tests
  {
    const int row_w = c0 + wm * 64 + 4 * (lane >> 5), col_w = n0 + wn * 64 + (lane & 31);
    float* const base = p.part + ((long)s * p.Cout + row_w) * p.Ncols + col_w;
    const long pitch = p.Ncols;
    const bool inside = c0 + 128 <= p.Cout && n0 + 128 <= p.Ncols;          // wave-uniform (the workgroup's tile)
    if (inside) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) base[(i * 32 + (e & 3) + 8 * (e >> 2)) * pitch + j * 32] = acc[i][j][e];
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int ro = i * 32 + (e & 3) + 8 * (e >> 2);
            if (row_w + ro < p.Cout && col_w + j * 32 < p.Ncols) base[ro * pitch + j * 32] = acc[i][j][e];
          }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant of the weight-gradient GEMM (the one that runs; the register-staged kernel above stays as the
// bit-identical cross-check, tscd_conv_wgrad_set_variant(0)).  Same tiles and the same k order per split, but
//   * both operands are fetched with global_load_lds_dwordx4 into a 4-deep ring (one wave-instruction = 4 pixel rows x
//     256 B): no staging registers, no ds_write pass, no per-tile integer divisions -- a lane's source pointers advance by
//     32 pixels per k-tile and its (ho, wo) image position is carried incrementally for the halo test;
//   * the LDS image is lane-linear (256-B rows, no padding), so the 16-B chunk c of row r lives in slot c ^ ((r&3) << 2):
//     the four rows of a transposing read then sit in the four 16-bank quarters;
//   * fragments of tile kt+1 are read into a second register set while the MFMAs of tile kt issue; one raw s_barrier and
//     one counted vmcnt per k-step (two younger tiles stay in flight).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) f16 g_wg_zero_chunk[8];

template <int N> __device__ __forceinline__ void wg_wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else static_assert(N < 0, "unsupported vmcnt");
}

// Transposing fragment fetch as inline asm: hipcc 7.2 orders the ds_read_tr16_b64 builtin behind EVERY pending LDS-DMA
// (s_waitcnt vmcnt(0) before the first read of each k-step -- the builtin carries no memory operand for its alias check),
// which drains the ring the loop keeps in flight.  The asm form is invisible to that pass; the loop below waits for its own
// reads with an explicit lgkmcnt(0) before the MFMAs that consume them.
template <int OFF> __device__ __forceinline__ f16x8 tr_frag_asm(unsigned addr) {
  s16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 4 * 256));
  f16x4 lo4 = __builtin_bit_cast(f16x4, lo), hi4 = __builtin_bit_cast(f16x4, hi);
  return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
}

// One 128 x 128 tile of one split: the body shared by the single-problem kernel and the job-table kernel below.
__device__ __forceinline__ void wgrad_glds_tile(const WgradParams& p, const int s, const int mt, const int nt) {
  constexpr int BM = 128, BN = 128, BK = 32, NS = 4, LD = 4;
  constexpr int T_BYTES = BK * 256, STAGE = 2 * T_BYTES;
  __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c0 = mt * BM, n0 = nt * BN;
  const int ktotal = (p.M + BK - 1) / BK;
  const int kt0 = s * p.ktiles_per_split;
  int kt1 = kt0 + p.ktiles_per_split;
  if (kt1 > ktotal) kt1 = ktotal;
  const int nk = kt1 - kt0;

  // ---- per-lane source state: a wave-instruction covers rows base..base+3; lane -> row lane>>4, slot lane&15 ----
  const int rsub = lane >> 4;                         // == row & 3 (bases are multiples of 4)
  const int chunk = (lane & 15) ^ (rsub << 2);
  const int ych = c0 + chunk * 8;
  const bool y_ok = ych < p.Cout;
  const int ncol = n0 + chunk * 8;
  const bool n_ok = ncol < p.Ncols;
  const int tap = n_ok ? ncol / p.Cin : 0, cin = n_ok ? ncol - tap * p.Cin : 0;
  const int dh = tap / p.KW - p.pad, dw = tap % p.KW - p.pad;
  const int step_q = BK / p.W, step_r = BK % p.W;
  const f16* y_src[2];
  const f16* x_src[2];
  int m_row[2], ho[2], wo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = kt0 * BK + wave * 8 + i * 4 + rsub;
    m_row[i] = m;
    const int rem = m % (p.H * p.W);
    ho[i] = rem / p.W;
    wo[i] = rem - ho[i] * p.W;
    y_src[i] = p.dy + (long)m * p.lddy + ych;
    x_src[i] = p.x + ((long)m + dh * p.W + dw) * p.ldx + cin;
  }
  auto issue = [&](int stage) {
    char* sbase = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool mok = m_row[i] < p.M;
      const f16* ys = (mok && y_ok) ? y_src[i] : g_wg_zero_chunk;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ys,
                                       (__attribute__((address_space(3))) void*)(sbase + (wave * 8 + i * 4) * 256), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = m_row[i] < p.M && n_ok && (p.KH == 1 || ((unsigned)(ho[i] + dh) < (unsigned)p.H && (unsigned)(wo[i] + dw) < (unsigned)p.W));
      const f16* xs = ok ? x_src[i] : g_wg_zero_chunk;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xs,
                                       (__attribute__((address_space(3))) void*)(sbase + T_BYTES + (wave * 8 + i * 4) * 256), 16, 0, 0);
      m_row[i] += BK;
      y_src[i] += (long)BK * p.lddy;
      x_src[i] += (long)BK * p.ldx;
      wo[i] += step_r;
      ho[i] += step_q;
      if (wo[i] >= p.W) { wo[i] -= p.W; ++ho[i]; }
      ho[i] = ho[i] >= p.H ? ho[i] - p.H : ho[i];      // next image (the launcher guarantees 32 / W + 1 <= H for 3x3)
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment byte offsets inside an operand tile: 16-lane group g -> channel block 16*(g&1), pixel block 8*(g>>1);
  // lane 4q+p of the group supplies row q, halves 4p..4p+3 of the block (see tr_frag)
  const int g = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
  int a_off[2], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ca = (wm * 64 + i * 32 + 16 * (g & 1)) / 8 + (fp >> 1);
    const int cb = (wn * 64 + i * 32 + 16 * (g & 1)) / 8 + (fp >> 1);
    a_off[i] = (8 * (g >> 1) + fq) * 256 + ((ca ^ (fq << 2)) * 16) + (fp & 1) * 8;
    b_off[i] = T_BYTES + (8 * (g >> 1) + fq) * 256 + ((cb ^ (fq << 2)) * 16) + (fp & 1) * 8;
  }

  if (nk > 0) {
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
      if (st < nk) issue(st);
    f16x8 fa[2][2][2], fb[2][2][2];     // [register set][ks][tile]
    auto wait_tile = [&](int t) __attribute__((always_inline)) {
      const int younger = nk - 1 - t;
      if (younger >= NS - 2) wg_wait_vmcnt<(NS - 2) * LD>();
      else if (younger == 1) wg_wait_vmcnt<LD>();
      else wg_wait_vmcnt<0>();
    };
    auto read_frags = [&](auto setc, int t) __attribute__((always_inline)) {
      constexpr int S = decltype(setc)::value;
      const unsigned sb = smem_base + (t % NS) * STAGE;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[S][0][i] = tr_frag_asm<0>(sb + a_off[i]);
        fa[S][1][i] = tr_frag_asm<16 * 256>(sb + a_off[i]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        fb[S][0][j] = tr_frag_asm<0>(sb + b_off[j]);
        fb[S][1][j] = tr_frag_asm<16 * 256>(sb + b_off[j]);
      }
    };
    auto drain_reads = [&]() __attribute__((always_inline)) {   // the asm reads are not tracked by the compiler's waitcnt pass
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto mfmas = [&](auto setc) __attribute__((always_inline)) {
      constexpr int S = decltype(setc)::value;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][ks][i], fb[S][ks][j], acc[i][j], 0, 0, 0);
    };
    auto phase = [&](auto setc, int kt) __attribute__((always_inline)) {
      constexpr int S = decltype(setc)::value;
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): this wave's fragments of tile kt are in registers
      wait_tile(kt + 1);
      __builtin_amdgcn_s_barrier();                            // everyone's are: tile kt's buffer may be refilled
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
      drain_reads();
      mfmas(std::integral_constant<int, 1>{});
    } else {
      drain_reads();
      mfmas(std::integral_constant<int, 0>{});
    }
  }
  // partial tile -> part[s][row][col]: one base pointer per lane, row offsets as multiples of the (wave-uniform) row pitch; tiles that
  // lie inside the matrix (all but the last row / column of tiles) skip the per-element bounds tests
  {
    const int row_w = c0 + wm * 64 + 4 * (lane >> 5), col_w = n0 + wn * 64 + (lane & 31);
    float* const base = p.part + ((long)s * p.Cout + row_w) * p.Ncols + col_w;
    const long pitch = p.Ncols;
    const bool inside = c0 + 128 <= p.Cout && n0 + 128 <= p.Ncols;          // wave-uniform (the workgroup's tile)
    if (inside) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) base[(i * 32 + (e & 3) + 8 * (e >> 2)) * pitch + j * 32] = acc[i][j][e];
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int ro = i * 32 + (e & 3) + 8 * (e >> 2);
            if (row_w + ro < p.Cout && col_w + j * 32 < p.Ncols) base[ro * pitch + j * 32] = acc[i][j][e];
          }
    }
  }
}

__global__ __launch_bounds__(256, 2) void conv_wgrad_glds_kernel(WgradParams p) {
  // XCD-aware order: the workgroups of one split (same pixel range) share an L2
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tiles = p.mtiles * p.ntiles;
  const int per_group = tiles * p.S;
  const int grp = lid / per_group;
  const int lid_g = lid - grp * per_group;
  p.x += grp * p.Cin;
  p.dy += grp * p.Cout;
  p.part += (long)grp * p.S * p.Cout * p.Ncols;
  const int s = lid_g / tiles, tile = lid_g - s * tiles;
  const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
  wgrad_glds_tile(p, s, mt, nt);
}

// Job-table form: SEVERAL independent weight-gradient GEMMs -- nn.Linear layers (1x1: dW_j[out_j][in_j] = dY_j^T X_j over M_j rows) and
// stride-1 convs (3x3 / 1x1, grouped or not) -- in ONE launch.  The proposal stage's ~25 linears have a few thousand rows each and the
// 36x36 / 18x18 pyramid levels' convs a few thousand pixels: launched one by one their weight gradients are launch- and latency-bound
// (4 GFLOP in 50 us, tools/wgrad_census.py); nothing in the backward pass waits for them, so the host collects them and flushes them
// together (tscd_amd/ops.py: deferred weight gradients).  A workgroup finds its job by its index (jobs are laid out back to back,
// longest walk first); unsplit 1x1 jobs write straight into dw (torch layout), the others fp32 slabs that ONE more launch reduces.
#define WGRAD_MAX_JOBS 40
struct WgradJob {
  const f16* x; const f16* dy; float* part;
  int B, H, W, Cin, Cout, KH, ldx, lddy, M, Ncols, S, ktiles_per_split, mtiles, ntiles, G, wg_begin;
};
struct WgradJobTable { WgradJob job[WGRAD_MAX_JOBS]; int njobs; };
__global__ __launch_bounds__(256, 2) void conv_wgrad_glds_jobs_kernel(WgradJobTable t) {
  const int bid = blockIdx.x;
  int j = 0;
  for (int i = 1; i < t.njobs; ++i) j = bid >= t.job[i].wg_begin ? i : j;        // uniform: scalar loads and compares
  const WgradJob q = t.job[j];
  WgradParams p;
  p.B = q.B; p.H = q.H; p.W = q.W; p.Cin = q.Cin; p.Cout = q.Cout; p.KH = q.KH; p.KW = q.KH; p.pad = (q.KH - 1) / 2;
  p.ldx = q.ldx; p.lddy = q.lddy;
  p.M = q.M; p.Ncols = q.Ncols; p.S = q.S; p.ktiles_per_split = q.ktiles_per_split; p.mtiles = q.mtiles; p.ntiles = q.ntiles;
  p.G = q.G;
  // XCD-aware order inside the job, as the single-GEMM launch above: workgroup ids go round-robin over the 8 XCDs, so the job's ids of
  // one residue class (one XCD) take a CONTIGUOUS range of (group, split, tile) indices -- the 36 tiles of a split of a 256-channel 3x3
  // conv read the same pixel rows of x and dy (1.5 MB), which then come from memory once per XCD that holds them instead of once
  // per XCD of the chip (the job kernel dealt consecutive tiles to consecutive XCDs: 568 MB of fabric reads per launch, 4.7 TB/s)
  const int tiles = q.mtiles * q.ntiles, per_group = tiles * q.S, nwg = per_group * q.G;
  const int rel = bid - q.wg_begin, cls = rel & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int lid = (cls < r8 ? cls * (q8 + 1) : r8 * (q8 + 1) + (cls - r8) * q8) + (rel >> 3);
  const int grp = lid / per_group, lid_g = lid - grp * per_group;
  p.x = q.x + grp * q.Cin;
  p.dy = q.dy + grp * q.Cout;
  p.part = q.part + (long)grp * q.S * q.Cout * q.Ncols;
  const int s = lid_g / tiles, tile = lid_g - s * tiles;
  const int mt = tile / q.ntiles, nt = tile - mt * q.ntiles;
  wgrad_glds_tile(p, s, mt, nt);
}

// dW[cout][cin][kh][kw] (fp32, torch layout) = sum_s part[s][cout][(kh*KW+kw)*Cin + cin].  Threads walk the slabs in SLAB
// order (a wave reads 256 contiguous bytes of every slab; the S-times smaller dw stream takes the scattered side), four
// wave-groups share the splits of 64 outputs and combine through LDS in a fixed order (deterministic, no atomics).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int S, int Cout, int Cin, int KH, int KW, int G,
                                                          float* __restrict__ dw) {
  __shared__ float red[4][64];
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int taps = KH * KW, Ncols = taps * Cin;
  const long total_g = (long)Cout * Ncols, total = total_g * G;
  for (long base = (long)blockIdx.x * 64; base < total; base += (long)gridDim.x * 64) {
    const long src = base + o;
    const long grp = src / total_g, r = src - grp * total_g;
    float a = 0.f;
    if (src < total) {
      const float* pg = part + grp * S * total_g + r;
#pragma unroll 4
      for (int s = g; s < S; s += 4) a += pg[(long)s * total_g];
    }
    red[g][o] = a;
    __syncthreads();
    if (g == 0 && src < total) {
      a = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
      const int co = (int)(r / Ncols), col = (int)(r - (long)co * Ncols);
      const int tap = col / Cin, ci = col - tap * Cin;
      dw[grp * total_g + ((long)co * Cin + ci) * taps + tap] = a;
    }
    __syncthreads();
  }
}

// Split-K over the pixel axis: as many splits as keep tiles x splits within the 512 resident workgroups, at least 8 k-tiles
// (256 pixels) per split so the fp32 slab traffic stays small next to the operand reads.
static int wgrad_splits(long M, int Cout, int Cin, int ksize, int G) {
  long kt = (M + 31) / 32;
  long tiles = (long)cdiv(Cout, 128) * cdiv((long)ksize * ksize * Cin, 128) * G;
  long s = 512 / tiles;                      // two workgroups per CU, never a second round (540 blocks take ~1.5x the time of 504)
  if (ksize == 1 && kt <= 24) return 1;      // proposal-stage linears (a few hundred rows): one pass, written straight into dw
  if (s > kt / 8) s = kt / 8;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return (int)s;
}
extern "C" int tscd_conv_wgrad_splits(long M, int Cout, int Cin, int ksize) { return wgrad_splits(M, Cout, Cin, ksize, 1); }
extern "C" long tscd_conv_wgrad_workspace_bytes(long M, int Cout, int Cin, int ksize) {
  return 4L * wgrad_splits(M, Cout, Cin, ksize, 1) * Cout * ksize * ksize * Cin + 256;
}
extern "C" long tscd_conv_wgrad_workspace_bytes_ex(long M, int Cout, int Cin, int ksize, int groups) {
  return 4L * groups * wgrad_splits(M, Cout, Cin, ksize, groups) * Cout * ksize * ksize * Cin + 256;
}

// groups > 1: `groups` weight gradients in one launch -- group g from channels [g*Cin, ...) of x and [g*Cout, ...) of dy into
// dw[g][Cout][Cin][k][k].  variant (tests only): 1 the LDS-DMA ring, 0 the register-staged loop kept as the cross-check (same tiles,
// splits and accumulation order: dw bit-identical).
static int wgrad_launch(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin, int Cout, int ksize, int ldx, int lddy,
                        int G, int variant, void* workspace, long workspace_bytes, void* stream) {
  TSCD_CHECK_ARG(x && dy && dw && workspace, "conv_wgrad: null pointer");
  TSCD_CHECK_ARG(ksize == 1 || ksize == 3, "conv_wgrad: ksize %d", ksize);
  TSCD_CHECK_ARG(G >= 1 && ldx >= G * Cin && lddy >= G * Cout, "conv_wgrad: groups=%d ldx=%d lddy=%d", G, ldx, lddy);
  TSCD_CHECK_ARG(Cin % 8 == 0 && Cout % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0, "conv_wgrad: channels / strides must be multiples of 8");
  TSCD_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "conv_wgrad: pointers must be 16-byte aligned");
  WgradParams p;
  p.x = (const f16*)x; p.dy = (const f16*)dy; p.part = (float*)workspace;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = ksize; p.KW = ksize; p.pad = (ksize - 1) / 2;
  p.ldx = ldx; p.lddy = lddy;
  long M = (long)B * H * W;
  TSCD_CHECK_ARG(M < (1L << 31), "conv_wgrad: M too large");
  p.M = (int)M; p.Ncols = ksize * ksize * Cin;
  p.G = G;
  p.S = wgrad_splits(M, Cout, Cin, ksize, G);
  TSCD_CHECK_ARG(workspace_bytes >= tscd_conv_wgrad_workspace_bytes_ex(M, Cout, Cin, ksize, G), "conv_wgrad: workspace too small");
  int ktotal = (p.M + 31) / 32;
  p.ktiles_per_split = (ktotal + p.S - 1) / p.S;
  p.mtiles = cdiv(Cout, 128); p.ntiles = cdiv(p.Ncols, 128);
  hipStream_t st = (hipStream_t)stream;
  const bool direct = p.S == 1 && ksize == 1;   // [Cout][Cin] slab == torch layout: no reduction pass
  if (direct) p.part = dw;
  // the LDS-DMA loop carries (ho, wo) incrementally with a single wrap per k-tile: images at least 32 / W + 1 rows high
  const bool dma_ok = ksize == 1 || 32 / W + 1 <= H;
  if (variant == 0 || !dma_ok)
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(p.mtiles * p.ntiles, p.S, G), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(conv_wgrad_glds_kernel, dim3(p.mtiles * p.ntiles * p.S * G), dim3(256), 0, st, p);
  TSCD_CHECK_LAUNCH("conv_wgrad");
  if (direct) return TSCD_OK;
  long total = (long)G * Cout * p.Ncols;
  long grid = (total + 63) / 64;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(grid > 8192 ? 8192 : grid)), dim3(256), 0, st, p.part, p.S, Cout, Cin, ksize, ksize, G, dw);
  TSCD_CHECK_LAUNCH("wgrad_reduce");
  return TSCD_OK;
}
extern "C" int tscd_conv_wgrad_nhwc_f16(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin, int Cout, int ksize,
                                        int ldx, int lddy, void* workspace, long workspace_bytes, void* stream) {
  return wgrad_launch(x, dy, dw, B, H, W, Cin, Cout, ksize, ldx, lddy, 1, 1, workspace, workspace_bytes, stream);
}
extern "C" int tscd_conv_wgrad_nhwc_f16_ex(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin, int Cout, int ksize,
                                           int ldx, int lddy, int groups, int variant, void* workspace, long workspace_bytes, void* stream) {
  return wgrad_launch(x, dy, dw, B, H, W, Cin, Cout, ksize, ldx, lddy, groups, variant, workspace, workspace_bytes, stream);
}

// Several weight gradients in one launch (see conv_wgrad_glds_jobs_kernel).  jobs: host array of njobs records
// {x, dy, dw, B, H, W, cin, cout, ksize, ldx, lddy, groups}: a stride-1 "same" conv of B x H x W pixels (an nn.Linear over M rows is
// B = H = 1, W = M, ksize = 1), dw [groups * cout][cin][k][k] fp32 (torch layout).  Every workgroup walks at most ~48 k-tiles (1536
// pixels): longer jobs are split over the pixels into fp32 slabs in ``workspace``, which a second launch adds in slab order and lays
// out as dw -- deterministic.  Unsplit ungrouped 1x1 jobs write dw directly.  At most 40 jobs per call (the job table travels as a kernel argument: 40 x 88 bytes).
struct WgradJobHost { const void* x; const void* dy; float* dw; int B, H, W, cin, cout, ksize, ldx, lddy, groups, pad_; };
struct WgradRedJob { const float* part; float* dw; int S, Cout, Cin, K, G, wg_begin; };
struct WgradRedTable { WgradRedJob job[WGRAD_MAX_JOBS]; int njobs; };
// one job's slabs -> dw.  A thread adds the S slabs of four consecutive input channels of one (cout, tap) -- 16-byte loads, coalesced
// along the slab rows -- in slab order.  1x1: the slab row IS the dw row.  3x3: a workgroup takes one cout and 64 input channels, stages
// its 9 x 64 sums in LDS and writes them as the 576 contiguous floats they are in the torch layout [cout][cin][3][3].
__global__ __launch_bounds__(256) void wgrad_jobs_reduce_kernel(WgradRedTable t) {
  __shared__ float tr[64 * 9];
  const int bid = blockIdx.x;
  int j = 0;
  for (int i = 1; i < t.njobs; ++i) j = bid >= t.job[i].wg_begin ? i : j;
  const WgradRedJob q = t.job[j];
  const int taps = q.K * q.K, Ncols = taps * q.Cin;
  const long total_g = (long)q.Cout * Ncols;
  const int lb = bid - q.wg_begin;
  if (taps == 1) {
    const long total = total_g * q.G;
    const long src = ((long)lb * 256 + threadIdx.x) * 4;
    if (src >= total) return;
    const long grp = src / total_g;
    const float* pg = q.part + grp * (q.S - 1) * total_g + src;          // slab 0 of the group that owns element `src`
    f32x4 a = *(const f32x4*)pg;
    for (int s = 1; s < q.S; ++s) a += *(const f32x4*)(pg + (long)s * total_g);
    *(f32x4*)(q.dw + src) = a;
    return;
  }
  // 3x3: workgroup = (group, cout, block of 64 input channels)
  const int cblocks = (q.Cin + 63) / 64;
  const int cb = lb % cblocks, row = lb / cblocks;                       // row = grp * Cout + co
  const int grp = row / q.Cout, co = row - grp * q.Cout;
  const int ci0 = cb * 64, nci = q.Cin - ci0 < 64 ? q.Cin - ci0 : 64;     // multiple of 8
  const int tid = threadIdx.x;
  const int tap = tid >> 4, c4 = (tid & 15) * 4;
  if (tap < taps && c4 < nci) {
    const float* pg = q.part + (long)grp * q.S * total_g + (long)co * Ncols + tap * q.Cin + ci0 + c4;
    f32x4 a = *(const f32x4*)pg;
    for (int s = 1; s < q.S; ++s) a += *(const f32x4*)(pg + (long)s * total_g);
    tr[(c4 + 0) * 9 + tap] = a[0]; tr[(c4 + 1) * 9 + tap] = a[1]; tr[(c4 + 2) * 9 + tap] = a[2]; tr[(c4 + 3) * 9 + tap] = a[3];
  }
  __syncthreads();
  float* d = q.dw + (long)grp * total_g + ((long)co * q.Cin + ci0) * 9;
  for (int i = tid; i < nci * 9; i += 256) d[i] = tr[i];
}
// Splits of a job: none up to 80 k-tiles (2560 pixels); beyond, ~48 k-tiles per workgroup, but never more than ~512 workgroups per
// job (the 72 x 72 level's convs: 7 slabs of 370 k-tiles instead of 54 -- their slab traffic would otherwise rival the operand reads).
static int wgrad_job_splits_t(long M, long tiles) {
  const long kt = (M + 31) / 32;
  if (kt <= 80) return 1;
  long s = (kt + 47) / 48, cap = 512 / (tiles > 0 ? tiles : 1);
  if (cap < 2) cap = 2;
  if (s > cap) s = cap;
  return (int)(s < 2 ? 2 : s);
}
static int wgrad_job_splits(const WgradJobHost& a) {
  return wgrad_job_splits_t((long)a.B * a.H * a.W, (long)cdiv(a.cout, 128) * cdiv(a.ksize * a.ksize * a.cin, 128) * a.groups);
}
static bool wgrad_job_direct(const WgradJobHost& a) { return a.ksize == 1 && a.groups == 1 && wgrad_job_splits(a) == 1; }
extern "C" int tscd_wgrad_batch_max_jobs(void) { return WGRAD_MAX_JOBS; }
extern "C" long tscd_wgrad_batch_workspace_bytes(const void* jobs_host, int njobs) {
  const WgradJobHost* h = (const WgradJobHost*)jobs_host;
  long b = 256;
  for (int i = 0; i < njobs; ++i)
    if (!wgrad_job_direct(h[i]))
      b += 4L * h[i].groups * wgrad_job_splits(h[i]) * h[i].cout * h[i].ksize * h[i].ksize * h[i].cin + 256;
  return b;
}
// 1 when the LDS-DMA loop can take the job (3x3: images at least 32 / W + 1 rows high; channels / strides multiples of 8)
extern "C" int tscd_wgrad_batch_supported(int B, int H, int W, int cin, int cout, int ksize, int ldx, int lddy, int groups) {
  if (!(ksize == 1 || ksize == 3) || groups < 1 || cin % 8 || cout % 8 || ldx % 8 || lddy % 8) return 0;
  if (ldx < groups * cin || lddy < groups * cout || (long)B * H * W >= (1L << 31) || (long)B * H * W <= 0) return 0;
  return (ksize == 1 || 32 / W + 1 <= H) ? 1 : 0;
}
extern "C" int tscd_wgrad_batch_f16(const void* jobs_host, int njobs, void* workspace, long workspace_bytes, void* stream) {
  TSCD_CHECK_ARG(jobs_host && njobs > 0 && njobs <= WGRAD_MAX_JOBS, "wgrad_batch: 1..%d jobs", WGRAD_MAX_JOBS);
  TSCD_CHECK_ARG(workspace_bytes >= tscd_wgrad_batch_workspace_bytes(jobs_host, njobs) && (workspace || workspace_bytes <= 256),
                 "wgrad_batch: workspace too small");
  const WgradJobHost* h = (const WgradJobHost*)jobs_host;
  // longest walk (k-tiles per workgroup) first: those workgroups start first and the short ones fill in behind them
  int order[WGRAD_MAX_JOBS];
  auto walk = [&](int i) {
    const long M = (long)h[i].B * h[i].H * h[i].W;
    const int S = wgrad_job_splits(h[i]);
    return ((M + 31) / 32 + S - 1) / S * (h[i].ksize == 3 ? 2 : 1);      // (3x3 k-tiles carry the halo test: a little longer)
  };
  for (int i = 0; i < njobs; ++i) order[i] = i;
  for (int i = 1; i < njobs; ++i) {
    int o = order[i], k = i;
    while (k > 0 && walk(order[k - 1]) < walk(o)) { order[k] = order[k - 1]; --k; }
    order[k] = o;
  }
  WgradJobTable t;
  WgradRedTable r;
  r.njobs = 0;
  long wg = 0, rwg = 0;
  char* ws = (char*)workspace;
  for (int i = 0; i < njobs; ++i) {
    const WgradJobHost& a = h[order[i]];
    TSCD_CHECK_ARG(a.x && a.dy && a.dw, "wgrad_batch: job %d: null pointer", order[i]);
    TSCD_CHECK_ARG(tscd_wgrad_batch_supported(a.B, a.H, a.W, a.cin, a.cout, a.ksize, a.ldx, a.lddy, a.groups),
                   "wgrad_batch: job %d: unsupported shape (B=%d H=%d W=%d cin=%d cout=%d k=%d groups=%d)", order[i], a.B, a.H, a.W, a.cin, a.cout,
                   a.ksize, a.groups);
    TSCD_CHECK_ARG(((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.dy & 15) == 0 && ((uintptr_t)a.dw & 15) == 0,
                   "wgrad_batch: job %d: pointers must be 16-byte aligned", order[i]);
    WgradJob& q = t.job[i];
    q.x = (const f16*)a.x; q.dy = (const f16*)a.dy;
    q.B = a.B; q.H = a.H; q.W = a.W; q.Cin = a.cin; q.Cout = a.cout; q.KH = a.ksize; q.ldx = a.ldx; q.lddy = a.lddy; q.G = a.groups;
    q.M = a.B * a.H * a.W; q.Ncols = a.ksize * a.ksize * a.cin;
    q.S = wgrad_job_splits(a);
    const int kt = (q.M + 31) / 32;
    q.ktiles_per_split = (kt + q.S - 1) / q.S;
    q.mtiles = cdiv(a.cout, 128); q.ntiles = cdiv(q.Ncols, 128);
    q.wg_begin = (int)wg;
    wg += (long)q.mtiles * q.ntiles * q.S * q.G;
    if (wgrad_job_direct(a)) {
      q.part = a.dw;
    } else {
      q.part = (float*)ws;
      const long n = (long)q.G * a.cout * q.Ncols;
      ws += (4L * q.S * n + 255) / 256 * 256;
      WgradRedJob& rj = r.job[r.njobs++];
      rj.part = q.part; rj.dw = a.dw; rj.S = q.S; rj.Cout = a.cout; rj.Cin = a.cin; rj.K = a.ksize; rj.G = q.G; rj.wg_begin = (int)rwg;
      rwg += a.ksize == 1 ? (n / 4 + 255) / 256 : (long)q.G * a.cout * ((a.cin + 63) / 64);
    }
  }
  TSCD_CHECK_ARG(wg < (1L << 31) && rwg < (1L << 31), "wgrad_batch: grid too large");
  t.njobs = njobs;
  hipLaunchKernelGGL(conv_wgrad_glds_jobs_kernel, dim3((unsigned)wg), dim3(256), 0, (hipStream_t)stream, t);
  TSCD_CHECK_LAUNCH("wgrad_batch");
  if (r.njobs > 0) {
    hipLaunchKernelGGL(wgrad_jobs_reduce_kernel, dim3((unsigned)rwg), dim3(256), 0, (hipStream_t)stream, r);
    TSCD_CHECK_LAUNCH("wgrad_batch (slab reduce)");
  }
  return TSCD_OK;
}

// Column sums of several dY matrices (the bias gradients of the pass's nn.Linear layers) in ONE launch: db_j[c] = sum_r dY_j[r][c].
// Workgroup = (job, 64 columns): 32 row lanes x 8 column chunks of 16 bytes, fp32 partial sums combined through LDS in row-lane order
// (deterministic).  jobs: host array of {dy [M][C] fp16 (row stride lddy), db [C] fp32, M, C, lddy}.
struct ColsumJobHost { const void* dy; float* db; int M, C, lddy, pad_; };
struct ColsumJob { const f16* dy; float* db; int M, C, lddy, wg_begin; };
struct ColsumTable { ColsumJob job[WGRAD_MAX_JOBS]; int njobs; };
__global__ __launch_bounds__(256) void colsum_jobs_kernel(ColsumTable t) {
  __shared__ float red[32][65];
  const int bid = blockIdx.x;
  int j = 0;
  for (int i = 1; i < t.njobs; ++i) j = bid >= t.job[i].wg_begin ? i : j;
  const ColsumJob q = t.job[j];
  const int c0 = (bid - q.wg_begin) * 64, chunk = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c = c0 + chunk * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (c < q.C) {                                   // C % 8 == 0: a chunk is inside or outside
    const f16* p = q.dy + (long)rl * q.lddy + c;
    for (int r = rl; r < q.M; r += 32, p += 32L * q.lddy) {
      const f16x8 v = *(const f16x8*)p;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][chunk * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 64 && c0 + threadIdx.x < q.C) {
    float a = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) a += red[r][threadIdx.x];
    q.db[c0 + threadIdx.x] = a;
  }
}
extern "C" int tscd_colsum_batch_f16(const void* jobs_host, int njobs, void* stream) {
  TSCD_CHECK_ARG(jobs_host && njobs > 0 && njobs <= WGRAD_MAX_JOBS, "colsum_batch: 1..%d jobs", WGRAD_MAX_JOBS);
  const ColsumJobHost* h = (const ColsumJobHost*)jobs_host;
  ColsumTable t;
  int wg = 0;
  for (int i = 0; i < njobs; ++i) {
    TSCD_CHECK_ARG(h[i].dy && h[i].db && h[i].M > 0 && h[i].C > 0 && h[i].C % 8 == 0 && h[i].lddy % 8 == 0 && h[i].lddy >= h[i].C &&
                   ((uintptr_t)h[i].dy & 15) == 0, "colsum_batch: job %d: bad pointer / shape", i);
    t.job[i].dy = (const f16*)h[i].dy; t.job[i].db = h[i].db; t.job[i].M = h[i].M; t.job[i].C = h[i].C; t.job[i].lddy = h[i].lddy;
    t.job[i].wg_begin = wg;
    wg += cdiv(h[i].C, 64);
  }
  t.njobs = njobs;
  hipLaunchKernelGGL(colsum_jobs_kernel, dim3(wg), dim3(256), 0, (hipStream_t)stream, t);
  TSCD_CHECK_LAUNCH("colsum_batch");
  return TSCD_OK;
}

// ---------------------------------------------------------------------------------------------- BN + SiLU backward
#define BNB_ROWS 32
// SiLU'(t) with the fast gate (see sigmoid_fast_)
__device__ __forceinline__ float silu_grad_(float t) {
  float sg = sigmoid_fast_(t);
  return sg * (1.f + t * (1.f - sg));
}
// partial [nblk][2][C]: sum dt, sum dt*xhat over the block's BNB_ROWS rows.  Block = (256/cv) row groups x cv chunk
// threads; the row groups are combined through LDS in a fixed order.
// dY may come as two channel segments (the two trainable towers hand their gradients over as separate tensors): channels
// [0, cseg) from dy0 (row stride ld0), [cseg, C) from dy1 (row stride ld1); raw / draw are channel slices with row strides of their own.
struct BnBwdSrc {
  const f16* dy0;
  const f16* dy1;
  const float* gamma0;
  const float* gamma1;
  long ld0, ld1, ldraw, lddraw;
  int cseg, gseg;      // dY splits at channel cseg, gamma at channel gseg (independently: one dY tensor may span two BatchNorms)
};
__device__ __forceinline__ const f16* bnb_dy(const BnBwdSrc& q, long r, int c) {
  return c < q.cseg ? q.dy0 + r * q.ld0 + c : q.dy1 + r * q.ld1 + (c - q.cseg);
}
__global__ __launch_bounds__(256) void bn_silu_bwd_stats_kernel(BnBwdSrc q, const f16* __restrict__ raw,
                                                                const float* __restrict__ ss, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, long M, int C, int cv, int rg,
                                                                float* __restrict__ partial) {
  extern __shared__ float red[];   // [rg][2][C]
  const int tid = threadIdx.x;
  const int c8 = tid % cv, g = tid / cv;
  const long r0 = (long)blockIdx.x * BNB_ROWS;
  if (g < rg) {
    for (int cc = c8; cc < (C >> 3); cc += cv) {
      const int c = cc << 3;
      float s1[8], s2[8], sc[8], sh[8], mu[8], iv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] = s2[e] = 0.f; sc[e] = ss[c + e]; sh[e] = ss[C + c + e]; mu[e] = mean[c + e]; iv[e] = invstd[c + e]; }
      for (long r = r0 + g; r < r0 + BNB_ROWS && r < M; r += rg) {
        f16x8 gy = *(const f16x8*)bnb_dy(q, r, c), x = *(const f16x8*)(raw + r * q.ldraw + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float xr = (float)x[e];
          float t = (float)(f16)(xr * sc[e] + sh[e]);
          float dt = (float)gy[e] * silu_grad_(t);
          s1[e] += dt;
          s2[e] += dt * ((xr - mu[e]) * iv[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(g * 2 + 0) * C + c + e] = s1[e];
        red[(g * 2 + 1) * C + c + e] = s2[e];
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * C; i += blockDim.x) {
    float a = 0.f;
    for (int k = 0; k < rg; ++k) a += red[k * 2 * C + i];
    partial[(long)blockIdx.x * 2 * C + i] = a;
  }
}
// out[c] = sum_t partial[t][c] over nrows x ncols fp32 partials, fixed order, fp64 accumulation: 64 columns per block (a wave
// reads 256 contiguous bytes per row), the 16 waves take rows t = g mod 16 and combine through LDS.
__global__ __launch_bounds__(1024) void colsum_partials_kernel(const float* __restrict__ partial, int nrows, int ncols, float* __restrict__ out) {
  __shared__ double red[16][64];
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 64 + o;
  double a = 0.0;
  if (c < ncols) {
#pragma unroll 8
    for (int t = g; t < nrows; t += 16) a += (double)partial[(long)t * ncols + c];
  }
  red[g][o] = a;
  __syncthreads();
  if (g == 0 && c < ncols) {
    double r = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) r += red[q][o];
    out[c] = (float)r;
  }
}
__global__ void bn_silu_bwd_apply_kernel(BnBwdSrc q, const f16* __restrict__ raw, const float* __restrict__ ss,
                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                         const float* __restrict__ sums, long M, int C, f16* __restrict__ draw) {
  const int cv = C >> 3;
  const long total = M * cv;
  const float invn = 1.0f / (float)M;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx / cv;
    int c = (int)(idx - r * cv) << 3;
    f16x8 g = *(const f16x8*)bnb_dy(q, r, c), x = *(const f16x8*)(raw + r * q.ldraw + c), o;
    const float* gam = c < q.gseg ? q.gamma0 + c : q.gamma1 + (c - q.gseg);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float xr = (float)x[e];
      float t = (float)(f16)(xr * ss[c + e] + ss[C + c + e]);
      float dt = (float)g[e] * silu_grad_(t);
      float xh = (xr - mean[c + e]) * invstd[c + e];
      o[e] = (f16)(gam[e] * invstd[c + e] * (dt - sums[c + e] * invn - xh * sums[C + c + e] * invn));
    }
    *(f16x8*)(draw + r * q.lddraw + c) = o;
  }
}

// dy, raw, draw: [M, C] fp16 (row strides C in the plain form).  sums_out [2][C] = (dbeta, dgamma).  workspace: floats [nblk][2][C].
extern "C" long tscd_bn_silu_bwd_workspace_bytes(long M, int C) { return 4L * ((M + BNB_ROWS - 1) / BNB_ROWS) * 2 * C + 256; }
static int bn_silu_bwd_launch(const BnBwdSrc& q, const void* raw, const float* scale_shift, const float* mean, const float* invstd, long M,
                              int C, void* draw, float* sums_out, void* workspace, long workspace_bytes, void* stream) {
  TSCD_CHECK_ARG(q.dy0 && raw && scale_shift && mean && invstd && q.gamma0 && draw && sums_out && workspace, "bn_silu_bwd: null pointer");
  TSCD_CHECK_ARG(C % 8 == 0 && M > 0 && q.cseg % 8 == 0 && q.cseg > 0 && q.cseg <= C, "bn_silu_bwd: C / segment must be multiples of 8");
  TSCD_CHECK_ARG((q.cseg == C || q.dy1) && (q.gseg == C || q.gamma1) && q.gseg % 8 == 0 && q.gseg > 0 && q.gseg <= C,
                 "bn_silu_bwd: second segment without pointers");
  TSCD_CHECK_ARG(q.ld0 % 8 == 0 && q.ld1 % 8 == 0 && q.ldraw % 8 == 0 && q.lddraw % 8 == 0, "bn_silu_bwd: row strides must be multiples of 8");
  TSCD_CHECK_ARG(workspace_bytes >= tscd_bn_silu_bwd_workspace_bytes(M, C), "bn_silu_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int nblk = (int)((M + BNB_ROWS - 1) / BNB_ROWS);
  int cv = C / 8;
  if (cv > 256) cv = 256;
  int rg = 256 / cv;
  if (rg > BNB_ROWS) rg = BNB_ROWS;
  hipLaunchKernelGGL(bn_silu_bwd_stats_kernel, dim3(nblk), dim3(256), (size_t)rg * 2 * C * sizeof(float), st, q, (const f16*)raw, scale_shift,
                     mean, invstd, M, C, cv, rg, (float*)workspace);
  hipLaunchKernelGGL(colsum_partials_kernel, dim3(cdiv(2 * C, 64)), dim3(1024), 0, st, (const float*)workspace, nblk, 2 * C, sums_out);
  long total = M * (C / 8);
  int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(bn_silu_bwd_apply_kernel, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, st, q, (const f16*)raw, scale_shift, mean, invstd,
                     sums_out, M, C, (f16*)draw);
  TSCD_CHECK_LAUNCH("bn_silu_bwd");
  return TSCD_OK;
}
extern "C" int tscd_bn_silu_bwd_f16(const void* dy, const void* raw, const float* scale_shift, const float* mean, const float* invstd,
                                    const float* gamma, long M, int C, void* draw, float* sums_out, void* workspace,
                                    long workspace_bytes, void* stream) {
  BnBwdSrc q;
  q.dy0 = (const f16*)dy; q.dy1 = nullptr; q.gamma0 = gamma; q.gamma1 = nullptr;
  q.ld0 = C; q.ld1 = 0; q.ldraw = C; q.lddraw = C; q.cseg = C; q.gseg = C;
  return bn_silu_bwd_launch(q, raw, scale_shift, mean, invstd, M, C, draw, sums_out, workspace, workspace_bytes, stream);
}
// The general form: dY in up to two channel segments ([0, cseg) from dy0, [cseg, C) from dy1, each with its own row stride and gamma
// vector: the two trainable towers of TSCDHead share one fused conv launch, their BatchNorms stay separate modules), raw / draw as
// channel slices (row strides ldraw / lddraw).
extern "C" int tscd_bn_silu_bwd_f16_ex(const void* dy0, long lddy0, const void* dy1, long lddy1, int cseg, const void* raw, long ldraw,
                                       const float* scale_shift, const float* mean, const float* invstd, const float* gamma0,
                                       const float* gamma1, int gseg, long M, int C, void* draw, long lddraw, float* sums_out,
                                       void* workspace, long workspace_bytes, void* stream) {
  BnBwdSrc q;
  q.dy0 = (const f16*)dy0; q.dy1 = (const f16*)dy1; q.gamma0 = gamma0; q.gamma1 = gamma1;
  q.ld0 = lddy0; q.ld1 = lddy1; q.ldraw = ldraw; q.lddraw = lddraw; q.cseg = cseg; q.gseg = gseg;
  return bn_silu_bwd_launch(q, raw, scale_shift, mean, invstd, M, C, draw, sums_out, workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------------------------- fused nn.Linear backward
// dX = dY W, dW = dY^T X, db = column sums of dY in ONE host call (the matcher's ~23 linears per step sit on a launch-bound
// stretch of the backward: their cost is the host's, not the device's).  Kernels: weight pack (data-gradient order) + the
// MFMA GEMM for dX, the pixel-axis wgrad GEMM (+ slab reduction unless it writes dW directly), one column-sum kernel.
extern "C" int tscd_pack_conv_weight(const void* w, int src_is_f32, void* out, int Cout, int Cin, int CinPad, int KH, int KW,
                                     int rows_pad, int Kpad, int mode, void* stream);
extern "C" int tscd_conv2d_nhwc_f16(const void* x, const void* wpk, void* y, const float* bias, const float* scale, const float* shift,
                                    float* bn_partial, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int ldx, int ldy,
                                    int epilogue, void* stream);
extern "C" int tscd_conv_tile_n(int cout);

// db[c] += sum over rows of g[r][c] (optionally masked by y[r][c] > 0, in which case the masked gradient is also written):
// the bias gradient of a Linear / biased conv and, with the mask, the backward of the fused ReLU.  A thread keeps one
// 8-channel chunk, strides the block's rows, the row groups are combined through LDS, one fp32 atomic per column and block.
__global__ __launch_bounds__(256) void bias_relu_bwd_kernel(const f16* __restrict__ g, int ldg, const f16* __restrict__ y, int ldy,
                                                            f16* __restrict__ gm, int ldm, long M, int C, int cv, int rpb,
                                                            float* __restrict__ partial) {
  // thread = (8-channel chunk c8, row group rg); a block walks chunks of 4*rpb rows grid-strided, the 4 rows of a thread are
  // loaded before any is used (4 x 2 x 16 B in flight per thread); the block's column sums go to partial[block][C] (no
  // atomics: a few hundred blocks adding into the same eight cache lines serialise in L2 -- measured 0.2 us per block)
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x, c8 = tid % cv, rg = tid / cv;
  const long chunk = 4L * rpb;
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int c0 = 0; c0 < (C >> 3); c0 += cv) {           // one pass unless C > 2048
    const int cc = c0 + c8, c = cc << 3;
    const bool live = rg < rpb && cc < (C >> 3);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (live) {
      for (long base = blockIdx.x * chunk; base < M; base += gridDim.x * chunk) {
        f16x8 v[4], yy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long r = base + rg + (long)u * rpb;
          v[u] = r < M ? *(const f16x8*)(g + r * ldg + c) : zero8;
          if (y) yy[u] = r < M ? *(const f16x8*)(y + r * ldy + c) : zero8;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long r = base + rg + (long)u * rpb;
          if (y) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[u][e] = (float)yy[u][e] > 0.f ? v[u][e] : (f16)0.f;
            if (r < M) *(f16x8*)(gm + r * ldm + c) = v[u];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += (float)v[u][e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = live ? acc[e] : 0.f;
    __syncthreads();
    // red is [rpb][cv*8]: fold the row groups pairwise (all threads), then one coalesced store of the block's sums
    const int ncol = cv * 8;
    for (int n = rpb; n > 1;) {
      const int h = (n + 1) >> 1;
      for (int idx = tid; idx < (n - h) * ncol; idx += 256) red[idx] += red[idx + h * ncol];
      __syncthreads();
      n = h;
    }
    for (int j = tid; j < ncol; j += 256)
      if (c0 * 8 + j < C) partial[(long)blockIdx.x * C + c0 * 8 + j] = red[j];
    __syncthreads();
  }
}
static int bias_relu_bwd_blocks(long M, int C) {
  int cv = C / 8;
  if (cv > 256) cv = 256;
  const int rpb = 256 / cv;
  long nblk = (M + 4L * rpb - 1) / (4L * rpb);
  return (int)(nblk > 512 ? 512 : (nblk < 1 ? 1 : nblk));
}
extern "C" long tscd_bias_relu_bwd_workspace_bytes(long M, int C) { return 4L * bias_relu_bwd_blocks(M, C) * C + 256; }
// g [M][C] (row stride ldg), optional y (row stride ldy) and masked-gradient output gm (row stride ldm); db [C] fp32.
// C % 8 == 0.  workspace: per-block column sums, reduced in a fixed order (deterministic).
extern "C" int tscd_bias_relu_bwd_f16(const void* g, int ldg, const void* y, int ldy, void* gm, int ldm, long M, int C, float* db,
                                      void* workspace, long workspace_bytes, void* stream) {
  TSCD_CHECK_ARG(g && db && workspace && C % 8 == 0 && ldg % 8 == 0 && (!y || (gm && ldy % 8 == 0 && ldm % 8 == 0)), "bias_relu_bwd: bad args");
  TSCD_CHECK_ARG(workspace_bytes >= tscd_bias_relu_bwd_workspace_bytes(M, C), "bias_relu_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    (void)hipMemsetAsync(db, 0, sizeof(float) * C, st);
    return TSCD_OK;
  }
  int cv = C / 8;
  if (cv > 256) cv = 256;
  const int rpb = 256 / cv;
  const int nblk = bias_relu_bwd_blocks(M, C);
  hipLaunchKernelGGL(bias_relu_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, st, (const f16*)g, ldg, (const f16*)y, ldy, (f16*)gm, ldm,
                     M, C, cv, rpb, (float*)workspace);
  hipLaunchKernelGGL(colsum_partials_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, st, (const float*)workspace, nblk, C, db);
  TSCD_CHECK_LAUNCH("bias_relu_bwd");
  return TSCD_OK;
}

static long linear_bwd_pack_bytes(int in_f, int outp) {
  const int bn = tscd_conv_tile_n(in_f);
  const long rows_pad = (long)cdiv(in_f, bn) * bn, kpad = (long)cdiv(outp, 32) * 32;
  return (rows_pad * kpad * 2 + 255) / 256 * 256;
}
// outp = out_f rounded up to 8 (dY arrives padded with zero columns)
extern "C" long tscd_linear_bwd_workspace_bytes(int n, int in_f, int outp) {
  return linear_bwd_pack_bytes(in_f, outp) + tscd_conv_wgrad_workspace_bytes(n, outp, in_f, 1) + tscd_bias_relu_bwd_workspace_bytes(n, outp);
}
// x [n][in_f] (row stride ldx), dY [n][outp] fp16 with row stride lddy (outp % 8 == 0; columns >= out_f zero), w [out_f][in_f]
// fp32 or fp16.  Outputs (each may be NULL): dx [n][in_f] fp16 contiguous, dw [outp][in_f] fp32, db [outp] fp32.
extern "C" int tscd_linear_bwd_packed_f16(const void* x, int ldx, const void* dy, int lddy, const void* w, int w_is_f32,
                                          const void* w_dgrad_packed, int n, int in_f, int out_f, int outp, void* dx, float* dw, float* db,
                                          void* workspace, long workspace_bytes, void* stream);
extern "C" int tscd_linear_bwd_f16(const void* x, int ldx, const void* dy, int lddy, const void* w, int w_is_f32, int n, int in_f,
                                   int out_f, int outp, void* dx, float* dw, float* db, void* workspace, long workspace_bytes,
                                   void* stream) {
  return tscd_linear_bwd_packed_f16(x, ldx, dy, lddy, w, w_is_f32, nullptr, n, in_f, out_f, outp, dx, dw, db, workspace, workspace_bytes,
                                    stream);
}
// w_dgrad_packed (optional): W already in data-gradient order (tscd_pack_conv_weight mode 1 of the [out_f][in_f][1][1] view, the copy
// the caller caches per parameter version) -- the call then launches no pack kernel.
extern "C" int tscd_linear_bwd_packed_f16(const void* x, int ldx, const void* dy, int lddy, const void* w, int w_is_f32,
                                          const void* w_dgrad_packed, int n, int in_f, int out_f, int outp, void* dx, float* dw, float* db,
                                          void* workspace, long workspace_bytes, void* stream) {
  TSCD_CHECK_ARG(x && dy && w && workspace && n > 0, "linear_bwd: bad args");
  TSCD_CHECK_ARG(outp % 8 == 0 && outp >= out_f && in_f % 8 == 0 && lddy % 8 == 0 && lddy >= outp,
                 "linear_bwd: in_f / padded out_f / lddy must be multiples of 8");
  TSCD_CHECK_ARG(workspace_bytes >= tscd_linear_bwd_workspace_bytes(n, in_f, outp), "linear_bwd: workspace too small");
  const long pack_bytes = linear_bwd_pack_bytes(in_f, outp);
  int st;
  if (dx) {
    const int bn = tscd_conv_tile_n(in_f);
    const int rows_pad = cdiv(in_f, bn) * bn, kpad = cdiv(outp, 32) * 32;
    // data-gradient packing of a 1x1 "conv": rows = in_f, k = out channel (CinPad carries the padded out count)
    if (!w_dgrad_packed &&
        (st = tscd_pack_conv_weight(w, w_is_f32, workspace, out_f, in_f, outp, 1, 1, rows_pad, kpad, 1, stream)) != TSCD_OK)
      return st;
    if ((st = tscd_conv2d_nhwc_f16(dy, w_dgrad_packed ? w_dgrad_packed : workspace, dx, nullptr, nullptr, nullptr, nullptr, 1, 1, n, outp, in_f, 1, 1, lddy, in_f, 0,
                                   stream)) != TSCD_OK)
      return st;
  }
  if (dw) {
    char* ws2 = (char*)workspace + pack_bytes;
    if ((st = tscd_conv_wgrad_nhwc_f16(x, dy, dw, 1, 1, n, in_f, outp, 1, ldx, lddy, ws2, workspace_bytes - pack_bytes, stream)) != TSCD_OK)
      return st;
  }
  if (db) {                                                                                             // db: outp floats
    const long off = pack_bytes + tscd_conv_wgrad_workspace_bytes(n, outp, in_f, 1);
    if ((st = tscd_bias_relu_bwd_f16(dy, lddy, nullptr, 0, nullptr, 0, n, outp, db, (char*)workspace + off, workspace_bytes - off, stream)) != TSCD_OK)
      return st;
  }
  return TSCD_OK;
}
