// Shared definitions of the MCA forward (mca.hip) and backward (mca_bwd.hip) kernels.
#pragma once
#include "common.h"

#define MCA_HEADS 4

struct McaTile { int row0, nrows, loc_start, loc_end; };

// The per-head L2-normalised q / k / v rows (zero padded to DP per head): written by the forward's prep launch, read in place by the
// backward (tscd_mca_backward_fws).  ONE definition of their layout for both files: six regions, each rounded up to 256 bytes.
struct McaRows { f16 *qn_c, *qn_r, *kn_c, *kn_r, *vn_c, *vn_r; };
static inline long mca_rows_pad(long bytes) { return (bytes + 255) / 256 * 256; }
static inline long mca_rows_bytes(int N, int N_L, int WP) {
  const long Npad = (N + 31) / 32 * 32 + 32;
  return 2 * mca_rows_pad(2L * N_L * WP) + 4 * mca_rows_pad(2L * Npad * WP);
}
static inline McaRows mca_rows_carve(void* base, int N, int N_L, int WP) {
  const long Npad = (N + 31) / 32 * 32 + 32;
  char* w = (char*)base;
  auto take = [&](long bytes) { char* r = w; w += mca_rows_pad(bytes); return (f16*)r; };
  McaRows r;
  r.qn_c = take(2L * N_L * WP); r.qn_r = take(2L * N_L * WP);
  r.kn_c = take(2L * Npad * WP); r.kn_r = take(2L * Npad * WP);
  r.vn_c = take(2L * Npad * WP); r.vn_r = take(2L * Npad * WP);
  return r;
}

// Attention logits.  F16L (the reference-faithful switch): the reference forms them as fp16 tensors -- the matmul output, x scale and
// x class score are each rounded to fp16 (post_trans.py:655-657 under autocast / .half()) -- before the (fp32) softmax; the default
// keeps them in fp32 (bounded in tests/test_deviation_bounds.py).
template <bool F16L> __device__ __forceinline__ float mca_logit_cls(float s, float kscore, float scale) {
  if constexpr (F16L) return (float)(f16)((float)(f16)((float)(f16)s * scale) * kscore);
  else return scale * kscore * s;
}
template <bool F16L> __device__ __forceinline__ float mca_logit_reg(float s, float scale) {
  if constexpr (F16L) return (float)(f16)((float)(f16)s * scale);
  else return scale * s;
}

// key blocks of a q-tile: first the 32-key blocks covering its own frame's rows, then those covering the global rows
__device__ __forceinline__ void block_range(const McaTile& t, int N_L, int N, int j, int& key0, int& lo, int& hi,
                                            int n_local) {
  if (j < n_local) {
    key0 = ((t.loc_start >> 5) + j) << 5;
    lo = t.loc_start; hi = t.loc_end;
  } else {
    key0 = ((N_L >> 5) + (j - n_local)) << 5;
    lo = N_L; hi = N;
  }
}

// XCD-aware workgroup order.  Workgroup ids go round-robin over the 8 XCDs (each with its own 4 MB L2), so ids with the same
// id % 8 share an L2: they take the affinity classes c = xcd, xcd + 8, ... (u = the units of a class, consecutive ids of the
// XCD), and what a class shares -- the key fragments of a key split, the dM / A blocks of a (tile, key block) pair that all four
// heads read -- is fetched from memory once per XCD instead of once per workgroup.  Grid = 8 * ceil(C / 8) * U; false = padding.
__device__ __forceinline__ bool xcd_remap(int id, int C, int U, int& c, int& u) {
  const int xcd = id & 7, k = id >> 3;
  c = xcd + 8 * (k / U);
  u = k % U;
  return c < C;
}
static inline unsigned xcd_grid(int C, int U) { return 8u * ((C + 7) / 8) * U; }

