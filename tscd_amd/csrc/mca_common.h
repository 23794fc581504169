// Shared definitions of the MCA forward (mca.hip) and backward (mca_bwd.hip) kernels.
#pragma once
#include "common.h"

#define MCA_HEADS 4

struct McaTile { int row0, nrows, loc_start, loc_end; };

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
