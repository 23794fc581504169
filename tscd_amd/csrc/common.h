// Shared helpers for the TSCD gfx950 kernels (C-ABI in include/tscd_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TSCD_OK 0
#define TSCD_ERR_ARG 1
#define TSCD_ERR_HIP 2

void tscd_set_error(const char* fmt, ...);

#define TSCD_CHECK_ARG(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      tscd_set_error(__VA_ARGS__);           \
      return TSCD_ERR_ARG;                   \
    }                                        \
  } while (0)

#define TSCD_CHECK_LAUNCH(name)                                                   \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      tscd_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return TSCD_ERR_HIP;                                                        \
    }                                                                             \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// SiLU gate of the BatchNorm+SiLU passes: v_exp_f32 + v_rcp_f32 (1 ulp each, far below the fp16 rounding of the result)
// instead of the IEEE division sequence -- those passes are VALU-bound otherwise.
__device__ __forceinline__ float sigmoid_fast_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
