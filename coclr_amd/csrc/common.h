// Shared device/host helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wavefront = 64 lanes everywhere in this tree; nothing here is written for
// 32-wide warps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define COCLR_WAVE 64

// Every C-ABI entry point returns 0 on success or a hipError_t cast to int.
#define COCLR_RETURN_IF(expr)                 \
  do {                                        \
    hipError_t _e = (expr);                   \
    if (_e != hipSuccess) return (int)_e;     \
  } while (0)

#define COCLR_LAUNCH_CHECK() \
  do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess) return (int)_e; } while (0)

// invalid-argument code shared by all entry points (== hipErrorInvalidValue)
#define COCLR_EINVAL 1

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4  = __attribute__((ext_vector_type(4))) float;

// Sum over the 32 lanes of one wave half (lanes 0-31 or 32-63); xor offsets
// < 32 never cross halves.
__device__ __forceinline__ float half_wave_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
  v += __shfl_xor(v, 32);
  return half_wave_sum(v);
}

__device__ __forceinline__ double wave_sum_d(double v) {
  v += __shfl_xor(v, 32);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

// Block-wide double sum for 256-thread blocks; `red` holds >= 4 doubles.
__device__ __forceinline__ double block256_sum_d(double v, double* red) {
  v = wave_sum_d(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
