// Shared device/host helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wavefront = 64 lanes everywhere in this tree; nothing here is written for
// 32-wide warps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

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

// Raise a kernel's dynamic-LDS limit once per DEVICE (the attribute belongs to the device's copy of
// the code object); `done` is a per-call-site bit mask of the devices already configured.  Safe to
// race: setting the attribute twice is idempotent.
static inline hipError_t ensure_dyn_lds(const void* kern, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

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
