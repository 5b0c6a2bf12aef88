// Internal helpers shared by the HIP translation units of liblgm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/lgm_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// thread-local last error string (lgm_last_error)
void lgm_set_error(const char* fmt, ...);

#define LGM_REQUIRE(cond, ...)              \
  do {                                      \
    if (!(cond)) {                          \
      lgm_set_error(__VA_ARGS__);           \
      return LGM_ERR_INVALID;               \
    }                                       \
  } while (0)

// returns hipError_t (>0) through the C-ABI when a launch fails
#define LGM_LAUNCH_CHECK()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) {                                 \
      lgm_set_error("%s: %s", __func__, hipGetErrorString(e__)); \
      return (int)e__;                                       \
    }                                                        \
  } while (0)

static inline int lgm_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

static inline bool lgm_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---- device helpers -------------------------------------------------------------
__device__ __forceinline__ float lgm_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float lgm_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); `sh` needs 16 floats.
// Deterministic: fixed shuffle tree + fixed-order combine.
__device__ __forceinline__ float lgm_block_sum(float v, float* sh) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = lgm_wave_sum(v);
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}
__device__ __forceinline__ float lgm_block_max(float v, float* sh) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = lgm_wave_max(v);
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, sh[i]);
  return r;
}
