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
// name (as rocprofv3 prints it, without the argument list) of the primary kernel the calling thread's last
// convolution-family entry point launched: lets bench.py attribute its HIP-event timings to profiler rows
void lgm_note_kernel(const char* name);
// Workgroup slots the launch planners count on: the chip's 256 CUs minus the margin left to the workgroups of a collective
// that is resident beside the step (LGM_CU_MARGIN / lgm_set_cu_margin, 16 under WORLD_SIZE > 1; elementwise.hip).  One-workgroup-per-CU kernels sized
// for all 256 CUs need a second round as soon as ONE CU is taken (tools/cu_hog_step.py).
int lgm_cu_budget();
// Every name handed to lgm_note_kernel goes through LGM_KNAME: the literal's address is also placed in the ELF section
// "lgm_knames" at link time (no code runs), so lgm_kernel_name(i) can list every name the library may ever note and
// tests/test_cabi.py checks each against the kernel symbols of liblgm_hip.so - a template argument added to a kernel
// can no longer silently detach bench.py's per-kernel attribution from the profiler's rows.
#define LGM_KNAME(lit)                                                                                          \
  ([]() -> const char* {                                                                                        \
    static const char* const lgm_kn_ __attribute__((section("lgm_knames"), used)) = lit;                        \
    return lgm_kn_;                                                                                             \
  }())

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
__device__ __forceinline__ bool lgm_aligned16_dev(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

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

// ---- wide epilogue stores for 32x32 MFMA accumulator tiles ------------------------------------
// The C/D layout puts one output COLUMN on a lane (16 rows in registers), so a direct epilogue is 16
// dword stores per lane, and 4-byte stores are issue-bound (~2 TB/s chip-wide measured).  Routing the
// tile through a wave-private LDS scratch (32 rows x LGM_TS_LD floats) lets every lane store 16
// contiguous bytes: 4 store instructions per tile, each covering 8 full 128-byte rows.
constexpr int LGM_TS_LD = 36;
constexpr int LGM_TS_FLOATS = 32 * LGM_TS_LD;

__device__ __forceinline__ void lgm_wave_lds_sync() {
  // Orders this wave's LDS writes before its later LDS reads of other lanes' data.  DS instructions
  // of one wave execute in order, so only the COMPILER must be kept from reordering: wavefront-scope
  // fences emit no s_waitcnt.  (A workgroup-scope release would also drain every outstanding global
  // load and store of the wave - measured 2 us per call inside the conv epilogues.)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// acc -> Ts (transposed staging).  lane = threadIdx & 63.
__device__ __forceinline__ void lgm_tile_to_lds(const f32x16& acc, float* Ts, int lane) {
  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) Ts[((r & 3) + 8 * (r >> 2) + 4 * lh) * LGM_TS_LD + lr] = acc[r];
}
// row j-th pass of the read-back: lane reads 4 consecutive columns of tile row (lane>>3) + 8*j
__device__ __forceinline__ f32x4 lgm_tile_row4(const float* Ts, int lane, int j) {
  return *reinterpret_cast<const f32x4*>(Ts + ((lane >> 3) + 8 * j) * LGM_TS_LD + (lane & 7) * 4);
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
