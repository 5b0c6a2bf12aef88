// Which ingredient of the conv3x3 inner loop costs MFMA issue rate?  One wave per SIMD.
//  bit0: A operands come from LDS (2 ds_read_b128 per 8 MFMAs, one group ahead)
//  bit1: B operands come from global memory (4 x 16 B per lane per 32 MFMAs, one step ahead)
//  bit2: one ds_write_b128 per 32 MFMAs
//  bit3: a workgroup barrier every 288 MFMAs
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(256, 1) void burn(float* out, const float* w, int phases) {
  __shared__ __align__(16) float lds[2 * 288 * 36];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 2 * 288 * 36; i += 256) lds[i] = 1e-3f * (i & 15);
  __syncthreads();
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const float* wl = w + (long)lr * 576 + lh * 4;
  f32x4 cb[4], nb[4];
  for (int kc = 0; kc < 4; ++kc) cb[kc] = *reinterpret_cast<const f32x4*>(wl + kc * 8);
  const float* a0 = lds + lr * 36 + lh * 4;
  const float* a1 = lds + (lr + 34) * 36 + lh * 4;
  f32x4 fa[2][2];
  fa[0][0] = *reinterpret_cast<const f32x4*>(a0);
  fa[0][1] = *reinterpret_cast<const f32x4*>(a1);
  fa[1][0] = fa[0][0]; fa[1][1] = fa[0][1];
  for (int ph = 0; ph < phases; ++ph) {
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      if (V & 2) {
        const float* src = wl + ((u + 1) % 9) * 32 + (ph & 1) * 288;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) nb[kc] = *reinterpret_cast<const f32x4*>(src + kc * 8);
      }
      if (V & 4) *reinterpret_cast<f32x4*>(lds + 288 * 36 + (tid / 8 + 32 * u) * 36 + (tid % 8) * 4) = cb[0];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const int g = u * 4 + kc;
        if (V & 1) {
          const int off = ((g + 1) % 36) / 4 * 36 * 2 + ((g + 1) % 4) * 8;
          fa[(g + 1) & 1][0] = *reinterpret_cast<const f32x4*>(a0 + off);
          fa[(g + 1) & 1][1] = *reinterpret_cast<const f32x4*>(a1 + off);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][0][s], cb[kc][s], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][1][s], cb[kc][s], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (V & 2) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) cb[kc] = nb[kc];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (V & 8) __syncthreads();
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  if (s == 123.456f) out[0] = s;
}

template <int V>
void run(const char* name) {
  float *d, *w;
  (void)hipMalloc(&d, 4);
  (void)hipMalloc(&w, 64 * 576 * 4 * 2);
  (void)hipMemset(w, 0, 64 * 576 * 4 * 2);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int phases = 2000;
  burn<V><<<256, 256>>>(d, w, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  burn<V><<<256, 256>>>(d, w, phases);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flop = 256.0 * 4 * phases * 288.0 * 4096.0;
  printf("V=%2d %-44s %.3f ms  %.1f TFLOP/s\n", V, name, ms, flop / (ms * 1e-3) / 1e12);
  (void)hipFree(d);
  (void)hipFree(w);
}

int main() {
  run<0>("registers only");
  run<1>("A from LDS");
  run<2>("B from global");
  run<3>("A from LDS + B from global");
  run<4>("ds_write per step");
  run<8>("barrier per phase");
  run<7>("A LDS + B global + ds_write");
  run<15>("everything");
  return 0;
}
