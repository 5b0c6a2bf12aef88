// Does the accumulator register class (ArchVGPR vs AccVGPR) change the fp32 MFMA issue rate?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int AGPR>
__global__ __launch_bounds__(256) void burn(float* out, int iters) {
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (AGPR) {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
      } else {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
      }
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  if (s == 123.456f) out[0] = s;
}

template <int AGPR>
void run(int blocks, int iters, const char* name) {
  float* d;
  (void)hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  burn<AGPR><<<blocks, 256>>>(d, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  burn<AGPR><<<blocks, 256>>>(d, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 4 * iters * 16.0 * 4096.0;
  printf("%s blocks=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flop / (ms * 1e-3) / 1e12);
  (void)hipFree(d);
}

int main() {
  run<1>(512, 20000, "acc in AGPR, 2 waves/SIMD");
  run<0>(512, 20000, "acc in VGPR, 2 waves/SIMD");
  run<1>(256, 20000, "acc in AGPR, 1 wave/SIMD ");
  run<0>(256, 20000, "acc in VGPR, 1 wave/SIMD ");
  return 0;
}
