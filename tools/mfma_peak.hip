// Sustained fp32 MFMA rate of the chip (no memory traffic): the practical ceiling that the
// 157.3 TFLOP/s paper number (256 CUs x 256 flop/clk x 2.4 GHz) turns into under load.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void burn(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}

template <int NACC>
void run(int blocks, int iters, const char* name) {
  float* d;
  hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  burn<NACC><<<blocks, 256>>>(d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  burn<NACC><<<blocks, 256>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 4 * iters * 8.0 * NACC * 4096.0;
  printf("%s blocks=%d nacc=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks, NACC, ms, flop / (ms * 1e-3) / 1e12);
  hipFree(d);
}

int main() {
  run<2>(256, 20000, "1 wave/SIMD ");
  run<2>(512, 20000, "2 waves/SIMD");
  run<4>(512, 10000, "2 waves/SIMD");
  run<1>(512, 40000, "2 waves/SIMD");
  run<1>(256, 40000, "1 wave/SIMD ");
  run<2>(512, 200000, "2 waves/SIMD long");
  return 0;
}
