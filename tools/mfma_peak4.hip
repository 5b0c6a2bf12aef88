// fp32 MFMA rate and core clock vs operand DATA: all-zero operands toggle almost nothing, random
// operands draw full power and the chip lowers its clock.  The random-data figure is the
// practical ceiling for real tensors.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void burn(float* out, const float* src, int iters) {
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = src[(threadIdx.x * 8 + i) % 4096];
    b[i] = src[(threadIdx.x * 8 + i + 2048 + blockIdx.x) % 4096];
  }
  const long long mt0 = __builtin_amdgcn_s_memtime(), rt0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + 3) & 7], acc1, 0, 0, 0);
    }
  }
  const long long mt1 = __builtin_amdgcn_s_memtime(), rt1 = wall_clock64();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  if (s == 123.456f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[2] = (float)((double)(mt1 - mt0) / (double)(rt1 - rt0) * 0.1);
}

void run(const char* name, int blocks, float scale) {
  float *d, *src;
  (void)hipMalloc(&d, 16);
  (void)hipMalloc(&src, 4096 * 4);
  float h[4096];
  unsigned x = 12345;
  for (int i = 0; i < 4096; ++i) {
    x = x * 1664525u + 1013904223u;
    h[i] = scale * ((float)(x >> 8) / 8388608.0f - 1.0f);
  }
  (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 40000;
  burn<<<blocks, 256>>>(d, src, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  burn<<<blocks, 256>>>(d, src, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms, res[4];
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(res, d, 16, hipMemcpyDeviceToHost);
  const double flop = (double)blocks * 4 * iters * 16.0 * 4096.0;
  printf("%-34s blocks=%d: %.2f ms  %.1f TFLOP/s  clock %.3f GHz\n", name, blocks, ms, flop / (ms * 1e-3) / 1e12, res[2]);
  (void)hipFree(d);
  (void)hipFree(src);
}

int main() {
  run("zero operands", 256, 0.f);
  run("random operands", 256, 1.f);
  run("random operands (small magnitude)", 256, 1e-3f);
  run("random operands, 2 waves/SIMD", 512, 1.f);
  run("zero operands, 2 waves/SIMD", 512, 0.f);
  return 0;
}
