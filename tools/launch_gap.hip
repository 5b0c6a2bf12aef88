// Inter-kernel gap on one stream: time from the last wave of kernel k finishing to the first wave
// of kernel k+1 starting (100 MHz wall clock), for different footprints.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void work(long long* stamps, int slot, float* out, long n_out, int spin) {
  extern __shared__ float lds[];
  const long long t0 = wall_clock64();
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (n_out > 0) {
    const long per = n_out / gridDim.x;
    float* o = out + (long)blockIdx.x * per;
    for (long i = threadIdx.x * 4; i < per; i += 256 * 4) *reinterpret_cast<f32x4*>(o + i) = f32x4{v, v, v, v};
  }
  if (v == 1.2345f) lds[threadIdx.x] = v;
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) {
    stamps[((long)slot * gridDim.x + blockIdx.x) * 2 + 0] = t0;
    stamps[((long)slot * gridDim.x + blockIdx.x) * 2 + 1] = t1;
  }
}

void run(const char* name, int blocks, size_t smem, long n_out, int spin) {
  const int K = 20;
  long long* d;
  float* out = nullptr;
  (void)hipMalloc(&d, (size_t)K * blocks * 2 * sizeof(long long));
  if (n_out) (void)hipMalloc(&out, n_out * 4);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(work), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    for (int k = 0; k < K; ++k) hipLaunchKernelGGL(work, dim3(blocks), dim3(256), smem, 0, d, k, out, n_out, spin);
    (void)hipDeviceSynchronize();
  }
  std::vector<long long> h((size_t)K * blocks * 2);
  (void)hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double gap = 0, dur = 0, spread = 0;
  for (int k = 0; k < K; ++k) {
    long long s0 = h[(size_t)k * blocks * 2], s1 = s0, e1 = h[(size_t)k * blocks * 2 + 1];
    for (int b = 0; b < blocks; ++b) {
      s0 = std::min(s0, h[((size_t)k * blocks + b) * 2]);
      s1 = std::max(s1, h[((size_t)k * blocks + b) * 2]);
      e1 = std::max(e1, h[((size_t)k * blocks + b) * 2 + 1]);
    }
    dur += (e1 - s0) * 0.01;
    spread += (s1 - s0) * 0.01;
    if (k + 1 < K) {
      long long n0 = h[(size_t)(k + 1) * blocks * 2];
      for (int b = 0; b < blocks; ++b) n0 = std::min(n0, h[((size_t)(k + 1) * blocks + b) * 2]);
      gap += (n0 - e1) * 0.01;
    }
  }
  printf("%-44s blocks=%4d lds=%3zuK out=%3ldMB: kernel %.1f us, start spread %.1f us, gap to next %.1f us\n", name, blocks,
         smem / 1024, n_out * 4 / 1000000, dur / K, spread / K, gap / (K - 1));
  (void)hipFree(d);
  if (out) (void)hipFree(out);
}

int main() {
  run("tiny", 8, 0, 0, 2000);
  run("256 blocks", 256, 0, 0, 2000);
  run("256 blocks, 100 KB LDS", 256, 100 * 1024, 0, 2000);
  run("512 blocks, 59 KB LDS", 512, 59 * 1024, 0, 2000);
  run("256 blocks, 100 KB LDS, 33 MB out", 256, 100 * 1024, 8 * 1024 * 1024, 2000);
  run("256 blocks, 100 KB LDS, 134 MB out", 256, 100 * 1024, 32 * 1024 * 1024, 2000);
  run("4096 blocks", 4096, 0, 0, 500);
  return 0;
}
