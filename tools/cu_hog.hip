// Diagnostic (not part of the library): occupy `k` workgroup slots for `cycles` shader clocks, the way a long-running
// collective kernel (RCCL: one 256-thread workgroup per channel) sits beside the training step on N > 1 GPUs.
//   hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libcu_hog.so tools/cu_hog.hip
#include <hip/hip_runtime.h>

template <int REGS>
__global__ __launch_bounds__(256) void hog_kernel(long long cycles, float* sink) {
  float v[REGS];
#pragma unroll
  for (int i = 0; i < REGS; ++i) v[i] = (float)(threadIdx.x + i);
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) {
#pragma unroll
    for (int i = 0; i < REGS; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    __builtin_amdgcn_s_sleep(8);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < REGS; ++i) s += v[i];
  if (s == 123.456f) sink[0] = s;
}

extern "C" int cu_hog(int k, int threads, int regs, long long cycles, float* sink, void* stream) {
  if (regs >= 96) hipLaunchKernelGGL(hog_kernel<96>, dim3(k), dim3(threads), 0, (hipStream_t)stream, cycles, sink);
  else hipLaunchKernelGGL(hog_kernel<32>, dim3(k), dim3(threads), 0, (hipStream_t)stream, cycles, sink);
  return (int)hipGetLastError();
}
