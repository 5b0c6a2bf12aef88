// Diagnostic (not part of the library): occupy `k` workgroup slots for `cycles` shader clocks, the way a long-running
// collective kernel (RCCL: one 256-thread workgroup per channel) sits beside the training step on N > 1 GPUs.
//   hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libcu_hog.so tools/cu_hog.hip
#include <hip/hip_runtime.h>

// REGS values stay live for the kernel's whole life (they are all summed at the end); WORK of them are updated per loop trip,
// then the wave sleeps: WORK = REGS is the busy guest of rounds 4 / 5 (40 - 65 % of its SIMD's VALU issue), WORK = 8 a quiet
// one that mostly waits - what a collective's workgroup does between its loads, stores and flag polls
template <int REGS, int WORK = REGS>
__global__ __launch_bounds__(256) void hog_kernel(long long cycles, float* sink) {
  float v[REGS];
#pragma unroll
  for (int i = 0; i < REGS; ++i) v[i] = (float)(threadIdx.x + i);
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) {
#pragma unroll
    for (int i = 0; i < WORK; ++i) v[i] = v[i] * 1.0001f + 0.5f;
#pragma unroll
    for (int i = WORK; i < REGS; ++i) asm volatile("" : "+v"(v[i]));     // no instruction: the value must sit in a VGPR here
    __builtin_amdgcn_s_sleep(8);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < REGS; ++i) s += v[i];
  if (s == 123.456f) sink[0] = s;
}

extern "C" int cu_hog(int k, int threads, int regs, long long cycles, float* sink, void* stream) {
  // 240 live values: the register class of RCCL's own kernel (rcclGenericKernel on gfx950: 261 - 280 VGPRs) -
  // a SIMD that hosts one of its waves has fewer than 256 registers left
  if (regs >= 240) hipLaunchKernelGGL(hog_kernel<240>, dim3(k), dim3(threads), 0, (hipStream_t)stream, cycles, sink);
  else if (regs >= 96) hipLaunchKernelGGL(hog_kernel<96>, dim3(k), dim3(threads), 0, (hipStream_t)stream, cycles, sink);
  else hipLaunchKernelGGL(hog_kernel<32>, dim3(k), dim3(threads), 0, (hipStream_t)stream, cycles, sink);
  return (int)hipGetLastError();
}

// the quiet guests: same register classes, eight updates per loop trip
extern "C" int cu_hog_quiet(int k, int threads, int regs, long long cycles, float* sink, void* stream) {
  // 280: more than half of a SIMD's 512 registers, like rcclGenericKernel (261 - 280): no 256-register wave fits beside it
  if (regs >= 280) hipLaunchKernelGGL((hog_kernel<280, 8>), dim3(k), dim3(threads), 0, (hipStream_t)stream, cycles, sink);
  else if (regs >= 240) hipLaunchKernelGGL((hog_kernel<240, 8>), dim3(k), dim3(threads), 0, (hipStream_t)stream, cycles, sink);
  else hipLaunchKernelGGL((hog_kernel<96, 8>), dim3(k), dim3(threads), 0, (hipStream_t)stream, cycles, sink);
  return (int)hipGetLastError();
}
