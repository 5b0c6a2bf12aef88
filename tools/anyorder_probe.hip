// Does a dispatch without the AQL barrier bit (hipExtAnyOrderLaunch) start before its predecessor on the same
// stream has drained - eagerly, and inside a captured graph?  Two independent whole-chip kernels A and B
// (one workgroup per CU: 512 threads, 120 KB of LDS, ~25 us of FMAs, each writes `out_mb` MB) alternate on one
// stream; a checker with the barrier bit set follows every pair and must see both results.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/anyorder_probe tools/anyorder_probe.hip && /tmp/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void work(float* out, long n_out, int spin, float tag) {
  extern __shared__ float lds[];
  float v = threadIdx.x * 1e-9f;
  for (int i = 0; i < spin; ++i) v = v * 0.999f + 1e-7f;
  if (v == 1.2345f) lds[threadIdx.x] = v;
  const long per = n_out / gridDim.x;
  float* o = out + (long)blockIdx.x * per;
  const float w = tag + (v > 1e30f ? 1.f : 0.f);
  for (long i = threadIdx.x * 4; i < per; i += 512 * 4) *reinterpret_cast<f32x4*>(o + i) = f32x4{w, w, w, w};
}

__global__ void check(const float* a, const float* b, long n, float ta, float tb, int* bad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (a[i] != ta || b[i] != tb)) atomicAdd(bad, 1);
}

static void launch(hipStream_t s, float* out, long n, int spin, float tag, int flags, size_t smem) {
  void* args[] = {&out, &n, &spin, &tag};
  CK(hipExtLaunchKernel(reinterpret_cast<const void*>(work), dim3(256), dim3(512), args, smem, s, nullptr, nullptr, flags));
}

int main(int argc, char** argv) {
  const int spin = argc > 1 ? atoi(argv[1]) : 12000;
  const long n = (argc > 2 ? atol(argv[2]) : 32) * 1048576 / 4;
  const size_t smem = 120 * 1024;
  const int pairs = 50, reps = 10;
  float *a, *b;
  int* bad;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(work), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode) {            // 0: B ordered, 1: B any-order
    for (int graph = 0; graph < 2; ++graph) {
      auto body = [&](int rep) {
        for (int p = 0; p < pairs; ++p) {
          const float ta = rep * 1000 + p, tb = ta + 0.5f;
          launch(s, a, n, spin, ta, 0, smem);
          launch(s, b, n, spin, tb, mode ? hipExtAnyOrderLaunch : 0, smem);
          hipLaunchKernelGGL(check, dim3((n + 255) / 256), dim3(256), 0, s, a, b, n, ta, tb, bad);
        }
      };
      float ms = 0;
      if (!graph) {
        body(0); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) body(r);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
      } else {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        body(7);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
      }
      int hbad = -1; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
      printf("B %-9s %-6s: %.2f us per (A, B, check) triple, mismatches %d\n", mode ? "any-order" : "ordered", graph ? "graph" : "eager",
             ms * 1000 / (pairs * reps), hbad);
      CK(hipMemset(bad, 0, 4));
    }
  }
  // the single-kernel time, for scale
  CK(hipEventRecord(e0, s));
  for (int r = 0; r < 100; ++r) launch(s, a, n, spin, 1.f, 0, smem);
  CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("A alone, back to back: %.2f us per launch\n", ms * 10);
  return 0;
}
