// 1x1 convolution / linear layers with a short reduction (K <= 608):  Y[M,N] = X[M,K] W[N,K]^T (+bias,+res)
//
// With K this small a tile-per-block GEMM is all prologue and epilogue.  Here a workgroup keeps a
// BM-row tile of X resident in LDS (loaded once, full K) and loops over ALL 64-column output tiles:
// per tile K/32 chunks of MFMAs whose weight fragments every lane streams straight from global
// memory (the whole weight matrix is <= a few hundred KB and L1/L2 resident), one chunk ahead,
// across tile boundaries — no barrier after the initial one.  X is read from HBM exactly once.
// Used for to_qkv / to_out / res_conv / Downsample convs (ddpm.py:103,187,213,215,252,253) forward,
// and for their input gradients through the transposed weight copy.
#include "lgm_common.h"

namespace {

struct RArgs {
  const float* x;     // [M, K] rows, pitch x_pitch
  const float* w;     // [N][K]
  const float* bias;  // [N] or null
  const float* res;   // [M, N] or null
  float* out;         // [M, N]
  long x_pitch, res_pitch, out_pitch;
  int M, N, K;
};

template <int TM>   // BM = 64 * TM rows per workgroup; waves 2 (m) x 2 (n), wave tile (32*TM) x 32
__global__ __launch_bounds__(256) void gemm_rows_kernel(const RArgs p) {
  constexpr int BM = 64 * TM;
  extern __shared__ __align__(16) float Xs[];
  const int LDX = p.K + 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * BM;
  const int k4 = p.K / 4;

  // ---- stage the X tile (8 loads in flight per thread) ----
  const int total = BM * k4;
  for (int base = tid; base < total; base += 256 * 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * 256;
      const int row = idx / k4, c4 = idx - row * k4;
      const bool ok = idx < total && (m0 + row) < p.M;
      v[u] = ok ? *reinterpret_cast<const f32x4*>(p.x + (long)(m0 + row) * p.x_pitch + c4 * 4)
                : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * 256;
      if (idx < total) {
        const int row = idx / k4, c4 = idx - row * k4;
        *reinterpret_cast<f32x4*>(Xs + row * LDX + c4 * 4) = v[u];
      }
    }
  }

  const int KQ = p.K / 32;
  const int NT = p.N / 64;
  const int nit = NT * KQ;
  const float* wbase = p.w + (long)(wn * 32 + lr) * p.K + lh * 4;
  auto load_b = [&](int it, f32x4 (&fb)[4]) {
    const int nt = it / KQ, q = it - nt * KQ;
    const float* src = wbase + (long)nt * 64 * p.K + q * 32;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) fb[kc] = *reinterpret_cast<const f32x4*>(src + kc * 8);
  };
  f32x4 cb[4], nb[4];
  load_b(0, cb);
  __syncthreads();

  const float* a_base = Xs + (wm * 32 * TM + lr) * LDX + lh * 4;
  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  for (int it = 0; it < nit; ++it) {
    const int nt = it / KQ, q = it - nt * KQ;
    if (it + 1 < nit) load_b(it + 1, nb);
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      f32x4 fa[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * LDX + q * 32 + kc * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], cb[kc][s], acc[i], 0, 0, 0);
    }
    if (q == KQ - 1) {   // tile finished: epilogue, then reset the accumulators
      const int n = nt * 64 + wn * 32 + lr;
      const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float rv[16];
        if (p.res) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            rv[r] = m < p.M ? p.res[(long)m * p.res_pitch + n] : 0.f;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < p.M) p.out[(long)m * p.out_pitch + n] = acc[i][r] + bv + rv[r];
          acc[i][r] = 0.f;
        }
      }
    }
    if (it + 1 < nit) {
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) cb[kc] = nb[kc];
    }
  }
}

}  // namespace

bool lgm_gemm_rows_supported(long M, int N, int K) {
  if (K % 32 != 0 || N % 64 != 0 || K > 608) return false;
  const int bm = (long)(K + 4) * 128 * 4 <= 140 * 1024 ? 128 : 64;
  return M / bm >= 96;   // enough row tiles to fill the chip; tiny M stays on the generic path
}

int lgm_gemm_rows_launch(const float* x, long x_pitch, const float* w, const float* bias, const float* res,
                         long res_pitch, float* out, long out_pitch, long M, int N, int K, hipStream_t s) {
  RArgs p{};
  p.x = x; p.w = w; p.bias = bias; p.res = res; p.out = out;
  p.x_pitch = x_pitch; p.res_pitch = res_pitch; p.out_pitch = out_pitch;
  p.M = (int)M; p.N = N; p.K = K;
  const bool big = (long)(K + 4) * 128 * 4 <= 140 * 1024;
  const int bm = big ? 128 : 64;
  const size_t smem = (size_t)bm * (K + 4) * sizeof(float);
  const unsigned nblocks = (unsigned)lgm_cdiv(M, bm);
  if (big) {
    static size_t attr = 0;
    if (smem > attr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rows_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr = smem;
    }
    hipLaunchKernelGGL(gemm_rows_kernel<2>, dim3(nblocks), dim3(256), smem, s, p);
  } else {
    static size_t attr = 0;
    if (smem > attr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rows_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr = smem;
    }
    hipLaunchKernelGGL(gemm_rows_kernel<1>, dim3(nblocks), dim3(256), smem, s, p);
  }
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
