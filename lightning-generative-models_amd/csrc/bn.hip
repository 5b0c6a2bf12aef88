// Train-mode BatchNorm2d building blocks for the DCGAN generator / critic (dcgan.py:86-90,
// 158-161) including what the WGAN-GP double backward (wgan.py:117-156) needs.
//
// Everything is expressed with three primitives over [M = B*H*W, C] NHWC matrices:
//   lgm_bn_stats     per-channel batch mean / rstd (block-local two-sweep sums combined exactly, deterministic) + running-stat update
//   lgm_bn_reduce3   per-channel  S1 = sum v1,  S2 = sum v1*xhat,  S3 = sum v1*v2
//   lgm_bn_affine3   out = A1[c]*v1 + A2[c]*v2 + A3[c]*xhat + A4[c]      (xhat = (a-mean)*rstd)
// plus the per-channel coefficient math, which lgm_bn_reduce3_coef runs inside the second reduction stage
// (lgm_bn_coef is the stand-alone form), and lgm_bn_affine3x2 = two affine3 results from one pass.  With them
//   forward        y = gamma*xhat + beta                       = affine3(A3 = gamma, A4 = beta)
//   backward       ga = T(gn) = c (gn - mean(gn) - xhat mean(gn xhat)),  c = gamma*rstd
//   GP 2nd order   adjoints of T w.r.t. its input, gamma and the batch statistics
// are all single launches of the same kernels.
#include "lgm_common.h"

namespace {

static inline long bn_rows(long rows) {   // rows per stage-1 block, <= 128 blocks per channel tile
  long r = (rows + 127) / 128;
  if (r < 64) r = 64;
  return (r + 3) / 4 * 4;
}

// stage 1: block = 64 channels x 4 row lanes; writes partial[blockIdx.y][k][c], k < 3
__global__ __launch_bounds__(256) void bn_reduce_stage1(const float* __restrict__ v1, long v1_pitch,
                                                        const float* __restrict__ v2, long v2_pitch,
                                                        const float* __restrict__ a, long a_pitch,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        int mode, long rows, int C, long rpb,
                                                        float* __restrict__ partial) {
  // mode 0: (sum v1, sum v1*xhat, sum v1*v2)   mode 1: (sum a, 0, 0)   mode 2: (sum (a-mean)^2, 0, 0)
  __shared__ float sh[3][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long r0 = (long)blockIdx.y * rpb;
  const long r1 = r0 + rpb < rows ? r0 + rpb : rows;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    const float mu = (mode != 1 && mean) ? mean[c] : 0.f;
    const float rs = (mode == 0 && rstd) ? rstd[c] : 0.f;
    // four independent row streams per thread (rows r, r+4, r+8, r+12): four loads in flight, fixed order
    float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f}, t3[4] = {0.f, 0.f, 0.f, 0.f};
    for (long rb = r0 + rl; rb < r1; rb += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long r = rb + 4 * u;
        if (r >= r1) continue;
        if (mode == 0) {
          const float x = v1[r * v1_pitch + c];
          t1[u] += x;
          if (a) t2[u] += x * ((a[r * a_pitch + c] - mu) * rs);
          if (v2) t3[u] += x * v2[r * v2_pitch + c];
        } else if (mode == 1) {
          t1[u] += a[r * a_pitch + c];
        } else {
          const float d = a[r * a_pitch + c] - mu;
          t1[u] += d * d;
        }
      }
    }
    s1 = (t1[0] + t1[1]) + (t1[2] + t1[3]);
    s2 = (t2[0] + t2[1]) + (t2[2] + t2[3]);
    s3 = (t3[0] + t3[1]) + (t3[2] + t3[3]);
  }
  sh[0][rl][cl] = s1;
  sh[1][rl][cl] = s2;
  sh[2][rl][cl] = s3;
  __syncthreads();
  if (rl < 3 && c < C) {
    const float v = (sh[rl][0][cl] + sh[rl][1][cl]) + (sh[rl][2][cl] + sh[rl][3][cl]);
    partial[((long)blockIdx.y * 3 + rl) * C + c] = v;
  }
}

// batch statistics, stage 1: the block sums its rows, then sums the squared deviations from ITS OWN mean (second
// sweep over the same rows, L2-resident); partial[b] = (sum_b, M2_b, n_b).  Stage 2 combines the blocks with
// M2 = sum_b M2_b + n_b (mean_b - mean)^2 - the accuracy of the two-pass formula from one launch pair.
__global__ __launch_bounds__(256) void bn_stats_stage1(const float* __restrict__ a, long a_pitch, long rows, int C,
                                                       long rpb, float* __restrict__ partial) {
  __shared__ float sh[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long r0 = (long)blockIdx.y * rpb;
  const long r1 = r0 + rpb < rows ? r0 + rpb : rows;
  const bool live = c < C;
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  if (live)
    for (long rb = r0 + rl; rb < r1; rb += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long r = rb + 4 * u;
        if (r < r1) t[u] += a[r * a_pitch + c];
      }
    }
  sh[rl][cl] = (t[0] + t[1]) + (t[2] + t[3]);
  __syncthreads();
  const float sum = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
  const float n = (float)(r1 - r0);
  const float mu = sum / n;
  __syncthreads();
  float q[4] = {0.f, 0.f, 0.f, 0.f};
  if (live)
    for (long rb = r0 + rl; rb < r1; rb += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long r = rb + 4 * u;
        if (r < r1) {
          const float d = a[r * a_pitch + c] - mu;
          q[u] += d * d;
        }
      }
    }
  sh[rl][cl] = (q[0] + q[1]) + (q[2] + q[3]);
  __syncthreads();
  if (rl == 0 && live) {
    float* o = partial + (long)blockIdx.y * 3 * C + c;
    o[0] = sum;
    o[C] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
    o[2 * C] = n;
  }
}

// stage 2: block = 64 of the 3C sums x 4 split lanes; lane l adds splits l, l+4, ... (two at a time,
// unconditionally - rows past nsplit are clamped and weighted 0), fixed-order LDS combine
__global__ __launch_bounds__(256) void bn_reduce_stage2(const float* __restrict__ partial, int nsplit, int C,
                                                        float* __restrict__ out3) {
  __shared__ float sh[4][64];
  const int cl = threadIdx.x & 63, jl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + cl;   // i in [0, 3C)
  float s0 = 0.f, s1 = 0.f;
  if (i < 3 * C) {
    const int k = i / C, c = i - k * C;
    const float* base = partial + (long)k * C + c;
    const long stride = 3L * C;
    for (int j = jl; j < nsplit; j += 8) {
      const int j2 = j + 4;
      const float a = base[(long)j * stride];
      const float b = base[(long)(j2 < nsplit ? j2 : j) * stride];
      s0 += a;
      s1 += j2 < nsplit ? b : 0.f;
    }
  }
  sh[jl][cl] = s0 + s1;
  __syncthreads();
  if (jl == 0 && i < 3 * C) out3[i] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}

// stage 2 with the per-channel epilogue fused in (fixed summation order => deterministic):
//   epi 1 (batch statistics from bn_stats_stage1's blocks): mean, M2 (Chan's combination), rstd, running stats
//   epi 2 (coefficients): bit 0 -> bn_coef mode 0 into coef, bit 1 -> bn_coef mode 1 into coef + 4C
struct BnEpi {
  int epi, modes;
  long M;
  float eps, momentum, beta_acc0, beta_acc1;
  float *mean, *rstd, *running_mean, *running_var;
  const float *gamma, *rstd_in, *saved_m;
  float *coef, *ggamma0, *gbeta0, *m_out, *ggamma1, *sums_out;
};

__global__ __launch_bounds__(256) void bn_stage2_epi_kernel(const float* __restrict__ partial, int nsplit, int C,
                                                            const BnEpi e) {
  // block = 16 channels x 16 split lanes: lane jl adds splits jl, jl+16, ... of the three sums (three loads in
  // flight per step); the 16 lane results are combined by a fixed tree
  __shared__ float sh[3][16][17];
  __shared__ float shm[16];
  const int cl = threadIdx.x & 15, jl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const long stride = 3L * C;
  float acc[3] = {0.f, 0.f, 0.f};
  if (c < C) {
    for (int j = jl; j < nsplit; j += 16) {
      const float* base = partial + (long)j * stride + c;
      acc[0] += base[0];
      acc[1] += base[C];
      acc[2] += base[2 * C];
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) sh[k][cl][jl] = acc[k];
  __syncthreads();
  auto tree = [&](int k) {
    float t[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) t[j] = sh[k][cl][j];
#pragma unroll
    for (int w = 8; w > 0; w >>= 1)
#pragma unroll
      for (int j = 0; j < w; ++j) t[j] += t[j + w];
    return t[0];
  };
  const float invM = 1.f / (float)e.M;
  if (e.epi == 1) {   // (sum_b, M2_b, n_b) blocks -> mean, then the between-block term of M2 in a second sweep
    if (jl == 0) shm[cl] = tree(0) * invM;
    __syncthreads();
    const float m = shm[cl];
    float btw = 0.f;
    if (c < C)
      for (int j = jl; j < nsplit; j += 16) {
        const float* base = partial + (long)j * stride + c;
        const float nb = base[2 * C];
        const float d = base[0] / nb - m;
        btw += nb * d * d;
      }
    sh[0][cl][jl] = btw;
    __syncthreads();
    if (jl != 0 || c >= C) return;
    const float m2 = tree(1) + tree(0);
    const float var = m2 * invM;
    e.mean[c] = m;
    e.rstd[c] = rsqrtf(var + e.eps);
    if (e.running_mean) e.running_mean[c] = (1.f - e.momentum) * e.running_mean[c] + e.momentum * m;
    if (e.running_var) {
      const float unb = e.M > 1 ? m2 / (float)(e.M - 1) : var;
      e.running_var[c] = (1.f - e.momentum) * e.running_var[c] + e.momentum * unb;
    }
    return;
  }
  if (jl != 0 || c >= C) return;
  float S[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) S[k] = tree(k);
  if (e.sums_out) {
    e.sums_out[c] = S[0];
    e.sums_out[C + c] = S[1];
    e.sums_out[2 * C + c] = S[2];
  }
  const float g = e.gamma[c], rs = e.rstd_in[c];
  const float cc = g * rs;
  if (e.modes & 1) {
    float* coef = e.coef;
    coef[c] = cc;
    coef[C + c] = 0.f;
    coef[2 * C + c] = -cc * S[1] * invM;
    coef[3 * C + c] = -cc * S[0] * invM;
    if (e.ggamma0) e.ggamma0[c] = (e.beta_acc0 != 0.f ? e.beta_acc0 * e.ggamma0[c] : 0.f) + S[1];
    if (e.gbeta0) e.gbeta0[c] = (e.beta_acc0 != 0.f ? e.beta_acc0 * e.gbeta0[c] : 0.f) + S[0];
    if (e.m_out) {
      e.m_out[c] = S[0] * invM;
      e.m_out[C + c] = S[1] * invM;
    }
  }
  if (e.modes & 2) {
    float* coef = e.coef + 4L * C;
    const float m1 = e.saved_m[c], m2 = e.saved_m[C + c];
    const float ubar = S[0] * invM, mux = S[1] * invM;
    const float Q = S[2] - m1 * S[0] - m2 * S[1];
    const float k = cc * rs;
    coef[c] = -k * m2;
    coef[C + c] = -k * mux;
    coef[2 * C + c] = k * 2.f * m2 * mux - g * Q * rs * rs * invM;
    coef[3 * C + c] = k * (m2 * ubar + m1 * mux);
    if (e.ggamma1) e.ggamma1[c] = (e.beta_acc1 != 0.f ? e.beta_acc1 * e.ggamma1[c] : 0.f) + Q * rs;
  }
}

__global__ __launch_bounds__(256) void bn_affine3_kernel(const float* __restrict__ v1, long v1_pitch,
                                                         const float* __restrict__ v2, long v2_pitch,
                                                         const float* __restrict__ a, long a_pitch,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ A1, const float* __restrict__ A2,
                                                         const float* __restrict__ A3, const float* __restrict__ A4,
                                                         float* __restrict__ out, long out_pitch, int accumulate,
                                                         int act, float slope, long rows, int C) {
  const int c4n = C / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * c4n) return;
  const long r = i / c4n;
  const int c = (int)(i % c4n) * 4;
  f32x4 o = A4 ? *reinterpret_cast<const f32x4*>(A4 + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  if (v1 && A1) o += *reinterpret_cast<const f32x4*>(v1 + r * v1_pitch + c) * *reinterpret_cast<const f32x4*>(A1 + c);
  if (v2 && A2) o += *reinterpret_cast<const f32x4*>(v2 + r * v2_pitch + c) * *reinterpret_cast<const f32x4*>(A2 + c);
  if (a && A3) {
    const f32x4 xh = (*reinterpret_cast<const f32x4*>(a + r * a_pitch + c) - *reinterpret_cast<const f32x4*>(mean + c)) *
                     *reinterpret_cast<const f32x4*>(rstd + c);
    o += xh * *reinterpret_cast<const f32x4*>(A3 + c);
  }
  if (act == 3) {          // ReLU
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.f ? o[k] : 0.f;
  } else if (act == 4) {   // LeakyReLU
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.f ? o[k] : o[k] * slope;
  }
  float* dst = out + r * out_pitch + c;
  if (accumulate) o += *reinterpret_cast<const f32x4*>(dst);
  *reinterpret_cast<f32x4*>(dst) = o;
}

// two affine3 results from one pass over (v1, v2, a): out_k = A1k*v1 + A2k*v2 + A3k*xhat + A4k, k = 0, 1
// (coef = [2][4][C]; the gradient-penalty adjoint of T needs both)
__global__ __launch_bounds__(256) void bn_affine3x2_kernel(const float* __restrict__ v1, long v1_pitch,
                                                           const float* __restrict__ v2, long v2_pitch,
                                                           const float* __restrict__ a, long a_pitch,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ coef, float* __restrict__ out0,
                                                           long out0_pitch, float* __restrict__ out1, long out1_pitch,
                                                           long rows, int C) {
  const int c4n = C / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * c4n) return;
  const long r = i / c4n;
  const int c = (int)(i % c4n) * 4;
  const f32x4 x1 = *reinterpret_cast<const f32x4*>(v1 + r * v1_pitch + c);
  const f32x4 x2 = *reinterpret_cast<const f32x4*>(v2 + r * v2_pitch + c);
  const f32x4 xh = (*reinterpret_cast<const f32x4*>(a + r * a_pitch + c) - *reinterpret_cast<const f32x4*>(mean + c)) *
                   *reinterpret_cast<const f32x4*>(rstd + c);
  auto K = [&](int k, int j) { return *reinterpret_cast<const f32x4*>(coef + ((long)k * 4 + j) * C + c); };
  // same association as bn_affine3_kernel: ((A4 + v1*A1) + v2*A2) + xhat*A3; A2 of set 0 is identically 0
  f32x4 o0 = K(0, 3);
  o0 += x1 * K(0, 0);
  o0 += xh * K(0, 2);
  f32x4 o1 = K(1, 3);
  o1 += x1 * K(1, 0);
  o1 += x2 * K(1, 1);
  o1 += xh * K(1, 2);
  *reinterpret_cast<f32x4*>(out0 + r * out0_pitch + c) = o0;
  *reinterpret_cast<f32x4*>(out1 + r * out1_pitch + c) = o1;
}

// coefficient kernels (one thread per channel), coef layout [4][C] = A1, A2, A3, A4
// mode 0 (operator T):    A1 = c, A2 = 0, A3 = -c*S2/M, A4 = -c*S1/M,   c = gamma*rstd
//                         optional grads: ggamma = beta_acc*ggamma + S2, gbeta = beta_acc*gbeta + S1
// mode 1 (GP statistics): sums (S1,S2,S3) of u; saved (m1, m2) = (mean gn, mean gn*xhat)
//                         Q = S3 - m1*S1 - m2*S2;  ggamma = beta_acc*ggamma + Q*rstd
//                         A1 = -(c*rstd)*m2, A2 = -(c*rstd)*mux, A3 = (c*rstd)*2*m2*mux - gamma*Q*rstd^2/M,
//                         A4 = (c*rstd)*(m2*ubar + m1*mux),   ubar = S1/M, mux = S2/M
__global__ void bn_coef_kernel(int mode, const float* __restrict__ sums, const float* __restrict__ gamma,
                               const float* __restrict__ rstd, const float* __restrict__ saved_m, int C, long M,
                               float* __restrict__ coef, float* __restrict__ ggamma, float* __restrict__ gbeta,
                               float beta_acc, float* __restrict__ m_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float S1 = sums[c], S2 = sums[C + c], S3 = sums[2 * C + c];
  const float g = gamma[c], rs = rstd[c];
  const float cc = g * rs;
  const float invM = 1.f / (float)M;
  if (mode == 0) {
    coef[c] = cc;
    coef[C + c] = 0.f;
    coef[2 * C + c] = -cc * S2 * invM;
    coef[3 * C + c] = -cc * S1 * invM;
    if (ggamma) ggamma[c] = (beta_acc != 0.f ? beta_acc * ggamma[c] : 0.f) + S2;
    if (gbeta) gbeta[c] = (beta_acc != 0.f ? beta_acc * gbeta[c] : 0.f) + S1;
    if (m_out) {
      m_out[c] = S1 * invM;
      m_out[C + c] = S2 * invM;
    }
  } else {
    const float m1 = saved_m[c], m2 = saved_m[C + c];
    const float ubar = S1 * invM, mux = S2 * invM;
    const float Q = S3 - m1 * S1 - m2 * S2;
    const float k = cc * rs;
    coef[c] = -k * m2;
    coef[C + c] = -k * mux;
    coef[2 * C + c] = k * 2.f * m2 * mux - g * Q * rs * rs * invM;
    coef[3 * C + c] = k * (m2 * ubar + m1 * mux);
    if (ggamma) ggamma[c] = (beta_acc != 0.f ? beta_acc * ggamma[c] : 0.f) + Q * rs;
  }
}

int check_mc(const void* p, long pitch, long rows, int C, const char* who) {
  LGM_REQUIRE(p && rows > 0 && C > 0 && C % 4 == 0 && pitch % 4 == 0 && pitch >= C && lgm_aligned16(p),
              "%s: need a 16B-aligned [rows, C] matrix with C, pitch multiples of 4", who);
  return LGM_OK;
}

int reduce3(int mode, const float* v1, long v1_pitch, const float* v2, long v2_pitch, const float* a, long a_pitch,
            const float* mean, const float* rstd, long rows, int C, float* out3, float* ws, hipStream_t s) {
  const long rpb = bn_rows(rows);
  const int ns = lgm_cdiv(rows, rpb);
  hipLaunchKernelGGL(bn_reduce_stage1, dim3(lgm_cdiv(C, 64), ns), dim3(256), 0, s, v1, v1_pitch, v2, v2_pitch, a,
                     a_pitch, mean, rstd, mode, rows, C, rpb, ws);
  hipLaunchKernelGGL(bn_reduce_stage2, dim3(lgm_cdiv(3 * C, 64)), dim3(256), 0, s, (const float*)ws, ns, C, out3);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

int reduce3_epi(int mode, const float* v1, long v1_pitch, const float* v2, long v2_pitch, const float* a, long a_pitch,
                const float* mean, const float* rstd, long rows, int C, float* ws, const BnEpi& e, hipStream_t s) {
  const long rpb = bn_rows(rows);
  const int ns = lgm_cdiv(rows, rpb);
  hipLaunchKernelGGL(bn_reduce_stage1, dim3(lgm_cdiv(C, 64), ns), dim3(256), 0, s, v1, v1_pitch, v2, v2_pitch, a,
                     a_pitch, mean, rstd, mode, rows, C, rpb, ws);
  hipLaunchKernelGGL(bn_stage2_epi_kernel, dim3(lgm_cdiv(C, 16)), dim3(256), 0, s, (const float*)ws, ns, C, e);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

}  // namespace

extern "C" int64_t lgm_bn_workspace(int64_t rows, int C) {
  return ((int64_t)lgm_cdiv(rows, bn_rows(rows)) * 3 * C + 3 * C) * (int64_t)sizeof(float) + 64;
}

extern "C" int lgm_bn_stats(const float* a, int64_t a_pitch, int64_t rows, int C, float eps, float momentum,
                            float* mean, float* rstd, float* running_mean, float* running_var, void* workspace,
                            void* stream) {
  if (int rc = check_mc(a, a_pitch, rows, C, "bn_stats")) return rc;
  LGM_REQUIRE(mean && rstd && workspace, "bn_stats: null pointer");
  // block-local sums and squared deviations, combined (and finalised) in the second stage: two launches
  BnEpi e{};
  e.epi = 1; e.M = (long)rows; e.eps = eps; e.momentum = momentum;
  e.mean = mean; e.rstd = rstd; e.running_mean = running_mean; e.running_var = running_var;
  const long rpb = bn_rows(rows);
  const int ns = lgm_cdiv(rows, rpb);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_stats_stage1, dim3(lgm_cdiv(C, 64), ns), dim3(256), 0, s, a, (long)a_pitch, (long)rows, C, rpb,
                     (float*)workspace);
  hipLaunchKernelGGL(bn_stage2_epi_kernel, dim3(lgm_cdiv(C, 16)), dim3(256), 0, s, (const float*)workspace, ns, C, e);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

// batch statistics from row-tile partials (sum, M2, n) a producing convolution left behind (lgm_conv_xy_stats /
// lgm_conv_yx_stats): the second reduction stage alone - the read pass over the activation is gone
extern "C" int lgm_bn_stats_from_tiles(const float* partial, int ntiles, int C, int64_t rows, float eps, float momentum,
                                       float* mean, float* rstd, float* running_mean, float* running_var,
                                       void* stream) {
  LGM_REQUIRE(partial && ntiles > 0 && C > 0 && rows > 0 && mean && rstd, "bn_stats_from_tiles: bad arguments");
  BnEpi e{};
  e.epi = 1; e.M = (long)rows; e.eps = eps; e.momentum = momentum;
  e.mean = mean; e.rstd = rstd; e.running_mean = running_mean; e.running_var = running_var;
  hipLaunchKernelGGL(bn_stage2_epi_kernel, dim3(lgm_cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, partial, ntiles, C,
                     e);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

// reduce3 + bn_coef in two launches: modes bit 0 -> coefficient set of mode 0 (coef8[0..4C), ggamma0/gbeta0/m_out),
// bit 1 -> set of mode 1 (coef8[4C..8C), ggamma1, needs saved_m).  sums3 (optional) also receives (S1, S2, S3).
extern "C" int lgm_bn_reduce3_coef(int modes, const float* v1, int64_t v1_pitch, const float* v2, int64_t v2_pitch,
                                   const float* a, int64_t a_pitch, const float* mean, const float* rstd,
                                   const float* gamma, const float* saved_m, int64_t rows, int C, float* coef8,
                                   float* ggamma0, float* gbeta0, float beta_acc0, float* m_out, float* ggamma1,
                                   float beta_acc1, float* sums3, void* workspace, void* stream) {
  if (int rc = check_mc(v1, v1_pitch, rows, C, "bn_reduce3_coef(v1)")) return rc;
  LGM_REQUIRE(a && mean && rstd && gamma && coef8 && workspace && (modes & 3) && (!(modes & 2) || saved_m),
              "bn_reduce3_coef: bad arguments");
  BnEpi e{};
  e.epi = 2; e.modes = modes; e.M = (long)rows; e.gamma = gamma; e.rstd_in = rstd; e.saved_m = saved_m;
  e.coef = coef8; e.ggamma0 = ggamma0; e.gbeta0 = gbeta0; e.beta_acc0 = beta_acc0; e.m_out = m_out;
  e.ggamma1 = ggamma1; e.beta_acc1 = beta_acc1; e.sums_out = sums3;
  return reduce3_epi(0, v1, v1_pitch, v2, v2_pitch, a, a_pitch, mean, rstd, rows, C, (float*)workspace, e,
                     (hipStream_t)stream);
}

// the second stage alone, over (sum v1, sum v1 * xhat, 0) rows a convolution's epilogue left per row tile (LgmPostOp.bn_*)
extern "C" int lgm_bn_reduce3_coef_tiles(int modes, const float* partial, int tiles, const float* gamma, const float* rstd,
                                         const float* saved_m, int64_t rows, int C, float* coef8, float* ggamma0,
                                         float* gbeta0, float beta_acc0, float* m_out, float* ggamma1, float beta_acc1,
                                         float* sums3, void* stream) {
  LGM_REQUIRE(partial && tiles > 0 && rstd && gamma && coef8 && rows > 0 && C > 0 && (modes & 3) && (!(modes & 2) || saved_m),
              "bn_reduce3_coef_tiles: bad arguments");
  BnEpi e{};
  e.epi = 2; e.modes = modes; e.M = (long)rows; e.gamma = gamma; e.rstd_in = rstd; e.saved_m = saved_m;
  e.coef = coef8; e.ggamma0 = ggamma0; e.gbeta0 = gbeta0; e.beta_acc0 = beta_acc0; e.m_out = m_out;
  e.ggamma1 = ggamma1; e.beta_acc1 = beta_acc1; e.sums_out = sums3;
  hipLaunchKernelGGL(bn_stage2_epi_kernel, dim3(lgm_cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, partial, tiles, C, e);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_bn_affine3x2(const float* v1, int64_t v1_pitch, const float* v2, int64_t v2_pitch, const float* a,
                                int64_t a_pitch, const float* mean, const float* rstd, const float* coef8, float* out0,
                                int64_t out0_pitch, float* out1, int64_t out1_pitch, int64_t rows, int C, void* stream) {
  if (int rc = check_mc(out0, out0_pitch, rows, C, "bn_affine3x2(out0)")) return rc;
  if (int rc = check_mc(out1, out1_pitch, rows, C, "bn_affine3x2(out1)")) return rc;
  LGM_REQUIRE(v1 && v2 && a && mean && rstd && coef8, "bn_affine3x2: null pointer");
  hipLaunchKernelGGL(bn_affine3x2_kernel, dim3(lgm_cdiv(rows * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, v1,
                     (long)v1_pitch, v2, (long)v2_pitch, a, (long)a_pitch, mean, rstd, coef8, out0, (long)out0_pitch,
                     out1, (long)out1_pitch, (long)rows, C);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_bn_reduce3(const float* v1, int64_t v1_pitch, const float* v2, int64_t v2_pitch, const float* a,
                              int64_t a_pitch, const float* mean, const float* rstd, int64_t rows, int C,
                              float* sums3, void* workspace, void* stream) {
  if (int rc = check_mc(v1, v1_pitch, rows, C, "bn_reduce3(v1)")) return rc;
  LGM_REQUIRE(sums3 && workspace && (!a || (mean && rstd)), "bn_reduce3: null pointer");
  return reduce3(0, v1, v1_pitch, v2, v2_pitch, a, a_pitch, mean, rstd, rows, C, sums3, (float*)workspace,
                 (hipStream_t)stream);
}

extern "C" int lgm_bn_coef(int mode, const float* sums3, const float* gamma, const float* rstd, const float* saved_m,
                           int C, int64_t M, float* coef4, float* ggamma, float* gbeta, float beta_acc, float* m_out,
                           void* stream) {
  LGM_REQUIRE(sums3 && gamma && rstd && coef4 && C > 0 && M > 0 && (mode == 0 || saved_m), "bn_coef: bad arguments");
  hipLaunchKernelGGL(bn_coef_kernel, dim3(lgm_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, mode, sums3, gamma, rstd,
                     saved_m, C, (long)M, coef4, ggamma, gbeta, beta_acc, m_out);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_bn_affine3(const float* v1, int64_t v1_pitch, const float* v2, int64_t v2_pitch, const float* a,
                              int64_t a_pitch, const float* mean, const float* rstd, const float* A1, const float* A2,
                              const float* A3, const float* A4, float* out, int64_t out_pitch, int accumulate, int act,
                              float slope, int64_t rows, int C, void* stream) {
  if (int rc = check_mc(out, out_pitch, rows, C, "bn_affine3(out)")) return rc;
  LGM_REQUIRE(!a || (mean && rstd), "bn_affine3: xhat term needs mean/rstd");
  hipLaunchKernelGGL(bn_affine3_kernel, dim3(lgm_cdiv(rows * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, v1,
                     (long)v1_pitch, v2, (long)v2_pitch, a, (long)a_pitch, mean, rstd, A1, A2, A3, A4, out,
                     (long)out_pitch, accumulate, act, slope, (long)rows, C);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
