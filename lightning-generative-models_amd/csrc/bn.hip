// Train-mode BatchNorm2d building blocks for the DCGAN generator / critic (dcgan.py:86-90,
// 158-161) including what the WGAN-GP double backward (wgan.py:117-156) needs.
//
// Everything is expressed with three primitives over [M = B*H*W, C] NHWC matrices:
//   lgm_bn_stats     per-channel batch mean / rstd (two-pass, deterministic) + running-stat update
//   lgm_bn_reduce3   per-channel  S1 = sum v1,  S2 = sum v1*xhat,  S3 = sum v1*v2
//   lgm_bn_affine3   out = A1[c]*v1 + A2[c]*v2 + A3[c]*xhat + A4[c]      (xhat = (a-mean)*rstd)
// plus two tiny coefficient kernels.  With them
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

// finalize batch statistics: mode 1 result -> mean; mode 2 result -> rstd (+ running stats, torch semantics:
// running_var uses the unbiased estimate; momentum 0.1)
__global__ void bn_finalize_kernel(const float* __restrict__ sums, int C, long M, int which, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                                   float* __restrict__ running_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (which == 0) {
    const float m = sums[c] / (float)M;
    mean[c] = m;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
  } else {
    const float var = sums[c] / (float)M;
    rstd[c] = rsqrtf(var + eps);
    if (running_var) {
      const float unb = M > 1 ? sums[c] / (float)(M - 1) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
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

}  // namespace

extern "C" int64_t lgm_bn_workspace(int64_t rows, int C) {
  return ((int64_t)lgm_cdiv(rows, bn_rows(rows)) * 3 * C + 3 * C) * (int64_t)sizeof(float) + 64;
}

extern "C" int lgm_bn_stats(const float* a, int64_t a_pitch, int64_t rows, int C, float eps, float momentum,
                            float* mean, float* rstd, float* running_mean, float* running_var, void* workspace,
                            void* stream) {
  if (int rc = check_mc(a, a_pitch, rows, C, "bn_stats")) return rc;
  LGM_REQUIRE(mean && rstd && workspace, "bn_stats: null pointer");
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  float* sums = ws + (long)lgm_cdiv(rows, bn_rows(rows)) * 3 * C;
  if (int rc = reduce3(1, nullptr, 0, nullptr, 0, a, a_pitch, nullptr, nullptr, rows, C, sums, ws, s)) return rc;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(lgm_cdiv(C, 256)), dim3(256), 0, s, (const float*)sums, C, (long)rows, 0,
                     eps, momentum, mean, rstd, running_mean, running_var);
  if (int rc = reduce3(2, nullptr, 0, nullptr, 0, a, a_pitch, mean, nullptr, rows, C, sums, ws, s)) return rc;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(lgm_cdiv(C, 256)), dim3(256), 0, s, (const float*)sums, C, (long)rows, 1,
                     eps, momentum, mean, rstd, running_mean, running_var);
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
