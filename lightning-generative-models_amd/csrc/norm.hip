// GroupNorm (+FiLM scale/shift, +SiLU, +residual) and RMSNorm for NHWC fp32 activations.
// HBM-bound kernels: float4 accesses, coalesced along channels, deterministic reductions
// (fixed-order LDS combines, no float atomics).
//
// Reference arithmetic: Block.forward ddpm.py:164-173 (GroupNorm(8, C, eps=1e-5) -> x*(scale+1)+shift
// -> SiLU), RMSNorm ddpm.py:107-113 (F.normalize(x, dim=1) * g * sqrt(C)).
#include <stdlib.h>

#include "lgm_common.h"

namespace {

__device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }
__device__ __forceinline__ float silu_grad(float z) {
  const float s = 1.f / (1.f + __expf(-z));
  return s * (1.f + z * (1.f - s));
}

// Blocks narrower than a 128-byte row (CB < 32 channels) share every cache line of x with their 32 / CB - 1 siblings.
// Consecutive block ids go to different XCDs, i.e. different L2s: every line was fetched from memory once per sibling
// (64 x 64 maps, B = 64: 268 MB instead of 134 MB per two-pass statistics launch, 44 us).  This map puts the siblings on
// consecutive slots of ONE XCD (block id = xcd + 8 slot), where the second reader hits the line the first one fetched.
__device__ __forceinline__ int gn_sibling_map(int bid, int total, int CB) {
  const int shr = 32 / CB;
  if (shr < 2 || total % (8 * shr)) return bid;
  const int xcd = bid & 7, slot = bid >> 3;
  return ((slot / shr) * 8 + xcd) * shr + slot % shr;
}

// ---------------------------------------------------------------------------------------
// GN statistics: one block per (image, channel block of CB channels); two in-kernel passes
// (mean, then centred second moment) so the variance does not suffer cancellation.
// ---------------------------------------------------------------------------------------
template <int NT>   // threads per block: 1024 on large maps (more loads in flight for an HBM-bound pass)
__global__ __launch_bounds__(NT) void gn_stats_kernel(const float* __restrict__ x, long pitch, int HW,
                                                       int C, int G, int CB, float eps,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ ss, long ss_pitch,
                                                       float* __restrict__ mean, float* __restrict__ rstd,
                                                       float* __restrict__ A, float* __restrict__ Bc) {
  __shared__ float sh[NT * 4];
  __shared__ float chs[256];
  __shared__ float gmean[64], grstd[64];
  const int nb = C / CB;
  const int bid = gn_sibling_map((int)blockIdx.x, (int)gridDim.x, CB);
  const int b = bid / nb, cb = bid % nb;
  const int c0 = cb * CB;
  const int Cg = C / G;
  const int tq = CB / 4;              // threads per pixel
  const int ppb = NT / tq;            // pixel lanes
  const int tid = threadIdx.x;
  const int q = tid % tq, pl = tid / tq;
  const bool active = pl < ppb;
  const float* xb = x + (long)b * HW * pitch + c0 + q * 4;
  const int ngl = CB / Cg;            // groups in this block
  const float inv_n = 1.f / ((float)Cg * (float)HW);

  // two-level fixed-order reduction of the per-thread partials in sh ([pixel lane][channel]):
  // one thread per channel over the pixel lanes, then one thread per group over its channels
  // fixed-order reduction of the per-thread partials in sh ([pixel lane][channel]) in two levels: every thread sums
  // the lanes pp = j, j + J, ... of one channel (J = NT / CB parts), then one thread per channel sums the J parts - a
  // chain of ppb / J + J additions instead of ppb (128 dependent LDS reads by 8 ... 32 threads were ~3 us per
  // reduction, twice per launch: a third of this kernel's time at small batches)
  auto group_reduce = [&](float* dst, bool second) {
    __syncthreads();
    {
      const int J = NT / CB, cch = tid % CB, j = tid / CB;
      float acc = 0.f;
      if (j < J)
        for (int pp = j; pp < ppb; pp += J) acc += sh[pp * CB + cch];
      __syncthreads();                       // everyone has read its sh entries: sh is reused for the parts
      if (j < J) sh[j * CB + cch] = acc;
      __syncthreads();
      if (tid < CB) {
        float a2 = 0.f;
        for (int jj = 0; jj < J; ++jj) a2 += sh[jj * CB + tid];
        chs[tid] = a2;
      }
    }
    __syncthreads();
    if (tid < ngl) {
      float acc = 0.f;
      for (int c = 0; c < Cg; ++c) acc += chs[tid * Cg + c];
      dst[tid] = second ? rsqrtf(acc * inv_n + eps) : acc * inv_n;
    }
    __syncthreads();
  };

  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    int p = pl;
    // large maps (64 x 64: 16 pixels per thread and pass): eight independent 16-byte loads in flight - with four, a block
    // of 1024 threads streamed its 256 KB slice at 3.2 TB/s chip-wide (two dependent round trips per pass more)
    for (; p + 7 * ppb < HW; p += 8 * ppb) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(xb + (long)(p + u * ppb) * pitch);
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; p + 3 * ppb < HW; p += 4 * ppb) {     // four independent loads in flight
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(xb + (long)p * pitch);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(xb + (long)(p + ppb) * pitch);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(xb + (long)(p + 2 * ppb) * pitch);
      const f32x4 v3 = *reinterpret_cast<const f32x4*>(xb + (long)(p + 3 * ppb) * pitch);
      s += (v0 + v1) + (v2 + v3);
    }
    for (; p < HW; p += ppb) s += *reinterpret_cast<const f32x4*>(xb + (long)p * pitch);
  }
  *reinterpret_cast<f32x4*>(&sh[tid * 4]) = s;
  group_reduce(gmean, false);
  f32x4 mu;
#pragma unroll
  for (int k = 0; k < 4; ++k) mu[k] = gmean[(q * 4 + k) / Cg];
  f32x4 s2 = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    int p = pl;
    for (; p + 7 * ppb < HW; p += 8 * ppb) {
      f32x4 d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) d[u] = *reinterpret_cast<const f32x4*>(xb + (long)(p + u * ppb) * pitch) - mu;
      s2 += ((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) + ((d[4] * d[4] + d[5] * d[5]) + (d[6] * d[6] + d[7] * d[7]));
    }
    for (; p + 3 * ppb < HW; p += 4 * ppb) {
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(xb + (long)p * pitch) - mu;
      const f32x4 d1 = *reinterpret_cast<const f32x4*>(xb + (long)(p + ppb) * pitch) - mu;
      const f32x4 d2 = *reinterpret_cast<const f32x4*>(xb + (long)(p + 2 * ppb) * pitch) - mu;
      const f32x4 d3 = *reinterpret_cast<const f32x4*>(xb + (long)(p + 3 * ppb) * pitch) - mu;
      s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    for (; p < HW; p += ppb) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(xb + (long)p * pitch) - mu;
      s2 += d * d;
    }
  }
  *reinterpret_cast<f32x4*>(&sh[tid * 4]) = s2;
  group_reduce(grstd, true);
  if (tid < ngl) {
    const int g = c0 / Cg + tid;
    mean[b * G + g] = gmean[tid];
    rstd[b * G + g] = grstd[tid];
  }
  // fused coefficient computation: z = x * A[b,c] + Bc[b,c]
  if (tid < CB) {
    const int c = c0 + tid;
    const float m = gmean[tid / Cg], rs = grstd[tid / Cg];
    float a = rs * gamma[c];
    float bb = beta[c] - m * a;
    if (ss) {
      const float sc = ss[(long)b * ss_pitch + c] + 1.f;
      const float shf = ss[(long)b * ss_pitch + C + c];
      a *= sc;
      bb = bb * sc + shf;
    }
    A[(long)b * C + c] = a;
    Bc[(long)b * C + c] = bb;
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, long x_pitch,
                                                       const float* __restrict__ A, const float* __restrict__ Bc,
                                                       const float* __restrict__ res, long res_pitch,
                                                       float* __restrict__ y, long y_pitch, long npix, int HW,
                                                       int C, int act) {
  const int c4n = C / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * c4n) return;
  const long pix = i / c4n;
  const int c = (int)(i % c4n) * 4;
  const int b = (int)(pix / HW);
  const f32x4 xv = *reinterpret_cast<const f32x4*>(x + pix * x_pitch + c);
  const f32x4 a = *reinterpret_cast<const f32x4*>(A + (long)b * C + c);
  const f32x4 bc = *reinterpret_cast<const f32x4*>(Bc + (long)b * C + c);
  f32x4 z = xv * a + bc;
  if (act) {
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] = silu_f(z[k]);
  }
  if (res) z += *reinterpret_cast<const f32x4*>(res + pix * res_pitch + c);
  *reinterpret_cast<f32x4*>(y + pix * y_pitch + c) = z;
}

// ---------------------------------------------------------------------------------------
// One-pass forward: statistics AND normalisation in one kernel.  A block owns the (image, CB
// channels) slice as above but keeps it in registers -- NV 16-byte loads per thread, all in flight
// at once -- so x is read from memory once instead of three times (mean pass, centred-moment pass,
// apply pass) and the second launch disappears.  Same thread -> (pixel lane, channel quad) map and
// the same fixed-order reductions as gn_stats_kernel.
// ---------------------------------------------------------------------------------------
template <int NT, int NV>
__global__ __launch_bounds__(NT) void gn_fused_fwd_kernel(const float* __restrict__ x, long pitch, int HW, int C, int G,
                                                           int CB, float eps, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ ss,
                                                           long ss_pitch, int act, const float* __restrict__ res,
                                                           long res_pitch, float* __restrict__ y, long y_pitch,
                                                           float* __restrict__ mean, float* __restrict__ rstd,
                                                           float* __restrict__ A, float* __restrict__ Bc,
                                                           const float* __restrict__ planes, long pstride, int splits,
                                                           const float* __restrict__ cbias, float* __restrict__ xw) {
  __shared__ float sh[NT * 4];
  __shared__ float chs[256];
  __shared__ float gmean[64], grstd[64];
  __shared__ __align__(16) float cA[256], cB[256];
  const int nb = C / CB;
  const int bid = gn_sibling_map((int)blockIdx.x, (int)gridDim.x, CB);
  const int b = bid / nb, cb = bid % nb;
  const int c0 = cb * CB;
  const int Cg = C / G;
  const int tq = CB / 4;              // threads per pixel
  const int ppb = NT / tq;            // pixel lanes (HW == NV * ppb, checked by the host)
  const int tid = threadIdx.x;
  const int q = tid % tq, pl = tid / tq;
  const float* xb = x + (long)b * HW * pitch + c0 + q * 4;
  const int ngl = CB / Cg;
  const float inv_n = 1.f / ((float)Cg * (float)HW);

  // fixed-order reduction of the per-thread partials in sh ([pixel lane][channel]) in two levels: every thread sums
  // the lanes pp = j, j + J, ... of one channel (J = NT / CB parts), then one thread per channel sums the J parts - a
  // chain of ppb / J + J additions instead of ppb (128 dependent LDS reads by 8 ... 32 threads were ~3 us per
  // reduction, twice per launch: a third of this kernel's time at small batches)
  auto group_reduce = [&](float* dst, bool second) {
    __syncthreads();
    {
      const int J = NT / CB, cch = tid % CB, j = tid / CB;
      float acc = 0.f;
      if (j < J)
        for (int pp = j; pp < ppb; pp += J) acc += sh[pp * CB + cch];
      __syncthreads();                       // everyone has read its sh entries: sh is reused for the parts
      if (j < J) sh[j * CB + cch] = acc;
      __syncthreads();
      if (tid < CB) {
        float a2 = 0.f;
        for (int jj = 0; jj < J; ++jj) a2 += sh[jj * CB + tid];
        chs[tid] = a2;
      }
    }
    __syncthreads();
    if (tid < ngl) {
      float acc = 0.f;
      for (int c = 0; c < Cg; ++c) acc += chs[tid * Cg + c];
      dst[tid] = second ? rsqrtf(acc * inv_n + eps) : acc * inv_n;
    }
    __syncthreads();
  };

  f32x4 v[NV];
  if (planes) {
    // x is still in pieces: the producing convolution split its reduction and left `splits` partial planes
    // [B*HW][C] (dense) instead of running its reducer.  Sum them here in the reducer's fixed order (plane 0, 1, ...,
    // then the convolution's bias), and write the finished x once - the backward pass reads it.
    const float* pb = planes + ((long)b * HW) * C + c0 + q * 4;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = *reinterpret_cast<const f32x4*>(pb + (long)(pl + k * ppb) * C);
    // PR planes per round: PR x NV independent 16-byte loads in flight (one plane per round left the block waiting for
    // `splits` dependent memory round trips - 16 of them on the 4x4 maps at small batches); the ADDITIONS stay in
    // plane order, so the sum is the reducer's
    constexpr int PR = NV >= 4 ? 2 : NV == 2 ? 4 : 8;
    int sidx = 1;
    for (; sidx + PR <= splits; sidx += PR) {
      const float* ps = pb + (long)sidx * pstride;
      f32x4 t[PR][NV];
#pragma unroll
      for (int j = 0; j < PR; ++j)
#pragma unroll
        for (int k = 0; k < NV; ++k) t[j][k] = *reinterpret_cast<const f32x4*>(ps + (long)j * pstride + (long)(pl + k * ppb) * C);
#pragma unroll
      for (int j = 0; j < PR; ++j)
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] += t[j][k];
    }
    for (; sidx < splits; ++sidx) {
      const float* ps = pb + (long)sidx * pstride;
#pragma unroll
      for (int k = 0; k < NV; ++k) v[k] += *reinterpret_cast<const f32x4*>(ps + (long)(pl + k * ppb) * C);
    }
    if (cbias) {
      const f32x4 cb4 = *reinterpret_cast<const f32x4*>(cbias + c0 + q * 4);
#pragma unroll
      for (int k = 0; k < NV; ++k) v[k] += cb4;
    }
    float* xo = xw + (long)b * HW * pitch + c0 + q * 4;
#pragma unroll
    for (int k = 0; k < NV; ++k) *reinterpret_cast<f32x4*>(xo + (long)(pl + k * ppb) * pitch) = v[k];
  } else {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = *reinterpret_cast<const f32x4*>(xb + (long)(pl + k * ppb) * pitch);
  }
  // Everything else the block will need from memory is requested NOW, before the two reductions: the residual rows
  // and this thread's gamma / beta / FiLM operands.  Issued where they are used they were three more dependent
  // memory round trips on the critical path of a kernel that is pure latency at small batches.
  f32x4 r[NV];
  if (res) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
      r[k] = *reinterpret_cast<const f32x4*>(res + ((long)b * HW + pl + k * ppb) * res_pitch + c0 + q * 4);
  }
  float pg = 0.f, pbeta = 0.f, psc = 1.f, psh = 0.f;
  if (tid < CB) {
    pg = gamma[c0 + tid];
    pbeta = beta[c0 + tid];
    if (ss) {
      psc = ss[(long)b * ss_pitch + c0 + tid] + 1.f;
      psh = ss[(long)b * ss_pitch + C + c0 + tid];
    }
  }
  f32x4 s = v[0];
#pragma unroll
  for (int k = 1; k < NV; ++k) s += v[k];
  *reinterpret_cast<f32x4*>(&sh[tid * 4]) = s;
  group_reduce(gmean, false);
  f32x4 mu;
#pragma unroll
  for (int k = 0; k < 4; ++k) mu[k] = gmean[(q * 4 + k) / Cg];
  f32x4 s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const f32x4 d = v[k] - mu;
    s2 += d * d;
  }
  *reinterpret_cast<f32x4*>(&sh[tid * 4]) = s2;
  group_reduce(grstd, true);
  if (tid < ngl) {
    const int g = c0 / Cg + tid;
    mean[b * G + g] = gmean[tid];
    rstd[b * G + g] = grstd[tid];
  }
  if (tid < CB) {      // z = x * A[b,c] + Bc[b,c]  (also kept for the backward pass)
    const int c = c0 + tid;
    const float m = gmean[tid / Cg], rs = grstd[tid / Cg];
    float a = rs * pg;
    float bb = pbeta - m * a;
    if (ss) {
      a *= psc;
      bb = bb * psc + psh;
    }
    A[(long)b * C + c] = a;
    Bc[(long)b * C + c] = bb;
    cA[tid] = a;
    cB[tid] = bb;
  }
  __syncthreads();
  const f32x4 a4 = *reinterpret_cast<const f32x4*>(&cA[q * 4]);
  const f32x4 b4 = *reinterpret_cast<const f32x4*>(&cB[q * 4]);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    f32x4 z = v[k] * a4 + b4;
    if (act) {
#pragma unroll
      for (int e = 0; e < 4; ++e) z[e] = silu_f(z[e]);
    }
    if (res) z += r[k];
    *reinterpret_cast<f32x4*>(y + ((long)b * HW + pl + k * ppb) * y_pitch + c0 + q * 4) = z;
  }
}

// backward pass 1: S1[b,c] = sum_hw gz, S2[b,c] = sum_hw gz * xhat
template <int NT>
__global__ __launch_bounds__(NT) void gn_bwd_reduce_kernel(const float* __restrict__ x, long x_pitch,
                                                            const float* __restrict__ gy, long gy_pitch,
                                                            const float* __restrict__ A, const float* __restrict__ Bc,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            int HW, int C, int G, int CB, int act,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ ss, long ss_pitch,
                                                            float* __restrict__ gss, long gss_pitch, float gss_beta,
                                                            float* __restrict__ S1, float* __restrict__ S2,
                                                            float* __restrict__ P, float* __restrict__ Qc,
                                                            float* __restrict__ Rc) {
  __shared__ float sh[NT * 8];
  const int nb = C / CB;
  const int bid = gn_sibling_map((int)blockIdx.x, (int)gridDim.x, CB);
  const int b = bid / nb, cb = bid % nb;
  const int c0 = cb * CB;
  const int Cg = C / G;
  const int tq = CB / 4, ppb = NT / tq;
  const int tid = threadIdx.x;
  const int q = tid % tq, pl = tid / tq;
  const bool active = pl < ppb;
  const int c = c0 + q * 4;
  const float* xb = x + (long)b * HW * x_pitch + c;
  const float* gb = gy + (long)b * HW * gy_pitch + c;
  const f32x4 a = *reinterpret_cast<const f32x4*>(A + (long)b * C + c);
  const f32x4 bc = *reinterpret_cast<const f32x4*>(Bc + (long)b * C + c);
  f32x4 mu, rs;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    mu[k] = mean[b * G + (c + k) / Cg];
    rs[k] = rstd[b * G + (c + k) / Cg];
  }
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    int p = pl;
    for (; p + 3 * ppb < HW; p += 4 * ppb) {       // four pixels = eight independent 16-byte loads in flight
      f32x4 xv[4], g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xv[u] = *reinterpret_cast<const f32x4*>(xb + (long)(p + u * ppb) * x_pitch);
        g[u] = *reinterpret_cast<const f32x4*>(gb + (long)(p + u * ppb) * gy_pitch);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (act) {
          const f32x4 z = xv[u] * a + bc;
#pragma unroll
          for (int k = 0; k < 4; ++k) g[u][k] *= silu_grad(z[k]);
        }
        s1 += g[u];
        s2 += g[u] * ((xv[u] - mu) * rs);
      }
    }
    for (; p < HW; p += ppb) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xb + (long)p * x_pitch);
      f32x4 g = *reinterpret_cast<const f32x4*>(gb + (long)p * gy_pitch);
      if (act) {
        const f32x4 z = xv * a + bc;
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] *= silu_grad(z[k]);
      }
      s1 += g;
      s2 += g * ((xv - mu) * rs);
    }
  }
  *reinterpret_cast<f32x4*>(&sh[tid * 8]) = s1;
  *reinterpret_cast<f32x4*>(&sh[tid * 8 + 4]) = s2;
  __syncthreads();
  __shared__ float wa1[256], wa2[256];
  float a1 = 0.f, a2 = 0.f, scv = 1.f;
  {   // two-level fixed-order sum over the pixel lanes (see gn_fused_fwd_kernel): J parts per channel, then the parts
    const int J = NT / CB, cch = tid % CB, j = tid / CB;
    const int qq = cch / 4, k = cch % 4;
    float p1 = 0.f, p2 = 0.f;
    if (j < J)
      for (int pp = j; pp < ppb; pp += J) {
        p1 += sh[(pp * tq + qq) * 8 + k];
        p2 += sh[(pp * tq + qq) * 8 + 4 + k];
      }
    __syncthreads();
    if (j < J) {
      sh[(j * CB + cch) * 2] = p1;
      sh[(j * CB + cch) * 2 + 1] = p2;
    }
    __syncthreads();
  }
  if (tid < CB) {  // one thread per channel
    for (int jj = 0; jj < NT / CB; ++jj) {
      a1 += sh[(jj * CB + tid) * 2];
      a2 += sh[(jj * CB + tid) * 2 + 1];
    }
    const int cc = c0 + tid;
    S1[(long)b * C + cc] = a1;
    S2[(long)b * C + cc] = a2;
    scv = ss ? ss[(long)b * ss_pitch + cc] + 1.f : 1.f;
    const float w = gamma[cc] * scv;
    wa1[tid] = w * a1;
    wa2[tid] = w * a2;
  }
  __syncthreads();
  // fused pass 2: coefficients for gx = P*gz + Qc + x*Rc and the FiLM scale/shift gradients
  if (tid < CB) {
    const int cc = c0 + tid;
    const int g0 = (tid / Cg) * Cg;
    float m1 = 0.f, m2 = 0.f;
    for (int j = 0; j < Cg; ++j) {
      m1 += wa1[g0 + j];
      m2 += wa2[g0 + j];
    }
    const float inv_n = 1.f / ((float)Cg * (float)HW);
    m1 *= inv_n;
    m2 *= inv_n;
    const float m = mean[b * G + cc / Cg], r = rstd[b * G + cc / Cg];
    const long i = (long)b * C + cc;
    P[i] = r * gamma[cc] * scv;
    const float R = -r * m2;          // multiplies xhat
    Rc[i] = R * r;                    // multiplies x
    Qc[i] = -r * m1 - m * r * R;
    if (gss) {
      float gsc = gamma[cc] * a2 + beta[cc] * a1;
      float gsh = a1;
      if (gss_beta != 0.f) {
        gsc += gss_beta * gss[(long)b * gss_pitch + cc];
        gsh += gss_beta * gss[(long)b * gss_pitch + C + cc];
      }
      gss[(long)b * gss_pitch + cc] = gsc;
      gss[(long)b * gss_pitch + C + cc] = gsh;
    }
  }
}

// gamma/beta gradients: reduce over the batch in a fixed order
__device__ __forceinline__ void gn_bwd_affine_body(int blk, const float* __restrict__ S1,
                                                   const float* __restrict__ S2, const float* __restrict__ ss,
                                                   long ss_pitch, int B, int C, float* __restrict__ ggamma,
                                                   float* __restrict__ gbeta, float beta_acc) {
  // block = 16 channels x 16 batch lanes (fixed-order LDS combine => deterministic)
  __shared__ float sg[16][17], sb[16][17];
  const int cl = threadIdx.x & 15, bl = threadIdx.x >> 4;
  const int c = blk * 16 + cl;
  float gg = 0.f, gb = 0.f;
  if (c < C)
    for (int b = bl; b < B; b += 16) {
      const float sc = ss ? ss[(long)b * ss_pitch + c] + 1.f : 1.f;
      gg += sc * S2[(long)b * C + c];
      gb += sc * S1[(long)b * C + c];
    }
  sg[bl][cl] = gg;
  sb[bl][cl] = gb;
  __syncthreads();
  if (bl == 0 && c < C) {
    gg = 0.f;
    gb = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      gg += sg[i][cl];
      gb += sb[i][cl];
    }
    if (beta_acc != 0.f) {
      gg += beta_acc * ggamma[c];
      gb += beta_acc * gbeta[c];
    }
    ggamma[c] = gg;
    gbeta[c] = gb;
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, long x_pitch,
                                                           const float* __restrict__ gy, long gy_pitch,
                                                           const float* __restrict__ A, const float* __restrict__ Bc,
                                                           const float* __restrict__ P, const float* __restrict__ Qc,
                                                           const float* __restrict__ Rc, float* __restrict__ gx,
                                                           long gx_pitch, long npix, int HW, int C, int act,
                                                           int accumulate, int apply_blocks,
                                                           const float* __restrict__ S1, const float* __restrict__ S2,
                                                           const float* __restrict__ ss, long ss_pitch, int B,
                                                           float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                           float affine_beta, float* add_out, long add_pitch) {
  const int affine_blocks = (int)gridDim.x - apply_blocks;
  if ((int)blockIdx.x < affine_blocks) {   // leading blocks: gamma/beta gradients, concurrent with the apply pass
    gn_bwd_affine_body((int)blockIdx.x, S1, S2, ss, ss_pitch, B, C, ggamma, gbeta, affine_beta);
    return;
  }
  const int c4n = C / 4;
  const long i = (long)((int)blockIdx.x - affine_blocks) * blockDim.x + threadIdx.x;
  if (i >= npix * c4n) return;
  const long pix = i / c4n;
  const int c = (int)(i % c4n) * 4;
  const long bc_off = (pix / HW) * C + c;
  const f32x4 xv = *reinterpret_cast<const f32x4*>(x + pix * x_pitch + c);
  f32x4 g = *reinterpret_cast<const f32x4*>(gy + pix * gy_pitch + c);
  if (add_out) {                       // add_out += gy (see gn_fused_bwd_kernel)
    float* ap = add_out + pix * add_pitch + c;
    *reinterpret_cast<f32x4*>(ap) = *reinterpret_cast<const f32x4*>(ap) + g;
  }
  if (act) {
    const f32x4 z = xv * *reinterpret_cast<const f32x4*>(A + bc_off) + *reinterpret_cast<const f32x4*>(Bc + bc_off);
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] *= silu_grad(z[k]);
  }
  f32x4 r = g * *reinterpret_cast<const f32x4*>(P + bc_off) + *reinterpret_cast<const f32x4*>(Qc + bc_off) +
            xv * *reinterpret_cast<const f32x4*>(Rc + bc_off);
  float* o = gx + pix * gx_pitch + c;
  if (accumulate) r += *reinterpret_cast<const f32x4*>(o);
  *reinterpret_cast<f32x4*>(o) = r;
}

// One-pass backward: the block's x and gy slices stay in registers (2 NV 16-byte values per thread),
// so both are read once instead of twice (reduce pass + apply pass).  Same maps, same fixed-order
// reductions and the same coefficient algebra as gn_bwd_reduce_kernel + gn_bwd_apply_kernel; the
// gamma/beta reduction over the batch (needs every block's S1/S2) is the small kernel below.
template <int NT, int NV>
__global__ __launch_bounds__(NT) void gn_fused_bwd_kernel(const float* __restrict__ x, long x_pitch,
                                                           const float* __restrict__ gy, long gy_pitch,
                                                           const float* __restrict__ A, const float* __restrict__ Bc,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           int HW, int C, int G, int CB, int act,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ ss, long ss_pitch,
                                                           float* __restrict__ gss, long gss_pitch, float gss_beta,
                                                           float* __restrict__ S1, float* __restrict__ S2,
                                                           float* __restrict__ gx, long gx_pitch, int accumulate,
                                                           float* __restrict__ T, const float* __restrict__ planes,
                                                           long pstride, int splits, float* add_out, long add_pitch) {
  __shared__ float sh[NT * 8];
  __shared__ float wa1[256], wa2[256];
  __shared__ __align__(16) float cP[256], cQ[256], cR[256];
  const int nb = C / CB;
  const int bid = gn_sibling_map((int)blockIdx.x, (int)gridDim.x, CB);
  const int b = bid / nb, cb = bid % nb;
  const int c0 = cb * CB;
  const int Cg = C / G;
  const int tq = CB / 4, ppb = NT / tq;     // HW == NV * ppb (checked by the host)
  const int tid = threadIdx.x;
  const int q = tid % tq, pl = tid / tq;
  const int c = c0 + q * 4;
  const float* xb = x + (long)b * HW * x_pitch + c;
  const float* gb = gy + (long)b * HW * gy_pitch + c;
  f32x4 xv[NV], g[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) xv[k] = *reinterpret_cast<const f32x4*>(xb + (long)(pl + k * ppb) * x_pitch);
  if (planes) {
    // gy = sum of the partial planes the producing input-gradient convolution left (fixed order, no bias): gy itself
    // has no other reader, so it is never written
    const float* pb = planes + ((long)b * HW) * C + c;
#pragma unroll
    for (int k = 0; k < NV; ++k) g[k] = *reinterpret_cast<const f32x4*>(pb + (long)(pl + k * ppb) * C);
    constexpr int PR = NV >= 4 ? 2 : NV == 2 ? 4 : 8;     // planes per round, additions in plane order (see the forward)
    int sidx = 1;
    for (; sidx + PR <= splits; sidx += PR) {
      const float* ps = pb + (long)sidx * pstride;
      f32x4 t[PR][NV];
#pragma unroll
      for (int j = 0; j < PR; ++j)
#pragma unroll
        for (int k = 0; k < NV; ++k) t[j][k] = *reinterpret_cast<const f32x4*>(ps + (long)j * pstride + (long)(pl + k * ppb) * C);
#pragma unroll
      for (int j = 0; j < PR; ++j)
#pragma unroll
        for (int k = 0; k < NV; ++k) g[k] += t[j][k];
    }
    for (; sidx < splits; ++sidx) {
      const float* ps = pb + (long)sidx * pstride;
#pragma unroll
      for (int k = 0; k < NV; ++k) g[k] += *reinterpret_cast<const f32x4*>(ps + (long)(pl + k * ppb) * C);
    }
  } else {
#pragma unroll
    for (int k = 0; k < NV; ++k) g[k] = *reinterpret_cast<const f32x4*>(gb + (long)(pl + k * ppb) * gy_pitch);
  }
  if (add_out) {
    // add_out += gy: the gradient of an identity residual that leaves the block beside this norm (ResnetBlock with
    // res_conv = Identity, reference ddpm.py:187,200) - gy is in registers here anyway, its own axpby launch is not needed.
    // Rounds of four rows (NV more quads at once would spill at NV = 8).
    float* ab = add_out + (long)b * HW * add_pitch + c;
    constexpr int RN = NV < 4 ? NV : 4;
#pragma unroll
    for (int k0 = 0; k0 < NV; k0 += RN) {
      f32x4 t[RN];
#pragma unroll
      for (int k = 0; k < RN; ++k) t[k] = *reinterpret_cast<const f32x4*>(ab + (long)(pl + (k0 + k) * ppb) * add_pitch);
#pragma unroll
      for (int k = 0; k < RN; ++k) *reinterpret_cast<f32x4*>(ab + (long)(pl + (k0 + k) * ppb) * add_pitch) = t[k] + g[k0 + k];
    }
  }
  const f32x4 a = *reinterpret_cast<const f32x4*>(A + (long)b * C + c);
  const f32x4 bc = *reinterpret_cast<const f32x4*>(Bc + (long)b * C + c);
  f32x4 mu, rs;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    mu[k] = mean[b * G + (c + k) / Cg];
    rs[k] = rstd[b * G + (c + k) / Cg];
  }
  // the per-channel operands of the coefficient stage, requested before the reduction (see the forward kernel)
  float pgam = 0.f, pbet = 0.f, pscv = 1.f, pm = 0.f, pr = 0.f, pgsc = 0.f, pgsh = 0.f;
  if (tid < CB) {
    const int cc = c0 + tid;
    pgam = gamma[cc];
    pbet = beta[cc];
    pscv = ss ? ss[(long)b * ss_pitch + cc] + 1.f : 1.f;
    pm = mean[b * G + cc / Cg];
    pr = rstd[b * G + cc / Cg];
    if (gss && gss_beta != 0.f) {
      pgsc = gss[(long)b * gss_pitch + cc];
      pgsh = gss[(long)b * gss_pitch + C + cc];
    }
  }
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (act) {
      const f32x4 z = xv[k] * a + bc;
#pragma unroll
      for (int e = 0; e < 4; ++e) g[k][e] *= silu_grad(z[e]);
    }
    s1 += g[k];
    s2 += g[k] * ((xv[k] - mu) * rs);
  }
  *reinterpret_cast<f32x4*>(&sh[tid * 8]) = s1;
  *reinterpret_cast<f32x4*>(&sh[tid * 8 + 4]) = s2;
  __syncthreads();
  float a1 = 0.f, a2 = 0.f, scv = 1.f;
  {   // two-level fixed-order sum over the pixel lanes (see gn_fused_fwd_kernel): J parts per channel, then the parts
    const int J = NT / CB, cch = tid % CB, j = tid / CB;
    const int qq = cch / 4, k = cch % 4;
    float p1 = 0.f, p2 = 0.f;
    if (j < J)
      for (int pp = j; pp < ppb; pp += J) {
        p1 += sh[(pp * tq + qq) * 8 + k];
        p2 += sh[(pp * tq + qq) * 8 + 4 + k];
      }
    __syncthreads();
    if (j < J) {
      sh[(j * CB + cch) * 2] = p1;
      sh[(j * CB + cch) * 2 + 1] = p2;
    }
    __syncthreads();
  }
  if (tid < CB) {  // one thread per channel
    for (int jj = 0; jj < NT / CB; ++jj) {
      a1 += sh[(jj * CB + tid) * 2];
      a2 += sh[(jj * CB + tid) * 2 + 1];
    }
    const int cc = c0 + tid;
    S1[(long)b * C + cc] = a1;
    S2[(long)b * C + cc] = a2;
    scv = pscv;
    const float w = pgam * scv;
    wa1[tid] = w * a1;
    wa2[tid] = w * a2;
    if (T) {   // deferred gamma/beta reduction: this image's row [sc*S2 | sc*S1] for lgm_wgrad_reduce_batch
      T[(long)b * 2 * C + cc] = scv * a2;
      T[(long)b * 2 * C + C + cc] = scv * a1;
    }
  }
  __syncthreads();
  if (tid < CB) {  // coefficients of gx = P*gz + Qc + x*Rc, and the FiLM scale/shift gradients
    const int cc = c0 + tid;
    const int g0 = (tid / Cg) * Cg;
    float m1 = 0.f, m2 = 0.f;
    for (int j = 0; j < Cg; ++j) {
      m1 += wa1[g0 + j];
      m2 += wa2[g0 + j];
    }
    const float inv_n = 1.f / ((float)Cg * (float)HW);
    m1 *= inv_n;
    m2 *= inv_n;
    const float m = pm, r = pr;
    cP[tid] = r * pgam * scv;
    const float R = -r * m2;          // multiplies xhat
    cR[tid] = R * r;                  // multiplies x
    cQ[tid] = -r * m1 - m * r * R;
    if (gss) {
      float gsc = pgam * a2 + pbet * a1;
      float gsh = a1;
      if (gss_beta != 0.f) {
        gsc += gss_beta * pgsc;
        gsh += gss_beta * pgsh;
      }
      gss[(long)b * gss_pitch + cc] = gsc;
      gss[(long)b * gss_pitch + C + cc] = gsh;
    }
  }
  __syncthreads();
  const f32x4 P4 = *reinterpret_cast<const f32x4*>(&cP[q * 4]);
  const f32x4 Q4 = *reinterpret_cast<const f32x4*>(&cQ[q * 4]);
  const f32x4 R4 = *reinterpret_cast<const f32x4*>(&cR[q * 4]);
  float* ob = gx + (long)b * HW * gx_pitch + c;
  // (the previous gx values are read next to their use: holding NV more quads would spill at NV = 8)
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    f32x4 r = g[k] * P4 + Q4 + xv[k] * R4;
    if (accumulate) r += *reinterpret_cast<const f32x4*>(ob + (long)(pl + k * ppb) * gx_pitch);
    *reinterpret_cast<f32x4*>(ob + (long)(pl + k * ppb) * gx_pitch) = r;
  }
}

__global__ __launch_bounds__(256) void gn_bwd_affine_kernel(const float* __restrict__ S1, const float* __restrict__ S2,
                                                            const float* __restrict__ ss, long ss_pitch, int B, int C,
                                                            float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                            float affine_beta) {
  gn_bwd_affine_body((int)blockIdx.x, S1, S2, ss, ss_pitch, B, C, ggamma, gbeta, affine_beta);
}

int gn_cb(int C, int G) {
  const int Cg = C / G;
  int cb = Cg * ((32 + Cg - 1) / Cg);
  if (cb > C) cb = C;
  return cb;
}

int gn_check(int B, int HW, int C, int G) {
  LGM_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "groupnorm: bad sizes B=%d HW=%d C=%d G=%d", B, HW, C, G);
  LGM_REQUIRE(C % 4 == 0 && G <= 64, "groupnorm: C %% 4 != 0 or G > 64");
  const int cb = gn_cb(C, G);
  LGM_REQUIRE(cb % 4 == 0 && C % cb == 0 && cb / 4 <= 256 && cb <= 256, "groupnorm: unsupported C=%d G=%d", C, G);
  return LGM_OK;
}

}  // namespace

// Block decomposition of the GroupNorm kernels.  A block owns (image, cb channels), cb a whole number of groups:
//   * by default >= 32 channels (full 128-byte rows per pixel);
//   * when that leaves fewer than 256 blocks (small per-GPU batches: B = 16 gave 32 blocks of 1024 threads on a
//     256-CU chip, 14 us of pure latency per launch) the blocks are narrowed, down to ONE group per block;
//   * one-pass kernels: nt threads = (cb / 4) channel quads x ppb pixel lanes, nv = HW / ppb 16-byte values per thread
//     in registers; nt is lowered until the pixel lanes divide the map (a 4x4 map has 16 pixels: 256 threads at
//     cb = 32 would be 32 lanes).  nv == 0: no one-pass kernel, the two-pass kernels run with the same cb.
struct GnPlan {
  int cb, nt, nv;
};
static GnPlan gn_plan_cb(int HW, int cb, bool bwd) {
  GnPlan p;
  p.cb = cb;
  p.nt = (long)HW * cb >= 16384 ? 1024 : 256;
  p.nv = 0;
  static const bool no_fused = getenv("LGM_GN_TWO_PASS") != nullptr;   // A/B switch
  if (no_fused) return p;
  const int cand[4] = {p.nt, 256, 128, 64};
  for (int i = 0; i < 4; ++i) {
    const int nt = cand[i];
    if (i > 0 && nt >= cand[0]) continue;
    const int tq = cb / 4;
    if (nt < cb || nt % tq) continue;
    const int ppb = nt / tq;
    if (HW % ppb) continue;
    const int nv = HW / ppb;
    if (nv == 1 || nv == 2 || nv == 4 || nv == 8 || (!bwd && nv == 16 && nt == 256)) {
      p.nt = nt;
      p.nv = nv;
      return p;
    }
  }
  return p;
}

static GnPlan gn_plan(int B, int HW, int C, int G, bool bwd) {
  const int Cg = C / G;
  int cb = gn_cb(C, G);
  static const bool wide_only = getenv("LGM_GN_WIDE") != nullptr;   // A/B switch: never narrow the blocks
  // tuning knob.  8-channel (one group, 32-byte row) blocks were 2 % slower at B = 32 while their siblings sat in
  // different L2s; with gn_sibling_map they are 0.3 % faster at B = 16 / 32 and neutral at B = 64
  static const int min_cb = getenv("LGM_GN_MINCB") ? atoi(getenv("LGM_GN_MINCB")) : 8;
  // blocks a launch should reach before the narrowing stops (tuning knob; 256 = one block per CU)
  static const long want = getenv("LGM_GN_BLOCKS") ? atol(getenv("LGM_GN_BLOCKS")) : 256;
  if (!wide_only)
    while ((long)B * (C / cb) < want && cb > Cg && cb / 2 >= min_cb && (cb / 2) % Cg == 0 && (cb / 2) % 4 == 0) cb /= 2;
  GnPlan p = gn_plan_cb(HW, cb, bwd);
  // (Large maps - 64 x 64 at 64 channels: 4096 pixels x 32 channels do not fit a block's registers - stay on the two-pass
  // kernels.  Narrowing the block to ONE group so that its slice fits was tried: one pass over x instead of two, but
  // 32-byte rows per pixel; the 64 x 64 DDPM step went from 3,170 to 3,060 images/s.)
  return p;
}

// 1 when the forward runs the one-pass (register-resident) kernel for this shape, 0 when it needs two passes over x
extern "C" int64_t lgm_gn_fwd_fused_supported(int B, int HW, int C, int G) {
  if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G || C % 4 || G > 64) return 0;
  const int cb = gn_cb(C, G);
  if (!(cb % 4 == 0 && C % cb == 0 && cb / 4 <= 256 && cb <= 256)) return 0;
  return gn_plan(B, HW, C, G, false).nv > 0 ? 1 : 0;
}

extern "C" int64_t lgm_gn_planes_supported(int B, int HW, int C, int G) {
  if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G || C % 4 || G > 64) return 0;
  const int cb = gn_cb(C, G);
  if (!(cb % 4 == 0 && C % cb == 0 && cb / 4 <= 256 && cb <= 256)) return 0;
  return (gn_plan(B, HW, C, G, false).nv > 0 && gn_plan(B, HW, C, G, true).nv > 0) ? 1 : 0;   // both one-pass kernels
}

static int gn_fwd_impl(const float* x, int64_t x_pitch, int B, int HW, int C, int G, float eps,
                       const float* gamma, const float* beta, const float* ss, int64_t ss_pitch,
                       int act, const float* res, int64_t res_pitch, float* y, int64_t y_pitch,
                       float* mean, float* rstd, float* coefA, float* coefB, const float* planes, long pstride,
                       int splits, const float* cbias, void* stream) {
  if (int rc = gn_check(B, HW, C, G)) return rc;
  LGM_REQUIRE(x && gamma && beta && y && mean && rstd && coefA && coefB, "gn_fwd: null pointer");
  LGM_REQUIRE(x_pitch % 4 == 0 && y_pitch % 4 == 0 && (!res || res_pitch % 4 == 0), "gn_fwd: pitch %% 4 != 0");
  hipStream_t s = (hipStream_t)stream;
  const GnPlan pln = gn_plan(B, HW, C, G, false);
  const int cb = pln.cb;
  static const bool small_only = getenv("LGM_GN_256") != nullptr;   // A/B switch
  if (pln.nv > 0) {   // one-pass kernel: the block's slice fits its registers (NV 16-byte values per thread)
    const int nt = pln.nt, nv = pln.nv;
#define GN_FUSED(NTV, NVV)                                                                                            \
  hipLaunchKernelGGL((gn_fused_fwd_kernel<NTV, NVV>), dim3(B * (C / cb)), dim3(NTV), 0, s, x, (long)x_pitch, HW, C, G, \
                     cb, eps, gamma, beta, ss, (long)ss_pitch, act, res, (long)res_pitch, y, (long)y_pitch, mean,    \
                     rstd, coefA, coefB, planes, pstride, splits, cbias, (float*)x)
#define GN_FUSED_NV(NTV)                                                                       \
  do {                                                                                         \
    if (nv == 1) GN_FUSED(NTV, 1); else if (nv == 2) GN_FUSED(NTV, 2); else if (nv == 4) GN_FUSED(NTV, 4); \
    else GN_FUSED(NTV, 8);                                                                     \
  } while (0)
    if (nt == 1024) GN_FUSED_NV(1024);
    else if (nt == 256) { if (nv == 16) GN_FUSED(256, 16); else GN_FUSED_NV(256); }
    else if (nt == 128) GN_FUSED_NV(128);
    else GN_FUSED_NV(64);
#undef GN_FUSED_NV
#undef GN_FUSED
    LGM_LAUNCH_CHECK();
    return LGM_OK;
  }
  LGM_REQUIRE(!planes, "gn_fwd_planes: this shape has no one-pass kernel (ask lgm_gn_planes_supported first)");
  if (!small_only && (long)HW * cb >= 16384)
    hipLaunchKernelGGL(gn_stats_kernel<1024>, dim3(B * (C / cb)), dim3(1024), 0, s, x, (long)x_pitch, HW, C, G, cb, eps,
                       gamma, beta, ss, (long)ss_pitch, mean, rstd, coefA, coefB);
  else
    hipLaunchKernelGGL(gn_stats_kernel<256>, dim3(B * (C / cb)), dim3(256), 0, s, x, (long)x_pitch, HW, C, G, cb, eps,
                       gamma, beta, ss, (long)ss_pitch, mean, rstd, coefA, coefB);
  const long npix = (long)B * HW;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(lgm_cdiv(npix * (C / 4), 256)), dim3(256), 0, s, x, (long)x_pitch, coefA,
                     coefB, res, (long)res_pitch, y, (long)y_pitch, npix, HW, C, act);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

// GroupNorm forward whose statistics come from the producing convolution's epilogue (lgm_conv3x3_wino4_stats): per image
// `parts` rows of (sum, sum of squares) per channel of the PRE-BIAS outputs, layout [b * parts + q][2][C].  One block per
// (image, group) adds them in float64 in a fixed order, shifts by the convolution bias, and leaves mean / rstd / the
// per-(image, channel) coefficients exactly as gn_stats_kernel does; gn_apply_kernel follows.  x is read ONCE.
__global__ __launch_bounds__(64) void gn_coef_from_stats_kernel(const float* __restrict__ stats, int parts, int C, int G,
                                                                long HW, const float* __restrict__ cbias, float eps,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ ss,
                                                                long ss_pitch, float* __restrict__ mean,
                                                                float* __restrict__ rstd, float* __restrict__ A,
                                                                float* __restrict__ Bc, const float* __restrict__ x,
                                                                long x_pitch) {
  __shared__ double sh1[64], sh2[64];
  __shared__ float smean, srstd;
  __shared__ int sredo;
  const int Cg = C / G;                    // host: Cg divides 64
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int t = threadIdx.x, c = t % Cg, j = t / Cg, J = 64 / Cg;
  const float* base = stats + ((long)b * parts) * 2 * C + g * Cg + c;
  double a1 = 0.0, a2 = 0.0;
  for (int q = j; q < parts; q += J) {
    a1 += (double)base[(long)q * 2 * C];
    a2 += (double)base[(long)q * 2 * C + C];
  }
  sh1[t] = a1;
  sh2[t] = a2;
  __syncthreads();
  if (t == 0) {
    const double n = (double)HW;
    double sx = 0.0, sxx = 0.0, pre2 = 0.0;
    for (int cc = 0; cc < Cg; ++cc) {
      double t1 = 0.0, t2 = 0.0;
      for (int jj = 0; jj < J; ++jj) {
        t1 += sh1[jj * Cg + cc];
        t2 += sh2[jj * Cg + cc];
      }
      const double bb = cbias ? (double)cbias[g * Cg + cc] : 0.0;
      sx += t1 + n * bb;
      sxx += t2 + 2.0 * bb * t1 + n * bb * bb;
      pre2 += t2 + 2.0 * fabs(bb * t1);       // what the fp32 rows' rounding errors scale with
    }
    const double cnt = n * (double)Cg;
    const double m = sx / cnt;
    double var = sxx / cnt - m * m;
    if (var < 0.0) var = 0.0;
    smean = (float)m;
    srstd = (float)(1.0 / sqrt(var + (double)eps));
    // The rows are fp32 sums (relative error ~1e-7 each): E[y^2] - mean^2 of the PRE-BIAS values loses what their mean
    // square exceeds the variance by.  Beyond a ratio of 1e3 (rstd would be off by > 5e-5; never seen in the UNet, whose
    // pre-bias outputs are zero-mean-ish) the block measures its slice of x itself, two passes in float64.
    sredo = (pre2 / cnt > 1e3 * (var + (double)eps)) ? 1 : 0;
  }
  __syncthreads();
  if (sredo) {                              // block-uniform
    const long img = (long)b * HW;
    double a = 0.0;
    for (long p = t; p < HW; p += 64) {
      const float* row = x + (img + p) * x_pitch + g * Cg;
      for (int cc = 0; cc < Cg; ++cc) a += (double)row[cc];
    }
    sh1[t] = a;
    __syncthreads();
    double m = 0.0;
    for (int i = 0; i < 64; ++i) m += sh1[i];
    m /= (double)HW * (double)Cg;
    double v = 0.0;
    for (long p = t; p < HW; p += 64) {
      const float* row = x + (img + p) * x_pitch + g * Cg;
      for (int cc = 0; cc < Cg; ++cc) {
        const double d = (double)row[cc] - m;
        v += d * d;
      }
    }
    sh2[t] = v;
    __syncthreads();
    if (t == 0) {
      double vv = 0.0;
      for (int i = 0; i < 64; ++i) vv += sh2[i];
      vv /= (double)HW * (double)Cg;
      smean = (float)m;
      srstd = (float)(1.0 / sqrt(vv + (double)eps));
    }
    __syncthreads();
  }
  if (t == 0) {
    mean[b * G + g] = smean;
    rstd[b * G + g] = srstd;
  }
  if (t < Cg) {
    const int ch = g * Cg + t;
    float a = srstd * gamma[ch];
    float bb = beta[ch] - smean * a;
    if (ss) {
      const float sc = ss[(long)b * ss_pitch + ch] + 1.f;
      const float shf = ss[(long)b * ss_pitch + C + ch];
      a *= sc;
      bb = bb * sc + shf;
    }
    A[(long)b * C + ch] = a;
    Bc[(long)b * C + ch] = bb;
  }
}

extern "C" int lgm_gn_fwd_stats(const float* stats, int parts_per_image, const float* conv_bias, const float* x,
                                int64_t x_pitch, int B, int HW, int C, int G, float eps, const float* gamma,
                                const float* beta, const float* ss, int64_t ss_pitch, int act, const float* res,
                                int64_t res_pitch, float* y, int64_t y_pitch, float* mean, float* rstd, float* coefA,
                                float* coefB, void* stream) {
  if (int rc = gn_check(B, HW, C, G)) return rc;
  LGM_REQUIRE(stats && parts_per_image > 0 && x && gamma && beta && y && mean && rstd && coefA && coefB,
              "gn_fwd_stats: null pointer");
  LGM_REQUIRE(x_pitch % 4 == 0 && y_pitch % 4 == 0 && (!res || res_pitch % 4 == 0) && C % 4 == 0 && 64 % (C / G) == 0,
              "gn_fwd_stats: pitch %% 4 != 0 or a group width that does not divide 64");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_coef_from_stats_kernel, dim3(B * G), dim3(64), 0, s, stats, parts_per_image, C, G, (long)HW, conv_bias,
                     eps, gamma, beta, ss, (long)ss_pitch, mean, rstd, coefA, coefB, x, (long)x_pitch);
  const long npix = (long)B * HW;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(lgm_cdiv(npix * (C / 4), 256)), dim3(256), 0, s, x, (long)x_pitch, coefA, coefB,
                     res, (long)res_pitch, y, (long)y_pitch, npix, HW, C, act);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_gn_fwd(const float* x, int64_t x_pitch, int B, int HW, int C, int G, float eps,
                          const float* gamma, const float* beta, const float* ss, int64_t ss_pitch,
                          int act, const float* res, int64_t res_pitch, float* y, int64_t y_pitch,
                          float* mean, float* rstd, float* coefA, float* coefB, void* stream) {
  return gn_fwd_impl(x, x_pitch, B, HW, C, G, eps, gamma, beta, ss, ss_pitch, act, res, res_pitch, y, y_pitch, mean, rstd,
                     coefA, coefB, nullptr, 0, 0, nullptr, stream);
}

extern "C" int lgm_gn_fwd_planes(const float* planes, int64_t plane_stride, int splits, const float* conv_bias,
                                 float* x, int64_t x_pitch, int B, int HW, int C, int G, float eps,
                                 const float* gamma, const float* beta, const float* ss, int64_t ss_pitch,
                                 int act, const float* res, int64_t res_pitch, float* y, int64_t y_pitch,
                                 float* mean, float* rstd, float* coefA, float* coefB, void* stream) {
  LGM_REQUIRE(planes && splits >= 1 && plane_stride >= (int64_t)B * HW * C && lgm_aligned16(planes) &&
                  plane_stride % 4 == 0 && (!conv_bias || lgm_aligned16(conv_bias)),
              "gn_fwd_planes: bad partial planes");
  return gn_fwd_impl(x, x_pitch, B, HW, C, G, eps, gamma, beta, ss, ss_pitch, act, res, res_pitch, y, y_pitch, mean, rstd,
                     coefA, coefB, planes, (long)plane_stride, splits, conv_bias, stream);
}

static int gn_bwd_impl(const float* x, int64_t x_pitch, const float* gy, int64_t gy_pitch, int B, int HW,
                       int C, int G, const float* gamma, const float* beta, const float* ss,
                       int64_t ss_pitch, int act, const float* mean, const float* rstd,
                       const float* coefA, const float* coefB, float* gx, int64_t gx_pitch,
                       int accumulate_gx, float* ggamma, float* gbeta, float affine_beta, float* gss,
                       int64_t gss_pitch, float gss_beta, float* workspace, float* rows, int64_t* desc,
                       void* stream, const float* planes = nullptr, long pstride = 0, int splits = 0,
                       float* add_out = nullptr, long add_pitch = 0) {
  if (desc) desc[6] = 0;      // nothing deferred unless the one-pass kernel below takes it
  if (int rc = gn_check(B, HW, C, G)) return rc;
  LGM_REQUIRE(x && (gy || planes) && gamma && beta && mean && rstd && coefA && coefB && gx && ggamma && gbeta && workspace,
              "gn_bwd: null pointer");
  LGM_REQUIRE(x_pitch % 4 == 0 && gy_pitch % 4 == 0 && gx_pitch % 4 == 0, "gn_bwd: pitch %% 4 != 0");
  hipStream_t s = (hipStream_t)stream;
  const long bc = (long)B * C;
  float* S1 = workspace;
  float* S2 = S1 + bc;
  float* P = S2 + bc;
  float* Qc = P + bc;
  float* Rc = Qc + bc;
  const GnPlan pln = gn_plan(B, HW, C, G, true);
  const int cb = pln.cb;
  static const bool small_only = getenv("LGM_GN_256") != nullptr;
  {   // one-pass kernel when the block's x and gy slices fit its registers
    const int nt = pln.nt, nv = pln.nv;
    if (nv > 0) {
#define GN_FUSED(NTV, NVV)                                                                                             \
  hipLaunchKernelGGL((gn_fused_bwd_kernel<NTV, NVV>), dim3(B * (C / cb)), dim3(NTV), 0, s, x, (long)x_pitch, gy,        \
                     (long)gy_pitch, coefA, coefB, mean, rstd, HW, C, G, cb, act, gamma, beta, ss, (long)ss_pitch, gss, \
                     (long)gss_pitch, gss_beta, S1, S2, gx, (long)gx_pitch, accumulate_gx, rows, planes, pstride, splits, \
                     add_out, add_pitch)
#define GN_FUSED_NV(NTV)                                                                       \
  do {                                                                                         \
    if (nv == 1) GN_FUSED(NTV, 1); else if (nv == 2) GN_FUSED(NTV, 2); else if (nv == 4) GN_FUSED(NTV, 4); \
    else GN_FUSED(NTV, 8);                                                                     \
  } while (0)
      if (nt == 1024) GN_FUSED_NV(1024);
      else if (nt == 256) GN_FUSED_NV(256);
      else if (nt == 128) GN_FUSED_NV(128);
      else GN_FUSED_NV(64);
#undef GN_FUSED_NV
#undef GN_FUSED
      if (rows && desc) {   // the caller sums the per-image rows of many layers with ONE lgm_wgrad_reduce_batch launch
        union { float f; int64_t i; } bb;
        bb.i = 0;
        bb.f = affine_beta;
        desc[0] = (int64_t)(uintptr_t)rows; desc[1] = 2L * C; desc[2] = (int64_t)(uintptr_t)ggamma; desc[3] = C;
        desc[4] = (int64_t)(uintptr_t)gbeta; desc[5] = C; desc[6] = B; desc[7] = bb.i;
      } else {
        hipLaunchKernelGGL(gn_bwd_affine_kernel, dim3(lgm_cdiv(C, 16)), dim3(256), 0, s, (const float*)S1,
                           (const float*)S2, ss, (long)ss_pitch, B, C, ggamma, gbeta, affine_beta);
      }
      LGM_LAUNCH_CHECK();
      return LGM_OK;
    }
  }
  LGM_REQUIRE(!planes, "gn_bwd_planes: this shape has no one-pass kernel (ask lgm_gn_planes_supported first)");
  if (!small_only && (long)HW * cb >= 16384)
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<1024>, dim3(B * (C / cb)), dim3(1024), 0, s, x, (long)x_pitch, gy,
                       (long)gy_pitch, coefA, coefB, mean, rstd, HW, C, G, cb, act, gamma, beta, ss, (long)ss_pitch, gss,
                       (long)gss_pitch, gss_beta, S1, S2, P, Qc, Rc);
  else
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<256>, dim3(B * (C / cb)), dim3(256), 0, s, x, (long)x_pitch, gy,
                       (long)gy_pitch, coefA, coefB, mean, rstd, HW, C, G, cb, act, gamma, beta, ss, (long)ss_pitch, gss,
                       (long)gss_pitch, gss_beta, S1, S2, P, Qc, Rc);
  const long npix = (long)B * HW;
  const int apply_blocks = lgm_cdiv(npix * (C / 4), 256);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(apply_blocks + lgm_cdiv(C, 16)), dim3(256), 0, s, x, (long)x_pitch, gy,
                     (long)gy_pitch, coefA, coefB, P, Qc, Rc, gx, (long)gx_pitch, npix, HW, C, act, accumulate_gx,
                     apply_blocks, (const float*)S1, (const float*)S2, ss, (long)ss_pitch, B, ggamma, gbeta,
                     affine_beta, add_out, add_pitch);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_gn_bwd(const float* x, int64_t x_pitch, const float* gy, int64_t gy_pitch, int B, int HW,
                          int C, int G, const float* gamma, const float* beta, const float* ss,
                          int64_t ss_pitch, int act, const float* mean, const float* rstd,
                          const float* coefA, const float* coefB, float* gx, int64_t gx_pitch,
                          int accumulate_gx, float* ggamma, float* gbeta, float affine_beta, float* gss,
                          int64_t gss_pitch, float gss_beta, float* workspace, void* stream) {
  return gn_bwd_impl(x, x_pitch, gy, gy_pitch, B, HW, C, G, gamma, beta, ss, ss_pitch, act, mean, rstd, coefA, coefB, gx,
                     gx_pitch, accumulate_gx, ggamma, gbeta, affine_beta, gss, gss_pitch, gss_beta, workspace, nullptr,
                     nullptr, stream);
}

extern "C" int lgm_gn_bwd_deferred(const float* x, int64_t x_pitch, const float* gy, int64_t gy_pitch, int B, int HW,
                                   int C, int G, const float* gamma, const float* beta, const float* ss,
                                   int64_t ss_pitch, int act, const float* mean, const float* rstd,
                                   const float* coefA, const float* coefB, float* gx, int64_t gx_pitch,
                                   int accumulate_gx, float* ggamma, float* gbeta, float affine_beta, float* gss,
                                   int64_t gss_pitch, float gss_beta, float* workspace, float* rows, int64_t* desc,
                                   void* stream) {
  LGM_REQUIRE(rows && desc && lgm_aligned16(rows) && lgm_aligned16(ggamma) && lgm_aligned16(gbeta),
              "gn_bwd_deferred: rows / descriptor missing or gradients not 16-byte aligned");
  return gn_bwd_impl(x, x_pitch, gy, gy_pitch, B, HW, C, G, gamma, beta, ss, ss_pitch, act, mean, rstd, coefA, coefB, gx,
                     gx_pitch, accumulate_gx, ggamma, gbeta, affine_beta, gss, gss_pitch, gss_beta, workspace, rows, desc,
                     stream);
}

/* lgm_gn_bwd / lgm_gn_bwd_deferred (rows and desc both NULL or both given) that ALSO adds gy to a second tensor:
 * add_out[b, p, c] += gy[b, p, c].  The backward of ResnetBlock's identity residual (reference ddpm.py:187,200:
 * `return h + self.res_conv(x)` with res_conv = nn.Identity) when the block's input gradient is accumulated into a tensor
 * that already holds another branch's gradient: gy is being read here anyway. */
extern "C" int lgm_gn_bwd_add(const float* x, int64_t x_pitch, const float* gy, int64_t gy_pitch, int B, int HW,
                              int C, int G, const float* gamma, const float* beta, const float* ss,
                              int64_t ss_pitch, int act, const float* mean, const float* rstd,
                              const float* coefA, const float* coefB, float* gx, int64_t gx_pitch,
                              int accumulate_gx, float* ggamma, float* gbeta, float affine_beta, float* gss,
                              int64_t gss_pitch, float gss_beta, float* workspace, float* rows, int64_t* desc,
                              float* add_out, int64_t add_pitch, void* stream) {
  LGM_REQUIRE((!rows && !desc) || (rows && desc && lgm_aligned16(rows) && lgm_aligned16(ggamma) && lgm_aligned16(gbeta)),
              "gn_bwd_add: rows / descriptor must come together, gradients 16-byte aligned");
  LGM_REQUIRE(gy && add_out && lgm_aligned16(add_out) && add_pitch % 4 == 0 && add_pitch >= C && add_out != gx &&
                  (const float*)add_out != gy,
              "gn_bwd_add: the second output must be a 16-byte aligned tensor of its own (pitch %% 4 == 0, >= C)");
  return gn_bwd_impl(x, x_pitch, gy, gy_pitch, B, HW, C, G, gamma, beta, ss, ss_pitch, act, mean, rstd, coefA, coefB, gx,
                     gx_pitch, accumulate_gx, ggamma, gbeta, affine_beta, gss, gss_pitch, gss_beta, workspace, rows, desc,
                     stream, nullptr, 0, 0, add_out, (long)add_pitch);
}

/* gy given as the split-K partial planes of the producing input-gradient convolution (see lgm_gn_fwd_planes);
 * deferred gamma / beta rows as lgm_gn_bwd_deferred when rows / desc are given */
extern "C" int lgm_gn_bwd_planes(const float* x, int64_t x_pitch, const float* gy_planes, int64_t plane_stride, int splits,
                                 int B, int HW, int C, int G, const float* gamma, const float* beta, const float* ss,
                                 int64_t ss_pitch, int act, const float* mean, const float* rstd,
                                 const float* coefA, const float* coefB, float* gx, int64_t gx_pitch,
                                 int accumulate_gx, float* ggamma, float* gbeta, float affine_beta, float* gss,
                                 int64_t gss_pitch, float gss_beta, float* workspace, float* rows, int64_t* desc,
                                 void* stream) {
  LGM_REQUIRE(gy_planes && splits >= 1 && plane_stride >= (int64_t)B * HW * C && lgm_aligned16(gy_planes) &&
                  plane_stride % 4 == 0, "gn_bwd_planes: bad partial planes");
  LGM_REQUIRE((!rows && !desc) || (rows && desc && lgm_aligned16(rows) && lgm_aligned16(ggamma) && lgm_aligned16(gbeta)),
              "gn_bwd_planes: rows / descriptor must come together, gradients 16-byte aligned");
  return gn_bwd_impl(x, x_pitch, nullptr, C, B, HW, C, G, gamma, beta, ss, ss_pitch, act, mean, rstd, coefA, coefB, gx,
                     gx_pitch, accumulate_gx, ggamma, gbeta, affine_beta, gss, gss_pitch, gss_beta, workspace, rows, desc,
                     stream, gy_planes, (long)plane_stride, splits);
}

// =====================================================================================
// RMSNorm over channels, one sub-wave group of L = min(64, C/4) lanes per pixel.
// y = x / max(||x||, 1e-12) * g * sqrt(C)  (+ res)
// =====================================================================================
namespace {

constexpr int RMS_MAXQ = 4;  // quads per lane (C <= 1024)

template <int Q>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const float* __restrict__ x, long x_pitch,
                                                          const float* __restrict__ g, const float* __restrict__ res,
                                                          long res_pitch, float* __restrict__ y, long y_pitch,
                                                          long npix, int C, int L) {
  const int ppb = 256 / L;  // pixels per block pass
  const int tid = threadIdx.x;
  const int l = tid % L, pg = tid / L;
  const float sqrtc = sqrtf((float)C);
  f32x4 gv[Q];
#pragma unroll
  for (int k = 0; k < Q; ++k) gv[k] = *reinterpret_cast<const f32x4*>(g + (l + k * L) * 4);
  for (long pix = (long)blockIdx.x * ppb + pg; pix < npix; pix += (long)gridDim.x * ppb) {
    f32x4 xv[Q];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      xv[k] = *reinterpret_cast<const f32x4*>(x + pix * x_pitch + (l + k * L) * 4);
      ss += xv[k][0] * xv[k][0] + xv[k][1] * xv[k][1] + xv[k][2] * xv[k][2] + xv[k][3] * xv[k][3];
    }
    for (int off = L >> 1; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float inv = sqrtc / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      f32x4 o = xv[k] * gv[k] * inv;
      if (res) o += *reinterpret_cast<const f32x4*>(res + pix * res_pitch + (l + k * L) * 4);
      *reinterpret_cast<f32x4*>(y + pix * y_pitch + (l + k * L) * 4) = o;
    }
  }
}

// backward: gx = (gxh - xh * dot(xh, gxh)) / n,  gxh = gy * g * sqrt(C);  gg partial per block.
template <int Q>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const float* __restrict__ x, long x_pitch,
                                                          const float* __restrict__ gy, long gy_pitch,
                                                          const float* __restrict__ g, float* __restrict__ gx,
                                                          long gx_pitch, int accumulate, const float* __restrict__ res,
                                                          long res_pitch, long npix, int C, int L,
                                                          float* __restrict__ gg_partial) {
  __shared__ float sh[256 * 4 * Q];
  const int ppb = 256 / L;
  const int tid = threadIdx.x;
  const int l = tid % L, pg = tid / L;
  const float sqrtc = sqrtf((float)C);
  f32x4 gv[Q], gacc[Q];
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    gv[k] = *reinterpret_cast<const f32x4*>(g + (l + k * L) * 4);
    gacc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (long pix = (long)blockIdx.x * ppb + pg; pix < npix; pix += (long)gridDim.x * ppb) {
    f32x4 xv[Q], gv_y[Q];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      xv[k] = *reinterpret_cast<const f32x4*>(x + pix * x_pitch + (l + k * L) * 4);
      gv_y[k] = *reinterpret_cast<const f32x4*>(gy + pix * gy_pitch + (l + k * L) * 4);
      ss += xv[k][0] * xv[k][0] + xv[k][1] * xv[k][1] + xv[k][2] * xv[k][2] + xv[k][3] * xv[k][3];
    }
    for (int off = L >> 1; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float nrm = sqrtf(ss);
    const bool clamped = nrm < 1e-12f;
    const float invn = 1.f / fmaxf(nrm, 1e-12f);
    float dot = 0.f;
    f32x4 gxh[Q], xh[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      xh[k] = xv[k] * invn;
      gxh[k] = gv_y[k] * gv[k] * sqrtc;
      gacc[k] += gv_y[k] * xh[k] * sqrtc;
      dot += xh[k][0] * gxh[k][0] + xh[k][1] * gxh[k][1] + xh[k][2] * gxh[k][2] + xh[k][3] * gxh[k][3];
    }
    for (int off = L >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
    if (clamped) dot = 0.f;  // x / eps is linear in x below the clamp
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      f32x4 o = (gxh[k] - xh[k] * dot) * invn;
      float* dst = gx + pix * gx_pitch + (l + k * L) * 4;
      if (accumulate) o += *reinterpret_cast<const f32x4*>(dst);
      if (res) o += *reinterpret_cast<const f32x4*>(res + pix * res_pitch + (l + k * L) * 4);
      *reinterpret_cast<f32x4*>(dst) = o;
    }
  }
  // combine the per-thread g-gradients over the pixel groups of this block (fixed order)
#pragma unroll
  for (int k = 0; k < Q; ++k) *reinterpret_cast<f32x4*>(&sh[(tid * Q + k) * 4]) = gacc[k];
  __syncthreads();
  if (tid < L) {
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      for (int p = 0; p < ppb; ++p) a += *reinterpret_cast<const f32x4*>(&sh[((p * L + tid) * Q + k) * 4]);
      *reinterpret_cast<f32x4*>(gg_partial + (long)blockIdx.x * C + (tid + k * L) * 4) = a;
    }
  }
}

int rms_plan(int C, int* L, int* Q) {
  LGM_REQUIRE(C % 4 == 0 && C >= 4 && C <= 1024, "rmsnorm: unsupported C=%d", C);
  const int c4 = C / 4;
  LGM_REQUIRE((c4 & (c4 - 1)) == 0, "rmsnorm: C/4 must be a power of two (C=%d)", C);
  *L = c4 < 64 ? c4 : 64;
  *Q = c4 / *L;
  return LGM_OK;
}

int rms_blocks(long npix, int L) {
  const int ppb = 256 / L;
  long nb = (npix + ppb - 1) / ppb;
  if (nb > 1024) nb = 1024;
  return (int)nb;
}

}  // namespace

extern "C" int lgm_rmsnorm_fwd(const float* x, int64_t x_pitch, const float* g, const float* res,
                               int64_t res_pitch, float* y, int64_t y_pitch, int64_t npix, int C, void* stream) {
  int L, Q;
  if (int rc = rms_plan(C, &L, &Q)) return rc;
  LGM_REQUIRE(x && g && y && npix > 0, "rmsnorm_fwd: null pointer / empty");
  LGM_REQUIRE(x_pitch % 4 == 0 && y_pitch % 4 == 0 && (!res || res_pitch % 4 == 0), "rmsnorm_fwd: pitch %% 4 != 0");
  hipStream_t s = (hipStream_t)stream;
  const int nb = rms_blocks(npix, L);
#define RMS_FWD(QQ)                                                                                             \
  hipLaunchKernelGGL(rmsnorm_fwd_kernel<QQ>, dim3(nb), dim3(256), 0, s, x, (long)x_pitch, g, res, (long)res_pitch, y, \
                     (long)y_pitch, (long)npix, C, L)
  if (Q == 1) RMS_FWD(1); else if (Q == 2) RMS_FWD(2); else RMS_FWD(4);
#undef RMS_FWD
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int64_t lgm_rmsnorm_bwd_workspace(int64_t npix, int C) {
  int L, Q;
  if (rms_plan(C, &L, &Q)) return -1;
  return (int64_t)rms_blocks(npix, L) * C * (int64_t)sizeof(float) + lgm_colsum_workspace(rms_blocks(npix, L), C);
}

static int rmsnorm_bwd_impl(const float* x, int64_t x_pitch, const float* gy, int64_t gy_pitch, const float* g,
                            float* gx, int64_t gx_pitch, int accumulate_gx, const float* res, int64_t res_pitch,
                            float* gg, float gg_beta, int64_t npix, int C, void* workspace, int64_t* desc,
                            void* stream) {
  int L, Q;
  if (int rc = rms_plan(C, &L, &Q)) return rc;
  LGM_REQUIRE(x && gy && g && gx && gg && workspace && npix > 0, "rmsnorm_bwd: null pointer / empty");
  LGM_REQUIRE(x_pitch % 4 == 0 && gy_pitch % 4 == 0 && gx_pitch % 4 == 0 && (!res || (res_pitch % 4 == 0 && lgm_aligned16(res))),
              "rmsnorm_bwd: pitch %% 4 != 0");
  hipStream_t s = (hipStream_t)stream;
  const int nb = rms_blocks(npix, L);
  float* partial = (float*)workspace;
#define RMS_BWD(QQ)                                                                                              \
  hipLaunchKernelGGL(rmsnorm_bwd_kernel<QQ>, dim3(nb), dim3(256), 0, s, x, (long)x_pitch, gy, (long)gy_pitch, g, gx, \
                     (long)gx_pitch, accumulate_gx, res, (long)res_pitch, (long)npix, C, L, partial)
  if (Q == 1) RMS_BWD(1); else if (Q == 2) RMS_BWD(2); else RMS_BWD(4);
#undef RMS_BWD
  LGM_LAUNCH_CHECK();
  if (desc) {   // deferred: the per-block partial rows are summed later by lgm_wgrad_reduce_batch (rows = "splits")
    union { float f; int64_t i; } bb;
    bb.i = 0;
    bb.f = gg_beta;
    desc[0] = (int64_t)(uintptr_t)partial; desc[1] = C; desc[2] = (int64_t)(uintptr_t)gg; desc[3] = C;
    desc[4] = 0; desc[5] = 0; desc[6] = nb; desc[7] = bb.i;
    return LGM_OK;
  }
  return lgm_colsum(partial, C, nb, C, gg, gg_beta, partial + (long)nb * C, stream);
}

extern "C" int lgm_rmsnorm_bwd(const float* x, int64_t x_pitch, const float* gy, int64_t gy_pitch,
                               const float* g, float* gx, int64_t gx_pitch, int accumulate_gx, const float* res,
                               int64_t res_pitch, float* gg, float gg_beta, int64_t npix, int C, void* workspace,
                               void* stream) {
  return rmsnorm_bwd_impl(x, x_pitch, gy, gy_pitch, g, gx, gx_pitch, accumulate_gx, res, res_pitch, gg, gg_beta, npix, C,
                          workspace, nullptr, stream);
}

extern "C" int lgm_rmsnorm_bwd_deferred(const float* x, int64_t x_pitch, const float* gy, int64_t gy_pitch,
                                        const float* g, float* gx, int64_t gx_pitch, int accumulate_gx,
                                        const float* res, int64_t res_pitch, float* gg, float gg_beta, int64_t npix,
                                        int C, void* workspace, int64_t* desc, void* stream) {
  LGM_REQUIRE(desc && C % 4 == 0 && lgm_aligned16(gg) && lgm_aligned16(workspace), "rmsnorm_bwd_deferred: bad arguments");
  return rmsnorm_bwd_impl(x, x_pitch, gy, gy_pitch, g, gx, gx_pitch, accumulate_gx, res, res_pitch, gg, gg_beta, npix, C,
                          workspace, desc, stream);
}
