// Vector-quantiser kernels (VQ-VAE): nearest-codebook search, deterministic segmented sums,
// EMA codebook update, gather + losses.  Reference: models/modules/vector_quantizer.py
//   _quantize :45-69  (dist = ||x||^2 + ||e||^2 - (2x).e, argmin -> int64 indices, embedding lookup)
//   losses :71-78, perplexity :80-88, STE :90-93, EMA update :128-147.
// The [N, K] distance and one-hot matrices of the reference are never materialised.
#include "lgm_common.h"

namespace {

constexpr int VQ_MAXD = 128;

// One team of 8 lanes per latent row; each lane scans K/8 codes (strided) with a sequential
// fp32 FMA chain over d, then the team reduces (dist, index) with lowest-index tie-break.
template <int D>
__global__ __launch_bounds__(256) void vq_assign_kernel(const float* __restrict__ x, long x_pitch,
                                                        const float* __restrict__ cb, int N, int K,
                                                        int64_t* __restrict__ idx_out, float* __restrict__ dist_out) {
  extern __shared__ float e2s[];   // ||e_k||^2
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(cb[(long)k * D + d], cb[(long)k * D + d], s);   // explicit: both searches agree bit for bit
    e2s[k] = s;
  }
  __syncthreads();
  const int team = threadIdx.x >> 3, tl = threadIdx.x & 7;
  const int row = blockIdx.x * 32 + team;
  if (row >= N) return;   // whole team exits together (row is per team)
  float xv[D];
  float x2 = 0.f;
  const float* xp = x + (long)row * x_pitch;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    xv[d] = xp[d];
    x2 = fmaf(xv[d], xv[d], x2);
    xv[d] *= 2.f;           // reference: - 2 * flat @ W.T  ==  (2 flat) @ W.T
  }
  float best = INFINITY;
  int bi = 0x7fffffff;
  for (int k = tl; k < K; k += 8) {
    const float* ep = cb + (long)k * D;
    float dot = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) dot = fmaf(xv[d], ep[d], dot);
    const float dist = (x2 + e2s[k]) - dot;
    if (dist < best) {      // ascending k per lane => first minimum kept
      best = dist;
      bi = k;
    }
  }
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ob < best || (ob == best && oi < bi)) {
      best = ob;
      bi = oi;
    }
  }
  if (tl == 0) {
    idx_out[row] = (int64_t)bi;
    if (dist_out) dist_out[row] = best;
  }
}

// The same search with the roles turned round: a LANE owns a code (its D floats stay in registers for the whole
// kernel), the block walks its rows, whose 2x values every lane reads from LDS as a broadcast.  The row-major
// version above streams the codebook through every team (64 scalar loads per code per lane) and ran at 5 % of the
// fp32 FMA rate; this one reads x once and the codebook once per block.  Arithmetic is IDENTICAL per (row, code) -
// the same sequential FMA chain over d, the same (x2 + e2) - dot, lowest index on ties - so indices and distances
// are bit-identical to the kernel above (tested).  Four rows at a time = four independent chains per lane.
// block = K threads (K a multiple of 64, <= 512: the code's D floats + four chains must fit 256 registers); LDS: rows 2x values, x2, per-wave partial minima.
template <int D>
__global__ __launch_bounds__(512) void vq_assign_bycode_kernel(const float* __restrict__ x, long x_pitch,
                                                                const float* __restrict__ cb, int N, int K, int rpb,
                                                                int64_t* __restrict__ idx_out,
                                                                float* __restrict__ dist_out) {
  extern __shared__ __align__(16) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int r0 = blockIdx.x * rpb, nrows = min(rpb, N - r0);
  if (nrows <= 0) return;
  float* xs = sm;                                   // [rpb][D]  (2 x)
  float* x2s = xs + rpb * D;                        // [rpb]
  float* redd = x2s + rpb;                          // [rpb][nwaves] best distance per wave
  int* redi = reinterpret_cast<int*>(redd + rpb * nwaves);
  // this lane's code
  float e[D];
  float e2 = 0.f;
  {
    const float* ep = cb + (long)tid * D;
#pragma unroll
    for (int d = 0; d < D; ++d) e[d] = ep[d];
    for (int d = 0; d < D; ++d) e2 = fmaf(e[d], e[d], e2);
  }
  // rows of the block: x2 and 2x exactly as the row-major kernel forms them
  for (int r = tid; r < nrows; r += blockDim.x) {
    const float* xp = x + (long)(r0 + r) * x_pitch;
    float x2 = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float v = xp[d];
      x2 = fmaf(v, v, x2);
      xs[r * D + d] = v * 2.f;
    }
    x2s[r] = x2;
  }
  __syncthreads();
  for (int rb = 0; rb < nrows; rb += 4) {
    float dot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < D; d += 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = rb + q < nrows ? rb + q : rb;          // clamp: the extra chains are discarded
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + r * D + d);   // broadcast read
#pragma unroll
        for (int j = 0; j < 4; ++j) dot[q] = fmaf(xv[j], e[d + j], dot[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (rb + q >= nrows) break;
      float best = (x2s[rb + q] + e2) - dot[q];
      int bi = tid;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ob < best || (ob == best && oi < bi)) {
          best = ob;
          bi = oi;
        }
      }
      if (lane == 0) {
        redd[(rb + q) * nwaves + wave] = best;
        redi[(rb + q) * nwaves + wave] = bi;
      }
    }
  }
  __syncthreads();
  for (int r = tid; r < nrows; r += blockDim.x) {
    float best = redd[r * nwaves];
    int bi = redi[r * nwaves];
    for (int w = 1; w < nwaves; ++w) {
      const float ob = redd[r * nwaves + w];
      const int oi = redi[r * nwaves + w];
      if (ob < best || (ob == best && oi < bi)) {
        best = ob;
        bi = oi;
      }
    }
    idx_out[r0 + r] = (int64_t)bi;
    if (dist_out) dist_out[r0 + r] = best;
  }
}

// Segmented sums by code: dw[k][:] = sum_{rows with idx == k} x[row][:], counts[k] = #rows.
// One block of VQ_SEG_WAVES waves per code (D <= 128: two floats per lane).  Wave w scans the w-th contiguous
// slice of the rows in ascending order, eight matching rows in flight at a time (a collapsed codebook puts
// every row on one code: the row loads of that block are then a latency chain, so they are issued in
// batches); the slices are combined in ascending order through LDS, so the result does not depend on timing.
constexpr int VQ_SEG_WAVES = 16;
__global__ __launch_bounds__(64 * VQ_SEG_WAVES) void vq_segment_sum_kernel(const float* __restrict__ x, long x_pitch,
                                                                           const int64_t* __restrict__ idx, int N,
                                                                           int D, float* __restrict__ dw,
                                                                           float* __restrict__ counts) {
  __shared__ float part[VQ_SEG_WAVES][128];
  __shared__ int pcnt[VQ_SEG_WAVES];
  const int k = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int per = ((N + VQ_SEG_WAVES - 1) / VQ_SEG_WAVES + 63) & ~63;
  const int lo = w * per, hi = min(N, lo + per);
  const bool c0 = lane < D, c1 = lane + 64 < D;
  float a0 = 0.f, a1 = 0.f;
  int cnt = 0;
  for (int r0 = lo; r0 < hi; r0 += 64) {
    const int r = r0 + lane;
    const bool m = r < hi && idx[r] == (int64_t)k;
    unsigned long long mask = __ballot(m);
    cnt += __popcll(mask);
    while (mask) {
      float v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v0[u] = 0.f;
        v1[u] = 0.f;
        if (mask) {
          const int j = __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          const float* xp = x + (long)(r0 + j) * x_pitch;
          if (c0) v0[u] = xp[lane];
          if (c1) v1[u] = xp[lane + 64];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 += v0[u];
        a1 += v1[u];
      }
    }
  }
  part[w][lane] = a0;
  part[w][lane + 64] = a1;
  if (lane == 0) pcnt[w] = cnt;
  __syncthreads();
  if (w == 0) {
    float s0 = 0.f, s1 = 0.f;
    int n = 0;
#pragma unroll
    for (int i = 0; i < VQ_SEG_WAVES; ++i) {
      s0 += part[i][lane];
      s1 += part[i][lane + 64];
      n += pcnt[i];
    }
    if (c0) dw[(long)k * D + lane] = s0;
    if (c1) dw[(long)k * D + lane + 64] = s1;
    if (lane == 0) counts[k] = (float)n;
  }
}

// EMA codebook update (vector_quantizer.py:128-147), single block.
__global__ __launch_bounds__(256) void vq_ema_update_kernel(float* __restrict__ cluster_size,
                                                            float* __restrict__ ema_emb, float* __restrict__ cb,
                                                            const float* __restrict__ counts,
                                                            const float* __restrict__ dw, int K, int D, float decay,
                                                            float eps) {
  __shared__ float sh[16];
  float part = 0.f;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const float cs = cluster_size[k] * decay + counts[k] * (1.f - decay);
    cluster_size[k] = cs;
    part += cs;
  }
  const float n = lgm_block_sum(part, sh);
  __syncthreads();
  for (int i = threadIdx.x; i < K * D; i += blockDim.x) {
    const int k = i / D;
    const float cw = (cluster_size[k] + eps) / (n + (float)K * eps) * n;
    const float e = ema_emb[i] * decay + dw[i] * (1.f - decay);
    ema_emb[i] = e;
    cb[i] = e / cw;
  }
}

// quantized[row] = cb[idx[row]]; per-block partial of sum (q - x)^2 (fixed order) -> partial[]
__global__ __launch_bounds__(256) void vq_gather_kernel(const float* __restrict__ x, long x_pitch,
                                                        const float* __restrict__ cb, const int64_t* __restrict__ idx,
                                                        int N, int D, float* __restrict__ q, long q_pitch,
                                                        float* __restrict__ partial) {
  __shared__ float sh[16];
  const int d4n = D / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float s = 0.f;
  if (i < (long)N * d4n) {
    const long row = i / d4n;
    const int d = (int)(i % d4n) * 4;
    const f32x4 e = *reinterpret_cast<const f32x4*>(cb + idx[row] * D + d);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + row * x_pitch + d);
    *reinterpret_cast<f32x4*>(q + row * q_pitch + d) = e;
    const f32x4 df = e - xv;
    s = (df[0] * df[0] + df[1] * df[1]) + (df[2] * df[2] + df[3] * df[3]);
  }
  s = lgm_block_sum(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// scalars: mse = sum(partial)/(N*D); vq_loss = (1 + commitment) * mse; perplexity from counts.
__global__ __launch_bounds__(256) void vq_scalars_kernel(const float* __restrict__ partial, int nparts,
                                                         const float* __restrict__ counts, int K, int N, int D,
                                                         float commitment, float* __restrict__ out3) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += partial[i];
  s = lgm_block_sum(s, sh);
  float h = 0.f;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const float p = counts[k] / (float)N;
    h += p * logf(p + 1e-10f);
  }
  h = lgm_block_sum(h, sh);
  if (threadIdx.x == 0) {
    const float mse = s / ((float)N * (float)D);
    out3[0] = mse + commitment * mse;   // e_latent_loss + commitment * q_latent_loss
    out3[1] = expf(-h);                 // perplexity
    out3[2] = mse;
  }
}

// backward: gx[row] (+)= gq[row] (straight-through) + g_vq * commitment * 2 (x - q) / (N D)
//           gcb[k]    = beta*gcb + g_vq * 2 (counts[k] cb[k] - dw[k]) / (N D)
// One launch: blocks [0, nbx) the rows' gradient, blocks [nbx, grid) the codebook's (two independent element-wise maps)
__global__ __launch_bounds__(256) void vq_bwd_kernel(const float* __restrict__ x, long x_pitch,
                                                     const float* __restrict__ q, long q_pitch,
                                                     const float* __restrict__ gq, long gq_pitch,
                                                     const float* __restrict__ gvq, float commitment, int N, int D,
                                                     float* __restrict__ gx, long gx_pitch, int nbx,
                                                     const float* __restrict__ cb, const float* __restrict__ dw,
                                                     const float* __restrict__ counts, int K, float* __restrict__ gcb,
                                                     float beta) {
  if ((int)blockIdx.x >= nbx) {
    const int i = ((int)blockIdx.x - nbx) * blockDim.x + threadIdx.x;
    if (i >= K * D) return;
    const int k = i / D;
    float g = gvq[0] * 2.f / ((float)N * (float)D) * (counts[k] * cb[i] - dw[i]);
    if (beta != 0.f) g += beta * gcb[i];
    gcb[i] = g;
    return;
  }
  const int d4n = D / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * d4n) return;
  const long row = i / d4n;
  const int d = (int)(i % d4n) * 4;
  const float c = gvq[0] * commitment * 2.f / ((float)N * (float)D);
  const f32x4 xv = *reinterpret_cast<const f32x4*>(x + row * x_pitch + d);
  const f32x4 qv = *reinterpret_cast<const f32x4*>(q + row * q_pitch + d);
  f32x4 g = (xv - qv) * c;
  if (gq) g += *reinterpret_cast<const f32x4*>(gq + row * gq_pitch + d);
  *reinterpret_cast<f32x4*>(gx + row * gx_pitch + d) = g;
}
}  // namespace

extern "C" int lgm_vq_assign(const float* x, int64_t x_pitch, const float* codebook, int N, int K, int D,
                             int64_t* indices, float* min_dist, void* stream) {
  LGM_REQUIRE(x && codebook && indices && N > 0 && K > 0, "vq_assign: bad arguments");
  LGM_REQUIRE(K * (int)sizeof(float) <= 60 * 1024, "vq_assign: K=%d too large", K);
  hipStream_t s = (hipStream_t)stream;
  static const bool row_major = getenv("LGM_VQ_ASSIGN_ROWS") != nullptr;      // A/B switch: the row-major search
  if (!row_major && K % 64 == 0 && K <= 512 && (D == 32 || D == 64) && N >= 4 * 256) {
    const int rpb = lgm_cdiv(N, 512);                                         // two blocks per CU's worth of rows
    const int nb = lgm_cdiv(N, rpb);
    const size_t sm = ((size_t)rpb * D + rpb + 2 * (size_t)rpb * (K / 64)) * sizeof(float);
    if (sm <= 64 * 1024) {
      if (D == 64)
        hipLaunchKernelGGL(vq_assign_bycode_kernel<64>, dim3(nb), dim3(K), sm, s, x, (long)x_pitch, codebook, N, K, rpb,
                           indices, min_dist);
      else
        hipLaunchKernelGGL(vq_assign_bycode_kernel<32>, dim3(nb), dim3(K), sm, s, x, (long)x_pitch, codebook, N, K, rpb,
                           indices, min_dist);
      LGM_LAUNCH_CHECK();
      return LGM_OK;
    }
  }
  const dim3 grid(lgm_cdiv(N, 32)), block(256);
  const size_t smem = (size_t)K * sizeof(float);
  switch (D) {
    case 16: hipLaunchKernelGGL(vq_assign_kernel<16>, grid, block, smem, s, x, (long)x_pitch, codebook, N, K, indices, min_dist); break;
    case 32: hipLaunchKernelGGL(vq_assign_kernel<32>, grid, block, smem, s, x, (long)x_pitch, codebook, N, K, indices, min_dist); break;
    case 64: hipLaunchKernelGGL(vq_assign_kernel<64>, grid, block, smem, s, x, (long)x_pitch, codebook, N, K, indices, min_dist); break;
    case 128: hipLaunchKernelGGL(vq_assign_kernel<128>, grid, block, smem, s, x, (long)x_pitch, codebook, N, K, indices, min_dist); break;
    default: lgm_set_error("vq_assign: embedding_dim=%d unsupported (16/32/64/128)", D); return LGM_ERR_UNSUPPORTED;
  }
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_vq_segment_sum(const float* x, int64_t x_pitch, const int64_t* indices, int N, int K, int D,
                                  float* dw, float* counts, void* stream) {
  LGM_REQUIRE(x && indices && dw && counts && N > 0 && K > 0 && D > 0 && D <= VQ_MAXD, "vq_segment_sum: bad arguments");
  hipLaunchKernelGGL(vq_segment_sum_kernel, dim3(K), dim3(64 * VQ_SEG_WAVES), 0, (hipStream_t)stream, x, (long)x_pitch, indices, N, D,
                     dw, counts);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_vq_ema_update(float* cluster_size, float* ema_embedding, float* codebook, const float* counts,
                                 const float* dw, int K, int D, float decay, float eps, void* stream) {
  LGM_REQUIRE(cluster_size && ema_embedding && codebook && counts && dw && K > 0 && D > 0, "vq_ema_update: bad arguments");
  hipLaunchKernelGGL(vq_ema_update_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, cluster_size, ema_embedding,
                     codebook, counts, dw, K, D, decay, eps);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int64_t lgm_vq_gather_workspace(int N, int D) {
  return (int64_t)lgm_cdiv((long)N * (D / 4), 256) * (int64_t)sizeof(float) + 16;
}

extern "C" int lgm_vq_gather_loss(const float* x, int64_t x_pitch, const float* codebook, const int64_t* indices,
                                  const float* counts, int N, int K, int D, float commitment, float* q,
                                  int64_t q_pitch, float* out3, void* workspace, void* stream) {
  LGM_REQUIRE(x && codebook && indices && counts && q && out3 && workspace && D % 4 == 0, "vq_gather_loss: bad arguments");
  LGM_REQUIRE(x_pitch % 4 == 0 && q_pitch % 4 == 0 && lgm_aligned16(x) && lgm_aligned16(q) && lgm_aligned16(codebook),
              "vq_gather_loss: alignment");
  hipStream_t s = (hipStream_t)stream;
  const int nb = lgm_cdiv((long)N * (D / 4), 256);
  hipLaunchKernelGGL(vq_gather_kernel, dim3(nb), dim3(256), 0, s, x, (long)x_pitch, codebook, indices, N, D, q,
                     (long)q_pitch, (float*)workspace);
  hipLaunchKernelGGL(vq_scalars_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, nb, counts, K, N, D,
                     commitment, out3);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_vq_bwd(const float* x, int64_t x_pitch, const float* q, int64_t q_pitch, const float* gq,
                          int64_t gq_pitch, const float* codebook, const float* dw, const float* counts,
                          const float* g_vq_loss, float commitment, int N, int K, int D, float* gx,
                          int64_t gx_pitch, float* gcodebook, float gcb_beta, void* stream) {
  LGM_REQUIRE(x && q && codebook && dw && counts && g_vq_loss && gx && gcodebook && D % 4 == 0, "vq_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int nbx = lgm_cdiv((long)N * (D / 4), 256), nbc = lgm_cdiv((long)K * D, 256);
  hipLaunchKernelGGL(vq_bwd_kernel, dim3(nbx + nbc), dim3(256), 0, s, x, (long)x_pitch, q, (long)q_pitch, gq, (long)gq_pitch,
                     g_vq_loss, commitment, N, D, gx, (long)gx_pitch, nbx, codebook, dw, counts, K, gcodebook, gcb_beta);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
