// Attention cores of the DDPM UNet on NHWC qkv tensors (dim_head fixed at 32 = half a wavefront).
//
//  LinearAttention (ddpm.py:217-239): q.softmax(d)*scale, k.softmax(n), context = k v^T, out = context^T q,
//      with num_mem_kv learned key/value columns prepended to k and v (mem_kv [2, heads, 32, M]).
//  Attention (ddpm.py:255-271 + modules/attend.py:97-126): softmax(q k^T * scale) v with M learned
//      memory rows prepended (mem_kv [2, heads, M, 32]).
//
// qkv layout: [B, n, 3*hidden] (pitch given), channel = which*hidden + head*32 + d, hidden = heads*32.
// Everything a (batch, head) pair needs lives in LDS/registers; reductions use fixed orders.
#include "lgm_common.h"

// MFMA kernels for the linear-attention contractions (linattn_mfma.hip)
int lgm_linattn_ctx_launch(int mode, const float* qkv, long pitch, const float* mem_kv, const float* gout,
                           long gout_pitch, const float* ctx_in, int B, int n, int heads, int M, float scale,
                           float* ctx_out, float* kmax, float* ksum, float* r_out, hipStream_t s,
                           const float* kmax_in = nullptr, const float* ksum_in = nullptr,
                           float* gmem_partial = nullptr);
// fused backward tail (linattn_fused.hip) and the single-layer slab reducer (conv_igemm.hip)
int lgm_linattn_bwd_fused_launch(const float* qkv, long pitch, const float* gout, long gout_pitch, const float* ctx,
                                 const float* gctx, const float* kmax, const float* ksum, const float* rvec,
                                 const float* xn, long xn_pitch, const float* wt, int B, int n, float scale,
                                 float* gxn, long gxn_pitch, float* slabs, int* blocks_out, hipStream_t s);
int lgm_wgrad_reduce_launch(const float* ws, long slab, float* gw, long n_w, float* gb, long n_b, int splits, float beta,
                            hipStream_t s);
int lgm_linattn_out_fused_launch(const float* qkv, long pitch, const float* ctx, const float* wout, const float* bout,
                                 const float* g, const float* x, long x_pitch, float* ao, long ao_pitch, float* o2,
                                 long o2_pitch, float* y, long y_pitch, int B, int n, int Cout, float scale, hipStream_t s);
int lgm_linattn_bwd_launch(const float* qkv, long pitch, const float* mem_kv, const float* gout, long gout_pitch,
                           const float* ctx, const float* gctx, const float* kmax, const float* ksum,
                           const float* rvec, int B, int n, int heads, int M, float scale, float* gqkv,
                           long gq_pitch, float* gmem_partial, hipStream_t s);

namespace {

constexpr int DH = 32;    // dim_head
constexpr int TI = 64;    // pixel tile

// =====================================================================================
// Linear attention
// =====================================================================================

// out[i, h*32+e] = sum_d ctx[d][e] * softmax_d(q[i,:])[d] * scale
__global__ __launch_bounds__(256) void linattn_out_kernel(const float* __restrict__ qkv, long pitch,
                                                          const float* __restrict__ ctx, int n, int heads, float scale,
                                                          float* __restrict__ out, long out_pitch) {
  __shared__ float Qs[TI][DH + 1];
  __shared__ __align__(16) float Cs[DH][DH];
  const int bh = blockIdx.x;
  const int b = bh / heads, h = bh % heads;
  const int i0 = blockIdx.y * TI;
  const int tid = threadIdx.x;
  const int d_l = tid % DH, pl = tid / DH;
  for (int k = tid; k < DH * DH; k += 256) Cs[k / DH][k % DH] = ctx[(long)bh * DH * DH + k];
  {
    // a pixel's 32 query channels are eight 16-byte loads; 32 pixels per pass
    const int c4 = (tid & 7) * 4, prow = tid >> 3;
    f32x4 q4[TI / 32];
#pragma unroll
    for (int u = 0; u < TI / 32; ++u) {
      const int i = i0 + prow + 32 * u;
      q4[u] = i < n ? *reinterpret_cast<const f32x4*>(qkv + ((long)b * n + i) * pitch + h * DH + c4)
                    : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < TI / 32; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) Qs[prow + 32 * u][c4 + k] = q4[u][k];
  }
  __syncthreads();
  const int r = tid / 4, part = tid % 4;
  {
    float v[8], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = Qs[r][part * 8 + k];
      mx = fmaxf(mx, v[k]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = __expf(v[k] - mx);
      sm += v[k];
    }
    sm += __shfl_xor(sm, 1, 64);
    sm += __shfl_xor(sm, 2, 64);
    const float inv = scale / sm;
#pragma unroll
    for (int k = 0; k < 8; ++k) Qs[r][part * 8 + k] = v[k] * inv;
  }
  __syncthreads();
  const int e0 = part * 8;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int d = 0; d < DH; ++d) {
    const float qv = Qs[r][d];
    a0 += *reinterpret_cast<const f32x4*>(&Cs[d][e0]) * qv;
    a1 += *reinterpret_cast<const f32x4*>(&Cs[d][e0 + 4]) * qv;
  }
  const int i = i0 + r;
  if (i < n) {
    float* o = out + ((long)b * n + i) * out_pitch + h * DH + e0;
    *reinterpret_cast<f32x4*>(o) = a0;
    *reinterpret_cast<f32x4*>(o + 4) = a1;
  }
}

// =====================================================================================
// Full softmax attention, n <= 128 query pixels, one block per (batch, head), one thread per row.
// =====================================================================================
constexpr int FA_MAXN = 128;
constexpr int FA_LD = DH + 1;

__global__ __launch_bounds__(128) void attn_fwd_kernel(const float* __restrict__ qkv, long pitch,
                                                       const float* __restrict__ mem_kv, int n, int heads, int M,
                                                       float scale, float* __restrict__ out, long out_pitch,
                                                       float* __restrict__ lse) {
  extern __shared__ float sm[];
  float* Ks = sm;                       // [(n+M)][33]
  float* Vs = sm + (n + M) * FA_LD;     // [(n+M)][33]
  const int bh = blockIdx.x;
  const int b = bh / heads, h = bh % heads;
  const int hidden = heads * DH;
  const int tid = threadIdx.x;
  const int nk = n + M;
  const float* memk = mem_kv + ((long)(0 * heads + h) * M) * DH;  // [j][d]
  const float* memv = mem_kv + ((long)(1 * heads + h) * M) * DH;
  for (int idx = tid; idx < nk * DH; idx += blockDim.x) {
    const int j = idx / DH, d = idx % DH;
    float kv, vv;
    if (j < M) {
      kv = memk[j * DH + d];
      vv = memv[j * DH + d];
    } else {
      const long row = (long)b * n + (j - M);
      kv = qkv[row * pitch + hidden + h * DH + d];
      vv = qkv[row * pitch + 2 * hidden + h * DH + d];
    }
    Ks[j * FA_LD + d] = kv;
    Vs[j * FA_LD + d] = vv;
  }
  __syncthreads();
  const int i = tid;
  if (i >= n) return;
  float q[DH], acc[DH];
  const float* qp = qkv + ((long)b * n + i) * pitch + h * DH;
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    q[d] = qp[d] * scale;
    acc[d] = 0.f;
  }
  float mx = -INFINITY, den = 0.f;
  for (int j = 0; j < nk; ++j) {
    float sdot = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) sdot += q[d] * Ks[j * FA_LD + d];
    const float nm = fmaxf(mx, sdot);
    const float corr = __expf(mx - nm);
    const float p = __expf(sdot - nm);
    den = den * corr + p;
#pragma unroll
    for (int d = 0; d < DH; ++d) acc[d] = acc[d] * corr + p * Vs[j * FA_LD + d];
    mx = nm;
  }
  const float inv = 1.f / den;
  float* o = out + ((long)b * n + i) * out_pitch + h * DH;
#pragma unroll
  for (int d = 0; d < DH; ++d) o[d] = acc[d] * inv;
  lse[(long)bh * n + i] = mx + __logf(den);
}

__global__ __launch_bounds__(192) void attn_bwd_kernel(const float* __restrict__ qkv, long pitch,
                                                       const float* __restrict__ mem_kv,
                                                       const float* __restrict__ out, long out_pitch,
                                                       const float* __restrict__ gout, long gout_pitch,
                                                       const float* __restrict__ lse, int n, int heads, int M,
                                                       float scale, float* __restrict__ gqkv, long gq_pitch,
                                                       float* __restrict__ gmem_partial) {
  extern __shared__ float sm[];
  const int nk = n + M;
  float* Ks = sm;                         // [nk][33]
  float* Vs = Ks + nk * FA_LD;            // [nk][33]
  float* Qs = Vs + nk * FA_LD;            // [n][33]  (pre-scaled by `scale`)
  float* Gs = Qs + n * FA_LD;             // [n][33]
  float* Ls = Gs + n * FA_LD;             // [n]  lse
  float* Ds = Ls + n;                     // [n]  D_i = gout_i . out_i
  const int bh = blockIdx.x;
  const int b = bh / heads, h = bh % heads;
  const int hidden = heads * DH;
  const int tid = threadIdx.x;
  const float* memk = mem_kv + ((long)(0 * heads + h) * M) * DH;
  const float* memv = mem_kv + ((long)(1 * heads + h) * M) * DH;
  for (int idx = tid; idx < nk * DH; idx += blockDim.x) {
    const int j = idx / DH, d = idx % DH;
    float kv, vv;
    if (j < M) {
      kv = memk[j * DH + d];
      vv = memv[j * DH + d];
    } else {
      const long row = (long)b * n + (j - M);
      kv = qkv[row * pitch + hidden + h * DH + d];
      vv = qkv[row * pitch + 2 * hidden + h * DH + d];
    }
    Ks[j * FA_LD + d] = kv;
    Vs[j * FA_LD + d] = vv;
  }
  for (int idx = tid; idx < n * DH; idx += blockDim.x) {
    const int i = idx / DH, d = idx % DH;
    const long row = (long)b * n + i;
    Qs[i * FA_LD + d] = qkv[row * pitch + h * DH + d] * scale;
    Gs[i * FA_LD + d] = gout[row * gout_pitch + h * DH + d];
  }
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) {
    const long row = (long)b * n + i;
    float dsum = 0.f;
    for (int d = 0; d < DH; ++d) dsum += Gs[i * FA_LD + d] * out[row * out_pitch + h * DH + d];
    Ds[i] = dsum;
    Ls[i] = lse[(long)bh * n + i];
  }
  __syncthreads();
  // phase 1: thread per query row -> gq
  if (tid < n) {
    const int i = tid;
    float gq[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) gq[d] = 0.f;
    for (int j = 0; j < nk; ++j) {
      float sdot = 0.f, ga = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        sdot += Qs[i * FA_LD + d] * Ks[j * FA_LD + d];
        ga += Gs[i * FA_LD + d] * Vs[j * FA_LD + d];
      }
      const float p = __expf(sdot - Ls[i]);
      const float gs = p * (ga - Ds[i]) * scale;
#pragma unroll
      for (int d = 0; d < DH; ++d) gq[d] += gs * Ks[j * FA_LD + d];
    }
    float* o = gqkv + ((long)b * n + i) * gq_pitch + h * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] = gq[d];
  }
  // phase 2: thread per key row -> gk, gv   (Qs already carries `scale`)
  if (tid < nk) {
    const int j = tid;
    float gk[DH], gv[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      gk[d] = 0.f;
      gv[d] = 0.f;
    }
    for (int i = 0; i < n; ++i) {
      float sdot = 0.f, ga = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        sdot += Qs[i * FA_LD + d] * Ks[j * FA_LD + d];
        ga += Gs[i * FA_LD + d] * Vs[j * FA_LD + d];
      }
      const float p = __expf(sdot - Ls[i]);
      const float gs = p * (ga - Ds[i]);
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        gk[d] += gs * Qs[i * FA_LD + d];   // includes scale
        gv[d] += p * Gs[i * FA_LD + d];
      }
    }
    if (j < M) {
      float* gm = gmem_partial + (long)b * 2 * heads * M * DH;  // [B][2][heads][M][32]
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        gm[((long)(0 * heads + h) * M + j) * DH + d] = gk[d];
        gm[((long)(1 * heads + h) * M + j) * DH + d] = gv[d];
      }
    } else {
      float* o = gqkv + ((long)b * n + (j - M)) * gq_pitch + h * DH;
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        o[hidden + d] = gk[d];
        o[2 * hidden + d] = gv[d];
      }
    }
  }
}

// ---- the same two passes for SMALL maps (n <= 64 query pixels: the 4 x 4 / 8 x 8 maps where the UNets use full attention) ----
// One thread per query row leaves 16 (64) of 128 threads busy and a 20 (68)-step dependent chain per thread (17 / 26 us per
// launch at 4 x 4 at every batch).  Here every thread works in every stage: scores for all (query, key) pairs, a four-thread
// softmax per row, then one 16-byte piece of an output row per work item; the backward builds P and dS once and reads each
// of gq, gk, gv off them.  Three barriers.  Sums run in index order (fixed).  256 threads, dynamic LDS sized by (n, M).
constexpr int SA_MAXN = 64;

__device__ __forceinline__ void sa_load_rows(const float* __restrict__ qkv, long pitch, const float* __restrict__ mem_kv,
                                             int b, int h, int n, int heads, int M, float scale, float* Qs, float* Ks,
                                             float* Vs) {
  const int hidden = heads * DH, nk = n + M, tid = threadIdx.x;
  const float* memk = mem_kv + ((long)(0 * heads + h) * M) * DH;  // [j][d]
  const float* memv = mem_kv + ((long)(1 * heads + h) * M) * DH;
  for (int e = tid; e < nk * 8; e += blockDim.x) {                  // a row's head slice = eight 16-byte pieces
    const int j = e >> 3, d = (e & 7) * 4;
    f32x4 kv, vv;
    if (j < M) {
      kv = *reinterpret_cast<const f32x4*>(memk + j * DH + d);
      vv = *reinterpret_cast<const f32x4*>(memv + j * DH + d);
    } else {
      const float* row = qkv + ((long)b * n + (j - M)) * pitch + h * DH + d;
      kv = *reinterpret_cast<const f32x4*>(row + hidden);
      vv = *reinterpret_cast<const f32x4*>(row + 2 * hidden);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      Ks[j * FA_LD + d + k] = kv[k];
      Vs[j * FA_LD + d + k] = vv[k];
    }
  }
  for (int e = tid; e < n * 8; e += blockDim.x) {
    const int i = e >> 3, d = (e & 7) * 4;
    const f32x4 qv = *reinterpret_cast<const f32x4*>(qkv + ((long)b * n + i) * pitch + h * DH + d);
#pragma unroll
    for (int k = 0; k < 4; ++k) Qs[i * FA_LD + d + k] = qv[k] * scale;
  }
}

// floats of dynamic LDS
size_t sa_fwd_floats(int n, int M) { return (size_t)(n + 2 * (n + M)) * FA_LD + (size_t)n * (n + M + 1); }
size_t sa_bwd_floats(int n, int M) { return (size_t)(2 * n + 2 * (n + M)) * FA_LD + (size_t)2 * n * (n + M + 1) + 2 * n; }

__global__ __launch_bounds__(256) void attn_small_fwd_kernel(const float* __restrict__ qkv, long pitch,
                                                             const float* __restrict__ mem_kv, int n, int heads, int M,
                                                             float scale, float* __restrict__ out, long out_pitch,
                                                             float* __restrict__ lse) {
  extern __shared__ float sm[];
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads, tid = threadIdx.x, nk = n + M, LS = nk + 1;
  float* Qs = sm;
  float* Ks = Qs + n * FA_LD;
  float* Vs = Ks + nk * FA_LD;
  float* S = Vs + nk * FA_LD;
  sa_load_rows(qkv, pitch, mem_kv, b, h, n, heads, M, scale, Qs, Ks, Vs);
  __syncthreads();
  for (int e = tid; e < n * nk; e += 256) {                          // scores
    const int i = e / nk, j = e - i * nk;
    float sd = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) sd += Qs[i * FA_LD + d] * Ks[j * FA_LD + d];
    S[i * LS + j] = sd;
  }
  __syncthreads();
  for (int r0 = 0; r0 < n; r0 += 64) {                               // softmax: four threads per row, 64 rows per pass
    const int i = r0 + (tid >> 2), part = tid & 3;
    float mx = -INFINITY;
    if (i < n)
      for (int j = part; j < nk; j += 4) mx = fmaxf(mx, S[i * LS + j]);
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float sum = 0.f;
    if (i < n)
      for (int j = part; j < nk; j += 4) {
        const float pv = __expf(S[i * LS + j] - mx);
        S[i * LS + j] = pv;
        sum += pv;
      }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    if (i < n) {
      const float inv = 1.f / sum;
      for (int j = part; j < nk; j += 4) S[i * LS + j] *= inv;
      if (part == 0) lse[(long)bh * n + i] = mx + __logf(sum);
    }
  }
  __syncthreads();
  for (int e = tid; e < n * 8; e += 256) {                           // out = P V
    const int i = e >> 3, d = (e & 7) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < nk; ++j) {
      const float pv = S[i * LS + j];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += pv * Vs[j * FA_LD + d + k];
    }
    *reinterpret_cast<f32x4*>(out + ((long)b * n + i) * out_pitch + h * DH + d) = acc;
  }
}

__global__ __launch_bounds__(256) void attn_small_bwd_kernel(const float* __restrict__ qkv, long pitch,
                                                             const float* __restrict__ mem_kv,
                                                             const float* __restrict__ out, long out_pitch,
                                                             const float* __restrict__ gout, long gout_pitch,
                                                             const float* __restrict__ lse, int n, int heads, int M,
                                                             float scale, float* __restrict__ gqkv, long gq_pitch,
                                                             float* __restrict__ gmem_partial) {
  extern __shared__ float sm[];
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads, tid = threadIdx.x, nk = n + M, hidden = heads * DH, LS = nk + 1;
  float* Qs = sm;
  float* Ks = Qs + n * FA_LD;
  float* Vs = Ks + nk * FA_LD;
  float* Gs = Vs + nk * FA_LD;
  float* P = Gs + n * FA_LD;
  float* dS = P + n * LS;
  float* Ds = dS + n * LS;
  float* Ls = Ds + n;
  sa_load_rows(qkv, pitch, mem_kv, b, h, n, heads, M, scale, Qs, Ks, Vs);
  for (int r0 = 0; r0 < n; r0 += 32) {                               // G rows and D_i = gout_i . out_i: eight threads per row
    const int i = r0 + (tid >> 3), d = (tid & 7) * 4;
    float ds = 0.f;
    if (i < n) {
      const long row = (long)b * n + i;
      const f32x4 gv = *reinterpret_cast<const f32x4*>(gout + row * gout_pitch + h * DH + d);
      const f32x4 ov = *reinterpret_cast<const f32x4*>(out + row * out_pitch + h * DH + d);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        Gs[i * FA_LD + d + k] = gv[k];
        ds += gv[k] * ov[k];
      }
    }
    ds += __shfl_xor(ds, 1, 64);
    ds += __shfl_xor(ds, 2, 64);
    ds += __shfl_xor(ds, 4, 64);
    if (i < n && (tid & 7) == 0) {
      Ds[i] = ds;
      Ls[i] = lse[(long)bh * n + i];
    }
  }
  __syncthreads();
  for (int e = tid; e < n * nk; e += 256) {                          // P and dS = P (G V^T - D)
    const int i = e / nk, j = e - i * nk;
    float sd = 0.f, ga = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      sd += Qs[i * FA_LD + d] * Ks[j * FA_LD + d];
      ga += Gs[i * FA_LD + d] * Vs[j * FA_LD + d];
    }
    const float pv = __expf(sd - Ls[i]);
    P[i * LS + j] = pv;
    dS[i * LS + j] = pv * (ga - Ds[i]);
  }
  __syncthreads();
  // 16-byte pieces of: gq rows [0, n), gk rows [n, n + nk), gv rows [n + nk, n + 2 nk)
  for (int e = tid; e < (n + 2 * nk) * 8; e += 256) {
    const int r = e >> 3, d = (e & 7) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (r < n) {                                                     // gq_i = scale * sum_j dS_ij K_j
      for (int j = 0; j < nk; ++j) {
        const float w = dS[r * LS + j];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += w * Ks[j * FA_LD + d + k];
      }
      *reinterpret_cast<f32x4*>(gqkv + ((long)b * n + r) * gq_pitch + h * DH + d) = acc * scale;
      continue;
    }
    const int which = r < n + nk ? 0 : 1;                            // 0: gk_j = sum_i dS_ij Qs_i (Qs carries scale), 1: gv_j = sum_i P_ij G_i
    const int j = r - n - which * nk;
    const float* W = which ? P : dS;
    const float* X = which ? Gs : Qs;
    for (int i = 0; i < n; ++i) {
      const float w = W[i * LS + j];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += w * X[i * FA_LD + d + k];
    }
    if (j < M)
      *reinterpret_cast<f32x4*>(gmem_partial + (long)b * 2 * heads * M * DH + ((long)(which * heads + h) * M + j) * DH + d) = acc;
    else
      *reinterpret_cast<f32x4*>(gqkv + ((long)b * n + (j - M)) * gq_pitch + (which + 1) * hidden + h * DH + d) = acc;
  }
}

int attn_check(int B, int n, int heads, int dim_head, int M) {
  LGM_REQUIRE(B > 0 && n > 0 && heads > 0 && M >= 0 && M <= 16, "attention: bad sizes B=%d n=%d heads=%d M=%d", B, n, heads, M);
  LGM_REQUIRE(dim_head == DH, "attention: dim_head=%d unsupported (kernels are built for 32)", dim_head);
  return LGM_OK;
}

}  // namespace

// ---- linear attention -----------------------------------------------------------------
extern "C" int lgm_linattn_fwd(const float* qkv, int64_t qkv_pitch, const float* mem_kv, int B, int n, int heads,
                               int dim_head, int M, float* out, int64_t out_pitch, float* ctx, float* kmax,
                               float* ksum, void* stream) {
  if (int rc = attn_check(B, n, heads, dim_head, M)) return rc;
  LGM_REQUIRE(qkv && mem_kv && out && ctx && kmax && ksum, "linattn_fwd: null pointer");
  LGM_REQUIRE(out_pitch % 4 == 0 && lgm_aligned16(out) && lgm_aligned16(ctx), "linattn_fwd: out must be 16B aligned");
  hipStream_t s = (hipStream_t)stream;
  const float scale = 1.f / sqrtf((float)dim_head);
  if (int rc = lgm_linattn_ctx_launch(0, qkv, (long)qkv_pitch, mem_kv, nullptr, 0L, nullptr, B, n, heads, M, scale, ctx,
                                      kmax, ksum, nullptr, s))
    return rc;
  hipLaunchKernelGGL(linattn_out_kernel, dim3(B * heads, lgm_cdiv(n, TI)), dim3(256), 0, s, qkv, (long)qkv_pitch,
                     (const float*)ctx, n, heads, scale, out, (long)out_pitch);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

// Forward with its tail fused (linattn_fused.hip): the context launch, then ONE launch for softmax_d(q), out = q ctx,
// to_out[0] (1x1 convolution + bias), to_out[1] (RMSNorm) and the residual: y = RMSNorm(to_out(out)) + x.  `out` and
// o2 = to_out[0](out) are written for the backward pass but not read back.
extern "C" int lgm_linattn_fwd_fused(const float* qkv, int64_t qkv_pitch, const float* mem_kv, int B, int n, int heads,
                                     int dim_head, int M, const float* wout, const float* bout, const float* g, int Cout,
                                     const float* x, int64_t x_pitch, float* out, int64_t out_pitch, float* o2,
                                     int64_t o2_pitch, float* y, int64_t y_pitch, float* ctx, float* kmax, float* ksum,
                                     void* stream) {
  if (int rc = attn_check(B, n, heads, dim_head, M)) return rc;
  LGM_REQUIRE(lgm_linattn_fwd_fused_supported(heads, dim_head, Cout), "linattn_fwd_fused: heads=%d dim_head=%d C=%d unsupported",
              heads, dim_head, Cout);
  LGM_REQUIRE(qkv && mem_kv && wout && bout && g && x && out && o2 && y && ctx && kmax && ksum, "linattn_fwd_fused: null pointer");
  LGM_REQUIRE(qkv_pitch % 4 == 0 && x_pitch % 4 == 0 && out_pitch % 4 == 0 && o2_pitch % 4 == 0 && y_pitch % 4 == 0 &&
                  lgm_aligned16(qkv) && lgm_aligned16(wout) && lgm_aligned16(bout) && lgm_aligned16(g) && lgm_aligned16(x) &&
                  lgm_aligned16(out) && lgm_aligned16(o2) && lgm_aligned16(y) && lgm_aligned16(ctx),
              "linattn_fwd_fused: 16-byte aligned operands required");
  hipStream_t s = (hipStream_t)stream;
  const float scale = 1.f / sqrtf((float)dim_head);
  if (int rc = lgm_linattn_ctx_launch(0, qkv, (long)qkv_pitch, mem_kv, nullptr, 0L, nullptr, B, n, heads, M, scale, ctx,
                                      kmax, ksum, nullptr, s))
    return rc;
  return lgm_linattn_out_fused_launch(qkv, (long)qkv_pitch, ctx, wout, bout, g, x, (long)x_pitch, out, (long)out_pitch, o2,
                                      (long)o2_pitch, y, (long)y_pitch, B, n, Cout, scale, s);
}

extern "C" int64_t lgm_linattn_bwd_workspace(int B, int heads, int dim_head, int M) {
  const int64_t bh = (int64_t)B * heads;
  const int64_t part = (int64_t)B * 2 * heads * dim_head * M;
  return (bh * dim_head * dim_head + bh * dim_head + part) * (int64_t)sizeof(float) +
         lgm_colsum_workspace(B, 2 * heads * dim_head * M) + 64;
}

namespace {
// gmem_desc == nullptr: the per-image partial rows of the mem_kv gradient (in `part`) are summed here, else the row of
// lgm_wgrad_reduce_batch that sums them is written to gmem_desc and `part` must outlive that launch
int linattn_bwd_impl(const float* qkv, int64_t qkv_pitch, const float* mem_kv, const float* gout, int64_t gout_pitch,
                     const float* ctx, const float* kmax, const float* ksum, int B, int n, int heads, int dim_head, int M,
                     float* gqkv, int64_t gqkv_pitch, float* gmem_kv, float gmem_beta, float* part, int64_t* gmem_desc,
                     void* workspace, void* stream) {
  if (int rc = attn_check(B, n, heads, dim_head, M)) return rc;
  LGM_REQUIRE(qkv && mem_kv && gout && ctx && kmax && ksum && gqkv && gmem_kv && workspace && part, "linattn_bwd: null pointer");
  LGM_REQUIRE(lgm_aligned16(workspace) && lgm_aligned16(part), "linattn_bwd: workspace must be 16B aligned");
  hipStream_t s = (hipStream_t)stream;
  const float scale = 1.f / sqrtf((float)dim_head);
  const long bh = (long)B * heads;
  float* gctx = (float*)workspace;
  float* rvec = gctx + bh * DH * DH;
  const long ncols = 2L * heads * DH * M;
  // the memory columns' gradient rides in the gctx launch when the fixed-order slab reducer can take it (16-byte rows)
  const bool in_ctx = M > 0 && lgm_aligned16(gmem_kv) && ncols % 4 == 0;
  if (int rc = lgm_linattn_ctx_launch(1, qkv, (long)qkv_pitch, mem_kv, gout, (long)gout_pitch, ctx, B, n, heads, M, scale,
                                      gctx, nullptr, nullptr, rvec, s, kmax, ksum, in_ctx ? part : nullptr))
    return rc;
  if (int rc = lgm_linattn_bwd_launch(qkv, (long)qkv_pitch, mem_kv, gout, (long)gout_pitch, ctx, gctx, kmax, ksum, rvec,
                                      B, n, heads, M, scale, gqkv, (long)gqkv_pitch, in_ctx ? nullptr : part, s))
    return rc;
  if (gmem_desc) gmem_desc[6] = 0;
  if (M <= 0) return LGM_OK;
  if (!in_ctx) {
    LGM_REQUIRE(!gmem_desc, "linattn_bwd_deferred: mem_kv gradient must be 16-byte aligned");
    return lgm_colsum(part, ncols, B, ncols, gmem_kv, gmem_beta, part + (long)B * ncols, stream);
  }
  if (gmem_desc) {
    union { float f; int64_t i; } bb;
    bb.i = 0; bb.f = gmem_beta;
    gmem_desc[0] = (int64_t)(uintptr_t)part; gmem_desc[1] = ncols; gmem_desc[2] = (int64_t)(uintptr_t)gmem_kv;
    gmem_desc[3] = ncols; gmem_desc[4] = 0; gmem_desc[5] = 0; gmem_desc[6] = B; gmem_desc[7] = bb.i;
    return LGM_OK;
  }
  return lgm_wgrad_reduce_launch(part, ncols, gmem_kv, ncols, nullptr, 0, B, gmem_beta, s);
}
}  // namespace

extern "C" int lgm_linattn_bwd(const float* qkv, int64_t qkv_pitch, const float* mem_kv, const float* gout,
                               int64_t gout_pitch, const float* ctx, const float* kmax, const float* ksum, int B,
                               int n, int heads, int dim_head, int M, float* gqkv, int64_t gqkv_pitch,
                               float* gmem_kv, float gmem_beta, void* workspace, void* stream) {
  LGM_REQUIRE(workspace, "linattn_bwd: null pointer");
  const long bh = (long)B * heads;
  float* part = (float*)workspace + bh * DH * DH + bh * DH;
  // (unaligned mem_kv gradients take the two-stage column sum, whose scratch follows the partial rows)
  return linattn_bwd_impl(qkv, qkv_pitch, mem_kv, gout, gout_pitch, ctx, kmax, ksum, B, n, heads, dim_head, M, gqkv,
                          gqkv_pitch, gmem_kv, gmem_beta, part, nullptr, workspace, stream);
}

// Deferred form: the partial rows of the mem_kv gradient are left in `gmem_part` (B*2*heads*32*M floats, must stay
// untouched until the caller's lgm_wgrad_reduce_batch has run) with their reducer row in gmem_desc (8 int64).
extern "C" int lgm_linattn_bwd_deferred(const float* qkv, int64_t qkv_pitch, const float* mem_kv, const float* gout,
                                        int64_t gout_pitch, const float* ctx, const float* kmax, const float* ksum,
                                        int B, int n, int heads, int dim_head, int M, float* gqkv, int64_t gqkv_pitch,
                                        float* gmem_kv, float gmem_beta, void* gmem_part, int64_t* gmem_desc,
                                        void* workspace, void* stream) {
  LGM_REQUIRE(gmem_desc && gmem_part && lgm_aligned16(gmem_kv), "linattn_bwd_deferred: descriptor / partial buffer / aligned gradient required");
  return linattn_bwd_impl(qkv, qkv_pitch, mem_kv, gout, gout_pitch, ctx, kmax, ksum, B, n, heads, dim_head, M, gqkv,
                          gqkv_pitch, gmem_kv, gmem_beta, (float*)gmem_part, gmem_desc, workspace, stream);
}

// Backward with the tail fused (linattn_fused.hip): gq / gk / gv never leave the chip - the kernel that computes them
// also produces to_qkv's input gradient gxn = gqkv Wqkv and to_qkv's weight gradient (slabs + descriptor for the
// fixed-order reducer).  The memory columns' gradient rides in the gctx launch.  `gw_desc` / `gmem_desc` (8 int64 each,
// rows of lgm_wgrad_reduce_batch): non-null = deferred, the caller reduces later and `slabs` / `gmem_part` must stay
// untouched until then.
extern "C" int64_t lgm_linattn_bwd_fused_workspace(int B, int heads, int dim_head) {
  const int64_t bh = (int64_t)B * heads;
  return (bh * dim_head * dim_head + bh * dim_head) * (int64_t)sizeof(float) + 64;
}

extern "C" int lgm_linattn_bwd_fused(const float* qkv, int64_t qkv_pitch, const float* mem_kv, const float* gout,
                                     int64_t gout_pitch, const float* ctx, const float* kmax, const float* ksum,
                                     const float* xn, int64_t xn_pitch, const float* wqkv_t, int C, int B, int n,
                                     int heads, int dim_head, int M, float* gxn, int64_t gxn_pitch, float* gwqkv,
                                     float gw_beta, void* slabs, int64_t slab_bytes, int64_t* gw_desc, float* gmem_kv,
                                     float gmem_beta, void* gmem_part, int64_t* gmem_desc, void* workspace,
                                     void* stream) {
  if (int rc = attn_check(B, n, heads, dim_head, M)) return rc;
  LGM_REQUIRE(lgm_linattn_bwd_fused_supported(heads, dim_head, C), "linattn_bwd_fused: heads=%d dim_head=%d C=%d unsupported",
              heads, dim_head, C);
  LGM_REQUIRE(qkv && mem_kv && gout && ctx && kmax && ksum && xn && wqkv_t && gxn && gwqkv && slabs && gmem_kv && gmem_part &&
                  workspace, "linattn_bwd_fused: null pointer");
  LGM_REQUIRE(qkv_pitch % 4 == 0 && gout_pitch % 4 == 0 && xn_pitch % 4 == 0 && gxn_pitch % 4 == 0 && lgm_aligned16(qkv) &&
                  lgm_aligned16(gout) && lgm_aligned16(ctx) && lgm_aligned16(kmax) && lgm_aligned16(ksum) &&
                  lgm_aligned16(xn) && lgm_aligned16(wqkv_t) && lgm_aligned16(gxn) && lgm_aligned16(gwqkv) &&
                  lgm_aligned16(slabs) && lgm_aligned16(workspace) && lgm_aligned16(gmem_kv) && lgm_aligned16(gmem_part),
              "linattn_bwd_fused: 16-byte aligned operands required");
  LGM_REQUIRE(slab_bytes >= lgm_linattn_bwd_fused_slabs(B, n, C), "linattn_bwd_fused: slab buffer too small");
  hipStream_t s = (hipStream_t)stream;
  const float scale = 1.f / sqrtf((float)dim_head);
  const long bh = (long)B * heads;
  float* gctx = (float*)workspace;
  float* rvec = gctx + bh * DH * DH;
  const long ncols = 2L * heads * DH * M;
  if (int rc = lgm_linattn_ctx_launch(1, qkv, (long)qkv_pitch, mem_kv, gout, (long)gout_pitch, ctx, B, n, heads, M, scale,
                                      gctx, nullptr, nullptr, rvec, s, kmax, ksum, (float*)gmem_part))
    return rc;
  int blocks = 0;
  if (int rc = lgm_linattn_bwd_fused_launch(qkv, (long)qkv_pitch, gout, (long)gout_pitch, ctx, gctx, kmax, ksum, rvec, xn,
                                            (long)xn_pitch, wqkv_t, B, n, scale, gxn, (long)gxn_pitch, (float*)slabs,
                                            &blocks, s))
    return rc;
  union { float f; int64_t i; } bb;
  if (M > 0) {
    if (gmem_desc) {
      bb.i = 0; bb.f = gmem_beta;
      gmem_desc[0] = (int64_t)(uintptr_t)gmem_part; gmem_desc[1] = ncols; gmem_desc[2] = (int64_t)(uintptr_t)gmem_kv;
      gmem_desc[3] = ncols; gmem_desc[4] = 0; gmem_desc[5] = 0; gmem_desc[6] = B; gmem_desc[7] = bb.i;
    } else if (int rc = lgm_wgrad_reduce_launch((const float*)gmem_part, ncols, gmem_kv, ncols, nullptr, 0, B, gmem_beta, s)) {
      return rc;
    }
  } else if (gmem_desc) {
    gmem_desc[6] = 0;
  }
  const long n_w = 3L * heads * DH * C;
  if (gw_desc) {
    bb.i = 0; bb.f = gw_beta;
    gw_desc[0] = (int64_t)(uintptr_t)slabs; gw_desc[1] = n_w; gw_desc[2] = (int64_t)(uintptr_t)gwqkv; gw_desc[3] = n_w;
    gw_desc[4] = 0; gw_desc[5] = 0; gw_desc[6] = blocks; gw_desc[7] = bb.i;
  } else if (int rc = lgm_wgrad_reduce_launch((const float*)slabs, n_w, gwqkv, n_w, nullptr, 0, blocks, gw_beta, s)) {
    return rc;
  }
  return LGM_OK;
}

// ---- full attention -------------------------------------------------------------------
extern "C" int lgm_attn_fwd(const float* qkv, int64_t qkv_pitch, const float* mem_kv, int B, int n, int heads,
                            int dim_head, int M, float* out, int64_t out_pitch, float* lse, void* stream) {
  if (int rc = attn_check(B, n, heads, dim_head, M)) return rc;
  LGM_REQUIRE(n <= FA_MAXN, "attn_fwd: n=%d > %d query pixels unsupported", n, FA_MAXN);
  LGM_REQUIRE(qkv && mem_kv && out && lse, "attn_fwd: null pointer");
  const size_t smem = (size_t)2 * (n + M) * FA_LD * sizeof(float);
  static const bool no_small = getenv("LGM_NO_SMALL_ATTN") != nullptr;          // A/B switch
  if (!no_small && n <= SA_MAXN && M <= 16 && qkv_pitch % 4 == 0 && out_pitch % 4 == 0 && lgm_aligned16(qkv) &&
      lgm_aligned16(out) && lgm_aligned16(mem_kv))
  {
    static bool attr = false;
    if (!attr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(attn_small_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)(sa_fwd_floats(SA_MAXN, 16) * sizeof(float)));
      attr = true;
    }
    hipLaunchKernelGGL(attn_small_fwd_kernel, dim3(B * heads), dim3(256), sa_fwd_floats(n, M) * sizeof(float),
                       (hipStream_t)stream, qkv, (long)qkv_pitch, mem_kv, n, heads, M, 1.f / sqrtf((float)dim_head), out,
                       (long)out_pitch, lse);
  }
  else
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * heads), dim3(128), smem, (hipStream_t)stream, qkv, (long)qkv_pitch,
                       mem_kv, n, heads, M, 1.f / sqrtf((float)dim_head), out, (long)out_pitch, lse);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int64_t lgm_attn_bwd_workspace(int B, int heads, int dim_head, int M) {
  const int64_t ncols = 2LL * heads * M * dim_head;
  return (int64_t)B * ncols * (int64_t)sizeof(float) + lgm_colsum_workspace(B, ncols) + 64;
}

namespace {
int attn_bwd_impl(const float* qkv, int64_t qkv_pitch, const float* mem_kv, const float* out, int64_t out_pitch,
                  const float* gout, int64_t gout_pitch, const float* lse, int B, int n, int heads, int dim_head, int M,
                  float* gqkv, int64_t gqkv_pitch, float* gmem_kv, float gmem_beta, float* part, int64_t* gmem_desc,
                  void* stream) {
  if (int rc = attn_check(B, n, heads, dim_head, M)) return rc;
  LGM_REQUIRE(n <= FA_MAXN, "attn_bwd: n=%d > %d query pixels unsupported", n, FA_MAXN);
  LGM_REQUIRE(qkv && mem_kv && out && gout && lse && gqkv && gmem_kv && part, "attn_bwd: null pointer");
  const size_t smem = ((size_t)2 * (n + M) * FA_LD + (size_t)2 * n * FA_LD + 2 * n) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)(((size_t)4 * (FA_MAXN + 16) * FA_LD + 2 * FA_MAXN) * sizeof(float)));
    attr_set = true;
  }
  const long ncols = 2L * heads * M * DH;
  static const bool no_small = getenv("LGM_NO_SMALL_ATTN") != nullptr;          // A/B switch
  if (!no_small && n <= SA_MAXN && M <= 16 && qkv_pitch % 4 == 0 && out_pitch % 4 == 0 && gout_pitch % 4 == 0 &&
      gqkv_pitch % 4 == 0 && lgm_aligned16(qkv) && lgm_aligned16(out) && lgm_aligned16(gout) && lgm_aligned16(gqkv) &&
      lgm_aligned16(mem_kv) && lgm_aligned16(part))
  {
    static bool attr = false;
    if (!attr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(attn_small_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)(sa_bwd_floats(SA_MAXN, 16) * sizeof(float)));
      attr = true;
    }
    hipLaunchKernelGGL(attn_small_bwd_kernel, dim3(B * heads), dim3(256), sa_bwd_floats(n, M) * sizeof(float),
                       (hipStream_t)stream, qkv, (long)qkv_pitch, mem_kv, out, (long)out_pitch, gout, (long)gout_pitch, lse, n,
                       heads, M, 1.f / sqrtf((float)dim_head), gqkv, (long)gqkv_pitch, part);
  }
  else
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * heads), dim3(192), smem, (hipStream_t)stream, qkv, (long)qkv_pitch,
                       mem_kv, out, (long)out_pitch, gout, (long)gout_pitch, lse, n, heads, M,
                       1.f / sqrtf((float)dim_head), gqkv, (long)gqkv_pitch, part);
  LGM_LAUNCH_CHECK();
  if (gmem_desc) gmem_desc[6] = 0;
  if (M <= 0) return LGM_OK;
  if (gmem_desc) {
    union { float f; int64_t i; } bb;
    bb.i = 0; bb.f = gmem_beta;
    gmem_desc[0] = (int64_t)(uintptr_t)part; gmem_desc[1] = ncols; gmem_desc[2] = (int64_t)(uintptr_t)gmem_kv;
    gmem_desc[3] = ncols; gmem_desc[4] = 0; gmem_desc[5] = 0; gmem_desc[6] = B; gmem_desc[7] = bb.i;
    return LGM_OK;
  }
  return lgm_colsum(part, ncols, B, ncols, gmem_kv, gmem_beta, part + (long)B * ncols, stream);
}
}  // namespace

extern "C" int lgm_attn_bwd(const float* qkv, int64_t qkv_pitch, const float* mem_kv, const float* out,
                            int64_t out_pitch, const float* gout, int64_t gout_pitch, const float* lse, int B, int n,
                            int heads, int dim_head, int M, float* gqkv, int64_t gqkv_pitch, float* gmem_kv,
                            float gmem_beta, void* workspace, void* stream) {
  return attn_bwd_impl(qkv, qkv_pitch, mem_kv, out, out_pitch, gout, gout_pitch, lse, B, n, heads, dim_head, M, gqkv,
                       gqkv_pitch, gmem_kv, gmem_beta, (float*)workspace, nullptr, stream);
}

// Deferred form (see lgm_linattn_bwd_deferred): gmem_part = B*2*heads*M*32 floats.
extern "C" int lgm_attn_bwd_deferred(const float* qkv, int64_t qkv_pitch, const float* mem_kv, const float* out,
                                     int64_t out_pitch, const float* gout, int64_t gout_pitch, const float* lse, int B,
                                     int n, int heads, int dim_head, int M, float* gqkv, int64_t gqkv_pitch,
                                     float* gmem_kv, float gmem_beta, void* gmem_part, int64_t* gmem_desc,
                                     void* stream) {
  LGM_REQUIRE(gmem_desc && gmem_part && lgm_aligned16(gmem_part) && lgm_aligned16(gmem_kv),
              "attn_bwd_deferred: descriptor / partial buffer / aligned gradient required");
  return attn_bwd_impl(qkv, qkv_pitch, mem_kv, out, out_pitch, gout, gout_pitch, lse, B, n, heads, dim_head, M, gqkv,
                       gqkv_pitch, gmem_kv, gmem_beta, (float*)gmem_part, gmem_desc, stream);
}
