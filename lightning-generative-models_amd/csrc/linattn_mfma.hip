// MFMA versions of the two heavy LinearAttention kernels (ddpm.py:217-239):
//   linattn_ctx   : ctx[d][e]  = sum_n softmax_n(k)[d,n] v[e,n]          (forward, mode 0)
//                   gctx[d][e] = sum_n (softmax_d(q) scale)[d,n] gout[e,n] (backward, mode 1)
//   linattn_bwd   : gq, gk, gv for a 128-pixel tile = three [128x32]x[32x32] products
// All contractions run on v_mfma_f32_32x32x2_f32; cross-wave sums use fixed orders (deterministic).
#include "lgm_common.h"

namespace {

constexpr int DH = 32;
constexpr int LDW = 33;   // padded row stride (softmax passes walk rows with a 4-thread team)
constexpr int TP = 128;   // pixels per tile (32 per wave)

template <int MODE>
__global__ __launch_bounds__(256) void linattn_ctx_mfma(const float* __restrict__ qkv, long pitch,
                                                        const float* __restrict__ mem_kv,
                                                        const float* __restrict__ gout, long gout_pitch,
                                                        const float* __restrict__ ctx_in, int n, int heads, int M,
                                                        float scale, float* __restrict__ ctx_out,
                                                        float* __restrict__ kmax_out, float* __restrict__ ksum_out,
                                                        float* __restrict__ r_out, const float* __restrict__ kmax_in,
                                                        const float* __restrict__ ksum_in,
                                                        float* __restrict__ gmem_partial) {
  __shared__ float Ws[TP * LDW];
  __shared__ __align__(16) float Us[TP * DH];
  __shared__ __align__(16) float Red[4][DH][DH];
  __shared__ __align__(16) float red32[32][DH];
  __shared__ float kmax_s[DH];
  const int bh = blockIdx.x;
  const int b = bh / heads, h = bh % heads;
  const int hidden = heads * DH;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  // staging map: a pixel's 32 channels are eight 16-byte loads; 32 pixels per pass of the block
  const int c4 = (tid & 7) * 4, prow = tid >> 3;
  const float* base = qkv + (long)b * n * pitch + h * DH;
  const float* memk = mem_kv + ((long)(0 * heads + h) * DH) * M;  // [d][j]
  const float* memv = mem_kv + ((long)(1 * heads + h) * DH) * M;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // MODE 0: softmax over the pixels with a RUNNING maximum (one pass over k instead of a maximum pass plus a product pass:
  // see DESIGN section 3.4.1 for the timings).  Per tile the column maxima of its 128 rows join the running maximum m; the
  // accumulators and the partial sums so far are rescaled by exp(m_old - m_new) (1 when the maximum did not move); at the
  // end m is the exact column maximum, which the backward kernels get as before.
  __shared__ float corr_s[DH];
  if (MODE == 0) {
    if (tid < DH) kmax_s[tid] = -INFINITY;
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32x4 wsum4 = zero4;
  const int total = n + (MODE == 0 ? M : 0);
  // register-staged tiles: tile t+1 is fetched while tile t is normalised and multiplied
  f32x4 w4[4], u4[4];
  auto fetch = [&](int i0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + prow + 32 * u;
      f32x4 wv = zero4, uv = zero4;
      if (i < n) {
        if (MODE == 0) {
          wv = *reinterpret_cast<const f32x4*>(base + (long)i * pitch + hidden + c4);
          uv = *reinterpret_cast<const f32x4*>(base + (long)i * pitch + 2 * hidden + c4);
        } else {
          wv = *reinterpret_cast<const f32x4*>(base + (long)i * pitch + c4);
          uv = *reinterpret_cast<const f32x4*>(gout + ((long)b * n + i) * gout_pitch + h * DH + c4);
        }
      } else if (i < total) {
        const int jm = i - n;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          wv[k] = memk[(c4 + k) * M + jm];
          uv[k] = memv[(c4 + k) * M + jm];
        }
      }
      w4[u] = wv;
      u4[u] = uv;
    }
  };
  fetch(0);
  for (int i0 = 0; i0 < total; i0 += TP) {
    if (MODE == 0) {
      f32x4 m4 = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + prow + 32 * u < total)
#pragma unroll
          for (int k = 0; k < 4; ++k) m4[k] = fmaxf(m4[k], w4[u][k]);
      *reinterpret_cast<f32x4*>(&red32[prow][c4]) = m4;
    }
    __syncthreads();          // (also: the previous tile's operand reads are done)
    if (MODE == 0) {
      if (tid < DH) {
        float m = red32[0][tid];
        for (int k = 1; k < 32; ++k) m = fmaxf(m, red32[k][tid]);
        const float mo = kmax_s[tid];
        const float mn = fmaxf(mo, m);            // every tile has a live row: mn is finite
        corr_s[tid] = __expf(mo - mn);            // first tile: exp(-inf) = 0 on zero accumulators
        kmax_s[tid] = mn;
      }
      __syncthreads();
      const f32x4 c4v = *reinterpret_cast<const f32x4*>(&corr_s[c4]);
      wsum4 *= c4v;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = prow + 32 * u;
      f32x4 wv = w4[u];
      if (MODE == 0) {
        const bool live = i0 + r < total;
#pragma unroll
        for (int k = 0; k < 4; ++k) wv[k] = live ? __expf(wv[k] - kmax_s[c4 + k]) : 0.f;
        wsum4 += wv;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) Ws[r * LDW + c4 + k] = wv[k];
      *reinterpret_cast<f32x4*>(&Us[r * DH + c4]) = u4[u];
    }
    __syncthreads();
    if (i0 + TP < total) fetch(i0 + TP);
    if (MODE == 1) {
      // softmax over d for every pixel row: 4 threads per row, 8 channels each, two passes of 64 rows
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int r = (tid >> 2) + 64 * ps, part = tid & 3;
        float v[8], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          v[k] = Ws[r * LDW + part * 8 + k];
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
        const float inv = (i0 + r < n) ? scale / sm : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) Ws[r * LDW + part * 8 + k] = v[k] * inv;
      }
      __syncthreads();
    }
    if (MODE == 0) {          // rows d of the accumulator: (r & 3) + 8 (r >> 2) + 4 lh
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] *= corr_s[(r & 3) + 8 * (r >> 2) + 4 * lh];
    }
    // MFMA: k = pixel; wave w owns pixels [32 w, 32 w + 32) of the tile.  A[i = d][k], B[k][j = e]
    const float* ap = Ws + (32 * wid + lh) * LDW + lr;
    const float* bp = Us + (32 * wid + lh) * DH + lr;
#pragma unroll
    for (int s = 0; s < 16; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s * LDW], bp[2 * s * DH], acc, 0, 0, 0);
  }
  // cross-wave reduction in a fixed order
#pragma unroll
  for (int r = 0; r < 16; ++r) Red[wid][(r & 3) + 8 * (r >> 2) + 4 * lh][lr] = acc[r];
  if (MODE == 0) *reinterpret_cast<f32x4*>(&red32[prow][c4]) = wsum4;
  __syncthreads();
  const int d_c = tid / 8, e0 = (tid % 8) * 4;
  f32x4 cv = (*reinterpret_cast<const f32x4*>(&Red[0][d_c][e0]) + *reinterpret_cast<const f32x4*>(&Red[1][d_c][e0])) +
             (*reinterpret_cast<const f32x4*>(&Red[2][d_c][e0]) + *reinterpret_cast<const f32x4*>(&Red[3][d_c][e0]));
  float* co = ctx_out + ((long)bh * DH + d_c) * DH + e0;
  if (MODE == 0) {
    float ws = 0.f;
    for (int k = 0; k < 32; ++k) ws += red32[k][d_c];
    *reinterpret_cast<f32x4*>(co) = cv * (1.f / ws);
    if ((tid % 8) == 0) {
      kmax_out[bh * DH + d_c] = kmax_s[d_c];
      ksum_out[bh * DH + d_c] = ws;
    }
  } else {
    *reinterpret_cast<f32x4*>(co) = cv;
    const f32x4 c = *reinterpret_cast<const f32x4*>(ctx_in + ((long)bh * DH + d_c) * DH + e0);
    float r = cv[0] * c[0] + cv[1] * c[1] + cv[2] * c[2] + cv[3] * c[3];
    r += __shfl_xor(r, 1, 64);
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 4, 64);
    if ((tid % 8) == 0) r_out[bh * DH + d_c] = r;
    if (gmem_partial != nullptr && M > 0) {
      // gradient of the M memory key / value columns (mem_kv, ddpm.py:211,226): they see the same gctx as the pixels -
      // gk[d][j] = ks[d][j] (sum_e gctx[d][e] v[e][j] - r[d]),  gv[e][j] = sum_d ks[d][j] gctx[d][e]
      __syncthreads();
      *reinterpret_cast<f32x4*>(&Red[0][d_c][e0]) = cv;
      if ((tid % 8) == 0) red32[0][d_c] = r;
      for (int t = tid; t < DH * M; t += 256) {
        const int d = t / M, j = t % M;
        Red[1][d][j] = __expf(memk[d * M + j] - kmax_in[bh * DH + d]) * (1.f / ksum_in[bh * DH + d]);
      }
      __syncthreads();
      float* gm = gmem_partial + (long)b * 2 * heads * DH * M;   // [B][2][heads][32][M]
      for (int t = tid; t < 2 * DH * M; t += 256) {
        const int which = t / (DH * M), c = (t % (DH * M)) / M, j = t % M;
        float a = 0.f;
        if (which == 0) {
          for (int e = 0; e < DH; ++e) a += Red[0][c][e] * memv[e * M + j];
          a = Red[1][c][j] * (a - red32[0][c]);
        } else {
          for (int d = 0; d < DH; ++d) a += Red[1][d][j] * Red[0][d][c];
        }
        gm[((long)(which * heads + h) * DH + c) * M + j] = a;
      }
    }
  }
}

// backward of q / k / v for one 128-pixel tile (blockIdx.y == ntiles: the M memory columns)
__global__ __launch_bounds__(256) void linattn_bwd_mfma(
    const float* __restrict__ qkv, long pitch, const float* __restrict__ mem_kv, const float* __restrict__ gout,
    long gout_pitch, const float* __restrict__ ctx, const float* __restrict__ gctx, const float* __restrict__ kmax,
    const float* __restrict__ ksum, const float* __restrict__ rvec, int n, int heads, int M, float scale,
    float* __restrict__ gqkv, long gq_pitch, float* __restrict__ gmem_partial) {
  extern __shared__ __align__(16) float sm[];
  float* Qs = sm;                    // [128][33]  q -> softmax_d(q)
  float* Ks = Qs + TP * LDW;         // ks = softmax_n(k)
  float* Vs = Ks + TP * LDW;         // v   -> T2 = V gctx^T
  float* Gs = Vs + TP * LDW;         // gout -> T1 = G ctx^T
  float* Cs = Gs + TP * LDW;         // ctx  [32][33]
  float* GCs = Cs + DH * LDW;        // gctx [32][33]
  float* kmx = GCs + DH * LDW;
  float* kinv = kmx + DH;
  float* rr = kinv + DH;
  const int bh = blockIdx.x;
  const int b = bh / heads, h = bh % heads;
  const int hidden = heads * DH;
  const int ntiles = (n + TP - 1) / TP;
  const bool is_mem = (int)blockIdx.y == ntiles;
  const int i0 = blockIdx.y * TP;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int d_l = tid % DH, pl = tid / DH;
  const float* memk = mem_kv + ((long)(0 * heads + h) * DH) * M;
  const float* memv = mem_kv + ((long)(1 * heads + h) * DH) * M;
  for (int k = tid; k < DH * DH; k += 256) {
    Cs[(k / DH) * LDW + (k % DH)] = ctx[(long)bh * DH * DH + k];
    GCs[(k / DH) * LDW + (k % DH)] = gctx[(long)bh * DH * DH + k];
  }
  if (tid < DH) {
    kmx[tid] = kmax[bh * DH + tid];
    kinv[tid] = 1.f / ksum[bh * DH + tid];
    rr[tid] = rvec[bh * DH + tid];
  }
  __syncthreads();
  const int rows = is_mem ? M : min(TP, n - i0);
  if (is_mem) {
#pragma unroll 4
    for (int j = 0; j < TP / 8; ++j) {
      const int r = pl + 8 * j;
      float kv = 0.f, vv = 0.f;
      if (r < rows) {
        kv = __expf(memk[d_l * M + r] - kmx[d_l]) * kinv[d_l];
        vv = memv[d_l * M + r];
      }
      Qs[r * LDW + d_l] = 0.f;
      Ks[r * LDW + d_l] = kv;
      Vs[r * LDW + d_l] = vv;
      Gs[r * LDW + d_l] = 0.f;
    }
  } else {
    // a pixel's 32 channels are eight 16-byte loads; 32 pixels per pass, all sixteen loads in flight
    const int c4 = (tid & 7) * 4, prow = tid >> 3;
    f32x4 q4[4], k4[4], v4[4], g4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = prow + 32 * u;
      const long row = (long)b * n + i0 + (r < rows ? r : 0);
      q4[u] = *reinterpret_cast<const f32x4*>(qkv + row * pitch + h * DH + c4);
      k4[u] = *reinterpret_cast<const f32x4*>(qkv + row * pitch + hidden + h * DH + c4);
      v4[u] = *reinterpret_cast<const f32x4*>(qkv + row * pitch + 2 * hidden + h * DH + c4);
      g4[u] = *reinterpret_cast<const f32x4*>(gout + row * gout_pitch + h * DH + c4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = prow + 32 * u;
      const bool live = r < rows;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        Qs[r * LDW + c4 + k] = live ? q4[u][k] : 0.f;
        Ks[r * LDW + c4 + k] = live ? __expf(k4[u][k] - kmx[c4 + k]) * kinv[c4 + k] : 0.f;
        Vs[r * LDW + c4 + k] = live ? v4[u][k] : 0.f;
        Gs[r * LDW + c4 + k] = live ? g4[u][k] : 0.f;
      }
    }
  }
  __syncthreads();
  // ---- phase A: s = softmax_d(q) in place ----
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int r = (tid >> 2) + 64 * ps, part = tid & 3;
    float v[8], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = Qs[r * LDW + part * 8 + k];
      mx = fmaxf(mx, v[k]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = __expf(v[k] - mx);
      sum += v[k];
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int k = 0; k < 8; ++k) Qs[r * LDW + part * 8 + k] = v[k] * inv;
  }
  // ---- phase B: wave w owns rows [32 w, 32 w + 32): three 32x32x32 products on MFMA ----
  f32x16 a1, a2, a3;
#pragma unroll
  for (int r = 0; r < 16; ++r) a1[r] = a2[r] = a3[r] = 0.f;
  {
    const float* gp = Gs + (32 * wid + lr) * LDW + lh;
    const float* vp = Vs + (32 * wid + lr) * LDW + lh;
    const float* kp = Ks + (32 * wid + lr) * LDW + lh;
    const float* cT = Cs + lr * LDW + lh;      // B[k = e][j = d] = ctx[d][e]
    const float* gT = GCs + lr * LDW + lh;     // B[k = e][j = d] = gctx[d][e]
    const float* gN = GCs + lh * LDW + lr;     // B[k = d][j = e] = gctx[d][e]
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(gp[2 * s], cT[2 * s], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[2 * s], gT[2 * s], a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * s], gN[2 * s * LDW], a3, 0, 0, 0);
    }
  }
  // T1 -> Gs, T2 -> Vs (rows owned by this wave only), gv -> global
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = 32 * wid + (r & 3) + 8 * (r >> 2) + 4 * lh;
    Gs[row * LDW + lr] = a1[r];
    Vs[row * LDW + lr] = a2[r];
    if (row < rows) {
      if (is_mem)
        gmem_partial[(long)b * 2 * heads * DH * M + ((long)(1 * heads + h) * DH + lr) * M + row] = a3[r];
      else
        gqkv[((long)b * n + i0 + row) * gq_pitch + 2 * hidden + h * DH + lr] = a3[r];
    }
  }
  __syncthreads();
  // ---- phase C: softmax backward for q, and gk = ks * (T2 - r) ----
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int r = (tid >> 2) + 64 * ps, part = tid & 3, c0 = part * 8;
    float s[8], g1[8], gk[8], dot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s[k] = Qs[r * LDW + c0 + k];
      g1[k] = Gs[r * LDW + c0 + k] * scale;
      dot += s[k] * g1[k];
      gk[k] = Ks[r * LDW + c0 + k] * (Vs[r * LDW + c0 + k] - rr[c0 + k]);
    }
    dot += __shfl_xor(dot, 1, 64);
    dot += __shfl_xor(dot, 2, 64);
    if (r < rows) {
      if (is_mem) {
        float* gm = gmem_partial + (long)b * 2 * heads * DH * M;
#pragma unroll
        for (int k = 0; k < 8; ++k) gm[((long)(0 * heads + h) * DH + c0 + k) * M + r] = gk[k];
      } else {
        float* o = gqkv + ((long)b * n + i0 + r) * gq_pitch + h * DH + c0;
        f32x4 oq[2], ok[2];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          oq[k >> 2][k & 3] = s[k] * (g1[k] - dot);
          ok[k >> 2][k & 3] = gk[k];
        }
        *reinterpret_cast<f32x4*>(o) = oq[0];
        *reinterpret_cast<f32x4*>(o + 4) = oq[1];
        *reinterpret_cast<f32x4*>(o + hidden) = ok[0];
        *reinterpret_cast<f32x4*>(o + hidden + 4) = ok[1];
      }
    }
  }
}

}  // namespace

int lgm_linattn_ctx_launch(int mode, const float* qkv, long pitch, const float* mem_kv, const float* gout,
                           long gout_pitch, const float* ctx_in, int B, int n, int heads, int M, float scale,
                           float* ctx_out, float* kmax, float* ksum, float* r_out, hipStream_t s,
                           const float* kmax_in, const float* ksum_in, float* gmem_partial) {
  LGM_REQUIRE(pitch % 4 == 0 && lgm_aligned16(qkv) && (mode == 0 || (gout_pitch % 4 == 0 && lgm_aligned16(gout))),
              "linattn_ctx: 16-byte aligned rows required");
  if (mode == 0)
    hipLaunchKernelGGL(linattn_ctx_mfma<0>, dim3(B * heads), dim3(256), 0, s, qkv, pitch, mem_kv, gout, gout_pitch,
                       ctx_in, n, heads, M, scale, ctx_out, kmax, ksum, r_out, kmax_in, ksum_in, gmem_partial);
  else
    hipLaunchKernelGGL(linattn_ctx_mfma<1>, dim3(B * heads), dim3(256), 0, s, qkv, pitch, mem_kv, gout, gout_pitch,
                       ctx_in, n, heads, M, scale, ctx_out, kmax, ksum, r_out, kmax_in, ksum_in, gmem_partial);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

int lgm_linattn_bwd_launch(const float* qkv, long pitch, const float* mem_kv, const float* gout, long gout_pitch,
                           const float* ctx, const float* gctx, const float* kmax, const float* ksum,
                           const float* rvec, int B, int n, int heads, int M, float scale, float* gqkv,
                           long gq_pitch, float* gmem_partial, hipStream_t s) {
  LGM_REQUIRE(pitch % 4 == 0 && gout_pitch % 4 == 0 && gq_pitch % 4 == 0 && lgm_aligned16(qkv) && lgm_aligned16(gout) &&
                  lgm_aligned16(gqkv),
              "linattn_bwd: 16-byte aligned rows required");
  const size_t smem = ((size_t)4 * TP * LDW + 2 * DH * LDW + 3 * DH) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(linattn_bwd_mfma), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)smem);
    attr = true;
  }
  // gmem_partial == nullptr: the memory columns' gradient came out of the gctx launch (linattn_ctx_mfma<1>'s epilogue)
  hipLaunchKernelGGL(linattn_bwd_mfma, dim3(B * heads, lgm_cdiv(n, TP) + (M > 0 && gmem_partial ? 1 : 0)), dim3(256), smem, s, qkv,
                     pitch, mem_kv, gout, gout_pitch, ctx, gctx, kmax, ksum, rvec, n, heads, M, scale, gqkv, gq_pitch,
                     gmem_partial);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
