// Fused tail of the LinearAttention backward pass (reference ddpm.py:203-239 under autograd; the forward it
// differentiates: q, k, v = to_qkv(norm(x)); out = (softmax_n(k) v^T)^T (softmax_d(q) scale)).
//
// linattn_bwd_mfma (linattn_mfma.hip) writes gq, gk, gv of every pixel - a [B n, 384] tensor, 201 MB at 32x32 maps and
// B = 128 - which the two GEMMs of to_qkv's backward then read back: gxn = gqkv Wqkv (input gradient) and
// dWqkv = gqkv^T xn (weight gradient).  All three launches are bound by that tensor's traffic.  Here the three
// [128 x 32] gradient tiles of a (pixel tile, head) stay in LDS:
//   * gxn[128 x C]  += [gq | gk | gv] Wqkv_h[96 x C]        accumulated over the four heads in registers,
//   * dWqkv_h[96 x 64] += [gq | gk | gv]^T xn[128 x 64]      (C = 64 only) accumulated over the block's pixel tiles in
//     registers and written once per block as a slab for the batched fixed-order reducer (lgm_wgrad_reduce_batch);
//     for C > 64 the accumulators do not fit and gqkv is written for the separate weight-gradient kernel as before.
// One persistent workgroup per CU walks a contiguous range of (image, 128-pixel tile) items; the next head's operand
// tiles are fetched into registers while the current head is multiplied.  Everything a wave needs between the staging
// barrier and the weight-gradient step lives in ITS 32 rows of the LDS tiles (softmax passes use two lanes per row), so
// a head costs three workgroup barriers.  Summation orders are fixed: results are run-to-run identical.
#include <type_traits>

#include "lgm_common.h"

int lgm_wgrad_reduce_launch(const float* ws, long slab, float* gw, long n_w, float* gb, long n_b, int splits, float beta,
                            hipStream_t s);

namespace {

constexpr int DH = 32;
constexpr int HEADS = 4;
constexpr int HID = HEADS * DH;
constexpr int LDW = 33;    // row stride of the [pixel][channel] tiles: conflict-free 32x32x2 operand reads
constexpr int TP = 128;    // pixels per tile (32 per wave)
constexpr int XLD = 80;    // row stride of the xn tile: the 4 pixel rows of a 16x16x4 operand read cover all banks twice
constexpr int WLD = 64;    // row stride of the staged weight chunk [96][64]

struct FArgs {
  const float* qkv;  long pitch;
  const float* gout; long gout_pitch;
  const float* ctx; const float* gctx; const float* kmax; const float* ksum; const float* rvec;
  const float* xn;   long xn_pitch;
  const float* w;                       // Wqkv [3 * HID][C]
  float* gxn;        long gxn_pitch;
  float* gqkv;       long gq_pitch;     // written when the weight gradient is not fused
  float* slabs;                         // [blocks][3 * HID * 64] (FUSE_DW)
  int n, tiles, items, per, C;
  float scale;
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int NCH, bool FUSE_DW>
__global__ __launch_bounds__(256, 1) void linattn_bwd_fused_kernel(const FArgs p) {
  static_assert(!FUSE_DW || NCH == 1, "the weight-gradient accumulators fit for 64 input channels only");
  extern __shared__ __align__(16) float sm[];
  float* Qs = sm;                    // q -> softmax_d(q) -> gq
  float* Ks = Qs + TP * LDW;         // softmax_n(k) -> gk
  float* Vs = Ks + TP * LDW;         // v -> T2 = V gctx^T -> gv
  float* Gs = Vs + TP * LDW;         // gout -> T1 = G ctx^T
  float* Cs = Gs + TP * LDW;         // ctx  [32][33]
  float* GCs = Cs + DH * LDW;        // gctx [32][33]
  float* rr = GCs + DH * LDW;        // r[d]
  float* Ws = rr + DH;               // Wqkv rows of the head, one 64-column chunk: [96][64]
  float* Xs = Ws + 3 * DH * WLD;     // xn tile [128][XLD] (FUSE_DW)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int l16 = lane & 15, lq = lane >> 4;
  const int c4 = (tid & 7) * 4, prow = tid >> 3;           // staging map: 8 threads x 16 bytes per pixel row, 32 rows per pass
  const int srow = 32 * wid + (lane >> 1), spart = (lane & 1) * 16;   // softmax map: two lanes per row of the wave's rows
  const int it0 = blockIdx.x * p.per, it1 = min(p.items, it0 + p.per);
  if (it0 >= it1) return;

  // ---- register-staged operands of the NEXT (item, head) ----
  f32x4 q4[4], k4[4], v4[4], g4[4], km4, ks4, cx4, gc4, w4[6];
  float rr1 = 0.f;
  auto issue = [&](int it, int h) {
    const int b = it / p.tiles, i0 = (it % p.tiles) * TP;
    const int rows = min(TP, p.n - i0);
    const long bh = (long)b * HEADS + h;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = prow + 32 * u;
      const long row = (long)b * p.n + i0 + (r < rows ? r : 0);
      const float* qp = p.qkv + row * p.pitch + h * DH + c4;
      q4[u] = *reinterpret_cast<const f32x4*>(qp);
      k4[u] = *reinterpret_cast<const f32x4*>(qp + HID);
      v4[u] = *reinterpret_cast<const f32x4*>(qp + 2 * HID);
      g4[u] = *reinterpret_cast<const f32x4*>(p.gout + row * p.gout_pitch + h * DH + c4);
    }
    km4 = *reinterpret_cast<const f32x4*>(p.kmax + bh * DH + c4);
    ks4 = *reinterpret_cast<const f32x4*>(p.ksum + bh * DH + c4);
    cx4 = *reinterpret_cast<const f32x4*>(p.ctx + bh * DH * DH + tid * 4);
    gc4 = *reinterpret_cast<const f32x4*>(p.gctx + bh * DH * DH + tid * 4);
    if (tid < DH) rr1 = p.rvec[bh * DH + tid];
    // weight chunk 0 of the head: rows part * HID + h * DH + d, 64 columns = 16 x 16 bytes; 1536 / 256 = 6 per thread
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int e = tid + 256 * u;
      const int wr = e >> 4, wc = (e & 15) * 4;
      w4[u] = *reinterpret_cast<const f32x4*>(p.w + (long)((wr >> 5) * HID + h * DH + (wr & 31)) * p.C + wc);
    }
  };
  auto stage_w = [&](int h, int cc) {           // chunks after the first: straight from global memory (small maps only)
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int e = tid + 256 * u;
      const int wr = e >> 4, wc = (e & 15) * 4;
      *reinterpret_cast<f32x4*>(Ws + wr * WLD + wc) =
          *reinterpret_cast<const f32x4*>(p.w + (long)((wr >> 5) * HID + h * DH + (wr & 31)) * p.C + cc * 64 + wc);
    }
  };

  // weight-gradient accumulators: wave w owns xn channels [16 w, 16 w + 16); per head 3 parts x 2 row blocks of 16
  f32x4 dacc[FUSE_DW ? HEADS : 1][6];
  if constexpr (FUSE_DW) {
#pragma unroll
    for (int h = 0; h < HEADS; ++h)
#pragma unroll
      for (int t = 0; t < 6; ++t) dacc[h][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  issue(it0, 0);
  for (int it = it0; it < it1; ++it) {
    const int b = it / p.tiles, i0 = (it % p.tiles) * TP;
    const int rows = min(TP, p.n - i0);
    f32x16 acc[2 * NCH];
#pragma unroll
    for (int a = 0; a < 2 * NCH; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    if constexpr (FUSE_DW) {
      // the item's xn tile: [128][64]; the previous item's weight-gradient step ended with a barrier
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = prow + 32 * u;
        const long row = (long)b * p.n + i0 + (r < rows ? r : 0);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          f32x4 xv = *reinterpret_cast<const f32x4*>(p.xn + row * p.xn_pitch + hf * 32 + c4);
          if (r >= rows) xv = f32x4{0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<f32x4*>(Xs + r * XLD + hf * 32 + c4) = xv;
        }
      }
    }
    auto head = [&](auto hc) {
      constexpr int h = decltype(hc)::value;
      // ---- staging: registers -> LDS ----
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = prow + 32 * u;
        const bool live = r < rows;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          Qs[r * LDW + c4 + k] = live ? q4[u][k] : 0.f;
          Ks[r * LDW + c4 + k] = live ? __expf(k4[u][k] - km4[k]) * (1.f / ks4[k]) : 0.f;
          Vs[r * LDW + c4 + k] = live ? v4[u][k] : 0.f;
          Gs[r * LDW + c4 + k] = live ? g4[u][k] : 0.f;
        }
      }
      {
        const int e = tid * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          Cs[((e + k) >> 5) * LDW + ((e + k) & 31)] = cx4[k];
          GCs[((e + k) >> 5) * LDW + ((e + k) & 31)] = gc4[k];
        }
      }
      if (tid < DH) rr[tid] = rr1;
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int e = tid + 256 * u;
        *reinterpret_cast<f32x4*>(Ws + (e >> 4) * WLD + (e & 15) * 4) = w4[u];
      }
      __syncthreads();
      // ---- the next (item, head) into the registers just freed ----
      if (h + 1 < HEADS) issue(it, h + 1);
      else if (it + 1 < it1) issue(it + 1, 0);

      // ---- phase A: s = softmax_d(q) for this wave's rows, two lanes per row; kept in registers for phase C ----
      float sv[16];
      {
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          sv[k] = Qs[srow * LDW + spart + k];
          mx = fmaxf(mx, sv[k]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          sv[k] = __expf(sv[k] - mx);
          sum += sv[k];
        }
        sum += __shfl_xor(sum, 1, 64);
        const float inv = 1.f / sum;
#pragma unroll
        for (int k = 0; k < 16; ++k) sv[k] *= inv;
      }
      // ---- phase B: T1 = G ctx^T, T2 = V gctx^T, gv = KS gctx for the wave's 32 rows ----
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
      lgm_wave_lds_sync();                          // the operand reads above precede the overwrites below
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * wid + (r & 3) + 8 * (r >> 2) + 4 * lh;
        Gs[row * LDW + lr] = a1[r];
        Vs[row * LDW + lr] = a2[r];
      }
      lgm_wave_lds_sync();
      // ---- phase C: gq = s (T1 scale - <s, T1 scale>), gk = ks (T2 - r); then gv replaces T2 ----
      {
        float g1[16], dot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          g1[k] = Gs[srow * LDW + spart + k] * p.scale;
          dot += sv[k] * g1[k];
        }
        dot += __shfl_xor(dot, 1, 64);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          Qs[srow * LDW + spart + k] = sv[k] * (g1[k] - dot);
          Ks[srow * LDW + spart + k] = Ks[srow * LDW + spart + k] * (Vs[srow * LDW + spart + k] - rr[spart + k]);
        }
      }
      lgm_wave_lds_sync();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * wid + (r & 3) + 8 * (r >> 2) + 4 * lh;
        Vs[row * LDW + lr] = a3[r];
      }
      lgm_wave_lds_sync();
      if constexpr (!FUSE_DW) {
        // gq | gk | gv of the wave's rows for the separate weight-gradient kernel: lane = (row, 16-channel half)
        const int r = srow;
        if (r < rows) {
          float* o = p.gqkv + ((long)b * p.n + i0 + r) * p.gq_pitch + h * DH + spart;
#pragma unroll
          for (int part = 0; part < 3; ++part) {
            const float* X = part == 0 ? Qs : part == 1 ? Ks : Vs;
#pragma unroll
            for (int k4i = 0; k4i < 4; ++k4i) {
              f32x4 ov;
#pragma unroll
              for (int k = 0; k < 4; ++k) ov[k] = X[r * LDW + spart + 4 * k4i + k];
              *reinterpret_cast<f32x4*>(o + part * HID + 4 * k4i) = ov;
            }
          }
        }
      }
      // ---- phase D: gxn[rows of the wave][C] += [gq | gk | gv] W_h ----
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc) {
        if (cc > 0) {
          __syncthreads();
          stage_w(h, cc);
          __syncthreads();
        }
#pragma unroll
        for (int part = 0; part < 3; ++part) {
          const float* X = (part == 0 ? Qs : part == 1 ? Ks : Vs) + (32 * wid + lr) * LDW + lh;
          const float* Wp = Ws + (part * DH + lh) * WLD + lr;
#pragma unroll
          for (int s = 0; s < 16; ++s) {
            const float av = X[2 * s];
            acc[2 * cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Wp[2 * s * WLD], acc[2 * cc], 0, 0, 0);
            acc[2 * cc + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Wp[2 * s * WLD + 32], acc[2 * cc + 1], 0, 0, 0);
          }
        }
      }
      __syncthreads();
      if constexpr (FUSE_DW) {
        // ---- phase E: dW_h[96][64] += [gq | gk | gv]^T xn over the tile's 128 pixels, on 16x16x4 tiles:
        // A[i = gradient channel][k = pixel], B[k = pixel][j = xn channel]; this wave's 16 xn channels
        const float* xb = Xs + lq * XLD + 16 * wid + l16;
#pragma unroll 4
        for (int s = 0; s < TP / 4; ++s) {
          const float bv = xb[4 * s * XLD];
#pragma unroll
          for (int part = 0; part < 3; ++part) {
            const float* X = (part == 0 ? Qs : part == 1 ? Ks : Vs) + (4 * s + lq) * LDW + l16;
            dacc[h][2 * part] = __builtin_amdgcn_mfma_f32_16x16x4f32(X[0], bv, dacc[h][2 * part], 0, 0, 0);
            dacc[h][2 * part + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(X[16], bv, dacc[h][2 * part + 1], 0, 0, 0);
          }
        }
        __syncthreads();
      }
    };
    head(std::integral_constant<int, 0>{});
    head(std::integral_constant<int, 1>{});
    head(std::integral_constant<int, 2>{});
    head(std::integral_constant<int, 3>{});
    // ---- the item's input gradient: rows of this wave, 32 consecutive channels per accumulator ----
#pragma unroll
    for (int a = 0; a < 2 * NCH; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * wid + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < rows) p.gxn[((long)b * p.n + i0 + row) * p.gxn_pitch + a * 32 + lr] = acc[a][r];
      }
  }
  if constexpr (FUSE_DW) {
    float* sl = p.slabs + (long)blockIdx.x * (3 * HID * 64);
#pragma unroll
    for (int h = 0; h < HEADS; ++h)
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int wrow = (t >> 1) * HID + h * DH + (t & 1) * 16 + 4 * lq + r;
          sl[(long)wrow * 64 + 16 * wid + l16] = dacc[h][t][r];
        }
  }
}

constexpr size_t smem_bytes(bool fuse) {
  return ((size_t)4 * TP * LDW + 2 * DH * LDW + DH + 3 * DH * WLD + (fuse ? TP * XLD : 0)) * sizeof(float);
}

template <int NCH, bool FUSE_DW>
int launch(const FArgs& a, int blocks, hipStream_t s) {
  const size_t smem = smem_bytes(FUSE_DW);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(linattn_bwd_fused_kernel<NCH, FUSE_DW>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  hipLaunchKernelGGL((linattn_bwd_fused_kernel<NCH, FUSE_DW>), dim3(blocks), dim3(256), smem, s, a);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

int plan_blocks(int B, int n, int* tiles, int* items, int* per) {
  *tiles = lgm_cdiv(n, TP);
  *items = B * *tiles;
  const int nb = *items < 256 ? *items : 256;
  *per = lgm_cdiv(*items, nb);
  return lgm_cdiv(*items, *per);
}

}  // namespace

extern "C" int64_t lgm_linattn_bwd_fused_supported(int heads, int dim_head, int C) {
  return heads == HEADS && dim_head == DH && (C == 64 || C == 128 || C == 256) ? 1 : 0;
}

// bytes of the weight-gradient slab buffer (0: the weight gradient is not fused for this C)
extern "C" int64_t lgm_linattn_bwd_fused_slabs(int B, int n, int C) {
  if (C != 64) return 0;
  int tiles, items, per;
  const int blocks = plan_blocks(B, n, &tiles, &items, &per);
  return (int64_t)blocks * 3 * HID * 64 * (int64_t)sizeof(float);
}

int lgm_linattn_bwd_fused_launch(const float* qkv, long pitch, const float* gout, long gout_pitch, const float* ctx,
                                 const float* gctx, const float* kmax, const float* ksum, const float* rvec,
                                 const float* xn, long xn_pitch, const float* w, int B, int n, int C, float scale,
                                 float* gxn, long gxn_pitch, float* gqkv, long gq_pitch, float* slabs, int* blocks_out,
                                 hipStream_t s) {
  FArgs a;
  a.qkv = qkv; a.pitch = pitch; a.gout = gout; a.gout_pitch = gout_pitch;
  a.ctx = ctx; a.gctx = gctx; a.kmax = kmax; a.ksum = ksum; a.rvec = rvec;
  a.xn = xn; a.xn_pitch = xn_pitch; a.w = w; a.gxn = gxn; a.gxn_pitch = gxn_pitch; a.gqkv = gqkv; a.gq_pitch = gq_pitch;
  a.slabs = slabs; a.n = n; a.C = C; a.scale = scale;
  const int blocks = plan_blocks(B, n, &a.tiles, &a.items, &a.per);
  *blocks_out = blocks;
  lgm_note_kernel(C == 64 ? "linattn_bwd_fused_kernel<1, true>" : C == 128 ? "linattn_bwd_fused_kernel<2, false>"
                                                                            : "linattn_bwd_fused_kernel<4, false>");
  if (C == 64) return launch<1, true>(a, blocks, s);
  if (C == 128) return launch<2, false>(a, blocks, s);
  return launch<4, false>(a, blocks, s);
}
