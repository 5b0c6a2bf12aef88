// Fused tail of the LinearAttention backward pass (reference ddpm.py:203-239 under autograd; the forward it
// differentiates: q, k, v = to_qkv(norm(x)); out = (softmax_n(k) v^T)^T (softmax_d(q) scale)).
//
// linattn_bwd_mfma (linattn_mfma.hip) writes gq, gk, gv of every pixel - a [B n, 384] tensor, 201 MB at 32x32 maps and
// B = 128 - which the two GEMMs of to_qkv's backward then read back: gxn = gqkv Wqkv (input gradient) and
// dWqkv = gqkv^T xn (weight gradient).  All three launches are bound by that tensor's traffic.  Here the three
// [128 x 32] gradient tiles of a (pixel tile, head) stay in LDS:
//   * gxn[128 x C]  += [gq | gk | gv] Wqkv_h[96 x C]        accumulated over the four heads in registers,
//   * dWqkv_h[96 x 64] += [gq | gk | gv]^T xn[128 x 64]      accumulated over the block's pixel tiles in registers and
//     written once per block as a slab for the batched fixed-order reducer (lgm_wgrad_reduce_batch).
// Built for to_qkv layers with 64 input channels (the 32x32 and the down-path 16x16 attention blocks of the UNet, where
// the gradient tensor is large); wider layers sit on small maps and keep the three-launch path.
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
constexpr int C = 64;      // to_qkv's input channels (the weight-gradient accumulators are sized for this)
constexpr int LD = 36;     // row stride of the [pixel][channel] tiles: 16-byte aligned rows; the 16 lanes one ds_read_b128
                           // services together (rows 36 floats apart) cover all 64 banks exactly once
constexpr int TP = 128;    // pixels per tile (32 per wave)
constexpr int XLD = 80;    // row stride of the xn tile: the 4 pixel rows of a 16x16x4 operand read cover all banks
constexpr int WLD = 100;   // row stride of the staged weight rows [c][96]: 100 = 36 (mod 64), same property as LD

struct FArgs {
  const float* qkv;  long pitch;
  const float* gout; long gout_pitch;
  const float* ctx; const float* gctx; const float* kmax; const float* ksum; const float* rvec;
  const float* xn;   long xn_pitch;
  const float* wt;                      // to_qkv's weight transposed: [C][3 * HID]
  float* gxn;        long gxn_pitch;
  float* slabs;                         // [blocks][3 * HID * C]
  int n, tiles, items, per;
  float scale;
};

// All contractions run with the reduction index dealt out as k = 16 * (lane >> 5) + step (any bijection of k works for a
// sum): a lane's sixteen operand values of a 32 x 32 x 32 product are then CONSECUTIVE floats of one tile row = four
// ds_read_b128 instead of sixteen ds_read_b32 (beside fp32 MFMAs every instruction of the wave costs MFMA issue time,
// DESIGN.md finding 11).  Products are oriented D[channel][pixel] - the small matrices are the A operand - so that a lane
// ends up with ONE pixel and groups of 4 consecutive channels: results go back into the [pixel][channel] tiles, and out
// to memory, as 16-byte accesses.
__device__ __forceinline__ void load16(const float* p, float (&v)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
    v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
  }
}

__global__ __launch_bounds__(256, 1) void linattn_bwd_fused_kernel(const FArgs p) {
  extern __shared__ __align__(16) float sm[];
  float* Qs = sm;                    // q -> gq
  float* Ks = Qs + TP * LD;          // softmax_n(k) -> gk
  float* Vs = Ks + TP * LD;          // v -> T2 = V gctx^T -> gv
  float* Gs = Vs + TP * LD;          // gout -> T1 = G ctx^T
  float* Cs = Gs + TP * LD;          // ctx  [d][e]
  float* GCs = Cs + DH * LD;         // gctx [d][e]
  float* GCt = GCs + DH * LD;        // gctx^T [e][d]
  float* rr = GCt + DH * LD;         // r[d]
  float* Ws = rr + DH;               // the head's weight rows, transposed: [c][part * 32 + d]
  float* Xs = Ws + C * WLD;          // xn tile [128][XLD]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int l16 = lane & 15, lq = lane >> 4;
  const int c4 = (tid & 7) * 4, prow = tid >> 3;           // staging map: 8 threads x 16 bytes per pixel row, 32 rows per pass
  const int srow = 32 * wid + (lane >> 1), spart = (lane & 1) * 16;   // softmax map: two lanes per row of the wave's rows
  const int it0 = blockIdx.x * p.per, it1 = min(p.items, it0 + p.per);
  if (it0 >= it1) return;

  // weight staging map: element e = tid + 256 u of [c][part][8 x 16 bytes]
  int w_src[6], w_dst[6];
#pragma unroll
  for (int u = 0; u < 6; ++u) {
    const int e = tid + 256 * u;
    const int c = e / 24, rem = e - 24 * c;
    const int part = rem >> 3, j = rem & 7;
    w_src[u] = c * (3 * HID) + part * HID + 4 * j;
    w_dst[u] = c * WLD + part * DH + 4 * j;
  }

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
#pragma unroll
    for (int u = 0; u < 6; ++u) w4[u] = *reinterpret_cast<const f32x4*>(p.wt + w_src[u] + h * DH);
  };

  // weight-gradient accumulators: wave w owns xn channels [16 w, 16 w + 16); per head 3 parts x 2 row blocks of 16
  f32x4 dacc[HEADS][6];
#pragma unroll
  for (int h = 0; h < HEADS; ++h)
#pragma unroll
    for (int t = 0; t < 6; ++t) dacc[h][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  issue(it0, 0);
  for (int it = it0; it < it1; ++it) {
    const int b = it / p.tiles, i0 = (it % p.tiles) * TP;
    const int rows = min(TP, p.n - i0);
    f32x16 acc[2];                     // gxn^T: [channel block of 32][the wave's 32 pixels]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    {
      // the item's xn tile: [128][64]; the previous item's weight-gradient step ended with a barrier
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = prow + 32 * u;
        const long row = (long)b * p.n + i0 + (r < rows ? r : 0);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          f32x4 xv = *reinterpret_cast<const f32x4*>(p.xn + row * p.xn_pitch + hf * 32 + c4);
          if (r >= rows) xv = zero4;
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
        f32x4 kv;
#pragma unroll
        for (int k = 0; k < 4; ++k) kv[k] = __expf(k4[u][k] - km4[k]) * (1.f / ks4[k]);
        *reinterpret_cast<f32x4*>(Qs + r * LD + c4) = live ? q4[u] : zero4;
        *reinterpret_cast<f32x4*>(Ks + r * LD + c4) = live ? kv : zero4;
        *reinterpret_cast<f32x4*>(Vs + r * LD + c4) = live ? v4[u] : zero4;
        *reinterpret_cast<f32x4*>(Gs + r * LD + c4) = live ? g4[u] : zero4;
      }
      {
        const int d = tid >> 3, e0 = (tid & 7) * 4;
        *reinterpret_cast<f32x4*>(Cs + d * LD + e0) = cx4;
        *reinterpret_cast<f32x4*>(GCs + d * LD + e0) = gc4;
#pragma unroll
        for (int k = 0; k < 4; ++k) GCt[(e0 + k) * LD + d] = gc4[k];
      }
      if (tid < DH) rr[tid] = rr1;
#pragma unroll
      for (int u = 0; u < 6; ++u) *reinterpret_cast<f32x4*>(Ws + w_dst[u]) = w4[u];
      __syncthreads();
      // ---- the next (item, head) into the registers just freed ----
      if (h + 1 < HEADS) issue(it, h + 1);
      else if (it + 1 < it1) issue(it + 1, 0);

      // ---- phase A: s = softmax_d(q) for this wave's rows, two lanes per row; kept in registers for phase C ----
      float sv[16];
      {
        load16(Qs + srow * LD + spart, sv);
        float mx = sv[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) mx = fmaxf(mx, sv[k]);
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
      // ---- phase B: T1^T = ctx G^T, T2^T = gctx V^T, gv^T = gctx^T KS^T for the wave's 32 pixels ----
      f32x16 a1, a2, a3;
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[r] = a2[r] = a3[r] = 0.f;
      {
        float ca[16], ga[16], ta[16], gb[16], vb[16], kb[16];
        load16(Cs + lr * LD + 16 * lh, ca);
        load16(GCs + lr * LD + 16 * lh, ga);
        load16(GCt + lr * LD + 16 * lh, ta);
        load16(Gs + (32 * wid + lr) * LD + 16 * lh, gb);
        load16(Vs + (32 * wid + lr) * LD + 16 * lh, vb);
        load16(Ks + (32 * wid + lr) * LD + 16 * lh, kb);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[s], gb[s], a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[s], vb[s], a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[s], kb[s], a3, 0, 0, 0);
        }
      }
      lgm_wave_lds_sync();                          // the operand reads above precede the overwrites below
      {
        float* gp = Gs + (32 * wid + lr) * LD + 4 * lh;
        float* vp = Vs + (32 * wid + lr) * LD + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          *reinterpret_cast<f32x4*>(gp + 8 * g) = f32x4{a1[4 * g], a1[4 * g + 1], a1[4 * g + 2], a1[4 * g + 3]};
          *reinterpret_cast<f32x4*>(vp + 8 * g) = f32x4{a2[4 * g], a2[4 * g + 1], a2[4 * g + 2], a2[4 * g + 3]};
        }
      }
      lgm_wave_lds_sync();
      // ---- phase C: gq = s (T1 scale - <s, T1 scale>), gk = ks (T2 - r); then gv replaces T2 ----
      {
        float g1[16], t2[16], kk[16], rv[16], dot = 0.f;
        load16(Gs + srow * LD + spart, g1);
        load16(Vs + srow * LD + spart, t2);
        load16(Ks + srow * LD + spart, kk);
        load16(rr + spart, rv);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          g1[k] *= p.scale;
          dot += sv[k] * g1[k];
        }
        dot += __shfl_xor(dot, 1, 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 oq, ok;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            oq[k] = sv[4 * q + k] * (g1[4 * q + k] - dot);
            ok[k] = kk[4 * q + k] * (t2[4 * q + k] - rv[4 * q + k]);
          }
          *reinterpret_cast<f32x4*>(Qs + srow * LD + spart + 4 * q) = oq;
          *reinterpret_cast<f32x4*>(Ks + srow * LD + spart + 4 * q) = ok;
        }
      }
      lgm_wave_lds_sync();
      {
        float* vp = Vs + (32 * wid + lr) * LD + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(vp + 8 * g) = f32x4{a3[4 * g], a3[4 * g + 1], a3[4 * g + 2], a3[4 * g + 3]};
      }
      lgm_wave_lds_sync();
      // ---- phase D: gxn^T[c][pixels of the wave] += W_h^T[c][96] [gq | gk | gv]^T ----
#pragma unroll
      for (int part = 0; part < 3; ++part) {
        float xb[16], w0[16], w1[16];
        load16((part == 0 ? Qs : part == 1 ? Ks : Vs) + (32 * wid + lr) * LD + 16 * lh, xb);
        load16(Ws + lr * WLD + part * DH + 16 * lh, w0);
        load16(Ws + (32 + lr) * WLD + part * DH + 16 * lh, w1);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[s], xb[s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[s], xb[s], acc[1], 0, 0, 0);
        }
      }
      __syncthreads();
      {
        // ---- phase E: dW_h[96][64] += [gq | gk | gv]^T xn over the tile's 128 pixels, on 16x16x4 tiles:
        // A[i = gradient channel][k = pixel], B[k = pixel][j = xn channel]; this wave's 16 xn channels
        const float* xb = Xs + lq * XLD + 16 * wid + l16;
#pragma unroll 4
        for (int s = 0; s < TP / 4; ++s) {
          const float bv = xb[4 * s * XLD];
#pragma unroll
          for (int part = 0; part < 3; ++part) {
            const float* X = (part == 0 ? Qs : part == 1 ? Ks : Vs) + (4 * s + lq) * LD + l16;
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
    // ---- the item's input gradient: this lane's pixel, groups of 4 consecutive channels ----
    if (32 * wid + lr < rows) {
      float* o = p.gxn + ((long)b * p.n + i0 + 32 * wid + lr) * p.gxn_pitch + 4 * lh;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(o + 32 * a + 8 * g) = f32x4{acc[a][4 * g], acc[a][4 * g + 1], acc[a][4 * g + 2], acc[a][4 * g + 3]};
    }
  }
  {
    float* sl = p.slabs + (long)blockIdx.x * (3 * HID * C);
#pragma unroll
    for (int h = 0; h < HEADS; ++h)
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int wrow = (t >> 1) * HID + h * DH + (t & 1) * 16 + 4 * lq + r;
          sl[(long)wrow * C + 16 * wid + l16] = dacc[h][t][r];
        }
  }
}

constexpr size_t kSmem = ((size_t)4 * TP * LD + 3 * DH * LD + DH + C * WLD + TP * XLD) * sizeof(float);

int plan_blocks(int B, int n, int* tiles, int* items, int* per) {
  *tiles = lgm_cdiv(n, TP);
  *items = B * *tiles;
  const int nb = *items < 256 ? *items : 256;
  *per = lgm_cdiv(*items, nb);
  return lgm_cdiv(*items, *per);
}

}  // namespace

extern "C" int64_t lgm_linattn_bwd_fused_supported(int heads, int dim_head, int Cin) {
  return heads == HEADS && dim_head == DH && Cin == C ? 1 : 0;
}

// bytes of the weight-gradient slab buffer
extern "C" int64_t lgm_linattn_bwd_fused_slabs(int B, int n, int Cin) {
  if (Cin != C) return 0;
  int tiles, items, per;
  const int blocks = plan_blocks(B, n, &tiles, &items, &per);
  return (int64_t)blocks * 3 * HID * C * (int64_t)sizeof(float);
}

int lgm_linattn_bwd_fused_launch(const float* qkv, long pitch, const float* gout, long gout_pitch, const float* ctx,
                                 const float* gctx, const float* kmax, const float* ksum, const float* rvec,
                                 const float* xn, long xn_pitch, const float* wt, int B, int n, float scale,
                                 float* gxn, long gxn_pitch, float* slabs, int* blocks_out, hipStream_t s) {
  FArgs a;
  a.qkv = qkv; a.pitch = pitch; a.gout = gout; a.gout_pitch = gout_pitch;
  a.ctx = ctx; a.gctx = gctx; a.kmax = kmax; a.ksum = ksum; a.rvec = rvec;
  a.xn = xn; a.xn_pitch = xn_pitch; a.wt = wt; a.gxn = gxn; a.gxn_pitch = gxn_pitch;
  a.slabs = slabs; a.n = n; a.scale = scale;
  const int blocks = plan_blocks(B, n, &a.tiles, &a.items, &a.per);
  *blocks_out = blocks;
  lgm_note_kernel(LGM_KNAME("linattn_bwd_fused_kernel"));
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linattn_bwd_fused_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem);
    attr = true;
  }
  hipLaunchKernelGGL(linattn_bwd_fused_kernel, dim3(blocks), dim3(256), kSmem, s, a);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

// =====================================================================================================================
// Fused tail of the LinearAttention FORWARD (ddpm.py:229-239 + the Residual around the block): for a 128-pixel tile,
//   qs = softmax_d(q) scale;  out[:, h] = qs ctx_h  (four heads);  o2 = to_out[0](out) + bias;  y = RMSNorm(o2) + x
// in ONE launch instead of three (linattn_out_kernel, the to_out GEMM, rmsnorm_fwd) and two round trips of `out` / o2
// through memory.  Products are oriented D[channel][pixel], so a lane owns ONE pixel: its query row comes straight from
// global memory into the lane's operand registers (no LDS staging), its softmax needs one exchange with lane ^ 32, and
// the RMSNorm over the produced channels is lane-local plus that same exchange.  The out^T accumulators of a head ARE
// the B operand of to_out's product: the MFMA contracts over k in whatever order the two operands agree on, and the
// weight fragments are read in the order the accumulator layout dictates (lane half lh, register 4 g + j <-> channel
// h 32 + 8 g + 4 lh + j), so `out` never goes through LDS either.  `out` (needed by to_out's weight gradient) and o2
// (needed by the RMSNorm backward) are still written, but never read back.  LDS holds only to_out's weight and the
// image's four ctx^T; waves are independent between the two barriers of an image change.
namespace {

constexpr int OTP = 128;    // pixels per block item (32 per wave)
constexpr int ALD = 132;    // row stride of the staged weight [c][128]: 132 = 4 (mod 64) -> conflict-free 16-byte rows

struct OArgs {
  const float* qkv;  long pitch;
  const float* ctx;
  const float* wout; const float* bout; const float* g;
  const float* x;    long x_pitch;
  float* ao;         long ao_pitch;
  float* o2;         long o2_pitch;
  float* y;          long y_pitch;
  int n, tiles, items, per;
  float scale;
};

template <int NC>       // produced channels / 32
__global__ __launch_bounds__(256, NC == 2 ? 3 : 1) void linattn_out_fused_kernel(const OArgs p) {
  constexpr int CO = 32 * NC;
  extern __shared__ __align__(16) float sm[];
  float* Ws = sm;                        // to_out[0].weight [CO][128], row stride ALD
  float* Ct = Ws + CO * ALD;             // ctx^T of the image's four heads: [h][e][d], row stride LD
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int it0 = blockIdx.x * p.per, it1 = min(p.items, it0 + p.per);
  if (it0 >= it1) return;
  const float sqrtc = sqrtf((float)CO);

  f32x4 qn[4];
  auto issue_q = [&](int it, int h) {
    const int b = it / p.tiles, i0 = (it % p.tiles) * OTP;
    const int pix = min(i0 + 32 * wid + lr, p.n - 1);
    const float* qp = p.qkv + ((long)b * p.n + pix) * p.pitch + h * DH + 16 * lh;
#pragma unroll
    for (int j = 0; j < 4; ++j) qn[j] = *reinterpret_cast<const f32x4*>(qp + 4 * j);
  };
  issue_q(it0, 0);
#pragma unroll
  for (int u = 0; u < NC * 4; ++u) {      // CO x 128 floats = CO * 32 16-byte pieces
    const int e = tid + 256 * u;
    const int c = e >> 5, k4 = (e & 31) * 4;
    *reinterpret_cast<f32x4*>(Ws + c * ALD + k4) = *reinterpret_cast<const f32x4*>(p.wout + (long)c * HID + k4);
  }
  int b_staged = -1;
  for (int it = it0; it < it1; ++it) {
    const int b = it / p.tiles, i0 = (it % p.tiles) * OTP;
    const int pix = i0 + 32 * wid + lr;
    const bool live = pix < p.n;
    const long row = (long)b * p.n + min(pix, p.n - 1);
    if (b != b_staged) {
      __syncthreads();                 // everybody is done with the previous image's ctx
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = (tid + 256 * u) * 4;             // flat index into [head][d][e]
        const int h = e >> 10, d = (e >> 5) & 31, e0 = e & 31;
        const f32x4 cv = *reinterpret_cast<const f32x4*>(p.ctx + (long)b * HEADS * DH * DH + e);
#pragma unroll
        for (int k = 0; k < 4; ++k) Ct[h * (DH * LD) + (e0 + k) * LD + d] = cv[k];
      }
      __syncthreads();
      b_staged = b;
    }
    // the residual rows of the lane's pixel: fetched a whole item ahead of their use
    f32x4 xres[NC <= 4 ? NC * 4 : 1];
    if constexpr (NC <= 4) {
#pragma unroll
      for (int a = 0; a < NC; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          xres[a * 4 + g] = *reinterpret_cast<const f32x4*>(p.x + row * p.x_pitch + a * 32 + 8 * g + 4 * lh);
    }
    f32x16 acc[NC];
#pragma unroll
    for (int a = 0; a < NC; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
#pragma unroll
    for (int h = 0; h < HEADS; ++h) {
      float qs[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        qs[4 * j] = qn[j][0]; qs[4 * j + 1] = qn[j][1]; qs[4 * j + 2] = qn[j][2]; qs[4 * j + 3] = qn[j][3];
      }
      if (h + 1 < HEADS) issue_q(it, h + 1);
      else if (it + 1 < it1) issue_q(it + 1, 0);
      float mx = qs[0];
#pragma unroll
      for (int k = 1; k < 16; ++k) mx = fmaxf(mx, qs[k]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        qs[k] = __expf(qs[k] - mx);
        sum += qs[k];
      }
      sum += __shfl_xor(sum, 32, 64);
      const float inv = p.scale / sum;
      float ca[16];
      load16(Ct + h * (DH * LD) + lr * LD + 16 * lh, ca);
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 16; ++s) o = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[s], qs[s] * inv, o, 0, 0, 0);
      // out^T[e][pixel]: this lane's pixel, register 4 g + j = channel h * 32 + 8 g + 4 lh + j
      if (live) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(p.ao + row * p.ao_pitch + h * DH + 8 * g + 4 * lh) =
              f32x4{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
      }
      // o2^T[c][pixel] += Wout[c][k] out^T[k][pixel] over the head's 32 k, in the accumulators' own order
#pragma unroll
      for (int a = 0; a < NC; ++a) {
        const float* wp = Ws + (a * 32 + lr) * ALD + h * DH + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + 8 * g);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j], o[4 * g + j], acc[a], 0, 0, 0);
        }
      }
    }
    // ---- bias, RMSNorm over the CO channels of the lane's pixel (the other half sits in lane ^ 32), residual ----
    float ss = 0.f;
#pragma unroll
    for (int a = 0; a < NC; ++a)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bout + a * 32 + 8 * g + 4 * lh);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[a][4 * g + k] += bv[k];
          ss += acc[a][4 * g + k] * acc[a][4 * g + k];
        }
      }
    ss += __shfl_xor(ss, 32, 64);
    const float rinv = sqrtc / fmaxf(sqrtf(ss), 1e-12f);
    if (live) {
#pragma unroll
      for (int a = 0; a < NC; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = a * 32 + 8 * g + 4 * lh;
          const f32x4 v = {acc[a][4 * g], acc[a][4 * g + 1], acc[a][4 * g + 2], acc[a][4 * g + 3]};
          const f32x4 gv = *reinterpret_cast<const f32x4*>(p.g + c);
          f32x4 xv;
          if constexpr (NC <= 4) xv = xres[a * 4 + g];
          else xv = *reinterpret_cast<const f32x4*>(p.x + row * p.x_pitch + c);
          *reinterpret_cast<f32x4*>(p.o2 + row * p.o2_pitch + c) = v;
          *reinterpret_cast<f32x4*>(p.y + row * p.y_pitch + c) = v * gv * rinv + xv;
        }
    }
  }
}

template <int NC>
int launch_out(const OArgs& a, int blocks, hipStream_t s) {
  const size_t smem = ((size_t)32 * NC * ALD + HEADS * DH * LD) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linattn_out_fused_kernel<NC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  hipLaunchKernelGGL(linattn_out_fused_kernel<NC>, dim3(blocks), dim3(256), smem, s, a);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

}  // namespace

// 1: built for this layer; 2: built AND measured faster than the three separate launches (256 produced channels sit on
// 8 x 8 maps, where a 128-pixel item is half empty and the 128 KB weight is staged per block: 35 vs 30 us)
extern "C" int64_t lgm_linattn_fwd_fused_supported(int heads, int dim_head, int Cout) {
  if (heads != HEADS || dim_head != DH) return 0;
  return Cout == 64 || Cout == 128 ? 2 : Cout == 256 ? 1 : 0;
}

int lgm_linattn_out_fused_launch(const float* qkv, long pitch, const float* ctx, const float* wout, const float* bout,
                                 const float* g, const float* x, long x_pitch, float* ao, long ao_pitch, float* o2,
                                 long o2_pitch, float* y, long y_pitch, int B, int n, int Cout, float scale, hipStream_t s) {
  OArgs a;
  a.qkv = qkv; a.pitch = pitch; a.ctx = ctx; a.wout = wout; a.bout = bout; a.g = g; a.x = x; a.x_pitch = x_pitch;
  a.ao = ao; a.ao_pitch = ao_pitch; a.o2 = o2; a.o2_pitch = o2_pitch; a.y = y; a.y_pitch = y_pitch;
  a.n = n; a.scale = scale;
  a.tiles = lgm_cdiv(n, OTP);
  a.items = B * a.tiles;
  const int slots = Cout == 64 ? 768 : 256;
  const int nb = a.items < slots ? a.items : slots;
  a.per = lgm_cdiv(a.items, nb);
  const int blocks = lgm_cdiv(a.items, a.per);
  lgm_note_kernel(Cout == 64 ? LGM_KNAME("linattn_out_fused_kernel<2>") : Cout == 128 ? LGM_KNAME("linattn_out_fused_kernel<4>")
                                                                             : LGM_KNAME("linattn_out_fused_kernel<8>"));
  if (Cout == 64) return launch_out<2>(a, blocks, s);
  if (Cout == 128) return launch_out<4>(a, blocks, s);
  return launch_out<8>(a, blocks, s);
}

// =====================================================================================================================
// Fused head of the attention blocks' forward (ddpm.py:224-225, :262-263): xn = RMSNorm(x), qkv = to_qkv(xn) (1x1
// convolution, no bias) in one launch.  Per-pixel work with the same conventions as above: a lane owns one pixel, its
// 16-byte pieces of the row arrive straight in registers, the norm is lane-local plus one exchange with lane ^ 32, and
// the normalised values ARE the B operand of the product (k dealt out as a 32 + 8 g + 4 lh + j, the weight fragments read
// in that order).  The 384 produced channels are split into three slices of 128 (q | k | v) over blockIdx.y so that a
// block's weight slice fits LDS several times per CU; every slice recomputes the norm of its pixels (x comes from L2 the
// second and third time), slice 0 writes xn (the weight gradient needs it).
namespace {

struct QArgs {
  const float* x;  long x_pitch;
  const float* g;
  const float* w;                 // [3 * HID][C]
  float* xn;       long xn_pitch;
  float* qkv;      long q_pitch;
  long P;                         // pixel rows
  int items, per;
};

template <int NCI>      // input channels / 32
__global__ __launch_bounds__(256, NCI == 2 ? 3 : NCI == 4 ? 2 : 1) void rms_qkv_fused_kernel(const QArgs p) {
  constexpr int CI = 32 * NCI;
  constexpr int WL = CI + 4;          // = 4 (mod 64): conflict-free 16-byte rows
  extern __shared__ __align__(16) float sm[];
  float* Ws = sm;                     // the slice's weight rows [128][WL]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  // the three slices of an item range read the same rows of x: on consecutive slots of ONE XCD (block id = xcd + 8 slot),
  // so that the second and third reader find them in that XCD's L2 (consecutive ids sit on different XCDs)
  int blk, slice;
  {
    const int nblk = (int)gridDim.x / 3, bid = (int)blockIdx.x;
    if (nblk % 8 == 0) {
      const int xcd = bid & 7, slot = bid >> 3;
      blk = (slot / 3) * 8 + xcd;
      slice = slot % 3;
    } else {
      blk = bid / 3;
      slice = bid % 3;
    }
  }
  const int it0 = blk * p.per, it1 = min(p.items, it0 + p.per);
  if (it0 >= it1) return;
  const float sqrtc = sqrtf((float)CI);
  f32x4 xr[NCI * 4];
  auto issue_x = [&](int it) {
    const long pix = (long)it * 128 + 32 * wid + lr;
    const long row = pix < p.P ? pix : p.P - 1;
#pragma unroll
    for (int a = 0; a < NCI; ++a)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        xr[a * 4 + g] = *reinterpret_cast<const f32x4*>(p.x + row * p.x_pitch + a * 32 + 8 * g + 4 * lh);
  };
  issue_x(it0);
#pragma unroll
  for (int u = 0; u < NCI * 4; ++u) {        // 128 rows x CI floats = 128 * CI / 4 16-byte pieces
    const int e = tid + 256 * u;
    const int r = e / (CI / 4), c4 = (e % (CI / 4)) * 4;
    *reinterpret_cast<f32x4*>(Ws + r * WL + c4) = *reinterpret_cast<const f32x4*>(p.w + (long)(slice * 128 + r) * CI + c4);
  }
  __syncthreads();
  for (int it = it0; it < it1; ++it) {
    const long pix = (long)it * 128 + 32 * wid + lr;
    const bool live = pix < p.P;
    const long row = live ? pix : p.P - 1;
    f32x4 xn[NCI * 4];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NCI * 4; ++k) {
      xn[k] = xr[k];
      ss += xn[k][0] * xn[k][0] + xn[k][1] * xn[k][1] + xn[k][2] * xn[k][2] + xn[k][3] * xn[k][3];
    }
    if (NCI <= 4 && it + 1 < it1) issue_x(it + 1);
    ss += __shfl_xor(ss, 32, 64);
    const float inv = sqrtc / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int a = 0; a < NCI; ++a)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(p.g + a * 32 + 8 * g + 4 * lh);
        xn[a * 4 + g] = xn[a * 4 + g] * gv * inv;
        if (slice == 0 && live)
          *reinterpret_cast<f32x4*>(p.xn + row * p.xn_pitch + a * 32 + 8 * g + 4 * lh) = xn[a * 4 + g];
      }
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {        // one output tile at a time: the operand fragments of all four would not fit
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* wp = Ws + (t * 32 + lr) * WL + 4 * lh;
#pragma unroll
      for (int a = 0; a < NCI; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + a * 32 + 8 * g);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j], xn[a * 4 + g][j], acc, 0, 0, 0);
        }
      // (D[pixel][channel] with dword stores that cover full 128-byte row segments was measured SLOWER here - 90 vs 77 us
      // at 32 x 32 maps, B = 128: sixteen store instructions per tile instead of four)
      if (live) {
        float* o = p.qkv + row * p.q_pitch + slice * 128 + t * 32 + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(o + 8 * g) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
      }
    }
    if (NCI > 4 && it + 1 < it1) issue_x(it + 1);      // wide layers: no register room for a row in flight
  }
}

template <int NCI>
int launch_qkv(const QArgs& a, int blocks, hipStream_t s) {
  const size_t smem = (size_t)128 * (32 * NCI + 4) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rms_qkv_fused_kernel<NCI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  hipLaunchKernelGGL(rms_qkv_fused_kernel<NCI>, dim3(blocks * 3), dim3(256), smem, s, a);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

}  // namespace

// 1: built for this layer; 2: built and measured faster than RMSNorm + GEMM as two launches (64 channels, where the
// GEMM is bound by its 6x larger output: 77 vs 87 us at 131 k pixel rows; wider layers sit on small maps: 50 vs 33 us)
extern "C" int64_t lgm_rms_qkv_fused_supported(int C, int N) {
  if (N != 3 * HID) return 0;
  return C == 64 ? 2 : C == 128 || C == 256 ? 1 : 0;
}

// xn = RMSNorm_g(x) (ddpm.py:115-121), qkv = xn W^T  (W = to_qkv.weight [384][C], no bias)
extern "C" int lgm_rms_qkv_fused(const float* x, int64_t x_pitch, const float* g, const float* w, int C, int N,
                                 int64_t npix, float* xn, int64_t xn_pitch, float* qkv, int64_t qkv_pitch, void* stream) {
  LGM_REQUIRE(lgm_rms_qkv_fused_supported(C, N), "rms_qkv_fused: C=%d N=%d unsupported", C, N);
  LGM_REQUIRE(x && g && w && xn && qkv && npix > 0, "rms_qkv_fused: null pointer / empty");
  LGM_REQUIRE(x_pitch % 4 == 0 && xn_pitch % 4 == 0 && qkv_pitch % 4 == 0 && lgm_aligned16(x) && lgm_aligned16(g) &&
                  lgm_aligned16(w) && lgm_aligned16(xn) && lgm_aligned16(qkv),
              "rms_qkv_fused: 16-byte aligned operands required");
  QArgs a;
  a.x = x; a.x_pitch = x_pitch; a.g = g; a.w = w; a.xn = xn; a.xn_pitch = xn_pitch; a.qkv = qkv; a.q_pitch = qkv_pitch;
  a.P = npix;
  a.items = lgm_cdiv(npix, 128);
  const int slots = C == 64 ? 256 : C == 128 ? 170 : 85;      // x 3 slices: 3 / 2 / 1 blocks per CU
  const int nb = a.items < slots ? a.items : slots;
  a.per = lgm_cdiv(a.items, nb);
  const int blocks = lgm_cdiv(a.items, a.per);
  lgm_note_kernel(C == 64 ? LGM_KNAME("rms_qkv_fused_kernel<2>") : C == 128 ? LGM_KNAME("rms_qkv_fused_kernel<4>") : LGM_KNAME("rms_qkv_fused_kernel<8>"));
  hipStream_t s = (hipStream_t)stream;
  if (C == 64) return launch_qkv<2>(a, blocks, s);
  if (C == 128) return launch_qkv<4>(a, blocks, s);
  return launch_qkv<8>(a, blocks, s);
}
