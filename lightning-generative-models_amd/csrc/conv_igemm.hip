// Implicit-GEMM convolution family for gfx950 on the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD = the fp32 roof of the chip).
//
//   lgm_conv_xy    : Y = conv(X, W)             GEMM  M = B*Ho*Wo, N = Nw, K = T*Cw
//   lgm_conv_yx    : X = conv_transpose(Y, W)   GEMM  M = B*H*W,   N = Cw, K = T*Nw
//   lgm_conv_wgrad : gW = Y^T * gather(X)       GEMM  M = Nw, N = T*Cw, K = B*Ho*Wo (split-K)
//
// Tiling (xy / yx): 256 threads = 4 waves in a 2x2 grid; each wave owns TM x TN MFMA tiles of
// 32x32; BK = 32.  Global -> register prefetch of chunk k+1 overlaps the MFMAs of chunk k; LDS is
// double buffered so one barrier per chunk suffices.  A (gathered activations) and B (weights,
// k-contiguous) are staged [row][BK+4] so every fragment read is one conflict-free ds_read_b128:
// lane (r = l&31, h = l>>5) reads k = 8*kc + 4*h .. +3 of its row, and MFMA step s consumes
// element s of that float4 from both operands (the k -> (half, step) assignment is arbitrary as
// long as A and B agree; fp32 accumulation order inside a chunk is fixed => deterministic).
#include <stdlib.h>
#include <string.h>

#include "lgm_common.h"

// specialised 3x3 / stride 1 / pad 1 kernels (conv3x3.hip)
bool lgm_conv3x3_supported(const LgmConvGeom* g, int gather_channels, int out_channels);
int lgm_conv3x3_splits(const LgmConvGeom* g, int gather_channels, int out_channels);
int lgm_splitk_reduce_launch(const float* ws, long ws_stride, int splits, const float* bias, const float* res,
                             long res_pitch, float* out, long out_pitch, long M, int N, hipStream_t s);
int lgm_conv3x3_launch(int mode, const LgmConvGeom* g, const float* a, long a_pitch, const float* w,
                       const float* bias, const float* res, long res_pitch, float* out, long out_pitch,
                       void* workspace, long workspace_bytes, hipStream_t s);
bool lgm_wgrad3x3_supported(const LgmConvGeom* g);
void lgm_wgrad3x3_plan(const LgmConvGeom* g, int* splits, int* tps, int* total_ts);
bool lgm_wgrad1x1_supported(const LgmConvGeom* g, long y_pitch, long x_pitch);
void lgm_wgrad1x1_plan(const LgmConvGeom* g, int* splits, int* chunks_per_split);
int lgm_wgrad1x1_launch(const LgmConvGeom* g, const float* y, long y_pitch, const float* x, long x_pitch, float* out,
                        float* bias_out, float beta, long slab, int splits, int chunks_per_split, hipStream_t s);
int lgm_wgrad3x3_launch(const LgmConvGeom* g, const float* y, long y_pitch, const float* x, long x_pitch,
                        float* out, float* bias_out, float beta, long slab, int splits, int tps, int total_ts,
                        hipStream_t s);

// Winograd F(2x2,3x3) weight gradient (winograd.hip): always through slabs + the fixed-order reducer
bool lgm_wino_wgrad_supported(const LgmConvGeom* g);
// F(4x4,3x3) weight gradient of the large-map layers (csrc/winograd4_wgrad.hip)
bool lgm_wino4_wgrad_use(const LgmConvGeom* g);
void lgm_wino4_wgrad_plan(const LgmConvGeom* g, long budget, int* splits, int* gps, int* total_groups);
int lgm_wino4_wgrad_launch(const LgmConvGeom* g, const float* y, long y_pitch, const float* x, long x_pitch, float* out,
                           int bias, long slab, int splits, int gps, int total, hipStream_t s);
void lgm_wino_wgrad_plan(const LgmConvGeom* g, int* splits, int* cps, int* total_chunks);
int lgm_wino_wgrad_launch(const LgmConvGeom* g, const float* y, long y_pitch, const float* x, long x_pitch, float* out,
                          int bias, long slab, int splits, int cps, int total_chunks, hipStream_t s);
static bool use_wino() {      // LGM_NO_WINO=1: direct fp32 MFMA kernels for the 3x3 layers (A/B comparisons)
  static const bool off = getenv("LGM_NO_WINO") != nullptr && getenv("LGM_NO_WINO")[0] == '1';
  return !off;
}

// 1x1 convolutions with a resident weight slice, X streamed (gemm_stream.hip)
bool lgm_gemm_stream_supported(long M, int N, int K, long x_pitch, long out_pitch, long res_pitch);
int lgm_gemm_stream_launch(const float* x, long x_pitch, const float* w, const float* bias, const float* res,
                           long res_pitch, float* out, long out_pitch, long M, int N, int K, hipStream_t s);
static bool use_gstream() {
  static const bool off = getenv("LGM_NO_GSTREAM") != nullptr;   // A/B switch
  return !off;
}
// short-reduction 1x1 convolutions with a resident activation tile (gemm_rows.hip)
bool lgm_gemm_rows_supported(long M, int N, int K);
int lgm_gemm_rows_launch(const float* x, long x_pitch, const float* w, const float* bias, const float* res,
                         long res_pitch, float* out, long out_pitch, long M, int N, int K, hipStream_t s);

// the specialised kernels store 16 bytes per lane: outputs / residual / bias must allow it
static bool wide_ok(const float* out, long out_pitch, const float* res, long res_pitch, const float* bias) {
  return lgm_aligned16(out) && out_pitch % 4 == 0 && (!res || (lgm_aligned16(res) && res_pitch % 4 == 0)) &&
         (!bias || lgm_aligned16(bias));
}

static bool use_3x3() {   // LGM_NO_3X3=1 forces the generic implicit-GEMM path (A/B comparisons)
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("LGM_NO_3X3");
    v = (e && e[0] == '1') ? 0 : 1;
  }
  return v == 1;
}

namespace {

constexpr int BK = 32;
constexpr int LDA = BK + 4;  // floats; 16B-aligned rows, conflict-free b128 reads (see header)

enum { MODE_XY = 0, MODE_YX = 1 };

__device__ __forceinline__ int lgm_xcd_swizzle(int bid, int nb) {
  return (nb % 8 == 0) ? (bid % 8) * (nb / 8) + bid / 8 : bid;
}
// opt-in (LGM_XCD_SWZ=1): measured 0.3 % SLOWER on the WGAN-GP and DDPM steps - the N-siblings' A tiles are served by the
// Infinity Cache already, and the contiguous ranges cost some balance at the tail
static int swz_on() {
  static const int on = getenv("LGM_XCD_SWZ") != nullptr;
  return on;
}

struct IgemmArgs {
  const float* a;     // gathered activations (X for XY, Y for YX)
  const float* w;     // [Nw][T][Cw]
  const float* bias;  // [N] or null
  const float* res;   // [M, N] pitch res_pitch or null
  float* out;         // [M, N] pitch out_pitch
  long a_pitch, res_pitch, out_pitch;
  int B, H, W, Cw, Ho, Wo, Nw, KH, KW, stride, pad;
  int M, N, K;        // GEMM sizes
  int Cg;             // channels of the gathered tensor (Cw for XY, Nw for YX)
  int tiles_m, tiles_n;
  int splits, kchunk; // split-K: `splits` partial products over K ranges of kchunk (multiple of BK) ...
  float* ws;          // ... written to ws[split][M][N], summed in fixed order by the reducer
  // Strided input gradient / transposed convolution (MODE_YX, stride s > 1) by PHASE decomposition: the
  // output pixels of one residue class (ih % s, iw % s) only ever meet the KH/s x KW/s taps of matching
  // parity, so each class is a dense GEMM with K = (KH/s)(KW/s)Nw instead of KH*KW*Nw with (1 - 1/s^2) of
  // the products multiplied by zero.  phases = s*s (or 1 = off); then M, K, tiles_m describe ONE class.
  int phases, KHs, KWs;
  int wide;           // output / residual / bias are 16-byte aligned with pitches % 4 == 0 (wide epilogue allowed)
  int swz;            // stand-alone launches: XCD-aware block order (igemm_kernel)
  // BatchNorm statistics of the OUTPUT folded into the epilogue (full tiles, no bias / residual / split-K): every
  // workgroup adds, per output column, the sum of its BM rows and their squared deviations from the tile's own mean
  // to stats[row tile][3][N] = (sum, M2, BM) - exactly what bn_stats_stage1 writes, so lgm_bn_stats_from_tiles
  // (Chan's combination in the second reduction stage) takes it from there and the read pass over the
  // activation disappears.
  float* stats;
  // post-op of the epilogue (lgm_conv_xy_post / lgm_conv_yx_post): v = act(acc + bias + res), then the ReLU /
  // LeakyReLU backward mask of a SAVED activation m: v *= (m > 0 ? 1 : mask_slope)
  int act;            // 0 none, 3 ReLU, 4 LeakyReLU(slope)
  float slope;
  const float* mask;
  long mask_pitch;
  float mask_slope;
  // BatchNorm backward sums of the finished output gn (LgmPostOp.bn_*): per full row tile (sum gn, sum gn * xhat, 0) per
  // column into bn_part[row tile][3][N], xhat = (bn_a - bn_mean) * bn_rstd at the output's own (row, column).  Host
  // guarantees: wide epilogue, full tiles everywhere, splits == 1.
  const float* bn_a;
  long bn_a_pitch;
  const float* bn_mean;
  const float* bn_rstd;
  float* bn_part;
};

__device__ __forceinline__ float lgm_post_act(float v, int act, float slope) {
  return act == 0 ? v : (v > 0.f ? v : (act == 4 ? slope * v : 0.f));
}

// kernel body as a device function of (arguments, logical block id): its own launch (igemm_kernel) or the first block
// range of gemm_bwd_pair_kernel (input gradient + weight gradient of one layer in ONE launch, see below)
template <int MODE, int BM, int BN, int TM, int TN, bool UNI>
__device__ __forceinline__ void igemm_body(const IgemmArgs& p, const int bidx) {
  static_assert(BM == 64 * TM && BN == 64 * TN, "2x2 wave grid");
  constexpr int A_PER = BM / 32;  // float4 per thread per chunk
  constexpr int B_PER = BN / 32;
  // B tile: XY -> [BN][LDA] (k contiguous), YX -> [BK][BN] (n contiguous)
  constexpr int A_TILE = BM * LDA;
  constexpr int B_TILE = (MODE == MODE_XY) ? BN * LDA : BK * BN;
  extern __shared__ __align__(16) float smem[];
  float* As = smem;                 // 2 * A_TILE
  float* Bs = smem + 2 * A_TILE;    // 2 * B_TILE

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;

  // tile mapping: consecutive blocks walk N first (share the A tile through L2)
  const int split = bidx % p.splits;
  int tile = bidx / p.splits;
  const int per_phase = p.tiles_m * p.tiles_n;
  const int pc = tile / per_phase;               // residue class (0 when the decomposition is off)
  tile -= pc * per_phase;
  const int ph = pc / p.stride, pw = pc - ph * p.stride;
  const int kh0 = (ph + p.pad) % p.stride, kw0 = (pw + p.pad) % p.stride;
  const int Hq = p.H / p.stride, Wq = p.W / p.stride;
  const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
  const int k_begin = split * p.kchunk;
  const int k_end = min(p.K, k_begin + p.kchunk);
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-thread A-row decode (fixed over the K loop) ---------------------------
  const int acol = tid & 7;    // float4 column inside the 32-wide chunk
  const int arow = tid >> 3;   // 0..31
  int r_pix[A_PER], r_h[A_PER], r_w[A_PER];
  bool r_ok[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int m = m0 + arow + 32 * i;
    r_ok[i] = m < p.M;
    const int mm = r_ok[i] ? m : 0;
    if (MODE == MODE_XY) {
      const int ow = mm % p.Wo, t = mm / p.Wo;
      const int oh = t % p.Ho, b = t / p.Ho;
      r_pix[i] = b * p.H * p.W;
      r_h[i] = oh * p.stride - p.pad;
      r_w[i] = ow * p.stride - p.pad;
    } else if (p.phases > 1) {
      const int iwq = mm % Wq, t = mm / Wq;
      const int ihq = t % Hq, b = t / Hq;
      r_pix[i] = b * p.Ho * p.Wo;
      r_h[i] = ihq * p.stride + ph + p.pad;
      r_w[i] = iwq * p.stride + pw + p.pad;
    } else {
      const int iw = mm % p.W, t = mm / p.W;
      const int ih = t % p.H, b = t / p.H;
      r_pix[i] = b * p.Ho * p.Wo;
      r_h[i] = ih + p.pad;
      r_w[i] = iw + p.pad;
    }
  }

  // ---- UNI: Cg % BK == 0, so a K chunk lies inside ONE tap: the tap decode is wave-uniform (SALU, advanced
  // incrementally) and every load is a raw buffer load  descriptor + per-thread byte offset + scalar offset.
  // Per chunk a thread spends ~6 VALU instructions per gathered row (two bound checks, one select) instead of
  // the ~30 of the general decode: beside fp32 MFMAs each VALU instruction costs issue slots the matrix pipe
  // wants (DESIGN finding 11).  Zero padding = an offset at the descriptor's range (hardware returns 0).
  unsigned va[A_PER], vb[B_PER];
  int q_h[A_PER], q_w[A_PER];
  unsigned nrec_a = 0;
  __amdgpu_buffer_rsrc_t rsrc_a, rsrc_b;
  int u_c0 = 0, u_ta = 0, u_tb = 0;        // scalar chunk state: channel origin, tap row / column
  if constexpr (UNI) {
    auto make_rsrc = [](const float* base, unsigned nrec) {
      const unsigned long long ab = reinterpret_cast<unsigned long long>(base);
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ab);
      const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(ab >> 32));
      return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                               __builtin_amdgcn_readfirstlane(nrec), 0x00020000);
    };
    const int KWu = (MODE == MODE_YX) ? p.KWs : p.KW;
    long shift;                             // descriptor origin = a - shift elements (keeps both offsets >= 0)
    if (MODE == MODE_XY) {
      shift = (long)(p.pad * p.W + p.pad) * p.a_pitch;
      nrec_a = (unsigned)((((long)p.B * p.H * p.W) * p.a_pitch + shift) * 4);
    } else {
      shift = (long)((p.KHs - 1) * p.Wo + (p.KWs - 1)) * p.a_pitch;
      nrec_a = (unsigned)((((long)p.B * p.Ho * p.Wo + (long)(p.KHs + 1) * p.Wo + p.KWs) * p.a_pitch + shift) * 4);
    }
    rsrc_a = make_rsrc(p.a - shift, nrec_a);
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      if (MODE == MODE_XY) {
        q_h[i] = r_h[i];
        q_w[i] = r_w[i];
        va[i] = (unsigned)((((long)r_pix[i] + (long)(r_h[i] + p.pad) * p.W + (r_w[i] + p.pad)) * p.a_pitch + acol * 4) * 4);
      } else {
        q_h[i] = p.phases > 1 ? (r_h[i] - kh0) / p.stride : r_h[i];
        q_w[i] = p.phases > 1 ? (r_w[i] - kw0) / p.stride : r_w[i];
        va[i] = (unsigned)((((long)r_pix[i] + (long)q_h[i] * p.Wo + q_w[i]) * p.a_pitch + acol * 4) * 4);
      }
      if (!r_ok[i]) {
        q_h[i] = -(1 << 20);               // fails every bound check
        q_w[i] = -(1 << 20);
      }
    }
    const unsigned nrec_b = (unsigned)((long)p.Nw * p.KH * p.KW * p.Cw * 4);
    rsrc_b = make_rsrc(p.w, nrec_b);
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      if (MODE == MODE_XY) {
        const int n = n0 + arow + 32 * i;
        vb[i] = n < p.N ? (unsigned)(((long)n * p.K + acol * 4) * 4) : nrec_b;
      } else {
        constexpr int TPR = BN / 4;
        constexpr int RPP = 256 / TPR;
        const int ncol = (tid % TPR) * 4, krow = tid / TPR;
        vb[i] = n0 + ncol < p.N ? (unsigned)(((long)(krow + RPP * i) * (p.KH * p.KW) * p.Cw + n0 + ncol) * 4) : nrec_b;
      }
    }
    const int tap0 = k_begin / p.Cg;
    u_c0 = k_begin - tap0 * p.Cg;
    u_ta = tap0 / KWu;
    u_tb = tap0 - u_ta * KWu;
  }

  f32x4 ra[A_PER], rb[B_PER];

  auto load_chunk_uni = [&]() {
    const int KWu = (MODE == MODE_YX) ? p.KWs : p.KW;
    unsigned soff_a, soff_b;
    if (MODE == MODE_XY) {
      soff_a = (unsigned)((((long)u_ta * p.W + u_tb) * p.a_pitch + u_c0) * 4);
      soff_b = (unsigned)(((u_ta * p.KW + u_tb) * p.Cg + u_c0) * 4);
    } else {
      soff_a = (unsigned)((((long)(p.KHs - 1 - u_ta) * p.Wo + (p.KWs - 1 - u_tb)) * p.a_pitch + u_c0) * 4);
      const int tp = p.phases > 1 ? (kh0 + u_ta * p.stride) * p.KW + kw0 + u_tb * p.stride : u_ta * p.KW + u_tb;
      soff_b = (unsigned)((((long)u_c0 * (p.KH * p.KW) + tp) * p.Cw) * 4);
    }
    soff_a = __builtin_amdgcn_readfirstlane(soff_a);
    soff_b = __builtin_amdgcn_readfirstlane(soff_b);
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      bool ok;
      if (MODE == MODE_XY)
        ok = (unsigned)(q_h[i] + u_ta) < (unsigned)p.H && (unsigned)(q_w[i] + u_tb) < (unsigned)p.W;
      else
        ok = (unsigned)(q_h[i] - u_ta) < (unsigned)p.Ho && (unsigned)(q_w[i] - u_tb) < (unsigned)p.Wo;
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? va[i] : nrec_a, soff_a, 0));
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, vb[i], soff_b, 0));
    u_c0 += BK;
    if (u_c0 == p.Cg) {
      u_c0 = 0;
      if (++u_tb == KWu) {
        u_tb = 0;
        ++u_ta;
      }
    }
  };

  auto load_chunk = [&](int k0) {
    // ---- A: one tap/channel decode per thread per chunk ----
    const int k = k0 + acol * 4;
    const bool kok = k < k_end;
    const int tap = kok ? k / p.Cg : 0;
    const int c = k - tap * p.Cg;
    int kh, kw;
    if (MODE == MODE_YX && p.phases > 1) {
      const int ta = tap / p.KWs;
      kh = kh0 + ta * p.stride;
      kw = kw0 + (tap - ta * p.KWs) * p.stride;
    } else {
      kh = tap / p.KW;
      kw = tap - kh * p.KW;
    }
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      bool ok = kok && r_ok[i];
      long off = 0;
      if (MODE == MODE_XY) {
        const int ih = r_h[i] + kh, iw = r_w[i] + kw;
        ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        off = (long)(r_pix[i] + ih * p.W + iw) * p.a_pitch + c;
      } else {
        int th = r_h[i] - kh, tw = r_w[i] - kw;
        ok = ok && th >= 0 && tw >= 0;
        if (p.stride != 1) {
          if (p.phases == 1) ok = ok && (th % p.stride == 0) && (tw % p.stride == 0);
          th /= p.stride;         // exact inside a residue class
          tw /= p.stride;
        }
        ok = ok && th < p.Ho && tw < p.Wo;
        off = (long)(r_pix[i] + th * p.Wo + tw) * p.a_pitch + c;
      }
      ra[i] = ok ? *reinterpret_cast<const f32x4*>(p.a + off) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- B ----
    if (MODE == MODE_XY) {
      const int kb = k0 + acol * 4;
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int n = n0 + arow + 32 * i;
        const bool ok = (n < p.N) && (kb < k_end);
        rb[i] = ok ? *reinterpret_cast<const f32x4*>(p.w + (long)n * p.K + kb)
                   : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
      constexpr int TPR = BN / 4;          // threads per k-row
      constexpr int RPP = 256 / TPR;       // k-rows per pass
      const int ncol = (tid % TPR) * 4;
      const int krow = tid / TPR;
      const int T = p.KH * p.KW;
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int kk = k0 + krow + RPP * i;
        const bool ok = (kk < k_end) && (n0 + ncol < p.N);
        int tp = ok ? kk / p.Cg : 0;
        const int nn = kk - tp * p.Cg;
        if (p.phases > 1) {
          const int ta = tp / p.KWs;
          tp = (kh0 + ta * p.stride) * p.KW + kw0 + (tp - ta * p.KWs) * p.stride;
        }
        rb[i] = ok ? *reinterpret_cast<const f32x4*>(p.w + ((long)nn * T + tp) * p.Cw + n0 + ncol)
                   : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  auto store_chunk = [&](int buf) {
    float* as = As + buf * A_TILE;
    float* bs = Bs + buf * B_TILE;
#pragma unroll
    for (int i = 0; i < A_PER; ++i)
      *reinterpret_cast<f32x4*>(as + (arow + 32 * i) * LDA + acol * 4) = ra[i];
    if (MODE == MODE_XY) {
#pragma unroll
      for (int i = 0; i < B_PER; ++i)
        *reinterpret_cast<f32x4*>(bs + (arow + 32 * i) * LDA + acol * 4) = rb[i];
    } else {
      constexpr int TPR = BN / 4;
      constexpr int RPP = 256 / TPR;
      const int ncol = (tid % TPR) * 4;
      const int krow = tid / TPR;
#pragma unroll
      for (int i = 0; i < B_PER; ++i)
        *reinterpret_cast<f32x4*>(bs + (krow + RPP * i) * BN + ncol) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (k_end - k_begin + BK - 1) / BK;
  if constexpr (UNI) load_chunk_uni(); else load_chunk(k_begin);
  store_chunk(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      if constexpr (UNI) load_chunk_uni(); else load_chunk(k_begin + (kt + 1) * BK);
    }

    const float* as = As + cur * A_TILE + (wm * 32 * TM + lr) * LDA + lh * 4;
    const float* bs;
    if (MODE == MODE_XY)
      bs = Bs + cur * B_TILE + (wn * 32 * TN + lr) * LDA + lh * 4;
    else
      bs = Bs + cur * B_TILE + (lh * 4) * BN + wn * 32 * TN + lr;

#pragma unroll
    for (int kc = 0; kc < BK / 8; ++kc) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDA + kc * 8);
      if (MODE == MODE_XY) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LDA + kc * 8);
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
          for (int s = 0; s < 4; ++s) fb[j][s] = bs[(kc * 8 + s) * BN + j * 32];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
    }

    if (kt + 1 < nk) store_chunk(cur ^ 1);
    __syncthreads();
  }

  // GEMM row -> output pixel (identity unless the rows enumerate one residue class)
  auto out_row = [&](int m) -> long {
    if (MODE == MODE_YX && p.phases > 1) {
      const int iwq = m % Wq, t = m / Wq;
      const int ihq = t % Hq, b = t / Hq;
      return (long)(b * p.H + ihq * p.stride + ph) * p.W + iwq * p.stride + pw;
    }
    return (long)m;
  };
  // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (p.splits > 1) {   // partial product -> workspace slab; bias/residual are added by the reducer
    float* slab = p.ws + (long)split * p.M * p.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * 32 * TN + j * 32 + lr;
      if (n >= p.N) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < p.M) slab[(long)m * p.N + n] = acc[i][j][r];
        }
    }
    return;
  }
  if (p.stats) {      // host guarantees: full tiles everywhere, wide epilogue, splits == 1, no bias / residual
    // C/D map: a lane holds, of column lr of tile (i, j), rows i*32 + (r&3) + 8*(r>>2) + 4*lh
    float* st = smem;                       // [2 passes][wm][wn][TN][32]; the K loop ended with a barrier
    float colsum[TN], part[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) v += acc[i][j][r];
      v += __shfl_xor(v, 32, 64);
      if (lh == 0) st[((wm * 2 + wn) * TN + j) * 32 + lr] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      colsum[j] = st[((0 * 2 + wn) * TN + j) * 32 + lr] + st[((1 * 2 + wn) * TN + j) * 32 + lr];
      const float mu = colsum[j] * (1.f / (float)BM);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = acc[i][j][r] - mu;
          q += d * d;
        }
      q += __shfl_xor(q, 32, 64);
      part[j] = q;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j)
      if (lh == 0) st[((wm * 2 + wn) * TN + j) * 32 + lr] = part[j];
    __syncthreads();
    if (wm == 0 && lh == 0) {
      float* o = p.stats + ((long)(pc * p.tiles_m + tm) * 3) * p.N;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * 32 * TN + j * 32 + lr;
        o[n] = colsum[j];
        o[p.N + n] = st[((0 * 2 + wn) * TN + j) * 32 + lr] + st[((1 * 2 + wn) * TN + j) * 32 + lr];
        o[2 * p.N + n] = (float)BM;
      }
    }
  }
  if (p.wide && m0 + BM <= p.M && n0 + BN <= p.N) {
    // full tile, 16-byte aligned output: wave-private LDS transpose (the A/B tiles are dead), residual
    // rows loaded up front, unconditional 16-byte stores (see lgm_common.h; a conditional dword store per
    // element serialises on s_waitcnt vmcnt(0) and is issue-bound)
    __syncthreads();
    float* Ts = smem + wid * LGM_TS_FLOATS;
    float* const Bst = smem + 4 * LGM_TS_FLOATS;       // BatchNorm sums: [2 sums][wm][wn][TN][32 columns] behind the four wave tiles
    // the epilogue's staging (four wave tiles + the sums) reuses the main loop's operand buffers: it must fit them
    static_assert(2 * (A_TILE + B_TILE) >= 4 * LGM_TS_FLOATS + 2 * 2 * 2 * TN * 32,
                  "igemm epilogue staging (wave tiles + BatchNorm sums) exceeds the operand buffers");
    const bool bnred = p.bn_part != nullptr;           // kernel argument: wave-uniform
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nc = n0 + wn * 32 * TN + j * 32 + (lane & 7) * 4;
      const f32x4 bv4 = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 bs1 = {0.f, 0.f, 0.f, 0.f}, bs2 = {0.f, 0.f, 0.f, 0.f}, bmu = bs1, brs = bs1;
      if (bnred) {
        bmu = *reinterpret_cast<const f32x4*>(p.bn_mean + nc);
        brs = *reinterpret_cast<const f32x4*>(p.bn_rstd + nc);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        long orow[4];
        f32x4 rv4[4], ba4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          orow[q] = out_row(m0 + wm * 32 * TM + i * 32 + (lane >> 3) + 8 * q);
          rv4[q] = p.res ? *reinterpret_cast<const f32x4*>(p.res + orow[q] * p.res_pitch + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
          if (bnred) ba4[q] = *reinterpret_cast<const f32x4*>(p.bn_a + orow[q] * p.bn_a_pitch + nc);
        }
        lgm_wave_lds_sync();
        lgm_tile_to_lds(acc[i][j], Ts, lane);
        lgm_wave_lds_sync();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v4 = lgm_tile_row4(Ts, lane, q) + bv4 + rv4[q];
          if (p.act | (p.mask != nullptr)) {       // wave-uniform: the plain epilogue pays one scalar test
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] = lgm_post_act(v4[e], p.act, p.slope);
            if (p.mask) {
              const f32x4 m4 = *reinterpret_cast<const f32x4*>(p.mask + orow[q] * p.mask_pitch + nc);
#pragma unroll
              for (int e = 0; e < 4; ++e) v4[e] *= m4[e] > 0.f ? 1.f : p.mask_slope;
            }
          }
          *reinterpret_cast<f32x4*>(p.out + orow[q] * p.out_pitch + nc) = v4;
          if (bnred) {                                  // fixed order: rows (lane >> 3) + 8 q of tile i, q then i ascending
            bs1 += v4;
            bs2 += v4 * ((ba4[q] - bmu) * brs);
          }
        }
      }
      if (bnred) {
        // the 8 row lanes of a column quad (lane & 7 fixed): a fixed butterfly; then this wave's 32 TM rows of column
        // quad (lane & 7) sit in lanes 0-7
#pragma unroll
        for (int m = 8; m < 64; m <<= 1)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            bs1[e] += __shfl_xor(bs1[e], m, 64);
            bs2[e] += __shfl_xor(bs2[e], m, 64);
          }
        if (lane < 8) {
          *reinterpret_cast<f32x4*>(Bst + (((0 * 2 + wm) * 2 + wn) * TN + j) * 32 + lane * 4) = bs1;
          *reinterpret_cast<f32x4*>(Bst + (((1 * 2 + wm) * 2 + wn) * TN + j) * 32 + lane * 4) = bs2;
        }
      }
    }
    if (bnred) {
      __syncthreads();
      if (wm == 0 && lane < 32) {       // the two row halves of the tile, in order
        float* o = p.bn_part + ((long)(pc * p.tiles_m + tm) * 3) * p.N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + wn * 32 * TN + j * 32 + lane;
          o[n] = Bst[(((0 * 2 + 0) * 2 + wn) * TN + j) * 32 + lane] + Bst[(((0 * 2 + 1) * 2 + wn) * TN + j) * 32 + lane];
          o[p.N + n] = Bst[(((1 * 2 + 0) * 2 + wn) * TN + j) * 32 + lane] + Bst[(((1 * 2 + 1) * 2 + wn) * TN + j) * 32 + lane];
          o[2 * p.N + n] = 0.f;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * 32 * TN + j * 32 + lr;
    if (n >= p.N) continue;
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float rv[16];
      if (p.res) {   // issue all residual loads before the first use (one wait, not sixteen)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          rv[r] = m < p.M ? p.res[out_row(m) * p.res_pitch + n] : 0.f;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < p.M) {
          float v = lgm_post_act(acc[i][j][r] + bv + rv[r], p.act, p.slope);
          if (p.mask) v *= p.mask[out_row(m) * p.mask_pitch + n] > 0.f ? 1.f : p.mask_slope;
          p.out[out_row(m) * p.out_pitch + n] = v;
        }
      }
    }
  }
}

template <int MODE, int BM, int BN, int TM, int TN, bool UNI>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmArgs p) {
  // "consecutive blocks walk N first (share the A tile through L2)" holds per XCD only: hardware deals consecutive block
  // ids out to the 8 XCDs round-robin, each with its own L2.  With the swizzle XCD x runs the CONTIGUOUS logical range
  // [x nb / 8, (x + 1) nb / 8): the N-siblings of a row tile (and the splits of a tile) meet in one L2.
  igemm_body<MODE, BM, BN, TM, TN, UNI>(p, p.swz ? lgm_xcd_swizzle((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x);
}

// ---- backward pair of the generic layers (1x1 convolutions, linears): lgm_conv_bwd_pair ---------------------------------
// While the context is active, the two launch sites that can share a grid - the 64 x 64 uniform-tap input-gradient
// GEMM and the generic weight-gradient kernel - RECORD their arguments instead of launching, and the reducers that would
// follow them are stashed; lgm_conv_bwd_pair then issues ONE launch for both (or each alone when only one recorded) and
// the stashed reducers after it.  Every other path launches as usual: whatever the dispatchers pick, nothing is lost.
struct PairCtx {
  bool active, rec_i, rec_w, red_i, red_w;
  IgemmArgs ig;
  size_t ig_smem;
  unsigned ig_blocks, wg_blocks;
  alignas(16) unsigned char wg[320];        // WgradArgs (defined below)
  // stashed split-K reducer of the input gradient
  const float* r_ws; long r_stride; int r_splits; const float* r_bias; const float* r_res; long r_res_pitch;
  float* r_out; long r_out_pitch; long r_M; int r_N;
  bool r_post; int r_act; float r_slope; const float* r_mask; long r_mask_pitch; float r_mask_slope;
  // stashed slab reducer of the weight gradient (non-deferred form)
  const float* w_ws; long w_slab; float* w_gw; long w_nw; float* w_gb; long w_nb; int w_splits; float w_beta;
};
static thread_local PairCtx t_pair = {};

struct IgemmName { char c[64]; };
constexpr IgemmName igemm_name(int mode, int bm, int bn, int tm, int tn, bool uni) {
  IgemmName r = {};
  int n = 0;
  const char* head = "igemm_kernel<";
  for (int i = 0; head[i]; ++i) r.c[n++] = head[i];
  const int v[5] = {mode, bm, bn, tm, tn};
  for (int k = 0; k < 5; ++k) {
    char d[8] = {};
    int nd = 0, x = v[k];
    do { d[nd++] = (char)('0' + x % 10); x /= 10; } while (x);
    while (nd) r.c[n++] = d[--nd];
    r.c[n++] = ',';
    r.c[n++] = ' ';
  }
  const char* tail = uni ? "true>" : "false>";
  for (int i = 0; tail[i]; ++i) r.c[n++] = tail[i];
  return r;
}

template <int MODE, int BM, int BN, int TM, int TN, bool UNI>
int launch_igemm_t(IgemmArgs& a, hipStream_t s) {
  a.tiles_m = lgm_cdiv(a.M, BM);
  a.tiles_n = lgm_cdiv(a.N, BN);
  constexpr int A_TILE = BM * LDA;
  constexpr int B_TILE = (MODE == MODE_XY) ? BN * LDA : BK * BN;
  const size_t smem = 2 * (A_TILE + B_TILE) * sizeof(float);
  auto kern = igemm_kernel<MODE, BM, BN, TM, TN, UNI>;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  // the instantiation's name as the profiler prints it, built at compile time so that it can sit in the name registry
  static constexpr IgemmName name = igemm_name(MODE, BM, BN, TM, TN, UNI);
  static const char* const name_reg __attribute__((section("lgm_knames"), used)) = name.c;
  lgm_note_kernel(name_reg);
  if (MODE == MODE_YX && BM == 64 && BN == 64 && UNI && t_pair.active && !t_pair.rec_i && !a.stats) {
    t_pair.ig = a;                                   // launched by lgm_conv_bwd_pair, together with the weight gradient
    t_pair.ig_smem = smem;
    t_pair.ig_blocks = (unsigned)(a.tiles_m * a.tiles_n * a.splits * a.phases);
    t_pair.rec_i = true;
    return LGM_OK;
  }
  a.swz = swz_on();
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.tiles_m * a.tiles_n * a.splits * a.phases)), dim3(256), smem, s, a);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

// UNI (uniform-tap chunks, buffer loads): gathered channel count a multiple of BK, strided input gradients only in
// their residue-class form, every byte offset below 2^31.  LGM_NO_IGEMM_UNI=1 keeps the general decode (A/B runs).
static bool igemm_uni_enabled() {
  static const int on = [] { const char* e = getenv("LGM_NO_IGEMM_UNI"); return !(e && e[0] == '1'); }();
  return on != 0;
}

template <int MODE, int BM, int BN, int TM, int TN>
int launch_igemm(IgemmArgs& a, hipStream_t s) {
  const long gathered = MODE == MODE_XY ? (long)a.B * a.H * a.W : (long)a.B * a.Ho * a.Wo;
  const long span = (gathered + (long)(a.KH + 2) * (a.W > a.Wo ? a.W : a.Wo) + a.KW) * a.a_pitch * 4;
  const bool uni = igemm_uni_enabled() && a.Cg % BK == 0 && a.K % BK == 0 && span < (1L << 31) &&
                   (long)a.Nw * a.KH * a.KW * a.Cw * 4 < (1L << 31) &&
                   (MODE == MODE_XY || a.stride == 1 || a.phases > 1);
  return uni ? launch_igemm_t<MODE, BM, BN, TM, TN, true>(a, s) : launch_igemm_t<MODE, BM, BN, TM, TN, false>(a, s);
}

// Split-K plan of the generic path: a GEMM with few output tiles and a long reduction (the fused FiLM
// projection's input gradient: M = batch, K = 4864) would otherwise run on a handful of CUs.
int igemm_splits(long M, int N, int K, int* kchunk) {
  const long tiles = (long)lgm_cdiv(M, 64) * lgm_cdiv(N, 64);
  const int nk = lgm_cdiv(K, BK);
  *kchunk = nk * BK;
  // up to 256 tiles a launch leaves at most one wave per SIMD: nothing hides the load -> LDS -> barrier latency
  // of a chunk.  Split the reduction until ~4 workgroups share a CU (>= 8 chunks each).
  if (tiles > 256 || N % 4 != 0) return 1;
  static const long t_big = getenv("LGM_IGEMM_TBIG") ? atol(getenv("LGM_IGEMM_TBIG")) : 1024;      // tuning knobs (A/B runs)
  static const long t_small = getenv("LGM_IGEMM_TSMALL") ? atol(getenv("LGM_IGEMM_TSMALL")) : 256;
  static const long c_min = getenv("LGM_IGEMM_CMIN") ? atol(getenv("LGM_IGEMM_CMIN")) : 0;          // 0: 8 / 4 chunks per split
  long s = tiles > 64 ? t_big / tiles : t_small / tiles;
  const long cmin = c_min > 0 ? c_min : (tiles > 64 ? 8 : 4);
  if (s > nk / cmin) s = nk / cmin;
  if (s < 2) return 1;
  const int per = lgm_cdiv(nk, s);
  *kchunk = per * BK;
  return lgm_cdiv(nk, per);
}

// ---- post-op plumbing (lgm_conv_xy_post / lgm_conv_yx_post) -----------------------------------------------------------
// The entry points park the caller's post-op here for the duration of ONE call on this thread; the launch paths that
// can apply it in their epilogue (implicit-GEMM kernels, their split-K reducer) take it and set `done`; for every
// other path the entry point runs post_kernel over the finished output, so the result is the same either way.
struct PostState {
  const LgmPostOp* post;
  bool done;
};
static thread_local PostState t_post = {nullptr, false};

__global__ __launch_bounds__(256) void splitk_reduce_post_kernel(const float* __restrict__ ws, long ws_stride, int splits,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ res, long res_pitch,
                                                                 float* __restrict__ out, long out_pitch, long M, int N,
                                                                 int act, float slope, const float* __restrict__ mask,
                                                                 long mask_pitch, float mask_slope) {
  const int n4 = N / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * n4) return;
  const long m = i / n4;
  const int n = (int)(i % n4) * 4;
  f32x4 s = *reinterpret_cast<const f32x4*>(ws + m * N + n);
  for (int k = 1; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(ws + (long)k * ws_stride + m * N + n);
  if (bias) s += *reinterpret_cast<const f32x4*>(bias + n);
  if (res) s += *reinterpret_cast<const f32x4*>(res + m * res_pitch + n);
#pragma unroll
  for (int e = 0; e < 4; ++e) s[e] = lgm_post_act(s[e], act, slope);
  if (mask) {
    const f32x4 m4 = *reinterpret_cast<const f32x4*>(mask + m * mask_pitch + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] *= m4[e] > 0.f ? 1.f : mask_slope;
  }
  *reinterpret_cast<f32x4*>(out + m * out_pitch + n) = s;
}

// in-place post-op over a finished [M][N] output (paths without an epilogue hook)
__global__ __launch_bounds__(256) void post_kernel(float* __restrict__ out, long out_pitch, long M, int N, int act,
                                                   float slope, const float* __restrict__ mask, long mask_pitch,
                                                   float mask_slope) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const long m = i / N;
  const int n = (int)(i % N);
  float v = lgm_post_act(out[m * out_pitch + n], act, slope);
  if (mask) v *= mask[m * mask_pitch + n] > 0.f ? 1.f : mask_slope;
  out[m * out_pitch + n] = v;
}

// stats / stats_tiles (optional): request the BatchNorm statistics of the output from the epilogue; *stats_tiles = the
// number of row tiles written (0: this launch could not produce them - split-K, ragged tiles, bias / residual)
template <int MODE>
int dispatch_igemm(IgemmArgs& a, void* workspace, int64_t workspace_bytes, bool wide, hipStream_t s,
                   float* stats = nullptr, int* stats_tiles = nullptr) {
  a.wide = wide && a.N % 4 == 0 ? 1 : 0;
  a.phases = 1;
  a.KHs = a.KH;
  a.KWs = a.KW;
  a.act = 0; a.slope = 0.f; a.mask = nullptr; a.mask_pitch = 0; a.mask_slope = 0.f;
  const LgmPostOp* post = t_post.post;
  a.bn_a = nullptr; a.bn_a_pitch = 0; a.bn_mean = a.bn_rstd = nullptr; a.bn_part = nullptr;
  if (post) {          // every launch of this dispatcher applies it: in the epilogue, or in the split-K reducer
    a.act = post->act; a.slope = post->slope; a.mask = post->mask; a.mask_pitch = post->mask_pitch;
    a.mask_slope = post->mask_slope;
    t_post.done = true;
  }
  // BatchNorm backward sums from the epilogue: full tiles of the kernel this dispatcher is about to pick, no split-K
  auto want_bn = [&](int bm, int bn) {
    if (!(post && post->bn_a && post->bn_tiles)) return;
    const long tiles = (long)(a.M / bm) * a.phases;
    if (a.wide && a.M % bm == 0 && a.N % bn == 0 && post->bn_partial && post->bn_mean && post->bn_rstd &&
        post->bn_a_pitch % 4 == 0 && lgm_aligned16(post->bn_a) && lgm_aligned16(post->bn_mean) &&
        lgm_aligned16(post->bn_rstd) && post->bn_partial_floats >= tiles * 3 * a.N) {
      a.bn_a = post->bn_a; a.bn_a_pitch = post->bn_a_pitch; a.bn_mean = post->bn_mean; a.bn_rstd = post->bn_rstd;
      a.bn_part = post->bn_partial;
      *post->bn_tiles = (int)tiles;
    }
  };
  if (MODE == MODE_YX && a.stride > 1 && a.KH % a.stride == 0 && a.KW % a.stride == 0 && a.H % a.stride == 0 &&
      a.W % a.stride == 0) {
    a.phases = a.stride * a.stride;
    a.KHs = a.KH / a.stride;
    a.KWs = a.KW / a.stride;
    a.M = a.B * (a.H / a.stride) * (a.W / a.stride);       // rows of ONE residue class
    a.K = a.KHs * a.KWs * a.Cg;
  }
  const long t128 = (long)lgm_cdiv(a.M, 128) * a.phases;
  a.splits = 1;
  a.kchunk = lgm_cdiv(a.K, BK) * BK;
  a.ws = nullptr;
  a.stats = nullptr;
  if (stats_tiles) *stats_tiles = 0;
  auto want_stats = [&](int bm, int bn) {
    if (stats && stats_tiles && a.wide && !a.bias && !a.res && a.M % bm == 0 && a.N % bn == 0) {
      a.stats = stats;
      *stats_tiles = (int)(a.M / bm) * a.phases;
    }
  };
  if (a.N > 64 && t128 * lgm_cdiv(a.N, 128) >= 384) {
    want_stats(128, 128);
    want_bn(128, 128);
    return launch_igemm<MODE, 128, 128, 2, 2>(a, s);
  }
  if (t128 * lgm_cdiv(a.N, 64) >= 384) {
    want_stats(128, 64);
    want_bn(128, 64);
    return launch_igemm<MODE, 128, 64, 2, 1>(a, s);
  }
  int kchunk;
  const int splits = a.phases > 1 ? 1 : igemm_splits(a.M, a.N, a.K, &kchunk);
  if (splits > 1 && wide && workspace && workspace_bytes >= (int64_t)splits * a.M * a.N * (int64_t)sizeof(float)) {
    a.splits = splits;
    a.kchunk = kchunk;
    a.ws = (float*)workspace;
    if (int rc = launch_igemm<MODE, 64, 64, 1, 1>(a, s)) return rc;      // partial products: the kernel skips the epilogue ops
    if (t_pair.active && t_pair.rec_i && t_pair.ig.ws == a.ws) {   // recorded, not launched: reduce after the pair
      t_pair.red_i = true;
      t_pair.r_ws = a.ws; t_pair.r_stride = (long)a.M * a.N; t_pair.r_splits = splits; t_pair.r_bias = a.bias;
      t_pair.r_res = a.res; t_pair.r_res_pitch = a.res_pitch; t_pair.r_out = a.out; t_pair.r_out_pitch = a.out_pitch;
      t_pair.r_M = a.M; t_pair.r_N = a.N;
      t_pair.r_post = post != nullptr; t_pair.r_act = a.act; t_pair.r_slope = a.slope; t_pair.r_mask = a.mask;
      t_pair.r_mask_pitch = a.mask_pitch; t_pair.r_mask_slope = a.mask_slope;
      return LGM_OK;
    }
    if (post) {
      const long items = (long)a.M * (a.N / 4);
      hipLaunchKernelGGL(splitk_reduce_post_kernel, dim3((unsigned)lgm_cdiv(items, 256)), dim3(256), 0, s,
                         (const float*)a.ws, (long)a.M * a.N, splits, a.bias, a.res, a.res_pitch, a.out, a.out_pitch,
                         (long)a.M, a.N, a.act, a.slope, a.mask, a.mask_pitch, a.mask_slope);
      LGM_LAUNCH_CHECK();
      return LGM_OK;
    }
    return lgm_splitk_reduce_launch(a.ws, (long)a.M * a.N, splits, a.bias, a.res, a.res_pitch, a.out, a.out_pitch, a.M,
                                    a.N, s);
  }
  want_stats(64, 64);
  want_bn(64, 64);
  return launch_igemm<MODE, 64, 64, 1, 1>(a, s);
}

int check_geom(const LgmConvGeom* g) {
  LGM_REQUIRE(g != nullptr, "conv: null geometry");
  LGM_REQUIRE(g->B > 0 && g->H > 0 && g->W > 0 && g->Cw > 0 && g->Ho > 0 && g->Wo > 0 && g->Nw > 0,
              "conv: non-positive dimension");
  LGM_REQUIRE(g->KH > 0 && g->KW > 0 && g->stride > 0 && g->pad >= 0, "conv: bad kernel/stride/pad");
  LGM_REQUIRE((g->H + 2 * g->pad - g->KH) / g->stride + 1 == g->Ho &&
                  (g->W + 2 * g->pad - g->KW) / g->stride + 1 == g->Wo,
              "conv: Ho/Wo inconsistent with H/W, kernel, stride, pad");
  LGM_REQUIRE((long)g->B * g->H * g->W < (1L << 31) && (long)g->B * g->Ho * g->Wo < (1L << 31),
              "conv: pixel count overflows int32");
  return LGM_OK;
}

}  // namespace

// bytes of workspace that let lgm_conv_xy (yx = 0) / lgm_conv_yx (yx = 1) use deterministic split-K
extern "C" int64_t lgm_conv_workspace(const LgmConvGeom* g, int yx) {
  if (check_geom(g) != LGM_OK) return -1;
  const int gc = yx ? g->Nw : g->Cw, oc = yx ? g->Cw : g->Nw;
  if (!use_3x3() || !lgm_conv3x3_supported(g, gc, oc)) {
    const long M = yx ? (long)g->B * g->H * g->W : (long)g->B * g->Ho * g->Wo;
    int kchunk;
    const int s = igemm_splits(M, oc, g->KH * g->KW * gc, &kchunk);
    return s > 1 ? (int64_t)s * M * oc * (int64_t)sizeof(float) : 0;
  }
  const int s = lgm_conv3x3_splits(g, gc, oc);
  return s > 1 ? (int64_t)s * g->B * g->H * g->W * oc * (int64_t)sizeof(float) : 0;
}

namespace {
// 1x1 convolution between 64 channels and <= 4 (the UNet's final_conv 64 -> 3, reference ddpm.py:422, and its input gradient).
// As a GEMM it has N = 4 and idles 15/16 of a 64-wide MFMA tile (igemm_kernel<0, 128, 64>: 25 us for 33.5 MB at B = 128); it
// is a streaming dot product: the 16 lanes of a DPP row own one pixel (lane = four channels, one 16-byte load), the four
// weight rows sit in registers, the row sums are four rotate-and-add steps on the DPP path (no LDS, no shuffle through the
// crossbar - DESIGN finding 19), lane 0 of the row stores the pixel's four outputs.
template <int CTRL>
__device__ __forceinline__ float lgm_dpp_row(float a) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lgm_row16_sum(float v) {
  v += lgm_dpp_row<0x128>(v);      // row_ror:8
  v += lgm_dpp_row<0x124>(v);      // row_ror:4
  v += lgm_dpp_row<0x122>(v);      // row_ror:2
  v += lgm_dpp_row<0x121>(v);      // row_ror:1
  return v;
}

__global__ __launch_bounds__(256) void narrow1x1_fwd_kernel(const float* __restrict__ x, long x_pitch,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ y, long y_pitch, long P) {
  const int cl = threadIdx.x & 15, slot = threadIdx.x >> 4;          // 16 pixels per block and trip
  f32x4 wr[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) wr[n] = *reinterpret_cast<const f32x4*>(w + n * 64 + cl * 4);
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv = *reinterpret_cast<const f32x4*>(bias);
  const long stride = (long)gridDim.x * 64;                          // four trips' pixels in flight per thread
  for (long p0 = (long)blockIdx.x * 64 + slot; p0 < P; p0 += stride) {
    f32x4 xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long pix = p0 + 16 * u;
      xv[u] = pix < P ? *reinterpret_cast<const f32x4*>(x + pix * x_pitch + cl * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long pix = p0 + 16 * u;
      f32x4 o;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const f32x4 t = xv[u] * wr[n];
        o[n] = lgm_row16_sum((t[0] + t[1]) + (t[2] + t[3]));
      }
      if (cl == 0 && pix < P) *reinterpret_cast<f32x4*>(y + pix * y_pitch) = o + bv;
    }
  }
}

__global__ __launch_bounds__(256) void narrow1x1_dgrad_kernel(const float* __restrict__ gy, long gy_pitch,
                                                              const float* __restrict__ w, float* __restrict__ gx,
                                                              long gx_pitch, long P) {
  const int cl = threadIdx.x & 15, slot = threadIdx.x >> 4;
  f32x4 wr[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) wr[n] = *reinterpret_cast<const f32x4*>(w + n * 64 + cl * 4);
  const long stride = (long)gridDim.x * 64;
  for (long p0 = (long)blockIdx.x * 64 + slot; p0 < P; p0 += stride) {
    f32x4 g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long pix = p0 + 16 * u;
      g[u] = pix < P ? *reinterpret_cast<const f32x4*>(gy + pix * gy_pitch) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long pix = p0 + 16 * u;
      if (pix < P)
        *reinterpret_cast<f32x4*>(gx + pix * gx_pitch + cl * 4) =
            (wr[0] * g[u][0] + wr[1] * g[u][1]) + (wr[2] * g[u][2] + wr[3] * g[u][3]);
    }
  }
}

// the layers these two kernels take: 1x1 / stride 1 / unpadded, 64 channels on the wide side, 4 (padded) on the narrow one
bool narrow1x1_geom(const LgmConvGeom* g) {
  static const bool off = getenv("LGM_NO_NARROW1X1") != nullptr;        // A/B switch
  return !off && g->KH == 1 && g->KW == 1 && g->stride == 1 && g->pad == 0 && g->Nw == 4 && g->Cw == 64;
}
unsigned narrow1x1_blocks(long P) {
  const long want = (P + 63) / 64;
  return (unsigned)(want < 2048 ? (want < 1 ? 1 : want) : 2048);
}
}  // namespace

static int conv_xy_impl(const LgmConvGeom* g, const float* x, int64_t x_pitch, const float* w,
                        const float* bias, const float* res, int64_t res_pitch, float* y,
                        int64_t y_pitch, void* workspace, int64_t workspace_bytes, float* stats, int* stats_tiles,
                        void* stream) {
  if (stats_tiles) *stats_tiles = 0;
  if (int rc = check_geom(g)) return rc;
  LGM_REQUIRE(x && w && y, "conv_xy: null pointer");
  LGM_REQUIRE(g->Cw % 4 == 0, "conv_xy: Cw=%d must be a multiple of 4 (pad channels)", g->Cw);
  LGM_REQUIRE(x_pitch % 4 == 0 && x_pitch >= g->Cw && lgm_aligned16(x) && lgm_aligned16(w),
              "conv_xy: x/w must be 16B aligned with pitch %% 4 == 0");
  LGM_REQUIRE(y_pitch >= g->Nw && (!res || res_pitch >= g->Nw), "conv_xy: output pitch < Nw");
  if (use_3x3() && narrow1x1_geom(g) && !res && !stats && !t_post.post && wide_ok(y, y_pitch, nullptr, 0, bias)) {
    const long P = (long)g->B * g->H * g->W;
    lgm_note_kernel(LGM_KNAME("narrow1x1_fwd_kernel"));
    hipLaunchKernelGGL(narrow1x1_fwd_kernel, dim3(narrow1x1_blocks(P)), dim3(256), 0, (hipStream_t)stream, x, (long)x_pitch, w,
                       bias, y, (long)y_pitch, P);
    LGM_LAUNCH_CHECK();
    return LGM_OK;
  }
  static const bool post_3x3 = getenv("LGM_POST_VIA_3X3") != nullptr;   // A/B switch: direct 3x3 kernel + elementwise post-op
  if (use_3x3() && (!t_post.post || post_3x3) && wide_ok(y, y_pitch, res, res_pitch, bias) && lgm_conv3x3_supported(g, g->Cw, g->Nw) &&
      ((long)g->B * g->H * g->W + g->W + 1) * x_pitch < (1L << 29))   // buffer offsets (bytes) below 2^31
    return lgm_conv3x3_launch(0, g, x, x_pitch, w, bias, res, res_pitch, y, y_pitch, workspace, workspace_bytes,
                              (hipStream_t)stream);
  if (use_3x3() && use_gstream() && wide_ok(y, y_pitch, res, res_pitch, bias) && g->KH == 1 && g->KW == 1 &&
      g->stride == 1 && g->pad == 0 && lgm_gemm_stream_supported((long)g->B * g->H * g->W, g->Nw, g->Cw, x_pitch, y_pitch, res ? res_pitch : 0))
    return lgm_note_kernel(LGM_KNAME("gemm_stream_kernel")), lgm_gemm_stream_launch(x, x_pitch, w, bias, res, res_pitch, y, y_pitch,
                                                                        (long)g->B * g->H * g->W, g->Nw, g->Cw, (hipStream_t)stream);
  if (use_3x3() && wide_ok(y, y_pitch, res, res_pitch, bias) && g->KH == 1 && g->KW == 1 && g->stride == 1 &&
      g->pad == 0 && lgm_gemm_rows_supported((long)g->B * g->H * g->W, g->Nw, g->Cw))
    return lgm_note_kernel(LGM_KNAME("gemm_rows_kernel")), lgm_gemm_rows_launch(x, x_pitch, w, bias, res, res_pitch, y, y_pitch,
                                                                    (long)g->B * g->H * g->W, g->Nw, g->Cw, (hipStream_t)stream);
  IgemmArgs a{};
  a.a = x; a.w = w; a.bias = bias; a.res = res; a.out = y;
  a.a_pitch = x_pitch; a.res_pitch = res_pitch; a.out_pitch = y_pitch;
  a.B = g->B; a.H = g->H; a.W = g->W; a.Cw = g->Cw; a.Ho = g->Ho; a.Wo = g->Wo; a.Nw = g->Nw;
  a.KH = g->KH; a.KW = g->KW; a.stride = g->stride; a.pad = g->pad;
  a.M = g->B * g->Ho * g->Wo; a.N = g->Nw; a.K = g->KH * g->KW * g->Cw; a.Cg = g->Cw;
  return dispatch_igemm<MODE_XY>(a, workspace, workspace_bytes, wide_ok(y, y_pitch, res, res_pitch, bias),
                                 (hipStream_t)stream, stats, stats_tiles);
}

extern "C" int lgm_conv_xy(const LgmConvGeom* g, const float* x, int64_t x_pitch, const float* w,
                           const float* bias, const float* res, int64_t res_pitch, float* y,
                           int64_t y_pitch, void* workspace, int64_t workspace_bytes, void* stream) {
  return conv_xy_impl(g, x, x_pitch, w, bias, res, res_pitch, y, y_pitch, workspace, workspace_bytes, nullptr, nullptr,
                      stream);
}

static int post_check(const LgmPostOp* post, const char* who) {
  LGM_REQUIRE(post && (post->act == 0 || post->act == 3 || post->act == 4), "%s: post-op activation must be none / ReLU / LeakyReLU", who);
  LGM_REQUIRE(!post->mask || (lgm_aligned16(post->mask) && post->mask_pitch % 4 == 0), "%s: mask must be 16B aligned, pitch %% 4 == 0", who);
  return LGM_OK;
}

// runs `call` with the post-op parked for the launch paths; applies it with one elementwise launch when the path
// taken had no epilogue hook
template <typename F>
static int with_post(const LgmPostOp* post, float* out, int64_t out_pitch, long M, int N, void* stream, F call) {
  if (post && post->bn_tiles) *post->bn_tiles = 0;      // set by the dispatcher when the epilogue takes the sums
  t_post.post = post;
  t_post.done = false;
  const int rc = call();
  const bool done = t_post.done;
  t_post.post = nullptr;
  t_post.done = false;
  if (rc != LGM_OK || done || (post->act == 0 && !post->mask)) return rc;
  hipLaunchKernelGGL(post_kernel, dim3((unsigned)lgm_cdiv(M * N, 256)), dim3(256), 0, (hipStream_t)stream, out,
                     (long)out_pitch, M, N, post->act, post->slope, post->mask, (long)post->mask_pitch, post->mask_slope);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_conv_xy_post(const LgmConvGeom* g, const float* x, int64_t x_pitch, const float* w,
                                const float* bias, const float* res, int64_t res_pitch, float* y, int64_t y_pitch,
                                void* workspace, int64_t workspace_bytes, const LgmPostOp* post, void* stream) {
  if (int rc = post_check(post, "conv_xy_post")) return rc;
  if (int rc = check_geom(g)) return rc;
  return with_post(post, y, y_pitch, (long)g->B * g->Ho * g->Wo, g->Nw, stream, [&] {
    return conv_xy_impl(g, x, x_pitch, w, bias, res, res_pitch, y, y_pitch, workspace, workspace_bytes, nullptr, nullptr,
                        stream);
  });
}

// lgm_conv_xy that also leaves the per-row-tile BatchNorm statistics of y in stats[tile][3][Nw] (see IgemmArgs::stats);
// *stats_tiles = tiles written, 0 when this geometry / path cannot (the caller then runs lgm_bn_stats on y).
// stats must hold lgm_conv_stats_floats(g, 0) floats.
extern "C" int lgm_conv_xy_stats(const LgmConvGeom* g, const float* x, int64_t x_pitch, const float* w, float* y,
                                 int64_t y_pitch, void* workspace, int64_t workspace_bytes, float* stats,
                                 int* stats_tiles, void* stream) {
  LGM_REQUIRE(stats && stats_tiles, "conv_xy_stats: null statistics buffer");
  return conv_xy_impl(g, x, x_pitch, w, nullptr, nullptr, 0, y, y_pitch, workspace, workspace_bytes, stats, stats_tiles,
                      stream);
}

extern "C" int64_t lgm_conv_stats_floats(const LgmConvGeom* g, int yx) {
  if (check_geom(g) != LGM_OK) return -1;
  const long rows = yx ? (long)g->B * g->H * g->W : (long)g->B * g->Ho * g->Wo;
  return (int64_t)(lgm_cdiv(rows, 64) + 4) * 3 * (yx ? g->Cw : g->Nw);      // 64-row tiles are the smallest
}

namespace {
// Batched weight transposition: for every table row (src_off, Nw, T, Cw, first_block) rewrite
// w[Nw][T][Cw] (at data + src_off) as wT[Cw][T][Nw] (at dst + src_off).  One 32x32 (n, c) tile of one
// tap per block; the owning layer is found by binary search over the block prefix.
__global__ __launch_bounds__(256) void transpose_weights_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                const int* __restrict__ table, int n_layers) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n_layers - 1;
  const int bid = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid * 5 + 4] <= bid) lo = mid; else hi = mid - 1;
  }
  const int* row = table + lo * 5;
  const long off = row[0];
  const int Nw = row[1], T = row[2], Cw = row[3];
  int lb = bid - row[4];
  const int tc = (Cw + 31) / 32, tn = (Nw + 31) / 32;
  const int ct = lb % tc;
  lb /= tc;
  const int nt = lb % tn, tap = lb / tn;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = nt * 32 + ty + 8 * i, c = ct * 32 + tx;
    tile[ty + 8 * i][tx] = (n < Nw && c < Cw) ? src[off + ((long)n * T + tap) * Cw + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = ct * 32 + ty + 8 * i, n = nt * 32 + tx;
    if (n < Nw && c < Cw) dst[off + ((long)c * T + tap) * Nw + n] = tile[tx][ty + 8 * i];
  }
}
}  // namespace

extern "C" int lgm_transpose_weights(const float* src, float* dst, const int32_t* table, int n_layers,
                                     int total_blocks, void* stream) {
  LGM_REQUIRE(src && dst && table && n_layers > 0 && total_blocks > 0, "transpose_weights: bad arguments");
  hipLaunchKernelGGL(transpose_weights_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, src, dst, table,
                     n_layers);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

namespace {
// Input gradient / transposed convolution with <= 4 produced channels (the image end of the DCGAN
// critic and generator: dD/dx for the gradient penalty, the generator's last layer).  As a GEMM this has
// N = 4 and would idle 15/16 of a 64-wide MFMA tile; it is a memory-bound dot product instead: one wave
// per output pixel, lane l owns gathered channels 4l..4l+3, the (tap, out-channel) weight vectors sit in
// LDS, four accumulators are combined with a fixed shuffle tree.  Only taps of the pixel's stride
// residue class are visited.
struct SmallNArgs {
  const float* y;     // gathered tensor [B,Ho,Wo,Nw] (pitch y_pitch)
  const float* w;     // [Nw][T][4]
  const float* bias;
  const float* res;
  float* x;           // [B,H,W,4]
  long y_pitch, res_pitch, x_pitch;
  int B, H, W, Ho, Wo, Nw, KH, KW, stride, pad;
  long npix;
};

// KHS x KWS = taps per residue class (KH/stride x KW/stride); LPP = Nw/4 lanes share one output pixel,
// 64/LPP pixels per wave.  blockIdx.y = residue class (ih % stride, iw % stride): every pixel a block visits
// meets the SAME taps, so their weights (for the lane's four gathered channels) live in registers for the
// whole kernel - the first version read them from LDS per pixel, where the four pixels of a wave (alternating
// parity) hit the same banks with different taps (4-way conflicts: LDS-bound at 1/13 of the FMA rate).
// Zero padding = raw buffer loads at an offset past the descriptor's range.
template <int KHS, int KWS>
__global__ __launch_bounds__(256) void smalln_yx_kernel(const SmallNArgs p) {
  const int T = p.KH * p.KW;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int LPP = p.Nw / 4;                       // power of two, <= 64
  const int ppw = 64 / LPP;
  const int sub = lane / LPP, cl = lane % LPP;   // pixel slot inside the wave, channel chunk
  const int ph = blockIdx.y / p.stride, pw = blockIdx.y % p.stride;
  const int kh0 = (ph + p.pad) % p.stride, kw0 = (pw + p.pad) % p.stride;
  const int dh0 = (ph + p.pad - kh0) / p.stride, dw0 = (pw + p.pad - kw0) / p.stride;
  const int Hq = p.H / p.stride, Wq = p.W / p.stride;
  const unsigned npq = (unsigned)(p.B * Hq * Wq);      // pixels of one class
  f32x4 wr[KHS * KWS][4];                               // [tap][output channel] over the lane's 4 gathered channels
#pragma unroll
  for (int ta = 0; ta < KHS; ++ta)
#pragma unroll
    for (int tb = 0; tb < KWS; ++tb) {
      const int tap = (kh0 + ta * p.stride) * p.KW + kw0 + tb * p.stride;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p.w + ((long)(cl * 4 + j) * T + tap) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) wr[ta * KWS + tb][c][j] = v[c];
      }
    }
  const unsigned nrec = (unsigned)((long)p.B * p.Ho * p.Wo * p.y_pitch * 4);
  __amdgpu_buffer_rsrc_t rsrc_y;
  {
    const unsigned long long ab = reinterpret_cast<unsigned long long>(p.y);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ab);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(ab >> 32));
    rsrc_y = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                               __builtin_amdgcn_readfirstlane(nrec), 0x00020000);
  }
  const unsigned ypb = (unsigned)p.y_pitch * 4u;
  const unsigned per_iter = gridDim.x * 4u * (unsigned)ppw;
  for (unsigned base = (blockIdx.x * 4u + (unsigned)wid) * (unsigned)ppw; base < npq; base += per_iter) {
    const unsigned pq = base + (unsigned)sub;
    const bool live = pq < npq;
    const unsigned t0 = pq / (unsigned)Wq;
    const int iwq = (int)(pq - t0 * (unsigned)Wq);
    const int b = (int)(t0 / (unsigned)Hq);
    const int ihq = (int)(t0 - (unsigned)b * (unsigned)Hq);
    f32x4 yv[KHS * KWS];
#pragma unroll
    for (int ta = 0; ta < KHS; ++ta)
#pragma unroll
      for (int tb = 0; tb < KWS; ++tb) {
        const int th = ihq + dh0 - ta, tw = iwq + dw0 - tb;
        const bool v = live && (unsigned)th < (unsigned)p.Ho && (unsigned)tw < (unsigned)p.Wo;
        const unsigned off = (unsigned)((b * p.Ho + th) * p.Wo + tw) * ypb + (unsigned)cl * 16u;
        yv[ta * KWS + tb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, v ? off : nrec, 0, 0));
      }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KHS * KWS; ++t) {
      const f32x4 y4 = yv[t];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 wv = wr[t][c];
        acc[c] += (y4[0] * wv[0] + y4[1] * wv[1]) + (y4[2] * wv[2] + y4[3] * wv[3]);
      }
    }
    for (int off = LPP >> 1; off > 0; off >>= 1) {   // fixed tree inside the pixel's lane group
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] += __shfl_xor(acc[c], off, 64);
    }
    if (cl == 0 && live) {
      const long pix = ((long)b * p.H + ihq * p.stride + ph) * p.W + iwq * p.stride + pw;
      if (p.bias) acc += *reinterpret_cast<const f32x4*>(p.bias);
      if (p.res) acc += *reinterpret_cast<const f32x4*>(p.res + pix * p.res_pitch);
      *reinterpret_cast<f32x4*>(p.x + pix * p.x_pitch) = acc;
    }
  }
}

// Lane-per-pixel form of the same operator (Nw a multiple of 32, maps with an even side >= 16): a workgroup owns a
// TR x TC tile of ONE residue class (256 output pixels, one per thread), stages the (TR+1) x (TC+1) patch of y it
// needs through LDS in 32-channel chunks (coalesced 16-byte loads; rows padded to 36 floats so that the 16-byte
// reads of neighbouring pixels fall on different banks) and every thread runs its own 4 x Nw x 4 FMA chain with the
// tap weights as scalar operands (uniform addresses: s_load).  No cross-lane reduction, no redundant work: ~22
// instructions per pixel against ~60 in the lane-group form above, which stays for the small maps.
template <int KHS, int KWS, int T>      // T = KH * KW as a constant: the weight offsets become s_load immediates
__global__ __launch_bounds__(256) void smalln_yx_lp_kernel(const SmallNArgs p, int lgc) {
  static_assert(KHS == 2 && KWS == 2, "patch = tile + 1 in both directions");
  extern __shared__ __align__(16) float ys[];            // [(TR + 1) * (TC + 1)][36], then the class's weights
  constexpr int CH = 32, LDY = 36;
  const int TC = 1 << lgc, TR = 256 >> lgc, PC = TC + 1, PR = TR + 1;
  float* wsm = ys + PR * PC * LDY;                       // [KHS * KWS][Nw][4]: every lane reads the same 16 bytes (LDS
                                                         // broadcast) - scalar loads shared lgkmcnt with the patch
                                                         // reads and drained it 21 times per chunk
  const int Hq = p.H / p.stride, Wq = p.W / p.stride;
  const int tiles_c = (Wq + TC - 1) >> lgc, tiles_r = (Hq + TR - 1) / TR;
  int t = blockIdx.x;
  const int tc = t % tiles_c;
  t /= tiles_c;
  const int tr = t % tiles_r, b = t / tiles_r;
  const int ph = blockIdx.y / p.stride, pw = blockIdx.y % p.stride;
  const int kh0 = (ph + p.pad) % p.stride, kw0 = (pw + p.pad) % p.stride;
  const int dh0 = (ph + p.pad - kh0) / p.stride, dw0 = (pw + p.pad - kw0) / p.stride;
  const int th0 = tr * TR + dh0 - 1, tw0 = tc * TC + dw0 - 1;      // y coordinates of patch position (0, 0)
  const int tid = threadIdx.x;
  const int r = tid >> lgc, c = tid & (TC - 1);
  const int ihq = tr * TR + r, iwq = tc * TC + c;
  const bool live = ihq < Hq && iwq < Wq;
  const unsigned nrec = (unsigned)((long)p.B * p.Ho * p.Wo * p.y_pitch * 4);
  __amdgpu_buffer_rsrc_t rsrc_y;
  {
    const unsigned long long ab = reinterpret_cast<unsigned long long>(p.y);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ab);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(ab >> 32));
    rsrc_y = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                               __builtin_amdgcn_readfirstlane(nrec), 0x00020000);
  }
  const unsigned ypb = (unsigned)p.y_pitch * 4u;
  // staging role: 8 threads per patch position (32 channels = 8 x 16 bytes), position k * 32 + sp in pass k.  The
  // positions do not change from chunk to chunk: their y offsets are decoded once, and a chunk issues all its loads
  // before the first LDS write (a load -> wait -> write loop serialised ten memory latencies per chunk)
  constexpr int KMAX = 10;                           // ceil((TR + 1) * (TC + 1) / 32) for both tile shapes
  const int sq = tid & 7, sp = tid >> 3;
  unsigned yoff[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int pos = k * 32 + sp;
    const int pr = pos / PC, pc = pos - pr * PC;
    const int th = th0 + pr, tw = tw0 + pc;
    const bool ok = pos < PR * PC && (unsigned)th < (unsigned)p.Ho && (unsigned)tw < (unsigned)p.Wo;
    yoff[k] = ok ? (unsigned)((b * p.Ho + th) * p.Wo + tw) * ypb + (unsigned)sq * 16u : nrec;
  }
  for (int i = tid; i < KHS * KWS * p.Nw; i += 256) {
    const int tp = i / p.Nw, n = i - tp * p.Nw;
    const int ta = tp / KWS, tb = tp - ta * KWS;
    const int tap = (kh0 + ta * p.stride) * p.KW + kw0 + tb * p.stride;
    *reinterpret_cast<f32x4*>(wsm + (long)i * 4) = *reinterpret_cast<const f32x4*>(p.w + ((long)n * T + tap) * 4);
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 st[KMAX];
  auto fetch = [&](int n0) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      st[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, yoff[k], (unsigned)(n0 * 4), 0));
  };
  fetch(0);
  for (int n0 = 0; n0 < p.Nw; n0 += CH) {
    if (n0) __syncthreads();
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int pos = k * 32 + sp;
      if (pos < PR * PC) *reinterpret_cast<f32x4*>(ys + pos * LDY + sq * 4) = st[k];
    }
    if (n0 + CH < p.Nw) fetch(n0 + CH);             // the next chunk travels while this one is multiplied
    __syncthreads();
#pragma unroll
    for (int ta = 0; ta < KHS; ++ta)
#pragma unroll
      for (int tb = 0; tb < KWS; ++tb) {
        const float* yp = ys + ((r + 1 - ta) * PC + (c + 1 - tb)) * LDY;
        const float* wp = wsm + ((ta * KWS + tb) * p.Nw + n0) * 4;
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
          const f32x4 y4 = *reinterpret_cast<const f32x4*>(yp + q * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wp + (q * 4 + j) * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fmaf(y4[j], w4[k], acc[k]);
          }
        }
      }
  }
  if (live) {
    const long pix = ((long)b * p.H + ihq * p.stride + ph) * p.W + iwq * p.stride + pw;
    if (p.bias) acc += *reinterpret_cast<const f32x4*>(p.bias);
    if (p.res) acc += *reinterpret_cast<const f32x4*>(p.res + pix * p.res_pitch);
    *reinterpret_cast<f32x4*>(p.x + pix * p.x_pitch) = acc;
  }
}
}  // namespace

static int conv_yx_impl(const LgmConvGeom* g, const float* y, int64_t y_pitch, const float* w,
                        const float* w_t, const float* bias, const float* res, int64_t res_pitch, float* x,
                        int64_t x_pitch, void* workspace, int64_t workspace_bytes, float* stats, int* stats_tiles,
                        void* stream) {
  if (stats_tiles) *stats_tiles = 0;
  if (int rc = check_geom(g)) return rc;
  LGM_REQUIRE(x && w && y, "conv_yx: null pointer");
  LGM_REQUIRE(g->Nw % 4 == 0 && g->Cw % 4 == 0, "conv_yx: Nw=%d, Cw=%d must be multiples of 4", g->Nw, g->Cw);
  LGM_REQUIRE(y_pitch % 4 == 0 && y_pitch >= g->Nw && lgm_aligned16(y) && lgm_aligned16(w),
              "conv_yx: y/w must be 16B aligned with pitch %% 4 == 0");
  LGM_REQUIRE(x_pitch >= g->Cw && (!res || res_pitch >= g->Cw), "conv_yx: output pitch < Cw");
  if (use_3x3() && narrow1x1_geom(g) && !res && !bias && !stats && !t_post.post && wide_ok(x, x_pitch, nullptr, 0, nullptr)) {
    const long P = (long)g->B * g->H * g->W;
    lgm_note_kernel(LGM_KNAME("narrow1x1_dgrad_kernel"));
    hipLaunchKernelGGL(narrow1x1_dgrad_kernel, dim3(narrow1x1_blocks(P)), dim3(256), 0, (hipStream_t)stream, y, (long)y_pitch, w,
                       x, (long)x_pitch, P);
    LGM_LAUNCH_CHECK();
    return LGM_OK;
  }
  static const bool post_3x3 = getenv("LGM_POST_VIA_3X3") != nullptr;   // A/B switch (see conv_xy_impl)
  if (use_3x3() && (!t_post.post || post_3x3) && wide_ok(x, x_pitch, res, res_pitch, bias) && lgm_conv3x3_supported(g, g->Nw, g->Cw) &&
      ((long)g->B * g->H * g->W + g->W + 1) * y_pitch < (1L << 29))
    return lgm_conv3x3_launch(w_t ? 2 : 1, g, y, y_pitch, w_t ? w_t : w, bias, res, res_pitch, x, x_pitch, workspace,
                              workspace_bytes, (hipStream_t)stream);
  if (use_3x3() && use_gstream() && w_t && wide_ok(x, x_pitch, res, res_pitch, bias) && g->KH == 1 && g->KW == 1 &&
      g->stride == 1 && g->pad == 0 && lgm_gemm_stream_supported((long)g->B * g->H * g->W, g->Cw, g->Nw, y_pitch, x_pitch, res ? res_pitch : 0))
    return lgm_gemm_stream_launch(y, y_pitch, w_t, bias, res, res_pitch, x, x_pitch, (long)g->B * g->H * g->W, g->Cw,
                                  g->Nw, (hipStream_t)stream);
  if (use_3x3() && w_t && wide_ok(x, x_pitch, res, res_pitch, bias) && g->KH == 1 && g->KW == 1 && g->stride == 1 &&
      g->pad == 0 && lgm_gemm_rows_supported((long)g->B * g->H * g->W, g->Cw, g->Nw))
    return lgm_gemm_rows_launch(y, y_pitch, w_t, bias, res, res_pitch, x, x_pitch, (long)g->B * g->H * g->W, g->Cw, g->Nw,
                                (hipStream_t)stream);
  const int lpp = g->Nw / 4;
  if (use_3x3() && g->Cw == 4 && g->Nw % 4 == 0 && lpp <= 64 && (lpp & (lpp - 1)) == 0 && g->stride == 2 &&
      g->KH == 4 && g->KW == 4 && g->H % 2 == 0 && g->W % 2 == 0 && wide_ok(x, x_pitch, res, res_pitch, bias) &&
      (long)g->B * g->Ho * g->Wo * y_pitch * 4 < (1L << 31)) {
    SmallNArgs q{};
    q.y = y; q.w = w; q.bias = bias; q.res = res; q.x = x;
    q.y_pitch = y_pitch; q.res_pitch = res_pitch; q.x_pitch = x_pitch;
    q.B = g->B; q.H = g->H; q.W = g->W; q.Ho = g->Ho; q.Wo = g->Wo; q.Nw = g->Nw;
    q.KH = g->KH; q.KW = g->KW; q.stride = g->stride; q.pad = g->pad;
    q.npix = (long)g->B * g->H * g->W;
    static const bool no_lp = getenv("LGM_SMALLN_LANES") != nullptr;      // A/B switch: the lane-group form everywhere
    const int Hq = g->H / 2, Wq = g->W / 2;
    if (!no_lp && g->Nw % 32 == 0 && Hq >= 8 && Wq >= 16) {
      const int lgc = Wq >= 32 ? 5 : 4;                                  // 8 x 32 or 16 x 16 pixels of a class
      const int TC = 1 << lgc, TR = 256 >> lgc;
      const size_t smem = ((size_t)(TR + 1) * (TC + 1) * 36 + (size_t)16 * g->Nw) * sizeof(float);
      const unsigned nbl = (unsigned)(g->B * lgm_cdiv(Hq, TR) * lgm_cdiv(Wq, TC));
      static size_t attr = 0;
      if (smem > attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(smalln_yx_lp_kernel<2, 2, 16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr = smem;
      }
      hipLaunchKernelGGL((smalln_yx_lp_kernel<2, 2, 16>), dim3(nbl, 4), dim3(256), smem, (hipStream_t)stream, q, lgc);
      LGM_LAUNCH_CHECK();
      return LGM_OK;
    }
    const long npq = q.npix / (g->stride * g->stride);
    const long want = (npq + 4 * (64 / lpp) - 1) / (4 * (64 / lpp));
    const unsigned nb = (unsigned)(want < 1024 ? want : 1024);
    hipLaunchKernelGGL((smalln_yx_kernel<2, 2>), dim3(nb, g->stride * g->stride), dim3(256), 0, (hipStream_t)stream, q);
    LGM_LAUNCH_CHECK();
    return LGM_OK;
  }
  IgemmArgs a{};
  a.a = y; a.w = w; a.bias = bias; a.res = res; a.out = x;
  a.a_pitch = y_pitch; a.res_pitch = res_pitch; a.out_pitch = x_pitch;
  a.B = g->B; a.H = g->H; a.W = g->W; a.Cw = g->Cw; a.Ho = g->Ho; a.Wo = g->Wo; a.Nw = g->Nw;
  a.KH = g->KH; a.KW = g->KW; a.stride = g->stride; a.pad = g->pad;
  a.M = g->B * g->H * g->W; a.N = g->Cw; a.K = g->KH * g->KW * g->Nw; a.Cg = g->Nw;
  return dispatch_igemm<MODE_YX>(a, workspace, workspace_bytes, wide_ok(x, x_pitch, res, res_pitch, bias),
                                 (hipStream_t)stream, stats, stats_tiles);
}

extern "C" int lgm_conv_yx(const LgmConvGeom* g, const float* y, int64_t y_pitch, const float* w,
                           const float* w_t, const float* bias, const float* res, int64_t res_pitch, float* x,
                           int64_t x_pitch, void* workspace, int64_t workspace_bytes, void* stream) {
  return conv_yx_impl(g, y, y_pitch, w, w_t, bias, res, res_pitch, x, x_pitch, workspace, workspace_bytes, nullptr,
                      nullptr, stream);
}

extern "C" int lgm_conv_yx_post(const LgmConvGeom* g, const float* y, int64_t y_pitch, const float* w,
                                const float* w_t, const float* bias, const float* res, int64_t res_pitch, float* x,
                                int64_t x_pitch, void* workspace, int64_t workspace_bytes, const LgmPostOp* post,
                                void* stream) {
  if (int rc = post_check(post, "conv_yx_post")) return rc;
  if (int rc = check_geom(g)) return rc;
  return with_post(post, x, x_pitch, (long)g->B * g->H * g->W, g->Cw, stream, [&] {
    return conv_yx_impl(g, y, y_pitch, w, w_t, bias, res, res_pitch, x, x_pitch, workspace, workspace_bytes, nullptr,
                        nullptr, stream);
  });
}

// the transposed convolution with the BatchNorm statistics of x left in stats (see lgm_conv_xy_stats)
extern "C" int lgm_conv_yx_stats(const LgmConvGeom* g, const float* y, int64_t y_pitch, const float* w,
                                 const float* w_t, float* x, int64_t x_pitch, void* workspace, int64_t workspace_bytes,
                                 float* stats, int* stats_tiles, void* stream) {
  LGM_REQUIRE(stats && stats_tiles, "conv_yx_stats: null statistics buffer");
  return conv_yx_impl(g, y, y_pitch, w, w_t, nullptr, nullptr, 0, x, x_pitch, workspace, workspace_bytes, stats,
                      stats_tiles, stream);
}

// =====================================================================================
// Weight gradient.  GEMM  M = Nw (rows n), N = T*Cw (cols q = tap*Cw + c), K = P = B*Ho*Wo.
// Both operands are pixel-major in memory, so the LDS tiles are [pixel][row] and every MFMA
// fragment is a conflict-free ds_read_b32 of 32 consecutive floats.
// =====================================================================================
namespace {

constexpr int WBK = 64;  // pixels per chunk (32 MFMAs per wave between barriers)

struct WgradArgs {
  const float* y;  // [P, Nw] rows
  const float* x;  // X-side NHWC
  float* out;      // gw (splits == 1) or workspace [splits][Nw*Q + Nw]
  float* bias_out; // gbias (splits == 1), workspace slab tail (splits > 1), or null
  float beta;      // only for splits == 1
  long slab;       // floats per split slab in the workspace
  long y_pitch, x_pitch;
  int B, H, W, Cw, Ho, Wo, Nw, KH, KW, stride, pad;
  int P, Q;        // pixels, T*Cw
  int tiles_m, tiles_n, splits, chunk;  // chunk = pixels per split (multiple of WBK)
  int swz;         // stand-alone launches: XCD-aware block order (wgrad_kernel)
  int lWo, lHo, lW, lH;                 // FAST: log2 of the (power-of-two) map sizes
};

// FAST: Ho, Wo, H, W powers of two, B*H*W < 2^24, byte offsets < 2^31: the pixel decode of the gather is shifts
// and masks and every load is a raw buffer load (zero fill = an offset at the descriptor's range) - ~12 VALU
// instructions per gathered row instead of ~60 (two integer divisions each), which beside fp32 MFMAs is the
// difference between a VALU-bound and an MFMA-bound loop (DESIGN finding 11).
// `lds`: 2 * WBK * (BM + BN) floats of workgroup memory, 16-byte aligned (the caller's: the stand-alone kernel's own
// array, or - gemm_bwd_pair_kernel - the dynamic allocation the input-gradient body uses in ITS blocks)
template <int BM, int BN, int TM, int TN, bool FAST>
__device__ __forceinline__ void wgrad_body(const WgradArgs& p, const int bidx, float* const lds) {
  static_assert(BM == 64 * TM && BN == 64 * TN, "2x2 wave grid");
  constexpr int A_TPR = BM / 4, A_RPP = 256 / A_TPR, A_PER = WBK / A_RPP > 0 ? WBK / A_RPP : 1;
  constexpr int B_TPR = BN / 4, B_RPP = 256 / B_TPR, B_PER = WBK / B_RPP > 0 ? WBK / B_RPP : 1;
  static_assert(A_RPP <= WBK && B_RPP <= WBK, "tile too narrow for 256 threads");
  float (*As)[WBK * BM] = reinterpret_cast<float (*)[WBK * BM]>(lds);
  float (*Bs)[WBK * BN] = reinterpret_cast<float (*)[WBK * BN]>(lds + 2 * WBK * BM);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;

  int bid = bidx;
  const int split = bid % p.splits;
  bid /= p.splits;
  const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
  const int m0 = tm * BM, q0 = tn * BN;
  const int p_begin = split * p.chunk;
  const int p_end = min(p.P, p_begin + p.chunk);

  // A (Y rows): thread -> (pixel row a_prow + A_RPP*i, float4 column a_ncol)
  const int a_ncol = (tid % A_TPR) * 4, a_prow = tid / A_TPR;
  const bool a_nok = (m0 + a_ncol) < p.Nw;
  // B (X gather): column q fixed per thread -> tap / channel decode once
  const int b_qcol = (tid % B_TPR) * 4, b_prow = tid / B_TPR;
  const int q = q0 + b_qcol;
  const bool b_qok = q < p.Q;
  const int tap = b_qok ? q / p.Cw : 0;
  const int cch = q - tap * p.Cw;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;

  f32x4 ra[A_PER], rb[B_PER];
  __amdgpu_buffer_rsrc_t rsrc_y, rsrc_x;
  unsigned nrec_y = 0, nrec_x = 0, ya[A_PER];
  const int khp = kh - p.pad, kwp = kw - p.pad;
  if constexpr (FAST) {
    auto make_rsrc = [](const float* base, unsigned nrec) {
      const unsigned long long ab = reinterpret_cast<unsigned long long>(base);
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ab);
      const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(ab >> 32));
      return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                               __builtin_amdgcn_readfirstlane(nrec), 0x00020000);
    };
    nrec_y = (unsigned)((long)p.P * p.y_pitch * 4);
    nrec_x = (unsigned)((long)p.B * p.H * p.W * p.x_pitch * 4);
    rsrc_y = make_rsrc(p.y, nrec_y);
    rsrc_x = make_rsrc(p.x, nrec_x);
#pragma unroll
    for (int i = 0; i < A_PER; ++i)
      ya[i] = a_nok ? (unsigned)(((long)(a_prow + A_RPP * i) * p.y_pitch + m0 + a_ncol) * 4) : nrec_y;
  }
  auto load_chunk_fast = [&](int pp0) {
    const unsigned soff_y = __builtin_amdgcn_readfirstlane((unsigned)((long)pp0 * p.y_pitch * 4));
    const int left = __builtin_amdgcn_readfirstlane(p.P - pp0);       // rows of this chunk that exist
#pragma unroll
    for (int i = 0; i < A_PER; ++i)
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            rsrc_y, (a_prow + A_RPP * i) < left ? ya[i] : nrec_y, soff_y, 0));
    const unsigned xpb = (unsigned)p.x_pitch * 4u;
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const unsigned pix = (unsigned)(pp0 + b_prow + B_RPP * i);
      const unsigned ow = pix & (unsigned)(p.Wo - 1), oh = (pix >> p.lWo) & (unsigned)(p.Ho - 1);
      const unsigned b = pix >> (p.lWo + p.lHo);                     // >= B past the last pixel: offset out of range
      const int ih = (int)oh * p.stride + khp, iw = (int)ow * p.stride + kwp;
      const bool ok = b_qok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const unsigned idx = (((b << p.lH) + (unsigned)ih) << p.lW) + (unsigned)iw;
      const unsigned off = (unsigned)__umul24(idx, xpb) + (unsigned)cch * 4u;
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, ok ? off : nrec_x, 0, 0));
    }
  };
  auto load_chunk = [&](int pp0) {
    if constexpr (FAST) {
      load_chunk_fast(pp0);
      return;
    }
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int pix = pp0 + a_prow + A_RPP * i;
      const bool ok = a_nok && pix < p_end;
      ra[i] = ok ? *reinterpret_cast<const f32x4*>(p.y + (long)pix * p.y_pitch + m0 + a_ncol)
                 : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int pix = pp0 + b_prow + B_RPP * i;
      bool ok = b_qok && pix < p_end;
      const int pc = ok ? pix : 0;
      const int ow = pc % p.Wo, t = pc / p.Wo;
      const int oh = t % p.Ho, b = t / p.Ho;
      const int ih = oh * p.stride - p.pad + kh, iw = ow * p.stride - p.pad + kw;
      ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      rb[i] = ok ? *reinterpret_cast<const f32x4*>(p.x + ((long)(b * p.H + ih) * p.W + iw) * p.x_pitch + cch)
                 : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i)
      *reinterpret_cast<f32x4*>(&As[buf][(a_prow + A_RPP * i) * BM + a_ncol]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      *reinterpret_cast<f32x4*>(&Bs[buf][(b_prow + B_RPP * i) * BN + b_qcol]) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (p_end - p_begin + WBK - 1) / WBK;
  if (nk > 0) {
    load_chunk(p_begin);
    store_chunk(0);
  }
  __syncthreads();
  // fused bias gradient: the tn == 0 blocks also column-sum their Y tile (thread -> column
  // tid % BM, pixel lane tid / BM; fixed order => deterministic)
  const bool do_bias = p.bias_out != nullptr && tn == 0;
  constexpr int BL = 256 / BM;          // pixel lanes
  const int bcol = tid % BM, blane = tid / BM;
  float bsum = 0.f;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_chunk(p_begin + (kt + 1) * WBK);
    if (do_bias) {
#pragma unroll
      for (int r = 0; r < WBK / BL; ++r) bsum += As[cur][(blane + BL * r) * BM + bcol];
    }
    const float* as = &As[cur][lh * BM + wm * 32 * TM + lr];
    const float* bs = &Bs[cur][lh * BN + wn * 32 * TN + lr];
#pragma unroll
    for (int s = 0; s < WBK / 2; ++s) {
      float fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = as[(2 * s) * BM + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = bs[(2 * s) * BN + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_chunk(cur ^ 1);
    __syncthreads();
  }

  float* out = p.out + (p.splits > 1 ? (long)split * p.slab : 0L);
  const bool acc_out = p.splits == 1 && p.beta != 0.f;
  if (m0 + BM <= p.Nw && q0 + BN <= p.Q && p.Q % 4 == 0 && lgm_aligned16_dev(out)) {
    // full tile: wave-private LDS transpose (the K loop ended with a barrier: As is free), previous values
    // loaded up front, unconditional 16-byte stores - the per-element conditional dword stores below made
    // the compiler wait for every previous store (s_waitcnt vmcnt(0) x 16 per tile)
    float* Ts = &As[0][0] + wid * LGM_TS_FLOATS;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int qc = q0 + wn * 32 * TN + j * 32 + (lane & 7) * 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        long o[4];
        f32x4 prev[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o[q] = (long)(m0 + wm * 32 * TM + i * 32 + (lane >> 3) + 8 * q) * p.Q + qc;
          prev[q] = acc_out ? *reinterpret_cast<const f32x4*>(out + o[q]) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        lgm_wave_lds_sync();
        lgm_tile_to_lds(acc[i][j], Ts, lane);
        lgm_wave_lds_sync();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v = lgm_tile_row4(Ts, lane, q);
          if (acc_out) v += prev[q] * p.beta;
          *reinterpret_cast<f32x4*>(out + o[q]) = v;
        }
      }
    }
    __syncthreads();     // the bias reduction below reuses As
  } else
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int qq = q0 + wn * 32 * TN + j * 32 + lr;
    if (qq >= p.Q) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < p.Nw) {
          const long o = (long)n * p.Q + qq;
          float v = acc[i][j][r];
          if (p.splits == 1 && p.beta != 0.f) v += p.beta * out[o];
          out[o] = v;
        }
      }
    }
  }
  if (do_bias) {   // uniform per block; the K loop ended with a barrier, so LDS is free
    As[0][blane * BM + bcol] = bsum;
    __syncthreads();
    if (tid < BM && m0 + tid < p.Nw) {
      float v = 0.f;
#pragma unroll
      for (int l = 0; l < BL; ++l) v += As[0][l * BM + tid];
      float* bo = p.bias_out + (p.splits > 1 ? (long)split * p.slab : 0L) + m0 + tid;
      if (p.splits == 1 && p.beta != 0.f) v += p.beta * bo[0];
      bo[0] = v;
    }
  }
}

template <int BM, int BN, int TM, int TN, bool FAST>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
  __shared__ __align__(16) float wlds[2 * WBK * (BM + BN)];
  wgrad_body<BM, BN, TM, TN, FAST>(p, p.swz ? lgm_xcd_swizzle((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x, wlds);
}

// Input gradient (blocks [0, n_ig): the 64 x 64 uniform-tap implicit GEMM) and weight gradient (the rest: the generic
// 64 x 64 kernel) of ONE layer in ONE launch - the 1x1 convolutions and linears of the UNet (to_qkv / to_out
// ddpm.py:214-223, res_conv :184, Downsample :100-104).  Same bodies as the separate kernels: bit-identical results.
// At the per-GPU batches of a strong-scaled run each of the two is a ~9 us launch on a quarter of the chip.
// ONE dynamic allocation serves whichever body a block runs (max of the two: 64 KB, two workgroups per CU).  With the
// weight-gradient body's operands in a static array of their own the kernel asked for 64 + 34 KB = 98 KB, ONE workgroup
// per CU - and its 288 ... 384-block grids ran a second, nearly empty round (profiles/r06_negative_results.txt, item 9).
__global__ __launch_bounds__(256) void gemm_bwd_pair_kernel(const IgemmArgs pi, const WgradArgs pw, const int n_ig) {
  extern __shared__ __align__(16) float smem[];
  if ((int)blockIdx.x < n_ig) igemm_body<MODE_YX, 64, 64, 1, 1, true>(pi, (int)blockIdx.x);
  else wgrad_body<64, 64, 1, 1, true>(pw, (int)blockIdx.x - n_ig, smem);
}

// The stand-alone weight gradients of two ... four layers in ONE launch: their grids one after the other, every layer with
// the plan (splits, slabs) it has on its own - same blocks, same sums, bit-identical results.  Nothing reads a weight
// gradient before the optimizer, so a deferred backward pass lets these launches wait for each other (the queue below):
// 13 launches of 5 ... 17 us in the B = 128 DDPM step become 5.
__global__ __launch_bounds__(256) void wgrad_group_kernel(const WgradArgs a, const WgradArgs b, const WgradArgs c,
                                                          const WgradArgs d, const int e0, const int e1, const int e2) {
  __shared__ __align__(16) float wlds[2 * WBK * (64 + 64)];
  const int bid = (int)blockIdx.x;
  if (bid < e0) wgrad_body<64, 64, 1, 1, true>(a, a.swz ? lgm_xcd_swizzle(bid, e0) : bid, wlds);
  else if (bid < e1) wgrad_body<64, 64, 1, 1, true>(b, b.swz ? lgm_xcd_swizzle(bid - e0, e1 - e0) : bid - e0, wlds);
  else if (bid < e2) wgrad_body<64, 64, 1, 1, true>(c, c.swz ? lgm_xcd_swizzle(bid - e1, e2 - e1) : bid - e1, wlds);
  else wgrad_body<64, 64, 1, 1, true>(d, d.swz ? lgm_xcd_swizzle(bid - e2, (int)gridDim.x - e2) : bid - e2, wlds);
}

// Per-thread queue of such launches.  Off unless the caller switches it on around a call (lgm_wgrad_queue_enable): only a
// caller that will call lgm_wgrad_queue_flush before anything reads the gradients - and keeps the operands alive until
// then - may let a launch wait (lgm_hip.nn.GradCtx does, for its deferred passes).
struct WgradQueue {
  WgradArgs a[4];
  unsigned nb[4];
  int n;
  bool on;
  hipStream_t s;
};
static thread_local WgradQueue t_wq = {};

static int wq_launch() {
  WgradQueue& q = t_wq;
  if (q.n == 0) return LGM_OK;
  if (q.n == 1) {
    lgm_note_kernel(LGM_KNAME("wgrad_kernel<64, 64, 1, 1, true>"));
    hipLaunchKernelGGL((wgrad_kernel<64, 64, 1, 1, true>), dim3(q.nb[0]), dim3(256), 0, q.s, q.a[0]);
  } else {
    for (int k = q.n; k < 4; ++k) {
      q.a[k] = q.a[q.n - 1];          // never reached: empty block range
      q.nb[k] = 0;
    }
    const int e0 = (int)q.nb[0], e1 = e0 + (int)q.nb[1], e2 = e1 + (int)q.nb[2];
    lgm_note_kernel(LGM_KNAME("wgrad_group_kernel"));
    hipLaunchKernelGGL(wgrad_group_kernel, dim3((unsigned)(e2 + (int)q.nb[3])), dim3(256), 0, q.s, q.a[0], q.a[1], q.a[2], q.a[3],
                       e0, e1, e2);
  }
  q.n = 0;
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

static int wq_push(const WgradArgs& a, unsigned blocks, hipStream_t s) {
  WgradQueue& q = t_wq;
  if (q.n > 0 && q.s != s)
    if (int rc = wq_launch()) return rc;
  q.s = s;
  q.a[q.n] = a;
  q.nb[q.n] = blocks;
  if (++q.n == 4) return wq_launch();
  return LGM_OK;
}

// Deterministic split-K reduction: out[i] = beta*out[i] + sum_s ws[s*slab + i] (fixed order:
// 4 split lanes summed in LDS in lane order).  i < n_w -> gw, n_w <= i < n_w + n_b -> gbias.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, long slab,
                                                           float* __restrict__ gw, long n_w,
                                                           float* __restrict__ gb, long n_b, int splits,
                                                           float beta) {
  __shared__ __align__(16) float sh[4][64 * 4];
  const int col = threadIdx.x & 63, lane = threadIdx.x >> 6;
  const long i = ((long)blockIdx.x * 64 + col) * 4;
  const long n = n_w + n_b;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    // four independent slab streams per thread (k, k+4, k+8, k+12): four 16-byte loads in flight;
    // slabs past `splits` are clamped and weighted 0 so the loop body has no branches
    f32x4 t0 = s, t1 = s, t2 = s, t3 = s;
    for (int k = lane; k < splits; k += 16) {
      const int k1 = k + 4, k2 = k + 8, k3 = k + 12;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(ws + (long)k * slab + i);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(ws + (long)(k1 < splits ? k1 : k) * slab + i);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(ws + (long)(k2 < splits ? k2 : k) * slab + i);
      const f32x4 v3 = *reinterpret_cast<const f32x4*>(ws + (long)(k3 < splits ? k3 : k) * slab + i);
      t0 += v0;
      t1 += v1 * (k1 < splits ? 1.f : 0.f);
      t2 += v2 * (k2 < splits ? 1.f : 0.f);
      t3 += v3 * (k3 < splits ? 1.f : 0.f);
    }
    s = (t0 + t1) + (t2 + t3);
  }
  *reinterpret_cast<f32x4*>(&sh[lane][col * 4]) = s;
  __syncthreads();
  if (lane == 0 && i < n) {
    f32x4 t = (*reinterpret_cast<const f32x4*>(&sh[0][col * 4]) + *reinterpret_cast<const f32x4*>(&sh[1][col * 4])) +
              (*reinterpret_cast<const f32x4*>(&sh[2][col * 4]) + *reinterpret_cast<const f32x4*>(&sh[3][col * 4]));
    float* dst = i < n_w ? gw + i : gb + (i - n_w);
    if (beta != 0.f) t += *reinterpret_cast<const f32x4*>(dst) * beta;
    *reinterpret_cast<f32x4*>(dst) = t;
  }
}

void wgrad_plan(const LgmConvGeom* g, int* splits, int* chunk) {
  const long P = (long)g->B * g->Ho * g->Wo;
  const long Q = (long)g->KH * g->KW * g->Cw;
  const long tiles = (long)lgm_cdiv(g->Nw, 64) * lgm_cdiv(Q, 64);
  // ~512 workgroups (two per CU): measured on the WGAN-GP step 256 / 512 / 1024 / 2048 / 4096 -> 24,550 / 25,100 /
  // 24,800 / 24,550 / 23,950 images/s (more splits = more slab traffic); no effect on the DDPM step
  static const long target = getenv("LGM_WGRAD_TARGET") ? atol(getenv("LGM_WGRAD_TARGET")) : 512;   // tuning knob
  long s = (target + tiles - 1) / tiles;
  const long max_s = (P + 255) / 256;            // at least 256 pixels per split
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  long c = (P + s - 1) / s;
  c = (c + WBK - 1) / WBK * WBK;
  s = (P + c - 1) / c;
  *splits = (int)s;
  *chunk = (int)c;
}

}  // namespace

extern "C" int64_t lgm_conv_wgrad_workspace(const LgmConvGeom* g) {
  if (check_geom(g) != LGM_OK) return -1;
  int splits, chunk;
  wgrad_plan(g, &splits, &chunk);          // the generic plan is the fallback of the 3x3 path (pitch limits)
  if (use_3x3() && lgm_wgrad3x3_supported(g)) {
    int s3, tps, total;
    lgm_wgrad3x3_plan(g, &s3, &tps, &total);
    if (s3 > splits) splits = s3;
  }
  if (use_3x3() && use_wino() && lgm_wino_wgrad_supported(g)) {
    int sw, cps, total;
    lgm_wino_wgrad_plan(g, &sw, &cps, &total);
    if (sw > splits) splits = sw;
  }
  if (use_3x3() && use_wino() && lgm_wino4_wgrad_use(g)) {
    int sw, gps, total;
    lgm_wino4_wgrad_plan(g, lgm_cu_budget(), &sw, &gps, &total);
    if (sw > splits) splits = sw;
  }
  if (g->KH == 1 && g->KW == 1 && g->Nw % 64 == 0 && g->Cw % 64 == 0) {
    int s1, per;
    lgm_wgrad1x1_plan(g, &s1, &per);
    if (s1 > splits) splits = s1;
  }
  if (splits == 1) return 16;
  const int64_t slab = (int64_t)g->Nw * g->KH * g->KW * g->Cw + g->Nw;
  return (int64_t)splits * slab * (int64_t)sizeof(float);
}

static int conv_wgrad_impl(const LgmConvGeom* g, const float* y, int64_t y_pitch, const float* x, int64_t x_pitch,
                           float* gw, float* gbias, float beta, void* workspace, int64_t workspace_bytes,
                           int64_t* desc, void* stream) {
  if (int rc = check_geom(g)) return rc;
  LGM_REQUIRE(y && x && gw, "conv_wgrad: null pointer");
  LGM_REQUIRE(g->Cw % 4 == 0 && g->Nw % 4 == 0, "conv_wgrad: Cw=%d, Nw=%d must be multiples of 4", g->Cw, g->Nw);
  LGM_REQUIRE(y_pitch % 4 == 0 && x_pitch % 4 == 0 && lgm_aligned16(y) && lgm_aligned16(x) && lgm_aligned16(gw) &&
                  (!gbias || lgm_aligned16(gbias)),
              "conv_wgrad: tensors must be 16B aligned with pitch %% 4 == 0");
  WgradArgs a{};
  a.y = y; a.x = x; a.beta = beta; a.y_pitch = y_pitch; a.x_pitch = x_pitch;
  a.B = g->B; a.H = g->H; a.W = g->W; a.Cw = g->Cw; a.Ho = g->Ho; a.Wo = g->Wo; a.Nw = g->Nw;
  a.KH = g->KH; a.KW = g->KW; a.stride = g->stride; a.pad = g->pad;
  a.P = g->B * g->Ho * g->Wo; a.Q = g->KH * g->KW * g->Cw;
  // 32-bit buffer offsets: bytes (+ the one-row halo shift of the X descriptor) stay below 2^31
  const bool fast3 = use_3x3() && lgm_wgrad3x3_supported(g) &&
                     ((long)g->B * g->H * g->W + g->W + 1) * x_pitch < (1L << 29) &&
                     (long)g->B * g->H * g->W * y_pitch < (1L << 29);
  static const bool no_w1x1 = getenv("LGM_NO_W1X1") != nullptr;   // A/B switch
  const bool fast1 = !fast3 && !no_w1x1 && lgm_wgrad1x1_supported(g, y_pitch, x_pitch);
  // Winograd: same operand limits as the direct 3x3 kernel (the kernel always writes slabs, the reducer applies beta)
  const bool fastw = fast3 && use_wino() && lgm_wino_wgrad_supported(g);
  const bool fastw4 = fastw && lgm_wino4_wgrad_use(g);          // large maps: F(4x4,3x3)
  int tps3 = 0, total3 = 0, per1 = 0, cpsw = 0, totalw = 0;
  if (fastw4)
    lgm_wino4_wgrad_plan(g, lgm_cu_budget(), &a.splits, &cpsw, &totalw);
  else if (fastw)
    lgm_wino_wgrad_plan(g, &a.splits, &cpsw, &totalw);
  else if (fast3)
    lgm_wgrad3x3_plan(g, &a.splits, &tps3, &total3);
  else if (fast1)
    lgm_wgrad1x1_plan(g, &a.splits, &per1);
  else
    wgrad_plan(g, &a.splits, &a.chunk);
  const long n_w = (long)a.Nw * a.Q;
  a.slab = n_w + a.Nw;
  if (a.splits > 1) {
    LGM_REQUIRE(workspace && workspace_bytes >= (int64_t)a.splits * a.slab * (int64_t)sizeof(float) && lgm_aligned16(workspace),
                "conv_wgrad: workspace too small (%lld bytes needed)", (long long)a.splits * a.slab * 4);
    a.out = (float*)workspace;
    a.bias_out = gbias ? (float*)workspace + n_w : nullptr;
  } else {
    a.out = gw;
    a.bias_out = gbias;
  }
  a.tiles_m = lgm_cdiv(a.Nw, 64);
  a.tiles_n = lgm_cdiv(a.Q, 64);
  hipStream_t s = (hipStream_t)stream;
  if (fastw4) {
    if (int rc = lgm_wino4_wgrad_launch(g, y, y_pitch, x, x_pitch, a.out, gbias ? 1 : 0, a.slab, a.splits, cpsw, totalw, s))
      return rc;
  } else if (fastw) {
    if (int rc = lgm_wino_wgrad_launch(g, y, y_pitch, x, x_pitch, a.out, gbias ? 1 : 0, a.slab, a.splits, cpsw, totalw, s))
      return rc;
  } else if (fast3) {
    lgm_note_kernel(LGM_KNAME("lgm3x3::wgrad3x3_kernel"));
    if (int rc = lgm_wgrad3x3_launch(g, y, y_pitch, x, x_pitch, a.out, a.bias_out, beta, a.slab, a.splits, tps3,
                                     total3, s))
      return rc;
  } else if (fast1) {
    lgm_note_kernel(LGM_KNAME("wgrad1x1_kernel"));
    if (int rc = lgm_wgrad1x1_launch(g, y, y_pitch, x, x_pitch, a.out, a.bias_out, beta, a.slab, a.splits, per1, s))
      return rc;
  } else {
    auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    a.lWo = lg2(a.Wo); a.lHo = lg2(a.Ho); a.lW = lg2(a.W); a.lH = lg2(a.H);
    static const bool no_fast = getenv("LGM_NO_WGRAD_FAST") != nullptr;   // A/B switch
    const bool fast = !no_fast && a.lWo >= 0 && a.lHo >= 0 && a.lW >= 0 && a.lH >= 0 &&
                      (long)a.B * a.H * a.W < (1L << 24) && x_pitch * 4 < (1L << 24) &&
                      (long)a.B * a.H * a.W * x_pitch * 4 < (1L << 31) && ((long)a.P + WBK) * y_pitch * 4 < (1L << 31);
    // Opt-in (LGM_WGRAD_BIG=1), measured SLOWER on the WGAN-GP step (24,150 vs 24,760 images/s): 128 output channels
    // per workgroup - two accumulators per wave, every gathered X fragment feeds two MFMAs - halves the workgroups,
    // and these layers need the parallelism more than the operand reuse.
    static const bool want_big = getenv("LGM_WGRAD_BIG") != nullptr;
    const bool big = fast && want_big && a.Nw % 128 == 0 && (long)(a.Nw / 128) * a.tiles_n * a.splits >= 512;
    lgm_note_kernel(big ? LGM_KNAME("wgrad_kernel<128, 64, 2, 1, true>")
                        : fast ? LGM_KNAME("wgrad_kernel<64, 64, 1, 1, true>") : LGM_KNAME("wgrad_kernel<64, 64, 1, 1, false>"));
    static_assert(sizeof(WgradArgs) <= sizeof(t_pair.wg), "PairCtx::wg too small");
    a.swz = swz_on();
    if (!big && fast && t_pair.active && !t_pair.rec_w) {       // launched by lgm_conv_bwd_pair, with the input gradient
      memcpy(t_pair.wg, &a, sizeof(WgradArgs));
      t_pair.wg_blocks = (unsigned)(a.tiles_m * a.tiles_n * a.splits);
      t_pair.rec_w = true;
    } else if (big) {
      a.tiles_m = a.Nw / 128;
      hipLaunchKernelGGL((wgrad_kernel<128, 64, 2, 1, true>), dim3((unsigned)(a.tiles_m * a.tiles_n * a.splits)), dim3(256), 0, s, a);
    } else if (fast && t_wq.on && desc) {            // deferred pass: the launch waits for up to three partners
      if (int rc = wq_push(a, (unsigned)(a.tiles_m * a.tiles_n * a.splits), s)) return rc;
    } else if (fast)
      hipLaunchKernelGGL((wgrad_kernel<64, 64, 1, 1, true>), dim3((unsigned)(a.tiles_m * a.tiles_n * a.splits)), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((wgrad_kernel<64, 64, 1, 1, false>), dim3((unsigned)(a.tiles_m * a.tiles_n * a.splits)), dim3(256), 0, s, a);
    LGM_LAUNCH_CHECK();
  }
  const long n_b = gbias ? a.Nw : 0;
  if (desc) {            // deferred: the caller reduces many layers' slabs with ONE lgm_wgrad_reduce_batch launch
    union { float f; int64_t i; } bb;
    bb.i = 0;
    bb.f = beta;
    desc[0] = (int64_t)(uintptr_t)workspace; desc[1] = a.slab; desc[2] = (int64_t)(uintptr_t)gw; desc[3] = n_w;
    desc[4] = (int64_t)(uintptr_t)gbias; desc[5] = n_b; desc[6] = a.splits; desc[7] = bb.i;
    return LGM_OK;
  }
  if (a.splits > 1) {
    if (t_pair.active && t_pair.rec_w) {          // the kernel has not run yet: reduce after the pair launch
      t_pair.red_w = true;
      t_pair.w_ws = (const float*)workspace; t_pair.w_slab = a.slab; t_pair.w_gw = gw; t_pair.w_nw = n_w;
      t_pair.w_gb = gbias; t_pair.w_nb = n_b; t_pair.w_splits = a.splits; t_pair.w_beta = beta;
      return LGM_OK;
    }
    const long groups = (n_w + n_b + 3) / 4;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)lgm_cdiv(groups, 64)), dim3(256), 0, s,
                       (const float*)workspace, a.slab, gw, n_w, gbias, n_b, a.splits, beta);
    LGM_LAUNCH_CHECK();
  }
  return LGM_OK;
}

// the single-layer slab reducer for other translation units (linattn_fused.hip)
int lgm_wgrad_reduce_launch(const float* ws, long slab, float* gw, long n_w, float* gb, long n_b, int splits, float beta,
                            hipStream_t s) {
  const long groups = (n_w + n_b + 3) / 4;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)lgm_cdiv(groups, 64)), dim3(256), 0, s, ws, slab, gw, n_w, gb, n_b,
                     splits, beta);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

// ---- backward pair of a 3x3 layer: input gradient + weight gradient in ONE launch (csrc/winograd.hip) -------------
bool lgm_wino_supported(const LgmConvGeom* g, int gather_channels, int out_channels);
struct WinoPairPlan {
  int csplits, wsplits, cps, total_chunks;
};
WinoPairPlan lgm_wino_pair_plan(const LgmConvGeom* g, bool fused);
int lgm_wino_pair_launch(const LgmConvGeom* g, const float* gy, long gy_pitch, const float* x, long x_pitch,
                         const float* u_b, const float* res, long res_pitch, float* gx, long gx_pitch, void* dws,
                         long dws_bytes, int64_t* partial, float* slabs, int bias, long slab, hipStream_t s);

/* out[0] / out[1]: bytes of the input gradient's split-K workspace / of the weight gradient's slab workspace that
 * lgm_conv3x3_wino_bwd needs for this geometry (the pair plans its two split counts jointly) */
extern "C" int lgm_conv3x3_wino_bwd_workspaces(const LgmConvGeom* g, int partial, int64_t* out) {
  LGM_REQUIRE(g && out && check_geom(g) == LGM_OK, "conv3x3_wino_bwd_workspaces: bad arguments");
  const WinoPairPlan pl = lgm_wino_pair_plan(g, partial != 0);
  out[0] = pl.csplits > 1 ? (int64_t)pl.csplits * g->B * g->H * g->W * g->Cw * (int64_t)sizeof(float) : 0;
  out[1] = (int64_t)pl.wsplits * ((int64_t)g->Nw * 9 * g->Cw + g->Nw) * (int64_t)sizeof(float);
  return LGM_OK;
}

extern "C" int64_t lgm_conv3x3_wino_bwd_supported(const LgmConvGeom* g, int64_t gy_pitch, int64_t x_pitch,
                                                  int64_t gx_pitch, int64_t res_pitch) {
  if (check_geom(g) != LGM_OK) return 0;
  static const bool no_pair = getenv("LGM_NO_PAIR") != nullptr;   // A/B switch: separate launches
  if (no_pair || !use_3x3() || !use_wino()) return 0;
  const long pix = (long)g->B * g->H * g->W + g->W + 1;
  if (!(lgm_wgrad3x3_supported(g) && lgm_wino_wgrad_supported(g) && lgm_wino_supported(g, g->Nw, g->Cw))) return 0;
  if (gy_pitch % 4 || x_pitch % 4 || gx_pitch % 4 || res_pitch % 4) return 0;
  return (pix * x_pitch < (1L << 29) && pix * gy_pitch < (1L << 29) && pix * gx_pitch < (1L << 29) &&
          pix * res_pitch < (1L << 29)) ? 1 : 0;
}

extern "C" int lgm_conv3x3_wino_bwd(const LgmConvGeom* g, const float* gy, int64_t gy_pitch, const float* x,
                                    int64_t x_pitch, const float* u_b, const float* res, int64_t res_pitch, float* gx,
                                    int64_t gx_pitch, void* dgrad_ws, int64_t dgrad_ws_bytes, int64_t* partial, float* gw,
                                    float* gbias, float beta, void* wgrad_ws, int64_t wgrad_ws_bytes, int64_t* desc,
                                    void* stream) {
  LGM_REQUIRE(g && gy && x && u_b && gx && gw, "conv3x3_wino_bwd: null pointer");
  LGM_REQUIRE(lgm_conv3x3_wino_bwd_supported(g, gy_pitch, x_pitch, gx_pitch, res ? res_pitch : 0),
              "conv3x3_wino_bwd: unsupported geometry / pitches (ask lgm_conv3x3_wino_bwd_supported first)");
  LGM_REQUIRE(lgm_aligned16(gy) && lgm_aligned16(x) && lgm_aligned16(u_b) && lgm_aligned16(gx) && lgm_aligned16(gw) &&
                  (!gbias || lgm_aligned16(gbias)) && (!res || lgm_aligned16(res)) && gy_pitch >= g->Nw &&
                  x_pitch >= g->Cw && gx_pitch >= g->Cw && (!res || res_pitch >= g->Cw) && !(partial && res),
              "conv3x3_wino_bwd: 16-byte aligned operands expected (and no residual with partial planes)");
  const WinoPairPlan pl = lgm_wino_pair_plan(g, partial != nullptr);
  const int splits = pl.wsplits;
  const long n_w = (long)g->Nw * 9 * g->Cw;
  const long slab = n_w + g->Nw;
  LGM_REQUIRE(wgrad_ws && wgrad_ws_bytes >= (int64_t)splits * slab * (int64_t)sizeof(float) && lgm_aligned16(wgrad_ws),
              "conv3x3_wino_bwd: weight-gradient workspace too small (%lld bytes needed)", (long long)splits * slab * 4);
  LGM_REQUIRE(pl.csplits == 1 || (dgrad_ws && lgm_aligned16(dgrad_ws) &&
                                  dgrad_ws_bytes >= (int64_t)pl.csplits * g->B * g->H * g->W * g->Cw * (int64_t)sizeof(float)),
              "conv3x3_wino_bwd: input-gradient workspace too small (ask lgm_conv3x3_wino_bwd_workspaces)");
  hipStream_t s = (hipStream_t)stream;
  if (int rc = lgm_wino_pair_launch(g, gy, gy_pitch, x, x_pitch, u_b, res, res_pitch, gx, gx_pitch, dgrad_ws,
                                    dgrad_ws_bytes, partial, (float*)wgrad_ws, gbias ? 1 : 0, slab, s))
    return rc;
  const long n_b = gbias ? g->Nw : 0;
  if (desc) {
    union { float f; int64_t i; } bb;
    bb.i = 0;
    bb.f = beta;
    desc[0] = (int64_t)(uintptr_t)wgrad_ws; desc[1] = slab; desc[2] = (int64_t)(uintptr_t)gw; desc[3] = n_w;
    desc[4] = (int64_t)(uintptr_t)gbias; desc[5] = n_b; desc[6] = splits; desc[7] = bb.i;
    return LGM_OK;
  }
  const long groups = (n_w + n_b + 3) / 4;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)lgm_cdiv(groups, 64)), dim3(256), 0, s, (const float*)wgrad_ws,
                     slab, gw, n_w, gbias, n_b, splits, beta);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

// Backward of a generic layer (any geometry): weight / bias gradient AND input gradient.  When the dispatchers pick
// the two kernels that can share a grid (see PairCtx) both run in ONE launch; otherwise this is exactly
// lgm_conv_wgrad[_deferred] followed by lgm_conv_yx.  dgrad_ws and wgrad_ws must not overlap (the kernels run side by side).
static int conv_bwd_pair_impl(const LgmConvGeom* g, const float* gy, int64_t gy_pitch, const float* x,
                              int64_t x_pitch, const float* w, const float* w_t, const float* res, int64_t res_pitch,
                              float* gx, int64_t gx_pitch, void* dgrad_ws, int64_t dgrad_ws_bytes, float* gw,
                              float* gbias, float beta, void* wgrad_ws, int64_t wgrad_ws_bytes, int64_t* desc,
                              const LgmPostOp* post, void* stream) {
  static const bool no_pair = getenv("LGM_NO_PAIR") != nullptr;   // A/B switch: separate launches
  hipStream_t s = (hipStream_t)stream;
  t_pair = PairCtx{};
  t_pair.active = !no_pair;
  int rc = conv_wgrad_impl(g, gy, gy_pitch, x, x_pitch, gw, gbias, beta, wgrad_ws, wgrad_ws_bytes, desc, stream);
  bool post_done = true;
  if (rc == LGM_OK) {
    // the post-op belongs to the input gradient only: parked for the duration of ITS dispatch (see with_post)
    if (post && post->bn_tiles) *post->bn_tiles = 0;
    t_post.post = post;
    t_post.done = false;
    rc = conv_yx_impl(g, gy, gy_pitch, w, w_t, nullptr, res, res_pitch, gx, gx_pitch, dgrad_ws, dgrad_ws_bytes, nullptr,
                      nullptr, stream);
    post_done = post == nullptr || t_post.done;
    t_post.post = nullptr;
    t_post.done = false;
  }
  const PairCtx c = t_pair;
  t_pair = PairCtx{};
  if (rc != LGM_OK) return rc;
  WgradArgs wa;
  memcpy(&wa, c.wg, sizeof(WgradArgs));
  // one launch only while both grids fit the chip together (67 KB of LDS per workgroup: two per CU): large layers keep
  // their own launches, where the input-gradient kernel alone gets four workgroups per CU (B = 128: 1 % slower paired)
  static const unsigned pair_max = getenv("LGM_PAIR_MAX") ? (unsigned)atoi(getenv("LGM_PAIR_MAX")) : 512u;
  const bool together = c.rec_i && c.rec_w && c.ig_blocks + c.wg_blocks <= pair_max;
  if (together) {
    static size_t attr = 0;
    constexpr size_t wg_smem = (size_t)2 * WBK * (64 + 64) * sizeof(float);      // wgrad_body<64, 64>: 64 KB
    const size_t pair_smem = c.ig_smem > wg_smem ? c.ig_smem : wg_smem;
    if (pair_smem > attr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bwd_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)pair_smem);
      attr = pair_smem;
    }
    lgm_note_kernel(LGM_KNAME("gemm_bwd_pair_kernel"));
    hipLaunchKernelGGL(gemm_bwd_pair_kernel, dim3(c.ig_blocks + c.wg_blocks), dim3(256), pair_smem, s, c.ig, wa,
                       (int)c.ig_blocks);
  } else {
    if (c.rec_w && t_wq.on && desc) {
      if (int r2 = wq_push(wa, c.wg_blocks, s)) return r2;
    } else if (c.rec_w) {
      lgm_note_kernel(LGM_KNAME("wgrad_kernel<64, 64, 1, 1, true>"));
      hipLaunchKernelGGL((wgrad_kernel<64, 64, 1, 1, true>), dim3(c.wg_blocks), dim3(256), 0, s, wa);
    }
    if (c.rec_i) {
      IgemmArgs ia = c.ig;
      t_pair = PairCtx{};                    // inactive: this time the launch site launches
      if (int r2 = launch_igemm_t<MODE_YX, 64, 64, 1, 1, true>(ia, s)) return r2;
    }
  }
  LGM_LAUNCH_CHECK();
  if (c.red_i) {
    if (c.r_post) {
      const long items = c.r_M * (c.r_N / 4);
      hipLaunchKernelGGL(splitk_reduce_post_kernel, dim3((unsigned)lgm_cdiv(items, 256)), dim3(256), 0, s, c.r_ws,
                         c.r_stride, c.r_splits, c.r_bias, c.r_res, c.r_res_pitch, c.r_out, c.r_out_pitch, c.r_M, c.r_N,
                         c.r_act, c.r_slope, c.r_mask, c.r_mask_pitch, c.r_mask_slope);
      LGM_LAUNCH_CHECK();
    } else if (int r2 = lgm_splitk_reduce_launch(c.r_ws, c.r_stride, c.r_splits, c.r_bias, c.r_res, c.r_res_pitch, c.r_out,
                                                 c.r_out_pitch, c.r_M, c.r_N, s)) {
      return r2;
    }
  }
  if (c.red_w) {
    const long groups = (c.w_nw + c.w_nb + 3) / 4;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)lgm_cdiv(groups, 64)), dim3(256), 0, s, c.w_ws, c.w_slab, c.w_gw,
                       c.w_nw, c.w_gb, c.w_nb, c.w_splits, c.w_beta);
    LGM_LAUNCH_CHECK();
  }
  if (!post_done) {        // the input gradient took a path without an epilogue hook: one elementwise pass
    const long M = (long)g->B * g->H * g->W;
    hipLaunchKernelGGL(post_kernel, dim3((unsigned)lgm_cdiv(M * g->Cw, 256)), dim3(256), 0, s, gx, (long)gx_pitch, M, g->Cw,
                       post->act, post->slope, post->mask, (long)post->mask_pitch, post->mask_slope);
    LGM_LAUNCH_CHECK();
  }
  return LGM_OK;
}

extern "C" int lgm_conv_bwd_pair(const LgmConvGeom* g, const float* gy, int64_t gy_pitch, const float* x,
                                 int64_t x_pitch, const float* w, const float* w_t, const float* res, int64_t res_pitch,
                                 float* gx, int64_t gx_pitch, void* dgrad_ws, int64_t dgrad_ws_bytes, float* gw,
                                 float* gbias, float beta, void* wgrad_ws, int64_t wgrad_ws_bytes, int64_t* desc,
                                 void* stream) {
  return conv_bwd_pair_impl(g, gy, gy_pitch, x, x_pitch, w, w_t, res, res_pitch, gx, gx_pitch, dgrad_ws, dgrad_ws_bytes, gw,
                            gbias, beta, wgrad_ws, wgrad_ws_bytes, desc, nullptr, stream);
}

// The same with a post-op on the INPUT gradient (lgm_conv_yx_post's: the backward mask of an activation that sat in
// front of this layer's input): gx = post(conv_yx(gy) + res).
extern "C" int lgm_conv_bwd_pair_post(const LgmConvGeom* g, const float* gy, int64_t gy_pitch, const float* x,
                                      int64_t x_pitch, const float* w, const float* w_t, const float* res,
                                      int64_t res_pitch, float* gx, int64_t gx_pitch, void* dgrad_ws,
                                      int64_t dgrad_ws_bytes, float* gw, float* gbias, float beta, void* wgrad_ws,
                                      int64_t wgrad_ws_bytes, int64_t* desc, const LgmPostOp* post, void* stream) {
  if (int rc = post_check(post, "conv_bwd_pair_post")) return rc;
  if (int rc = check_geom(g)) return rc;
  return conv_bwd_pair_impl(g, gy, gy_pitch, x, x_pitch, w, w_t, res, res_pitch, gx, gx_pitch, dgrad_ws, dgrad_ws_bytes, gw,
                            gbias, beta, wgrad_ws, wgrad_ws_bytes, desc, post, stream);
}

extern "C" int lgm_wgrad_queue_enable(int on) {
  const int was = t_wq.on ? 1 : 0;
  if (on < 0) t_wq.n = 0;      // discard: entries an abandoned pass left behind must never be launched (their operands are gone)
  t_wq.on = on > 0;
  return was;
}

extern "C" int lgm_wgrad_queue_flush(void) { return wq_launch(); }

extern "C" int lgm_conv_wgrad(const LgmConvGeom* g, const float* y, int64_t y_pitch, const float* x,
                              int64_t x_pitch, float* gw, float* gbias, float beta, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  return conv_wgrad_impl(g, y, y_pitch, x, x_pitch, gw, gbias, beta, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int lgm_conv_wgrad_deferred(const LgmConvGeom* g, const float* y, int64_t y_pitch, const float* x,
                                       int64_t x_pitch, float* gw, float* gbias, float beta, void* workspace,
                                       int64_t workspace_bytes, int64_t* desc, void* stream) {
  LGM_REQUIRE(desc, "conv_wgrad_deferred: null descriptor");
  return conv_wgrad_impl(g, y, y_pitch, x, x_pitch, gw, gbias, beta, workspace, workspace_bytes, desc, stream);
}

namespace {
// One launch for the slab reductions of many layers.  Table rows (int64 x 9): slab base, slab stride (floats),
// gw, n_w, gbias, n_b, splits, beta (float bits), first block; the owning row is found by binary search.
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const long long* __restrict__ table, int n_entries) {
  __shared__ __align__(16) float sh[4][64 * 4];
  int lo = 0, hi = n_entries - 1;
  const long long bid = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid * 9 + 8] <= bid) lo = mid; else hi = mid - 1;
  }
  const long long* row = table + lo * 9;
  const float* ws = reinterpret_cast<const float*>(row[0]);
  const long slab = row[1];
  float* gw = reinterpret_cast<float*>(row[2]);
  const long n_w = row[3];
  float* gb = reinterpret_cast<float*>(row[4]);
  const long n_b = row[5];
  const int splits = (int)row[6];
  union { float f; long long i; } bb;
  bb.i = row[7];
  const float beta = bb.f;
  const int col = threadIdx.x & 63, lane = threadIdx.x >> 6;
  const long i = ((long)(bid - row[8]) * 64 + col) * 4;
  const long n = n_w + n_b;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    f32x4 t0 = s, t1 = s, t2 = s, t3 = s;
    for (int k = lane; k < splits; k += 16) {
      const int k1 = k + 4, k2 = k + 8, k3 = k + 12;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(ws + (long)k * slab + i);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(ws + (long)(k1 < splits ? k1 : k) * slab + i);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(ws + (long)(k2 < splits ? k2 : k) * slab + i);
      const f32x4 v3 = *reinterpret_cast<const f32x4*>(ws + (long)(k3 < splits ? k3 : k) * slab + i);
      t0 += v0;
      t1 += v1 * (k1 < splits ? 1.f : 0.f);
      t2 += v2 * (k2 < splits ? 1.f : 0.f);
      t3 += v3 * (k3 < splits ? 1.f : 0.f);
    }
    s = (t0 + t1) + (t2 + t3);
  }
  *reinterpret_cast<f32x4*>(&sh[lane][col * 4]) = s;
  __syncthreads();
  if (lane == 0 && i < n) {
    f32x4 t = (*reinterpret_cast<const f32x4*>(&sh[0][col * 4]) + *reinterpret_cast<const f32x4*>(&sh[1][col * 4])) +
              (*reinterpret_cast<const f32x4*>(&sh[2][col * 4]) + *reinterpret_cast<const f32x4*>(&sh[3][col * 4]));
    float* dst = i < n_w ? gw + i : gb + (i - n_w);
    if (beta != 0.f) t += *reinterpret_cast<const f32x4*>(dst) * beta;
    *reinterpret_cast<f32x4*>(dst) = t;
  }
}
}  // namespace

extern "C" int lgm_wgrad_reduce_batch(const int64_t* table, int n_entries, int64_t total_blocks, void* stream) {
  LGM_REQUIRE(table && n_entries > 0 && total_blocks > 0, "wgrad_reduce_batch: bad arguments");
  lgm_note_kernel(LGM_KNAME("wgrad_reduce_batch_kernel"));
  hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(table), n_entries);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
