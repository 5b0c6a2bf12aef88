// Specialised 3x3 / stride 1 / pad 1 convolution kernels (98 % of the DDPM UNet's FLOPs,
// SURVEY.md §8a) on v_mfma_f32_32x32x2_f32.
//
// Idea: a workgroup owns a spatial tile of 128 output pixels (TH x TW pixels of NI images) and
// stages the (TH+2) x (TW+2) halo patch of the gathered tensor in LDS ONCE per 64-channel chunk.
// All nine filter taps then read their A fragments straight from that patch at a wave-uniform
// offset ((kh*(TW+2)+kw) positions): no per-chunk gather, no bounds checks and no address
// arithmetic in the MFMA loop, and the activation bytes cross L2 once instead of nine times.
// Only the weight tile (64 x 32 floats per chunk) is streamed global -> registers -> LDS (double
// buffered, one barrier per chunk).
//
//   conv3x3  MODE_XY : y[pix][n] = sum_{tap,c} x[pix + tap - 1][c] * w[n][tap][c]
//            MODE_YX : x[pix][c] = sum_{tap,n} y[pix + 1 - tap][n] * w[n][tap][c]   (input gradient)
//   wgrad3x3         : gw[n][tap][c] = sum_pix y[pix][n] * x[pix + tap - 1][c]      (+ fused bias grad)
//
// Supported: H, W powers of two >= 4 (W <= 32 or W % 32 == 0), gathered channels % 32 == 0,
// output channels % 64 == 0.  Everything else takes the generic implicit-GEMM path.
#include "lgm_common.h"

namespace lgm3x3 {

constexpr int BN = 64;       // output-channel tile
constexpr int BK = 32;       // k per weight chunk
constexpr int LDB = BK + 4;  // weight tile row stride (XY)

enum { MODE_XY = 0, MODE_YX = 1, MODE_YXT = 2 };   // YXT: input gradient reading TRANSPOSED weights [Cw][9][Nw]

struct Args {
  const float* a;     // gathered activations, NHWC, C channels
  const float* w;     // [Nw][9][Cw]
  const float* bias;
  const float* res;
  float* out;
  long a_pitch, res_pitch, out_pitch;
  int B, H, W;
  int C;              // gathered channels (reduction)
  int N;              // output channels
  int Wn;             // inner dim of the weight tensor (Cw)
  int TH, TW, NI, lgTW, lgTT;   // tile: TH x TW pixels of NI images; lgTT = log2(TH*TW)
  int tiles_h, tiles_w, tiles_n, NP;
  int splits;         // split-K over the (channel chunk, tap) sequence; > 1 => partials to ws
  float* ws;          // [splits][B*H*W][N]
  long ws_stride;
};

// out[m][n] = sum_s ws[s][m][n] + bias[n] + res[m][n]   (fixed order => deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, long ws_stride, int splits,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ res, long res_pitch,
                                                            float* __restrict__ out, long out_pitch, long M, int N) {
  const int n4 = N / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * n4) return;
  const long m = i / n4;
  const int n = (int)(i % n4) * 4;
  f32x4 s = *reinterpret_cast<const f32x4*>(ws + m * N + n);
  for (int k = 1; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(ws + (long)k * ws_stride + m * N + n);
  if (bias) s += *reinterpret_cast<const f32x4*>(bias + n);
  if (res) s += *reinterpret_cast<const f32x4*>(res + m * res_pitch + n);
  *reinterpret_cast<f32x4*>(out + m * out_pitch + n) = s;
}

__device__ __forceinline__ int xcd_swizzle(int bid, int nb) {
  // blocks b, b+8, b+16, ... share an XCD (and its L2): give them consecutive logical ids
  return (nb % 8 == 0) ? (bid % 8) * (nb / 8) + bid / 8 : bid;
}

template <int MODE, int CK>
__global__ __launch_bounds__(256) void conv3x3_kernel(const Args p) {
  constexpr int LDP = CK + 4;            // patch row stride (floats)
  constexpr int TPP = CK / 4;            // threads per patch position
  constexpr int PPP = 256 / TPP;         // positions per pass
  constexpr int NJ = (288 + PPP - 1) / PPP;
  constexpr int KS = CK / BK;            // weight chunks per (chunk, tap)
  extern __shared__ __align__(16) float smem[];
  float* Ps = smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;

  const int L = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tn = L % p.tiles_n;
  int ts = L / p.tiles_n;
  const int split = ts % p.splits;
  ts /= p.splits;
  const int twi = ts % p.tiles_w;
  ts /= p.tiles_w;
  const int thi = ts % p.tiles_h;
  const int b0 = (ts / p.tiles_h) * p.NI;
  const int h0 = thi * p.TH, w0 = twi * p.TW;
  const int n0 = tn * BN;
  const int PW = p.TW + 2, PP1 = (p.TH + 2) * PW;

  // ---- patch loader bookkeeping: pixel index (or -1) of every position this thread fills ----
  const int c4 = (tid % TPP) * 4;
  int gpix[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int pos = tid / TPP + PPP * j;
    int g = -1;
    if (pos < p.NP) {
      const int img = pos / PP1, rem = pos - img * PP1;
      const int py = rem / PW, px = rem - py * PW;
      const int ih = h0 + py - 1, iw = w0 + px - 1, b = b0 + img;
      if (b < p.B && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) g = (b * p.H + ih) * p.W + iw;
    }
    gpix[j] = g;
  }
  auto load_patch = [&](int cc) {
    const float* src = p.a + cc * CK + c4;
#pragma unroll
    for (int jj = 0; jj < NJ; jj += 6) {
      f32x4 v[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int j = jj + u;
        if (j < NJ)
          v[u] = (gpix[j] >= 0) ? *reinterpret_cast<const f32x4*>(src + (long)gpix[j] * p.a_pitch)
                                : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int j = jj + u;
        const int pos = tid / TPP + PPP * j;
        if (j < NJ && pos < p.NP) *reinterpret_cast<f32x4*>(Ps + pos * LDP + c4) = v[u];
      }
    }
  };

  // ---- A fragment bases (tile-local pixel -> patch position) ----
  int abase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wm * 64 + i * 32 + lr;
    const int img = r >> p.lgTT, rr = r & ((1 << p.lgTT) - 1);
    const int ph = rr >> p.lgTW, pw = rr & (p.TW - 1);
    abase[i] = ((img * (p.TH + 2) + ph) * PW + pw) * LDP + lh * 4;
  }

  // ---- weight fragments of chunk q = (cc, tap, ks): every lane fetches ITS OWN B operand values
  // straight from global memory (the weight tile is tiny and L1/L2 resident), one chunk ahead.
  // No LDS staging of weights => no barrier in the main loop: the four waves run decoupled.
  constexpr bool KCONTIG = (MODE != MODE_YX);   // weight rows contiguous along the reduction index
  constexpr bool FLIP = (MODE != MODE_XY);      // input gradient: taps are mirrored
  const float* wlane = KCONTIG ? p.w + (long)(n0 + wn * 32 + lr) * (9 * p.C) + lh * 4
                               : p.w + (long)(lh * 4) * 9 * p.Wn + n0 + wn * 32 + lr;
  auto load_b = [&](int q, f32x4 (&fb)[4]) {
    const int ks = q % KS, t2 = q / KS;
    const int tap = t2 % 9, cc = t2 / 9;
    if (KCONTIG) {
      const float* src = wlane + tap * p.C + cc * CK + ks * BK;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) fb[kc] = *reinterpret_cast<const f32x4*>(src + kc * 8);
    } else {
      const float* src = wlane + ((long)(cc * CK + ks * BK) * 9 + tap) * p.Wn;
      const long kstride = (long)9 * p.Wn;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
#pragma unroll
        for (int s = 0; s < 4; ++s) fb[kc][s] = src[(kc * 8 + s) * kstride];
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int ncc = p.C / CK;
  const int nq = ncc * 9 * KS;
  const int cps = (nq + p.splits - 1) / p.splits;
  const int q_begin = split * cps;
  const int q_end = min(nq, q_begin + cps);

  f32x4 cb[4], nb[4];
  if (q_begin < q_end) {
    load_b(q_begin, cb);
    load_patch((q_begin / KS) / 9);
  }
  __syncthreads();

  for (int q = q_begin; q < q_end; ++q) {
    const int ks = q % KS, t2 = q / KS;
    const int tap = t2 % 9, cc = t2 / 9;
    const int kh = tap / 3, kw = tap - kh * 3;
    const bool more = q + 1 < q_end;
    if (more) load_b(q + 1, nb);

    const int tapoff = FLIP ? ((2 - kh) * PW + (2 - kw)) * LDP : (kh * PW + kw) * LDP;
    const float* a0 = Ps + abase[0] + tapoff + ks * BK;
    const float* a1 = Ps + abase[1] + tapoff + ks * BK;
#pragma unroll
    for (int kc = 0; kc < BK / 8; ++kc) {
      const f32x4 fa0 = *reinterpret_cast<const f32x4*>(a0 + kc * 8);
      const f32x4 fa1 = *reinterpret_cast<const f32x4*>(a1 + kc * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[s], cb[kc][s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[s], cb[kc][s], acc[1], 0, 0, 0);
      }
    }
    if (more) {
      if ((ks == KS - 1) && (tap == 8)) {   // next chunk starts a new channel chunk: swap the patch
        __syncthreads();
        load_patch(cc + 1);
        __syncthreads();
      }
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) cb[kc] = nb[kc];
    }
  }

  // ---- epilogue: wave-private LDS transpose (the patch is dead by now), then 16-byte stores ----
  __syncthreads();                                   // every wave has finished reading the patch
  float* Ts = smem + wid * LGM_TS_FLOATS;
  const int nc = n0 + wn * 32 + (lane & 7) * 4;
  float* dst = p.out;
  long dpitch = p.out_pitch;
  const bool partial = p.splits > 1;                 // split-K: plain partial sums, reducer adds bias / res
  if (partial) {
    dst = p.ws + (long)split * p.ws_stride;
    dpitch = p.N;
  }
  const f32x4 bv = (!partial && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    lgm_wave_lds_sync();
    lgm_tile_to_lds(acc[i], Ts, lane);
    lgm_wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rt = wm * 64 + i * 32 + (lane >> 3) + 8 * j;
      const int img = rt >> p.lgTT, rr = rt & ((1 << p.lgTT) - 1);
      const int oh = h0 + (rr >> p.lgTW), ow = w0 + (rr & (p.TW - 1)), b = b0 + img;
      if (b < p.B && oh < p.H && ow < p.W) {
        const long m = (long)((b * p.H + oh) * p.W + ow);
        f32x4 v = lgm_tile_row4(Ts, lane, j) + bv;
        if (!partial && p.res) v += *reinterpret_cast<const f32x4*>(p.res + m * p.res_pitch + nc);
        *reinterpret_cast<f32x4*>(dst + m * dpitch + nc) = v;
      }
    }
  }
}

// =====================================================================================
// wgrad: block = (pixel range, 64 n, 64 c); per spatial tile the Y tile [128 pix][64 n] and the X
// halo patch are staged once, then 9 taps x 64 pixel-pairs of MFMAs (k = pixel) accumulate into
// nine 32x32 accumulators per wave.
// =====================================================================================
struct WArgs {
  const float* y;
  const float* x;
  float* out;        // gw or workspace slabs
  float* bias_out;   // or null
  float beta;
  long slab, y_pitch, x_pitch;
  int B, H, W, Nw, Cw;
  int TH, lgTH, NI, tiles_h, tiles_w, tiles_n, tiles_c, splits, tps, total_ts;
};

template <int TW>
__global__ __launch_bounds__(256) void wgrad3x3_kernel(const WArgs p) {
  constexpr int LDP = 64;
  constexpr int PW = TW + 2;
  constexpr int lgTW = TW == 32 ? 5 : TW == 16 ? 4 : TW == 8 ? 3 : 2;
  extern __shared__ __align__(16) float smem[];
  const int PP1 = (p.TH + 2) * PW;
  const int NP = p.NI * PP1;
  float* As = smem;                // [128][64]
  float* Ps = smem + 128 * 64;     // [NP][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;

  int bid = blockIdx.x;
  const int split = bid % p.splits;
  bid /= p.splits;
  const int tc = bid % p.tiles_c, tn = bid / p.tiles_c;
  const int n0 = tn * 64, c0 = tc * 64;
  const int lgTT = lgTW + p.lgTH;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const bool do_bias = p.bias_out != nullptr && tc == 0;
  float bsum = 0.f;

  const int ts_begin = split * p.tps;
  const int ts_end = min(p.total_ts, ts_begin + p.tps);
  // Register-staged tiles: the global loads of tile t+1 are issued right after tile t has been
  // committed to LDS, so they are in flight during tile t's 576 MFMAs per wave.
  constexpr int NJW = 18;              // patch positions per thread (NP <= 288, 16 positions per pass)
  const int c4 = (tid & 15) * 4;
  f32x4 ry[8], rp[NJW];
  auto load_tile = [&](int ts) {
    int t = ts;
    const int twi = t % p.tiles_w;
    t /= p.tiles_w;
    const int thi = t % p.tiles_h;
    const int b0 = (t / p.tiles_h) * p.NI;
    const int h0 = thi * p.TH, w0 = twi * TW;
#pragma unroll
    for (int j = 0; j < 8; ++j) {      // Y tile: thread -> (row tid/16 + 16 j, float4 column tid%16)
      const int r = (tid >> 4) + 16 * j;
      const int img = r >> lgTT, rr = r & ((1 << lgTT) - 1);
      const int oh = h0 + (rr >> lgTW), ow = w0 + (rr & (TW - 1)), b = b0 + img;
      const bool ok = b < p.B && oh < p.H && ow < p.W;
      ry[j] = ok ? *reinterpret_cast<const f32x4*>(p.y + (long)((b * p.H + oh) * p.W + ow) * p.y_pitch + n0 + c4)
                 : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < NJW; ++u) {    // X halo patch
      const int pos = (tid >> 4) + 16 * u;
      bool ok = pos < NP;
      long off = 0;
      if (ok) {
        const int img = pos / PP1, rem = pos - img * PP1;
        const int py = rem / PW, px = rem - py * PW;
        const int ih = h0 + py - 1, iw = w0 + px - 1, b = b0 + img;
        ok = b < p.B && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        off = (long)((b * p.H + ih) * p.W + iw) * p.x_pitch + c0 + c4;
      }
      rp[u] = ok ? *reinterpret_cast<const f32x4*>(p.x + off) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if (ts_begin < ts_end) load_tile(ts_begin);
  for (int ts = ts_begin; ts < ts_end; ++ts) {
    __syncthreads();   // previous tile fully consumed
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(As + ((tid >> 4) + 16 * j) * 64 + c4) = ry[j];
#pragma unroll
    for (int u = 0; u < NJW; ++u) {
      const int pos = (tid >> 4) + 16 * u;
      if (pos < NP) *reinterpret_cast<f32x4*>(Ps + pos * LDP + c4) = rp[u];
    }
    __syncthreads();
    if (ts + 1 < ts_end) load_tile(ts + 1);
    if (do_bias) {
#pragma unroll 8
      for (int r = 0; r < 32; ++r) bsum += As[((tid >> 6) + 4 * r) * 64 + (tid & 63)];
    }
    // ---- MFMAs: k = pixel.  16-pixel groups; inside a group all patch offsets are constants ----
    const float* ap = As + lh * 64 + wm * 32 + lr;
    const float* bp = Ps + wn * 32 + lr;
    for (int g = 0; g < 8; ++g) {
      const int pix0 = g * 16;
      const int img = pix0 >> lgTT, rr = pix0 & ((1 << lgTT) - 1);
      const int pos0 = (img * (p.TH + 2) + (rr >> lgTW)) * PW + (rr & (TW - 1));
      const float* bg = bp + pos0 * LDP;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        // pixel j = 2 s + lh inside the group -> (row j / TW, col j % TW)
        const float a = ap[(pix0 + 2 * s) * 64];
        constexpr int dummy = 0;
        (void)dummy;
        const int j0 = 2 * s;
        const int rowoff = (j0 / TW) * PW + (j0 % TW);   // lh adds +1 column (TW is even)
        const float* bb = bg + (rowoff + lh) * LDP;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          const float b = bb[((tp / 3) * PW + (tp % 3)) * LDP];
          acc[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tp], 0, 0, 0);
        }
      }
    }
  }

  float* out = p.out + (p.splits > 1 ? (long)split * p.slab : 0L);
  const int c = c0 + wn * 32 + lr;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const long o = ((long)n * 9 + tp) * p.Cw + c;
      float v = acc[tp][r];
      if (p.splits == 1 && p.beta != 0.f) v += p.beta * out[o];
      out[o] = v;
    }
  }
  if (do_bias) {
    __syncthreads();
    As[(tid >> 6) * 64 + (tid & 63)] = bsum;
    __syncthreads();
    if (tid < 64) {
      float v = (As[tid] + As[64 + tid]) + (As[128 + tid] + As[192 + tid]);
      float* bo = p.bias_out + (p.splits > 1 ? (long)split * p.slab : 0L) + n0 + tid;
      if (p.splits == 1 && p.beta != 0.f) v += p.beta * bo[0];
      bo[0] = v;
    }
  }
}

static inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// spatial tiling of 128 output pixels
bool plan_tile(int H, int W, int* TH, int* TW, int* NI) {
  if (!pow2(H) || !pow2(W) || H < 4 || W < 4) {
    // W a multiple of 32 with any H % 4 == 0 also tiles exactly
    if (!(W % 32 == 0 && H % 4 == 0)) return false;
  }
  *TW = W < 32 ? W : 32;
  int th = 128 / *TW;
  if (th > H) th = H;
  if (!pow2(th) || H % th != 0 || W % *TW != 0) return false;
  *TH = th;
  *NI = 128 / (th * *TW);
  if (*NI > 1 && (th != H || *TW != W)) return false;
  return true;
}

}  // namespace lgm3x3

// -------------------------------------------------------------------------------------------
// host side (called from conv_igemm.hip's dispatchers)
// -------------------------------------------------------------------------------------------
bool lgm_conv3x3_supported(const LgmConvGeom* g, int gather_channels, int out_channels) {
  using namespace lgm3x3;
  int TH, TW, NI;
  if (!(g->KH == 3 && g->KW == 3 && g->stride == 1 && g->pad == 1)) return false;
  if (gather_channels % 32 != 0 || out_channels % 64 != 0) return false;
  return plan_tile(g->H, g->W, &TH, &TW, &NI);
}

// split-K factor for a given geometry (1 = none); also the workspace it needs
int lgm_conv3x3_splits(const LgmConvGeom* g, int gather_channels, int out_channels) {
  using namespace lgm3x3;
  int TH, TW, NI;
  if (!plan_tile(g->H, g->W, &TH, &TW, &NI)) return 1;
  const long base = (long)lgm_cdiv(g->B, NI) * (g->H / TH) * (g->W / TW) * (out_channels / BN);
  if (base >= 400) return 1;
  const int ck = gather_channels % 64 == 0 ? 64 : 32;
  const int nq = (gather_channels / ck) * 9 * (ck / BK);
  // Two workgroups fit on a CU (512 resident).  Time ~ rounds(base*s) / s: pick the split that
  // minimises it, with a small penalty per split for the partial-sum traffic.
  long smax = nq / 6;                 // at least 6 chunks (192 k) per split
  if (smax > 8) smax = 8;
  if (smax < 1) smax = 1;
  long s = 1;
  double best = 1e30;
  for (long c = 1; c <= smax; ++c) {
    const double rounds = (double)((base * c + 511) / 512);
    const double cost = rounds / (double)c + 0.02 * (double)(c - 1);
    if (cost < best - 1e-9) {
      best = cost;
      s = c;
    }
  }
  return s < 1 ? 1 : (int)s;
}

int lgm_conv3x3_launch(int mode, const LgmConvGeom* g, const float* a, long a_pitch, const float* w,
                       const float* bias, const float* res, long res_pitch, float* out, long out_pitch,
                       void* workspace, long workspace_bytes, hipStream_t s) {
  using namespace lgm3x3;
  Args p{};
  p.a = a; p.w = w; p.bias = bias; p.res = res; p.out = out;
  p.a_pitch = a_pitch; p.res_pitch = res_pitch; p.out_pitch = out_pitch;
  p.B = g->B; p.H = g->H; p.W = g->W;
  p.C = mode == MODE_XY ? g->Cw : g->Nw;   // reduction (gathered) channels
  p.N = mode == MODE_XY ? g->Nw : g->Cw;   // produced channels
  p.Wn = g->Cw;
  plan_tile(g->H, g->W, &p.TH, &p.TW, &p.NI);
  p.lgTW = ilog2(p.TW);
  p.lgTT = ilog2(p.TH * p.TW);
  p.tiles_h = g->H / p.TH;
  p.tiles_w = g->W / p.TW;
  p.tiles_n = p.N / BN;
  p.NP = p.NI * (p.TH + 2) * (p.TW + 2);
  const int groups = lgm_cdiv(g->B, p.NI);
  const long M = (long)g->B * g->H * g->W;
  p.splits = lgm_conv3x3_splits(g, p.C, p.N);
  if (p.splits > 1) {
    const long need = (long)p.splits * M * p.N * (long)sizeof(float);
    const bool aligned = lgm_aligned16(out) && out_pitch % 4 == 0 && (!res || (lgm_aligned16(res) && res_pitch % 4 == 0)) &&
                         lgm_aligned16(workspace);
    if (!workspace || workspace_bytes < need || !aligned) p.splits = 1;
  }
  p.ws = (float*)workspace;
  p.ws_stride = M * p.N;
  const unsigned nblocks = (unsigned)((long)groups * p.tiles_h * p.tiles_w * p.tiles_n * p.splits);
  const bool ck64 = (p.C % 64 == 0);
  const size_t smem = (size_t)p.NP * (ck64 ? 68 : 36) * sizeof(float);
#define LGM_C3_LAUNCH(M, CKV)                                                                          \
  do {                                                                                                 \
    auto kern = conv3x3_kernel<M, CKV>;                                                                \
    static size_t attr = 0;                                                                            \
    if (smem > attr) {                                                                                 \
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
      attr = smem;                                                                                     \
    }                                                                                                  \
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), smem, s, p);                                    \
  } while (0)
  if (mode == MODE_XY) {
    if (ck64) LGM_C3_LAUNCH(MODE_XY, 64); else LGM_C3_LAUNCH(MODE_XY, 32);
  } else if (mode == MODE_YX) {
    if (ck64) LGM_C3_LAUNCH(MODE_YX, 64); else LGM_C3_LAUNCH(MODE_YX, 32);
  } else {
    if (ck64) LGM_C3_LAUNCH(MODE_YXT, 64); else LGM_C3_LAUNCH(MODE_YXT, 32);
  }
#undef LGM_C3_LAUNCH
  if (p.splits > 1) {
    const long items = M * (p.N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)lgm_cdiv(items, 256)), dim3(256), 0, s,
                       (const float*)p.ws, p.ws_stride, p.splits, bias, res, res_pitch, out, out_pitch, M, p.N);
  }
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

int lgm_splitk_reduce_launch(const float* ws, long ws_stride, int splits, const float* bias, const float* res,
                             long res_pitch, float* out, long out_pitch, long M, int N, hipStream_t s) {
  const long items = M * (N / 4);
  hipLaunchKernelGGL(lgm3x3::splitk_reduce_kernel, dim3((unsigned)lgm_cdiv(items, 256)), dim3(256), 0, s, ws, ws_stride,
                     splits, bias, res, res_pitch, out, out_pitch, M, N);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

bool lgm_wgrad3x3_supported(const LgmConvGeom* g) {
  using namespace lgm3x3;
  int TH, TW, NI;
  if (!(g->KH == 3 && g->KW == 3 && g->stride == 1 && g->pad == 1)) return false;
  if (g->Cw % 64 != 0 || g->Nw % 64 != 0) return false;
  return plan_tile(g->H, g->W, &TH, &TW, &NI);
}

void lgm_wgrad3x3_plan(const LgmConvGeom* g, int* splits, int* tps, int* total_ts) {
  using namespace lgm3x3;
  int TH, TW, NI;
  plan_tile(g->H, g->W, &TH, &TW, &NI);
  const int total = lgm_cdiv(g->B, NI) * (g->H / TH) * (g->W / TW);
  const long tiles = (long)(g->Nw / 64) * (g->Cw / 64);
  // LDS allows ONE resident workgroup per CU: never exceed 256 workgroups (a 257th would run alone)
  long s = 256 / tiles;
  if (s > total) s = total;
  if (s < 1) s = 1;
  const int t = lgm_cdiv(total, s);
  *tps = t;
  *splits = lgm_cdiv(total, t);
  *total_ts = total;
}

int lgm_wgrad3x3_launch(const LgmConvGeom* g, const float* y, long y_pitch, const float* x, long x_pitch,
                        float* out, float* bias_out, float beta, long slab, int splits, int tps, int total_ts,
                        hipStream_t s) {
  using namespace lgm3x3;
  WArgs p{};
  int TW;
  p.y = y; p.x = x; p.out = out; p.bias_out = bias_out; p.beta = beta; p.slab = slab;
  p.y_pitch = y_pitch; p.x_pitch = x_pitch;
  p.B = g->B; p.H = g->H; p.W = g->W; p.Nw = g->Nw; p.Cw = g->Cw;
  plan_tile(g->H, g->W, &p.TH, &TW, &p.NI);
  p.lgTH = ilog2(p.TH);
  p.tiles_h = g->H / p.TH; p.tiles_w = g->W / TW;
  p.tiles_n = g->Nw / 64; p.tiles_c = g->Cw / 64;
  p.splits = splits; p.tps = tps; p.total_ts = total_ts;
  const int NP = p.NI * (p.TH + 2) * (TW + 2);
  const size_t smem = (size_t)(128 * 64 + NP * 64) * sizeof(float);
  const unsigned nblocks = (unsigned)((long)p.tiles_n * p.tiles_c * splits);
#define LGM_W3_LAUNCH(TWV)                                                                             \
  do {                                                                                                 \
    auto kern = wgrad3x3_kernel<TWV>;                                                                  \
    static size_t attr = 0;                                                                            \
    if (smem > attr) {                                                                                 \
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
      attr = smem;                                                                                     \
    }                                                                                                  \
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), smem, s, p);                                    \
  } while (0)
  switch (TW) {
    case 32: LGM_W3_LAUNCH(32); break;
    case 16: LGM_W3_LAUNCH(16); break;
    case 8: LGM_W3_LAUNCH(8); break;
    case 4: LGM_W3_LAUNCH(4); break;
    default: lgm_set_error("wgrad3x3: unsupported tile width %d", TW); return LGM_ERR_UNSUPPORTED;
  }
#undef LGM_W3_LAUNCH
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
