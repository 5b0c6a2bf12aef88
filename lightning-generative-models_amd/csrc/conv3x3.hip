// Specialised 3x3 / stride 1 / pad 1 convolution kernels (98 % of the DDPM UNet's FLOPs,
// SURVEY.md §8a): reference Block.proj ddpm.py:160-171 and everything autograd derives from it.
//
//   conv3x3_kernel<MODE>   exact fp32, v_mfma_f32_32x32x2_f32
//       MODE_XY  : y[pix][n] = sum_{tap,c} x[pix + tap - 1][c] * w[n][tap][c]
//       MODE_YXT : x[pix][c] = sum_{tap,n} y[pix + 1 - tap][n] * wT[c][tap][n]   (input gradient, transposed
//                  weight copy); MODE_YX the same from the untransposed weights
//   conv3x3_b3_kernel      the same two products in split precision (opt-in, see below)
//   wgrad3x3_kernel<TW>    gw[n][tap][c] = sum_pix y[pix][n] * x[pix + tap - 1][c]   (+ fused bias gradient)
//
// A workgroup owns spatial tiles of 128 output pixels (TH x TW pixels of NI images) and stages the
// (TH+2) x (TW+2) halo patch of the gathered tensor in LDS once per 32-channel chunk; all nine taps read
// their A fragments from that patch at a wave-uniform offset, so the MFMA loop has no gather, no bounds
// checks and no address arithmetic, and the activation bytes cross L2 once instead of nine times.  The
// pipelining around that idea is described at each kernel.
//
// Supported: H, W powers of two >= 4 (W <= 32 or W % 32 == 0), gathered channels % 32 == 0, produced
// channels % 64 == 0, batch a multiple of the images per tile, tensors < 2^30 elements.  Everything else
// takes the generic implicit-GEMM path (conv_igemm.hip).
#include <type_traits>
#include "lgm_common.h"

namespace lgm3x3 {
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BN = 64;       // output-channel tile
constexpr int BK = 32;       // k per weight chunk

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
  int units, per;     // work units = tiles x splits x n-tiles; units per (persistent) workgroup
  int splits, pps;    // split-K over whole phases (pps phases per split); > 1 => partials to ws
  float* ws;          // [splits][B*H*W][N]
  long ws_stride;
  const unsigned short* wb;   // split-precision path: bf16 planes [3][w_plane] of the (k-contiguous) weights
  long w_plane;
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

// One persistent workgroup per CU (one wave per SIMD), everything software-pipelined INSIDE the
// wave.  Measured on MI355X: while a wave streams MFMAs back to back, the VALU instructions of a
// co-resident wave on the same SIMD issue at roughly one per MFMA slot, so "let another workgroup
// hide my prologue / epilogue" does not work for MFMA-dense code; two barrier-coupled workgroups
// per CU ran no faster than one.  Hence:
//   * a workgroup walks a contiguous range of UNITS (spatial tile, split, n-tile); a unit is a
//     sequence of PHASES, one per 32-channel chunk of the reduction, each = one halo patch in LDS
//     + 9 MFMA steps (tap, 32 k; 32 MFMAs per wave);
//   * the LDS patch is double buffered.  During phase i the patch of phase i+1 (already in
//     registers) is committed to the other buffer one position per step, and the patch of phase
//     i+2 is fetched from global memory into the freed register - one 16-byte load per step;
//   * weight fragments come straight from global memory (per-lane B operands, L2 resident), two
//     steps ahead in a ring of three register sets, across phase and unit boundaries;
//   * A fragments are read from LDS one k-group (8 MFMAs) ahead;
//   * the step body is branch-free (clamped addresses, selects instead of predicated loads), which
//     keeps the compiler's s_waitcnt placement exact, and cheap: per-position byte offsets and
//     border flags are precomputed, bases are scalar - a clump of ~20 VALU instructions (64-bit
//     multiplies, exec-mask juggling) between two MFMAs cost ~150 idle MFMA cycles per step;
//   * one barrier per phase.  Split-K splits on whole phases.
struct Phase {
  int L, cc, cc_end;          // unit index, channel chunk, end of the unit's chunk range
  int tn, split, twi, thi, bg;   // unit coordinates (n-tile, split, tile column, tile row, image group)
  int n0, b0, h0, w0;
  unsigned border;            // which image borders the tile touches (+ bit 4: "never valid" positions)
  long abase;                 // BYTE offset of patch position (image b0, row h0 - 1, column w0 - 1), channel chunk 0
  bool valid;
};

template <int MODE>
__global__ __launch_bounds__(256, 1) void conv3x3_kernel(const Args p) {
  constexpr int CK = 32;                 // channels per phase
  constexpr int LDP = CK + 4;            // patch row stride (floats)
  constexpr int TPP = CK / 4;            // threads per patch position
  constexpr int PPP = 256 / TPP;         // positions per pass
  constexpr int U = 9;                   // MFMA steps per phase (one per tap)
  constexpr int NJ = 288 / PPP;          // patch positions per thread (a buffer holds all 288 positions)
  static_assert(NJ == U, "one patch position per step");
  constexpr int PBUF = 288 * LDP;
  extern __shared__ __align__(16) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;
  float* Ts = smem + 2 * PBUF + wid * LGM_TS_FLOATS;   // wave-private epilogue scratch

  const int Lb = xcd_swizzle(blockIdx.x, gridDim.x);
  const int L0 = Lb * p.per;
  const int L1 = min(p.units, L0 + p.per);
  if (L0 >= L1) return;

  const int PW = p.TW + 2, PP1 = (p.TH + 2) * PW;
  const int ncc_total = p.C / CK;

  // unit coordinates -> tile origin, border mask, scalar base offset (all wave-uniform: SALU only)
  auto place = [&](Phase& ph) {
    ph.n0 = ph.tn * BN;
    ph.w0 = ph.twi * p.TW;
    ph.h0 = ph.thi * p.TH;
    ph.b0 = ph.bg * p.NI;
    ph.cc = ph.split * p.pps;
    ph.cc_end = min(ncc_total, ph.cc + p.pps);
    ph.border = 16u | (ph.h0 == 0 ? 1u : 0u) | (ph.h0 + p.TH == p.H ? 2u : 0u) | (ph.w0 == 0 ? 4u : 0u) |
                (ph.w0 + p.TW == p.W ? 8u : 0u);
    ph.abase = ((long)((ph.b0 * p.H + ph.h0) * p.W + ph.w0) * p.a_pitch) * 4;   // relative to the shifted descriptor
  };
  auto decode = [&](Phase& ph, int L) {    // once per workgroup (integer divisions)
    ph.L = L;
    ph.tn = L % p.tiles_n;
    int ts = L / p.tiles_n;
    ph.split = ts % p.splits;
    ts /= p.splits;
    ph.twi = ts % p.tiles_w;
    ts /= p.tiles_w;
    ph.thi = ts % p.tiles_h;
    ph.bg = ts / p.tiles_h;
    place(ph);
  };
  auto advance = [&](Phase& ph) {          // past the end: stays on the last phase (harmless re-reads)
    if (!ph.valid) return;
    if (ph.cc + 1 < ph.cc_end) {
      ++ph.cc;
    } else if (ph.L + 1 < L1) {            // next unit: odometer increment, no divisions
      ++ph.L;
      if (++ph.tn == p.tiles_n) {
        ph.tn = 0;
        if (++ph.split == p.splits) {
          ph.split = 0;
          if (++ph.twi == p.tiles_w) {
            ph.twi = 0;
            if (++ph.thi == p.tiles_h) {
              ph.thi = 0;
              ++ph.bg;
            }
          }
        }
      }
      place(ph);
    } else {
      ph.valid = false;
    }
  };

  // ---- patch bookkeeping.  Per owned position j: pdelta[j] = BYTE offset from the patch origin
  // (>= 0), and a 5-bit flag group (bit 0/1: top/bottom halo row, bit 2/3: left/right halo
  // column, bit 4: position does not exist); a position is zero padding iff flags & border != 0.
  // The per-step cost is then 4 simple VALU instructions + one scalar-base load.
  const int c4 = (tid % TPP) * 4;
  unsigned pdelta[NJ];
  unsigned pflagA = 0, pflagB = 0;       // positions 0..5 / 6..8, five bits each
  {
    // walk pos = tid / TPP + PPP * j without divisions: (img, py, px) advance by PPP columns per j
    int px = tid / TPP, py = 0, img = 0;
    while (px >= PW) { px -= PW; ++py; }
    while (py >= p.TH + 2) { py -= p.TH + 2; ++img; }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      unsigned d = 0, f = 16u;
      if (img < p.NI) {
        d = (unsigned)(((img * p.H + py) * p.W + px) * (int)p.a_pitch + c4) * 4u;
        f = (py == 0 ? 1u : 0u) | (py == p.TH + 1 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == p.TW + 1 ? 8u : 0u);
      }
      pdelta[j] = d;
      if (j < 6) pflagA |= f << (5 * j);
      else pflagB |= f << (5 * (j - 6));
      px += PPP;
      while (px >= PW) { px -= PW; ++py; }
      while (py >= p.TH + 2) { py -= p.TH + 2; ++img; }
    }
  }
  // The patch is fetched with raw buffer loads through a descriptor that starts one row and one
  // column BEFORE the tensor (so the halo origin of every tile is a non-negative offset): per-lane
  // 32-bit offset + scalar (tile, channel chunk) offset, and a padding position gets an offset past
  // the descriptor's range, for which the hardware returns zeros -- no select when the value is
  // committed to LDS, no 64-bit address arithmetic.  (The host keeps all offsets below 2^31.)
  const unsigned nrec_a = (unsigned)(((long)p.B * p.H * p.W + p.W + 1) * p.a_pitch * 4);
  __amdgpu_buffer_rsrc_t rsrc_a;
  {
    const unsigned long long ab = reinterpret_cast<unsigned long long>(p.a - (long)(p.W + 1) * p.a_pitch);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ab);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(ab >> 32));
    rsrc_a = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                               __builtin_amdgcn_readfirstlane(nrec_a), 0x00020000);
  }
  u32x4 rp[NJ];
  auto fetch_addr = [&](int j, const Phase& ph, unsigned& off) -> unsigned {
    const unsigned fl = (j < 6 ? pflagA : pflagB) & (ph.border << (5 * (j < 6 ? j : j - 6)));
    off = fl == 0u ? pdelta[j] : nrec_a;
    return 0u;
  };
  auto fetch_issue = [&](int j, const Phase& ph, unsigned off) {
    const unsigned soff = (unsigned)(ph.abase + (long)ph.cc * (CK * 4));   // wave-uniform
    rp[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, off, soff, 0);
  };
  auto fetch = [&](int j, const Phase& ph) -> unsigned {
    unsigned off;
    const unsigned ok = fetch_addr(j, ph, off);
    fetch_issue(j, ph, off);
    return ok;
  };
  auto commit = [&](int j, float* buf, unsigned mask) {
    const int pos = tid / TPP + PPP * j;
    (void)mask;
    *reinterpret_cast<u32x4*>(buf + pos * LDP + c4) = rp[j];
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

  // ---- weight fragments: every lane fetches ITS OWN B operand values from global memory ----
  constexpr bool KCONTIG = (MODE != MODE_YX);   // weight rows contiguous along the reduction index
  constexpr bool FLIP = (MODE != MODE_XY);      // input gradient: taps are mirrored
  // (scalar base + 32-bit per-lane byte offset: no 64-bit VALU address arithmetic per step)
  const unsigned wlane = KCONTIG ? (unsigned)((wn * 32 + lr) * 9 * p.C + lh * 4) * 4u
                                 : (unsigned)(lh * 4 * 9 * p.Wn + wn * 32 + lr) * 4u;
  auto load_b = [&](const Phase& ph, int tap, f32x4 (&fb)[4]) {
    if (KCONTIG) {
      const char* sbase = reinterpret_cast<const char*>(p.w) + ((long)ph.n0 * 9 * p.C + tap * p.C + ph.cc * CK) * 4;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) fb[kc] = *reinterpret_cast<const f32x4*>(sbase + wlane + kc * 32);
    } else {
      const char* sbase = reinterpret_cast<const char*>(p.w) + (((long)ph.cc * CK * 9 + tap) * p.Wn + ph.n0) * 4;
      const long kstride = (long)9 * p.Wn * 4;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          fb[kc][s] = *reinterpret_cast<const float*>(sbase + wlane + (kc * 8 + s) * kstride);
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // ---- prologue: patch of the first phase -> buffer 0 (the one exposed load), patch of the
  // second phase into registers, first weight fragment ----
  Phase cur;
  cur.valid = true;
  decode(cur, L0);
  Phase nx1 = cur;
  advance(nx1);
  Phase nx2 = nx1;
  advance(nx2);
  // Weight fragments run TWO steps ahead in a ring of three register sets.  vmcnt retires in
  // order, so the wait for a weight fragment also waits for every older load - in particular the
  // patch fetch (HBM latency) issued just before it; two steps (~2 us) of distance cover that.
  f32x4 wq[3][4];
  load_b(cur, 0, wq[0]);
  load_b(cur, 1, wq[1]);
  unsigned mrp = 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) mrp |= fetch(j, cur) << j;
#pragma unroll
  for (int j = 0; j < NJ; ++j) commit(j, smem, mrp);
  mrp = 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) mrp |= fetch(j, nx1) << j;
  __syncthreads();

  int mrow[2][4];
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  f32x4 rv[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mrow[i][j] = 0;
      rv[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  int buf = 0;
  for (;;) {
    const float* Pc = smem + buf * PBUF;
    float* Pn = smem + (buf ^ 1) * PBUF;
    auto read_frag = [&](int gidx, f32x4 (&f)[2]) {
      const int tap = gidx / 4, kc = gidx % 4;
      const int kh = tap / 3, kw = tap - kh * 3;
      const int tapoff = FLIP ? ((2 - kh) * PW + (2 - kw)) * LDP : (kh * PW + kw) * LDP;
      f[0] = *reinterpret_cast<const f32x4*>(Pc + abase[0] + tapoff + kc * 8);
      f[1] = *reinterpret_cast<const f32x4*>(Pc + abase[1] + tapoff + kc * 8);
    };
    f32x4 fa[2][2];
    read_frag(0, fa[0]);
    unsigned mnew = 0;
    // Last phase of a unit: the epilogue's bias / residual loads are issued NOW, a whole phase
    // before they are used, so that waiting for them later does not drain the (younger) weight and
    // patch prefetches - vmcnt retires in order.
    const bool last_of_unit = cur.cc + 1 >= cur.cc_end;
    if (last_of_unit) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int rt = wm * 64 + i * 32 + (lane >> 3) + 8 * j;
          const int img = rt >> p.lgTT, rr = rt & ((1 << p.lgTT) - 1);
          const int oh = cur.h0 + (rr >> p.lgTW), ow = cur.w0 + (rr & (p.TW - 1)), b = cur.b0 + img;
          mrow[i][j] = (b * p.H + oh) * p.W + ow;   // tiles divide B (image groups), H and W exactly
        }
      if (p.splits == 1) {
        const int nc = cur.n0 + wn * 32 + (lane & 7) * 4;
        if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + nc);
        if (p.res) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              rv[i][j] = *reinterpret_cast<const f32x4*>(p.res + (long)mrow[i][j] * p.res_pitch + nc);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // The step's bookkeeping is dealt out over its four k-groups so that no more than ~10
      // non-MFMA instructions sit between two groups of 8 MFMAs: a longer clump drains the MFMA
      // pipe (measured ~340 idle cycles per step when everything was issued up front).
      unsigned foff = 0, fok = 0;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const int gidx = u * 4 + kc;
        if (kc == 0) {                       // weights two steps ahead
          if (u + 2 < U) load_b(cur, u + 2, wq[(u + 2) % 3]);
          else load_b(nx1, u + 2 - U, wq[(u + 2) % 3]);
        } else if (kc == 1) {                // position u of the next phase's patch -> other buffer
          commit(u, Pn, mrp);
        } else if (kc == 2) {                // address of position u of the phase after
          fok = fetch_addr(u, nx2, foff);
          mnew |= fok << u;
        } else {                             // ... and its register is refilled
          fetch_issue(u, nx2, foff);
        }
        if (gidx + 1 < 4 * U) read_frag(gidx + 1, fa[(gidx + 1) & 1]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[gidx & 1][0][s], wq[u % 3][kc][s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[gidx & 1][1][s], wq[u % 3][kc][s], acc[1], 0, 0, 0);
        }
        // interleave: one MFMA, then up to two of the group's other instructions (VALU / VMEM / DS)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x096, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    mrp = mnew;
    __syncthreads();   // everyone is done reading Pc and writing Pn

    if (last_of_unit) {
      // ---- epilogue: wave-private LDS transpose, then unconditional 16-byte stores ----
      const int nc = cur.n0 + wn * 32 + (lane & 7) * 4;
      float* dst = p.out;
      long dpitch = p.out_pitch;
      if (p.splits > 1) {                    // split-K: plain partial sums, the reducer adds bias / res
        dst = p.ws + (long)cur.split * p.ws_stride;
        dpitch = p.N;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        lgm_wave_lds_sync();
        lgm_tile_to_lds(acc[i], Ts, lane);
        lgm_wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4*>(dst + (long)mrow[i][j] * dpitch + nc) = lgm_tile_row4(Ts, lane, j) + bv + rv[i][j];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      }
    }
    if (!nx1.valid) break;
    cur = nx1;
    nx1 = nx2;
    advance(nx2);
    buf ^= 1;
  }
}

// =====================================================================================
// Split-precision variant (opt-in, SURVEY.md "bf16x3"): every fp32 operand is split EXACTLY into three
// bf16 pieces x = h + m + l (truncation splits of the 24-bit significand) and the product is evaluated as
// hh' + hm' + mh' + hl' + lh' + mm' on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the dropped terms
// are <= 2^-24 relative: the result carries fp32-level error), at 6 x 32-cycle MFMAs per 16 k instead of
// 8 x 64-cycle fp32 MFMAs.  Same work decomposition, pipelining and epilogue as conv3x3_kernel; the LDS
// patch holds three bf16 planes per position (row = 3 x 64 B + 16 B pad = 208 B, conflict-free
// ds_read_b128), activations are split while being committed to LDS, weights are pre-split per step by
// lgm_split_bf16x3.  FLIP = input gradient (mirrored taps) on the transposed weight copy.
// =====================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split3(const f32x4& v, u32x2& H, u32x2& M, u32x2& L) {
  unsigned hb[4], mb[4], lb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned xb = __float_as_uint(v[i]);
    hb[i] = xb & 0xFFFF0000u;
    const float r1 = v[i] - __uint_as_float(hb[i]);
    mb[i] = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(mb[i]);
    lb[i] = __float_as_uint(r2) & 0xFFFF0000u;
  }
  H[0] = (hb[0] >> 16) | hb[1]; H[1] = (hb[2] >> 16) | hb[3];
  M[0] = (mb[0] >> 16) | mb[1]; M[1] = (mb[2] >> 16) | mb[3];
  L[0] = (lb[0] >> 16) | lb[1]; L[1] = (lb[2] >> 16) | lb[3];
}

template <bool FLIP>
__global__ __launch_bounds__(256, 1) void conv3x3_b3_kernel(const Args p) {
  constexpr int CK = 32;
  constexpr int ROWB = 208;              // bytes per patch position: planes h | m | l (64 B each) + pad
  constexpr int U = 9, NJ = 9, TPP = 8, PPP = 32;
  constexpr int PBUFB = 288 * ROWB;
  extern __shared__ __align__(16) float smem[];
  char* smemb = reinterpret_cast<char*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;
  float* Ts = reinterpret_cast<float*>(smemb + 2 * PBUFB) + wid * LGM_TS_FLOATS;

  const int Lb = xcd_swizzle(blockIdx.x, gridDim.x);
  const int L0 = Lb * p.per;
  const int L1 = min(p.units, L0 + p.per);
  if (L0 >= L1) return;

  const int PW = p.TW + 2;
  const int ncc_total = p.C / CK;

  auto place = [&](Phase& ph) {
    ph.n0 = ph.tn * BN;
    ph.w0 = ph.twi * p.TW;
    ph.h0 = ph.thi * p.TH;
    ph.b0 = ph.bg * p.NI;
    ph.cc = ph.split * p.pps;
    ph.cc_end = min(ncc_total, ph.cc + p.pps);
    ph.border = 16u | (ph.h0 == 0 ? 1u : 0u) | (ph.h0 + p.TH == p.H ? 2u : 0u) | (ph.w0 == 0 ? 4u : 0u) |
                (ph.w0 + p.TW == p.W ? 8u : 0u);
    ph.abase = ((long)((ph.b0 * p.H + ph.h0 - 1) * p.W + ph.w0 - 1) * p.a_pitch) * 4;
  };
  auto decode = [&](Phase& ph, int L) {
    ph.L = L;
    ph.tn = L % p.tiles_n;
    int ts = L / p.tiles_n;
    ph.split = ts % p.splits;
    ts /= p.splits;
    ph.twi = ts % p.tiles_w;
    ts /= p.tiles_w;
    ph.thi = ts % p.tiles_h;
    ph.bg = ts / p.tiles_h;
    place(ph);
  };
  auto advance = [&](Phase& ph) {
    if (!ph.valid) return;
    if (ph.cc + 1 < ph.cc_end) {
      ++ph.cc;
    } else if (ph.L + 1 < L1) {
      ++ph.L;
      if (++ph.tn == p.tiles_n) {
        ph.tn = 0;
        if (++ph.split == p.splits) {
          ph.split = 0;
          if (++ph.twi == p.tiles_w) {
            ph.twi = 0;
            if (++ph.thi == p.tiles_h) {
              ph.thi = 0;
              ++ph.bg;
            }
          }
        }
      }
      place(ph);
    } else {
      ph.valid = false;
    }
  };

  const int c4 = (tid % TPP) * 4;
  unsigned pdelta[NJ];
  unsigned pflagA = 0, pflagB = 0;
  {
    int px = tid / TPP, py = 0, img = 0;
    while (px >= PW) { px -= PW; ++py; }
    while (py >= p.TH + 2) { py -= p.TH + 2; ++img; }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      unsigned d = 0, f = 16u;
      if (img < p.NI) {
        d = (unsigned)(((img * p.H + py) * p.W + px) * (int)p.a_pitch + c4) * 4u;
        f = (py == 0 ? 1u : 0u) | (py == p.TH + 1 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == p.TW + 1 ? 8u : 0u);
      }
      pdelta[j] = d;
      if (j < 6) pflagA |= f << (5 * j);
      else pflagB |= f << (5 * (j - 6));
      px += PPP;
      while (px >= PW) { px -= PW; ++py; }
      while (py >= p.TH + 2) { py -= p.TH + 2; ++img; }
    }
  }
  const unsigned safe_delta = (unsigned)((p.W + 1) * (int)p.a_pitch) * 4u;
  f32x4 rp[NJ];
  auto fetch_addr = [&](int j, const Phase& ph, unsigned& off) -> unsigned {
    const unsigned fl = (j < 6 ? pflagA : pflagB) & (ph.border << (5 * (j < 6 ? j : j - 6)));
    const bool ok = fl == 0u;
    off = ok ? pdelta[j] : safe_delta;
    return ok ? 1u : 0u;
  };
  auto fetch_issue = [&](int j, const Phase& ph, unsigned off) {
    const char* sbase = reinterpret_cast<const char*>(p.a) + ph.abase + (long)ph.cc * (CK * 4);
    rp[j] = *reinterpret_cast<const f32x4*>(sbase + off);
  };
  auto fetch = [&](int j, const Phase& ph) -> unsigned {
    unsigned off;
    const unsigned ok = fetch_addr(j, ph, off);
    fetch_issue(j, ph, off);
    return ok;
  };
  // split the fp32 patch value into its three bf16 planes while committing it
  auto commit = [&](int j, char* buf, unsigned mask) {
    const int pos = tid / TPP + PPP * j;
    const f32x4 v = ((mask >> j) & 1u) ? rp[j] : f32x4{0.f, 0.f, 0.f, 0.f};
    u32x2 H, M, L;
    split3(v, H, M, L);
    char* row = buf + pos * ROWB + c4 * 2;
    *reinterpret_cast<u32x2*>(row) = H;
    *reinterpret_cast<u32x2*>(row + 64) = M;
    *reinterpret_cast<u32x2*>(row + 128) = L;
  };

  int apos[2];                            // patch position of the lane's two pixels (tile rows i = 0, 1)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wm * 64 + i * 32 + lr;
    const int img = r >> p.lgTT, rr = r & ((1 << p.lgTT) - 1);
    const int ph = rr >> p.lgTW, pw = rr & (p.TW - 1);
    apos[i] = (img * (p.TH + 2) + ph) * PW + pw;
  }

  // weight fragments in FRAGMENT-MAJOR order (written by lgm_split_bf16x3): per plane
  // [n/32][tap][k/16][lane][8], lane = n%32 + 32*((k%16)/8) - a wave's B operand of one (tap, 16 k) is
  // one contiguous 1 KB read (the row-major layout made every load touch 32 cache lines and the
  // texture-address path became the limiter)
  auto load_b = [&](const Phase& ph, int tap, bf16x8 (&fb)[2][3]) {
    const long frag = (((long)(ph.n0 / 32 + wn) * 9 + tap) * (p.C / 16) + ph.cc * 2) * 64 + lane;
    const char* sbase = reinterpret_cast<const char*>(p.wb) + frag * 16;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        fb[q][pl] = *reinterpret_cast<const bf16x8*>(sbase + (long)pl * p.w_plane * 2 + q * 1024);
  };

  f32x16 acc[2], accs[2];      // large-term and small-term partial sums per pixel tile
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[i][r] = 0.f;
      accs[i][r] = 0.f;
    }

  Phase cur;
  cur.valid = true;
  decode(cur, L0);
  Phase nx1 = cur;
  advance(nx1);
  Phase nx2 = nx1;
  advance(nx2);
  bf16x8 wq[3][2][3];
  load_b(cur, 0, wq[0]);
  load_b(cur, 1, wq[1]);
  unsigned mrp = 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) mrp |= fetch(j, cur) << j;
#pragma unroll
  for (int j = 0; j < NJ; ++j) commit(j, smemb, mrp);
  mrp = 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) mrp |= fetch(j, nx1) << j;
  __syncthreads();

  int mrow[2][4];
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  f32x4 rv[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mrow[i][j] = 0;
      rv[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  int buf = 0;
  for (;;) {
    const char* Pc = smemb + buf * PBUFB;
    char* Pn = smemb + (buf ^ 1) * PBUFB;
    // group g = (tap, q): fragments of the two pixel tiles, three planes each
    auto read_frag = [&](int gidx, bf16x8 (&f)[2][3]) {
      const int tap = gidx / 2, q = gidx % 2;
      const int kh = tap / 3, kw = tap - kh * 3;
      const int tapoff = FLIP ? (2 - kh) * PW + (2 - kw) : kh * PW + kw;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          f[i][pl] = *reinterpret_cast<const bf16x8*>(Pc + (apos[i] + tapoff) * ROWB + pl * 64 + q * 32 + lh * 16);
    };
    bf16x8 fa[2][2][3];
    read_frag(0, fa[0]);
    unsigned mnew = 0;
    const bool last_of_unit = cur.cc + 1 >= cur.cc_end;
    if (last_of_unit) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int rt = wm * 64 + i * 32 + (lane >> 3) + 8 * j;
          const int img = rt >> p.lgTT, rr = rt & ((1 << p.lgTT) - 1);
          const int oh = cur.h0 + (rr >> p.lgTW), ow = cur.w0 + (rr & (p.TW - 1)), b = cur.b0 + img;
          mrow[i][j] = (b * p.H + oh) * p.W + ow;
        }
      if (p.splits == 1) {
        const int nc = cur.n0 + wn * 32 + (lane & 7) * 4;
        if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + nc);
        if (p.res) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              rv[i][j] = *reinterpret_cast<const f32x4*>(p.res + (long)mrow[i][j] * p.res_pitch + nc);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      unsigned foff = 0, fok = 0;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int gidx = u * 2 + q;
        if (q == 0) {
          if (u + 2 < U) load_b(cur, u + 2, wq[(u + 2) % 3]);
          else load_b(nx1, u + 2 - U, wq[(u + 2) % 3]);
          fok = fetch_addr(u, nx2, foff);
          mnew |= fok << u;
        } else {
          commit(u, Pn, mrp);
          fetch_issue(u, nx2, foff);
        }
        if (gidx + 1 < 2 * U) read_frag(gidx + 1, fa[(gidx + 1) & 1]);
        {
          // four independent accumulator chains (tile 0/1 x small/large terms): consecutive MFMAs never
          // depend on each other (a dependent 8-pass MFMA cannot issue back to back)
          const bf16x8 Bh = wq[u % 3][q][0], Bm = wq[u % 3][q][1], Bl = wq[u % 3][q][2];
          const bf16x8 A0h = fa[gidx & 1][0][0], A0m = fa[gidx & 1][0][1], A0l = fa[gidx & 1][0][2];
          const bf16x8 A1h = fa[gidx & 1][1][0], A1m = fa[gidx & 1][1][1], A1l = fa[gidx & 1][1][2];
          accs[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0l, Bh, accs[0], 0, 0, 0);
          accs[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1l, Bh, accs[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0m, Bh, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1m, Bh, acc[1], 0, 0, 0);
          accs[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0h, Bl, accs[0], 0, 0, 0);
          accs[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1h, Bl, accs[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0h, Bm, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1h, Bm, acc[1], 0, 0, 0);
          accs[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0m, Bm, accs[0], 0, 0, 0);
          accs[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1m, Bm, accs[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0h, Bh, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1h, Bh, acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x296, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    mrp = mnew;
    __syncthreads();

    if (last_of_unit) {
      const int nc = cur.n0 + wn * 32 + (lane & 7) * 4;
      float* dst = p.out;
      long dpitch = p.out_pitch;
      if (p.splits > 1) {
        dst = p.ws + (long)cur.split * p.ws_stride;
        dpitch = p.N;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        lgm_wave_lds_sync();
        acc[i] += accs[i];
        lgm_tile_to_lds(acc[i], Ts, lane);
        lgm_wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4*>(dst + (long)mrow[i][j] * dpitch + nc) = lgm_tile_row4(Ts, lane, j) + bv + rv[i][j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[i][r] = 0.f;
          accs[i][r] = 0.f;
        }
      }
    }
    if (!nx1.valid) break;
    cur = nx1;
    nx1 = nx2;
    advance(nx2);
    buf ^= 1;
  }
}

// fp32 weights [rows][T][K] -> three bf16 planes in fragment-major order (see conv3x3_b3_kernel::load_b).
// Table rows: (offset, rows, T, K, first_chunk); one thread per 8-element chunk; rows % 32 == 0, K % 16 == 0.
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                           const int* __restrict__ table, int n_slots, long total_chunks,
                                                           long plane) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total_chunks) return;
  int lo = 0, hi = n_slots - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((long)table[mid * 5 + 4] <= gid) lo = mid; else hi = mid - 1;
  }
  const int* row = table + lo * 5;
  const long off = row[0];
  const int T = row[2], K = row[3];
  long ch = gid - row[4];                 // chunk inside the slot: (((nt * T + tap) * (K/16) + kq) * 64 + lane)
  const int lane = (int)(ch % 64);
  long t = ch / 64;
  const int kq = (int)(t % (K / 16));
  t /= K / 16;
  const int tap = (int)(t % T), nt = (int)(t / T);
  const int n = nt * 32 + (lane & 31), c = kq * 16 + (lane >> 5) * 8;
  const float* sp = src + off + ((long)n * T + tap) * K + c;
  u32x2 H0, M0, L0, H1, M1, L1;
  split3(*reinterpret_cast<const f32x4*>(sp), H0, M0, L0);
  split3(*reinterpret_cast<const f32x4*>(sp + 4), H1, M1, L1);
  unsigned short* dp = dst + off + ch * 8;
  *reinterpret_cast<u32x4*>(dp) = u32x4{H0[0], H0[1], H1[0], H1[1]};
  *reinterpret_cast<u32x4*>(dp + plane) = u32x4{M0[0], M0[1], M1[0], M1[1]};
  *reinterpret_cast<u32x4*>(dp + 2 * plane) = u32x4{L0[0], L0[1], L1[0], L1[1]};
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
  int gsplit;        // sub-blocks per spatial tile along its eight 16-pixel groups (small problems)
};

template <int TW, int GS>
__global__ __launch_bounds__(256) void wgrad3x3_kernel(const WArgs p) {
  constexpr int NG = 8 / GS;       // 16-pixel groups this block runs per tile
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
  const int slab_id = bid % p.splits;      // slab = (tile split, group split)
  const int gs = slab_id % GS, split = slab_id / GS;
  const int g_lo = gs * NG, g_hi = g_lo + NG;
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
  // committed to LDS, so they are in flight during tile t's 576 MFMAs per wave.  Everything about a
  // thread's 8 Y rows and 18 patch positions that does not depend on the tile is computed once
  // (byte offsets from the tile origin, border flags), so issuing a tile costs ~3 VALU
  // instructions per load instead of two integer divisions.
  constexpr int NJW = 18;              // patch positions per thread (NP <= 288, 16 positions per pass)
  const int c4 = (tid & 15) * 4;
  unsigned ydelta[8], xdelta[NJW];
  unsigned xflag[3] = {0u, 0u, 0u};    // 5 flag bits per position, six positions per word
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = (tid >> 4) + 16 * j;
    const int img = r >> lgTT, rr = r & ((1 << lgTT) - 1);
    ydelta[j] = (unsigned)(((img * p.H + (rr >> lgTW)) * p.W + (rr & (TW - 1))) * (int)p.y_pitch + c4) * 4u;
  }
  // Tiles are fetched with raw buffer loads: 32-bit per-lane offset (the precomputed deltas) plus a
  // scalar tile offset, and a position outside the image is given an offset past the descriptor's
  // range, for which the hardware returns zeros -- no 64-bit address arithmetic, no zero-fill selects.
  // The X descriptor starts one row and one column before the tensor so the halo origin is >= 0.
  const long pixels = (long)p.B * p.H * p.W;
  const unsigned nrec_y = (unsigned)((pixels * p.y_pitch - n0) * 4);
  const unsigned nrec_x = (unsigned)(((pixels + p.W + 1) * p.x_pitch - c0) * 4);
  auto make_rsrc = [](const float* base, unsigned nrec) {   // descriptor pinned to scalar registers
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(nrec), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsrc_y = make_rsrc(p.y + n0, nrec_y);
  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(p.x + c0 - (long)(p.W + 1) * p.x_pitch, nrec_x);
  u32x4 ry[8], rp[NJW];
  unsigned soff_y = 0, soff_x = 0, border = 16u;
  auto tile_base = [&](int ts, bool exists) {
    int t = ts;
    const int twi = t % p.tiles_w;
    t /= p.tiles_w;
    const int thi = t % p.tiles_h;
    const int b0 = (t / p.tiles_h) * p.NI;
    const int h0 = thi * p.TH, w0 = twi * TW;
    border = 16u | (h0 == 0 ? 1u : 0u) | (h0 + p.TH == p.H ? 2u : 0u) | (w0 == 0 ? 4u : 0u) |
             (w0 + TW == p.W ? 8u : 0u);
    const unsigned pix = (unsigned)((b0 * p.H + h0) * p.W + w0);
    // a tile past this block's range: offsets beyond both descriptors, every load returns zeros
    soff_y = exists ? pix * (unsigned)p.y_pitch * 4u : 0x80000000u;
    soff_x = exists ? pix * (unsigned)p.x_pitch * 4u : 0x80000000u;
  };
  auto load_one = [&](int idx) {   // idx is a constant after unrolling: 0..7 Y rows, 8..25 patch positions
    if (idx < 8) {
      ry[idx] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, ydelta[idx], soff_y, 0);
    } else {
      const int u = idx - 8;
      const bool ok = (xflag[u / 6] & (border << (5 * (u % 6)))) == 0u;
      rp[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, ok ? xdelta[u] : nrec_x, soff_x, 0);
    }
  };
  // first tile: the Y loads go out before the patch offsets are worked out
  if (ts_begin < ts_end) {
    tile_base(ts_begin, true);
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) load_one(idx);
  }
  {
    // pos = tid / 16 + 16 u  ->  (image, patch row, patch column); positions are < 512, so the
    // quotient by the runtime patch size is exact in fp32 with a half-unit bias
    const float inv_pp1 = 1.0f / (float)PP1;
#pragma unroll
    for (int u = 0; u < NJW; ++u) {
      const int pos = (tid >> 4) + 16 * u;
      const int img = (int)(((float)pos + 0.5f) * inv_pp1);
      const int rem = pos - img * PP1;
      const int py = rem / PW, px = rem - py * PW;   // PW is a compile-time constant
      unsigned d = 0, f = 16u;
      if (img < p.NI) {
        d = (unsigned)(((img * p.H + py) * p.W + px) * (int)p.x_pitch + c4) * 4u;
        f = (py == 0 ? 1u : 0u) | (py == p.TH + 1 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == TW + 1 ? 8u : 0u);
      }
      xdelta[u] = d;
      xflag[u / 6] |= f << (5 * (u % 6));
    }
  }
  if (ts_begin < ts_end) {
#pragma unroll
    for (int idx = 8; idx < 8 + NJW; ++idx) load_one(idx);
  }
  for (int ts = ts_begin; ts < ts_end; ++ts) {
    __syncthreads();   // previous tile fully consumed
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(As + ((tid >> 4) + 16 * j) * 64 + c4) = ry[j];
#pragma unroll
    for (int u = 0; u < NJW; ++u) {
      const int pos = (tid >> 4) + 16 * u;
      if (pos < 288)
        *reinterpret_cast<u32x4*>(Ps + pos * LDP + c4) = rp[u];
    }
    __syncthreads();
    // The next tile's 26 loads ride inside the MFMA steps (a burst here would be TA-rate bound,
    // ~2k cycles per tile); after the last tile they all fall out of range and fetch nothing.
    tile_base(ts + 1 < ts_end ? ts + 1 : ts, ts + 1 < ts_end);
    if (do_bias) {   // column sums of this block's Y rows: branch-free, four reads in flight
      const float* col = As + (g_lo * 16 + (tid >> 6)) * 64 + (tid & 63);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int r = 0; r < g_hi - g_lo; ++r) {
        s0 += col[(16 * r) * 64];
        s1 += col[(16 * r + 4) * 64];
        s2 += col[(16 * r + 8) * 64];
        s3 += col[(16 * r + 12) * 64];
      }
      bsum += (s0 + s1) + (s2 + s3);
    }
    // ---- MFMAs: k = pixel.  16-pixel groups; inside a group all patch offsets are constants ----
    const float* ap = As + lh * 64 + wm * 32 + lr;
    const float* bp = Ps + wn * 32 + lr;
    // Operands one step ahead: step (g, s) issues the ten LDS reads of step (g, s + 1) -- or of
    // (g + 1, 0) -- before its own nine MFMAs, so no LDS latency is exposed inside a tile.
    auto group_base = [&](int g) {
      const int pix0 = g * 16;
      const int img = pix0 >> lgTT, rr = pix0 & ((1 << lgTT) - 1);
      const int pos0 = (img * (p.TH + 2) + (rr >> lgTW)) * PW + (rr & (TW - 1));
      return pos0;
    };
    float a_cur, b_cur[9], a_nxt, b_nxt[9];
    auto read_step = [&](const float* ag, const float* bg, int s, float& a, float (&b)[9]) {
      const int j0 = 2 * s;
      const int rowoff = (j0 / TW) * PW + (j0 % TW);   // lh adds +1 column (TW is even)
      a = ag[j0 * 64];
      const float* bb = bg + (rowoff + lh) * LDP;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) b[tp] = bb[((tp / 3) * PW + (tp % 3)) * LDP];
    };
    const float* ag = ap + g_lo * 16 * 64;
    const float* bg = bp + group_base(g_lo) * LDP;
    read_step(ag, bg, 0, a_cur, b_cur);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int gn = g_lo + (gi + 1 < NG ? gi + 1 : gi);
      const float* ag_n = ap + gn * 16 * 64;
      const float* bg_n = bp + group_base(gn) * LDP;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        // loads of this step: 26 spread over the block's 8 NG steps
        const int q = gi * 8 + s;
        const int first = GS == 1 ? q / 2 : q * (GS / 2);
        const int cnt = GS == 1 ? ((q & 1) == 0 ? 1 : 0) : GS / 2;
        int issued = 0;
        if (s < 7) read_step(ag, bg, s + 1, a_nxt, b_nxt);
        else read_step(ag_n, bg_n, 0, a_nxt, b_nxt);
#pragma unroll
        for (int i = 0; i < cnt; ++i)
          if (first + i < 8 + NJW) {
            load_one(first + i);
            ++issued;
          }
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) acc[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[tp], acc[tp], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);   // next step's LDS reads first ...
        if (issued == 0) {
          __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);  // ... then this step's MFMAs
        } else if (issued == 1) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 7, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
        a_cur = a_nxt;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) b_cur[tp] = b_nxt[tp];
      }
      ag = ag_n;
      bg = bg_n;
      __builtin_amdgcn_sched_barrier(0);   // keep the scheduler's read-ahead inside one group
    }
  }

  // ---- epilogue: nine 32x32 tiles per wave through a wave-private LDS transpose, 16-byte stores
  float* out = p.out + (p.splits > 1 ? (long)slab_id * p.slab : 0L);
  __syncthreads();                                   // every wave is done with As / Ps
  float* Ts = smem + wid * LGM_TS_FLOATS;
  const int cc = c0 + wn * 32 + (lane & 7) * 4;
  const bool acc_out = p.splits == 1 && p.beta != 0.f;
  // Two copies of the loop: with the (rare) read-modify-write of gw inside it, every tap would wait
  // for its loads and with them -- the vector-memory queue retires in order -- for the previous
  // tap's stores.
  if (acc_out) {
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      f32x4 prev[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wm * 32 + (lane >> 3) + 8 * j;
        prev[j] = *reinterpret_cast<const f32x4*>(out + ((long)n * 9 + tp) * p.Cw + cc);
      }
      lgm_wave_lds_sync();
      lgm_tile_to_lds(acc[tp], Ts, lane);
      lgm_wave_lds_sync();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wm * 32 + (lane >> 3) + 8 * j;
        *reinterpret_cast<f32x4*>(out + ((long)n * 9 + tp) * p.Cw + cc) = lgm_tile_row4(Ts, lane, j) + p.beta * prev[j];
      }
    }
  } else {
    // Straight from the accumulator layout: register r of a 32x32 tile is row (r & 3) + 8 (r >> 2) + 4 lh
    // (n), column lr (c), so one dword store writes two full 128-byte segments of the slab -- no LDS
    // transpose, no dependent write -> read -> store chain per tap; lane offset + scalar (row, tap)
    // offset through a buffer descriptor over this block's slab.
    __amdgpu_buffer_rsrc_t rsrc_o;
    {
      const unsigned long long ob = reinterpret_cast<unsigned long long>(out);
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ob);
      const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(ob >> 32));
      rsrc_o = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane((unsigned)(p.Nw * 9 * p.Cw) * 4u),
                                                 0x00020000);
    }
    const unsigned row_bytes = (unsigned)(9 * p.Cw) * 4u;
    const unsigned vo = (unsigned)(n0 + wm * 32 + 4 * lh) * row_bytes + (unsigned)(c0 + wn * 32 + lr) * 4u;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[tp][r];
        asm volatile("" : "+v"(v));   // (a direct bit_cast of the accumulator element stored element 0 sixteen times)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_o, vo,
                                              (unsigned)((r & 3) + 8 * (r >> 2)) * row_bytes + (unsigned)(tp * p.Cw) * 4u,
                                              0);
      }
  }
  if (do_bias) {
    __syncthreads();
    As[(tid >> 6) * 64 + (tid & 63)] = bsum;
    __syncthreads();
    if (tid < 64) {
      float v = (As[tid] + As[64 + tid]) + (As[128 + tid] + As[192 + tid]);
      float* bo = p.bias_out + (p.splits > 1 ? (long)slab_id * p.slab : 0L) + n0 + tid;
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
  if (!plan_tile(g->H, g->W, &TH, &TW, &NI)) return false;
  return g->B % NI == 0;   // whole image groups only (the launcher also needs 32-bit element offsets)
}

// split-K factor for a given geometry (1 = none); also the workspace it needs
int lgm_conv3x3_splits(const LgmConvGeom* g, int gather_channels, int out_channels) {
  using namespace lgm3x3;
  int TH, TW, NI;
  if (!plan_tile(g->H, g->W, &TH, &TW, &NI)) return 1;
  const long base = (long)lgm_cdiv(g->B, NI) * (g->H / TH) * (g->W / TW) * (out_channels / BN);
  if (base >= 1024) return 1;
  const int phases = gather_channels / 32;   // split on whole phases (288 k each)
  // One persistent workgroup per CU (256 resident).  Time ~ rounds(base*s) * phases per split: pick
  // the split that minimises it, with a small penalty per split for the partial-sum traffic.
  long smax = phases < 8 ? phases : 8;
  long s = 1;
  double best = 1e30;
  for (long c = 1; c <= smax; ++c) {
    const long pps = (phases + c - 1) / c;
    if ((phases + pps - 1) / pps != c) continue;          // would leave an empty split
    const double rounds = (double)((base * c + lgm_cu_budget() - 1) / lgm_cu_budget());
    const double cost = rounds * (double)pps / (double)phases + 0.02 * (double)(c - 1);
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
  p.pps = lgm_cdiv(p.C / 32, p.splits);
  p.splits = lgm_cdiv(p.C / 32, p.pps);
  p.units = (int)((long)groups * p.tiles_h * p.tiles_w * p.tiles_n * p.splits);
  p.per = lgm_cdiv(p.units, lgm_cu_budget());     // one persistent workgroup per CU walks its unit range
  const unsigned nblocks = (unsigned)lgm_cdiv(p.units, p.per);
  const size_t smem = ((size_t)2 * 288 * 36 + 4 * LGM_TS_FLOATS) * sizeof(float);
#define LGM_C3_LAUNCH(M)                                                                               \
  do {                                                                                                 \
    auto kern = conv3x3_kernel<M>;                                                                     \
    static bool attr = false;                                                                          \
    if (!attr) {                                                                                       \
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
      attr = true;                                                                                     \
    }                                                                                                  \
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), smem, s, p);                                    \
  } while (0)
  lgm_note_kernel(mode == MODE_XY ? LGM_KNAME("lgm3x3::conv3x3_kernel<0>") : mode == MODE_YX ? LGM_KNAME("lgm3x3::conv3x3_kernel<1>") : LGM_KNAME("lgm3x3::conv3x3_kernel<2>"));
  if (mode == MODE_XY) LGM_C3_LAUNCH(MODE_XY);
  else if (mode == MODE_YX) LGM_C3_LAUNCH(MODE_YX);
  else LGM_C3_LAUNCH(MODE_YXT);
#undef LGM_C3_LAUNCH
  if (p.splits > 1) {
    const long items = M * (p.N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)lgm_cdiv(items, 256)), dim3(256), 0, s,
                       (const float*)p.ws, p.ws_stride, p.splits, bias, res, res_pitch, out, out_pitch, M, p.N);
  }
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

// ---- split-precision (bf16x3) path: same planning as the fp32 launcher, other kernel and LDS footprint
extern "C" int64_t lgm_conv3x3_bf16x3_supported(const LgmConvGeom* g, int mode, int64_t a_pitch) {
  if (!g) return 0;
  const int gc = mode == 0 ? g->Cw : g->Nw, oc = mode == 0 ? g->Nw : g->Cw;
  return lgm_conv3x3_supported(g, gc, oc) && (long)g->B * g->H * g->W * a_pitch < (1L << 30) ? 1 : 0;
}

extern "C" int lgm_split_bf16x3(const float* src, uint16_t* dst, const int32_t* table, int n_slots, int64_t total_chunks,
                                int64_t plane_elems, void* stream) {
  LGM_REQUIRE(src && dst && table && n_slots > 0 && total_chunks > 0 && plane_elems % 8 == 0 && lgm_aligned16(src) &&
                  lgm_aligned16(dst),
              "split_bf16x3: bad arguments");
  hipLaunchKernelGGL(lgm3x3::split_bf16x3_kernel, dim3((unsigned)lgm_cdiv(total_chunks, 256)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, table, n_slots, (long)total_chunks, (long)plane_elems);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_conv3x3_bf16x3(int mode, const LgmConvGeom* g, const float* a, int64_t a_pitch,
                                  const uint16_t* w_planes, int64_t plane_elems, const float* bias, const float* res,
                                  int64_t res_pitch, float* out, int64_t out_pitch, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
  using namespace lgm3x3;
  LGM_REQUIRE(g && a && w_planes && out && (mode == 0 || mode == 1), "conv3x3_bf16x3: bad arguments");
  LGM_REQUIRE(lgm_conv3x3_bf16x3_supported(g, mode, a_pitch), "conv3x3_bf16x3: unsupported geometry");
  LGM_REQUIRE(lgm_aligned16(a) && lgm_aligned16(out) && a_pitch % 4 == 0 && out_pitch % 4 == 0 &&
                  (((uintptr_t)w_planes) & 15u) == 0 && plane_elems % 8 == 0 &&
                  (!res || (lgm_aligned16(res) && res_pitch % 4 == 0)) && (!bias || lgm_aligned16(bias)),
              "conv3x3_bf16x3: 16-byte aligned tensors with pitch %% 4 == 0 required");
  hipStream_t s = (hipStream_t)stream;
  Args p{};
  p.a = a; p.w = nullptr; p.wb = w_planes; p.w_plane = plane_elems; p.bias = bias; p.res = res; p.out = out;
  p.a_pitch = a_pitch; p.res_pitch = res_pitch; p.out_pitch = out_pitch;
  p.B = g->B; p.H = g->H; p.W = g->W;
  p.C = mode == 0 ? g->Cw : g->Nw;
  p.N = mode == 0 ? g->Nw : g->Cw;
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
    if (!workspace || workspace_bytes < need || !lgm_aligned16(workspace)) p.splits = 1;
  }
  p.ws = (float*)workspace;
  p.ws_stride = M * p.N;
  p.pps = lgm_cdiv(p.C / 32, p.splits);
  p.splits = lgm_cdiv(p.C / 32, p.pps);
  p.units = (int)((long)groups * p.tiles_h * p.tiles_w * p.tiles_n * p.splits);
  p.per = lgm_cdiv(p.units, lgm_cu_budget());
  const unsigned nblocks = (unsigned)lgm_cdiv(p.units, p.per);
  const size_t smem = (size_t)2 * 288 * 208 + 4 * LGM_TS_FLOATS * sizeof(float);
  if (mode == 0) {
    static bool attr = false;
    if (!attr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_b3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)smem);
      attr = true;
    }
    hipLaunchKernelGGL(conv3x3_b3_kernel<false>, dim3(nblocks), dim3(256), smem, s, p);
  } else {
    static bool attr = false;
    if (!attr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_b3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)smem);
      attr = true;
    }
    hipLaunchKernelGGL(conv3x3_b3_kernel<true>, dim3(nblocks), dim3(256), smem, s, p);
  }
  if (p.splits > 1) {
    const long items = M * (p.N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)lgm_cdiv(items, 256)), dim3(256), 0, s, (const float*)p.ws,
                       p.ws_stride, p.splits, bias, res, res_pitch, out, out_pitch, M, p.N);
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
  if (!plan_tile(g->H, g->W, &TH, &TW, &NI)) return false;
  return g->B % NI == 0;   // whole image groups only
}

void lgm_wgrad3x3_plan(const LgmConvGeom* g, int* splits, int* tps, int* total_ts) {
  using namespace lgm3x3;
  int TH, TW, NI;
  plan_tile(g->H, g->W, &TH, &TW, &NI);
  const int total = lgm_cdiv(g->B, NI) * (g->H / TH) * (g->W / TW);
  const long tiles = (long)(g->Nw / 64) * (g->Cw / 64);
  // LDS allows ONE resident workgroup per CU: never exceed 256 workgroups (a 257th would run alone)
  long s = lgm_cu_budget() / tiles;
  if (s > total) s = total;
  if (s < 1) s = 1;
  const int t = lgm_cdiv(total, s);
  *tps = t;
  const int tsplits = lgm_cdiv(total, t);
  // few workgroups (small batch / small maps): also split each tile's eight 16-pixel groups, so
  // that a workgroup's MFMA chain (576 per tile and wave) is shorter and more CUs take part
  const long wgs = tiles * tsplits;
  static const bool no_gsplit = getenv("LGM_NO_GSPLIT") != nullptr;   // A/B switch
  const int gsplit = no_gsplit ? 1 : wgs <= 64 ? 4 : wgs <= 128 ? 2 : 1;
  *splits = tsplits * gsplit;
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
  p.gsplit = splits / lgm_cdiv(total_ts, tps);
  const int NP = p.NI * (p.TH + 2) * (TW + 2);
  (void)NP;
  const size_t smem = (size_t)(128 * 64 + 288 * 64) * sizeof(float);   // all 288 patch positions are written
  const unsigned nblocks = (unsigned)((long)p.tiles_n * p.tiles_c * splits);
#define LGM_W3_LAUNCH1(TWV, GSV)                                                                       \
  do {                                                                                                 \
    auto kern = wgrad3x3_kernel<TWV, GSV>;                                                             \
    static size_t attr = 0;                                                                            \
    if (smem > attr) {                                                                                 \
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
      attr = smem;                                                                                     \
    }                                                                                                  \
    lgm_note_kernel(LGM_KNAME("lgm3x3::wgrad3x3_kernel<" #TWV ", " #GSV ">"));                                    \
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), smem, s, p);                                    \
  } while (0)
#define LGM_W3_LAUNCH(TWV)                                                                             \
  do {                                                                                                 \
    if (p.gsplit == 4) LGM_W3_LAUNCH1(TWV, 4);                                                         \
    else if (p.gsplit == 2) LGM_W3_LAUNCH1(TWV, 2);                                                    \
    else LGM_W3_LAUNCH1(TWV, 1);                                                                       \
  } while (0)
  if (p.gsplit != 1 && p.gsplit != 2 && p.gsplit != 4) {
    lgm_set_error("wgrad3x3: unsupported group split %d", p.gsplit);
    return LGM_ERR_UNSUPPORTED;
  }
  switch (TW) {
    case 32: LGM_W3_LAUNCH(32); break;
    case 16: LGM_W3_LAUNCH(16); break;
    case 8: LGM_W3_LAUNCH(8); break;
    case 4: LGM_W3_LAUNCH(4); break;
    default: lgm_set_error("wgrad3x3: unsupported tile width %d", TW); return LGM_ERR_UNSUPPORTED;
  }
#undef LGM_W3_LAUNCH
#undef LGM_W3_LAUNCH1
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
