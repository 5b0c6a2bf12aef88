// Winograd F(4x4, 3x3) WEIGHT GRADIENT in fp32 on v_mfma_f32_32x32x2_f32 — the 3x3 / stride 1 / pad 1 layers of the DDPM
// UNet on the large maps (W % 16 == 0, H % 4 == 0; reference: autograd's weight / bias gradient of Block.proj,
// ddpm.py:157-173).  Transposes the forward algorithm of winograd4.hip:
//
//   dU[xi][n][c] = sum over tiles t of  Yt[xi][t][n] * Xt[xi][t][c],     Yt = A dY A^T (4x4 -> 6x6),  Xt = B^T x B (6x6),
//   dw = G^T dU G  (6x6 -> 3x3),        bias gradient = sum of dY
//
// 36 products per 16 output pixels instead of 64 (the F(2x2) weight gradient of winograd.hip) or 144 (direct).
//
// Work decomposition.  WORKGROUP = 64 output channels x 32 input channels x a range of tile GROUPS (a group = four tiles
// in a row = 4 x 16 output pixels; one group per PHASE); it writes one slab [Nw][9][Cw] region (+ bias sums) for the
// batched fixed-order slab reducer, exactly as the other weight-gradient kernels do.  512 threads = 8 waves:
//   * MFMA role (all waves): wave = (9 of the 36 xi) x (32 of the 64 output channels): per xi two 8-byte LDS reads
//     (A = Yt: output channels x tiles, B = Xt: tiles x input channels) and two MFMAs into one 32x32 accumulator;
//   * waves 0-3 also build Yt: thread = (tile, output channel): 16 dword loads (64 consecutive channels per wave:
//     coalesced), A . A^T in registers (80 operations), 36 values to LDS;
//   * waves 4-7 also build Xt: thread = (tile, input channel, half of the six rows): 30 dword loads with the zero padding
//     as out-of-range buffer offsets, B^T . B (54 operations), 18 values to LDS.
// Operands double-buffered (2 x 55 KB), one barrier per phase; the raw values of group p + 2 are requested as soon as the
// registers of group p + 1 have been transformed.  Epilogue: accumulators through LDS in two rounds (the 36 xi of an
// (n, c) sit in four waves), thread = (input channel, four output channels, tap rows) applies G^T . G and stores 128-byte
// segments of the slab.  Fixed summation orders: run-to-run identical.
#include <stdlib.h>

#include <type_traits>

#include "lgm_common.h"

// W4W_EXP (attribution builds only, -DW4W_EXP=n into a separate library, LGM_LIB=<path>): bit 0 drops the Yt transform,
// bit 1 the Xt transform, bit 2 the raw loads, bit 3 the operand reads of the MFMA steps -- wrong results, timing valid.
#ifndef W4W_EXP
#define W4W_EXP 0
#endif

namespace lgmwino4w {
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NXI = 36;
constexpr int YB = NXI * 256;      // floats of a Yt buffer: [xi][tile pair][n 64][2]
constexpr int XB = NXI * 128;      // floats of an Xt buffer: [xi][tile pair][c 32][2]
constexpr int OPB = YB + XB;       // one operand buffer
constexpr int MH = NXI * 512;      // floats of the epilogue exchange per half: [xi][n group 4][c 32][4 n]

struct WArgs {
  const float* y;      // output gradient, NHWC
  const float* x;      // layer input, NHWC
  float* out;          // slabs: split k at out + k * slab
  long y_pitch, x_pitch, slab;
  int bias;            // 1: sums of y into slab[n_w + n]
  int B, H, W, Nw, Cw;
  int tiles_c;         // Cw / 32
  int splits, gps, total_groups;
  int grow;            // groups per group row (W / 16, or W / 8 for square groups)
  int trows;           // group rows per image (H / 4, or H / 8)
  int xcd_order;       // 1: sharers of a pixel range on one XCD (see the block decode)
  int sq;              // group shape: 0 = four tiles in a row (4 x 16 output pixels), 1 = 2 x 2 tiles (8 x 8: the 8 x 8 maps)
};

__device__ __forceinline__ f32x4 add4(const f32x4 a, const f32x4 b) { return a + b; }
__device__ __forceinline__ f32x4 sub4(const f32x4 a, const f32x4 b) {
  f32x2 lo, hi;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]"
      : "=v"(lo)
      : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(__builtin_shufflevector(b, b, 0, 1)));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]"
      : "=v"(hi)
      : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(__builtin_shufflevector(b, b, 2, 3)));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 fma4(const float c, const f32x4 a, const f32x4 b) {   // c * a + b
  return __builtin_elementwise_fma(f32x4{c, c, c, c}, a, b);
}
__device__ __forceinline__ f32x2 fma2(const float c, const f32x2 a, const f32x2 b) {
  return __builtin_elementwise_fma(f32x2{c, c}, a, b);
}

// The kernel body as a device function of (arguments, logical block id): runs as its own launch or as a block range of a
// grouped launch (several layers' weight gradients in one grid).
__device__ __forceinline__ void wino4_wgrad_body(const WArgs& p, const int bidx) {
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;

  // Block -> (split, input-channel block, output-channel block).  The tiles_c * tiles_n workgroups of a split read the SAME
  // pixels (all of them dY's 64-channel slice, pairs of them the same 32 channels of x): they get consecutive logical ids
  // inside one XCD's share of the range (blocks with equal id & 7 sit on one XCD whatever the range's first block is), so
  // that one of them fetches and the others hit that XCD's L2.  With the split index fastest (the former order) the
  // sharers sat on different XCDs and dY came from memory tiles_c times (404 MB fetched per four-layer launch against 268
  // MB of operands, profiles/r05_pmc_traffic.json).
  int L = bidx;
  const int inner = p.tiles_c * (p.Nw >> 6);
  if (p.xcd_order) {
    const int nblk = inner * p.splits, v = bidx & 7, r = nblk & 7;
    L = v * (nblk >> 3) + (v < r ? v : r) + (bidx >> 3);
  }
  const int split = p.xcd_order ? L / inner : L % p.splits;
  const int bid = p.xcd_order ? L % inner : L / p.splits;
  const int tc = bid % p.tiles_c, tn = bid / p.tiles_c;
  const int n0 = tn * 64, c0 = tc * 32;
  const int g_begin = split * p.gps;
  const int g_end = min(p.total_groups, g_begin + p.gps);
  const int nph = g_end - g_begin;

  // MFMA role
  const int xg = wid & 3, nh = wid >> 2;
  const int ard = xg * (9 * 256) + lh * 128 + (nh * 32 + lr) * 2;             // A fragments (Yt), floats
  const int brd = YB + xg * (9 * 128) + lh * 64 + lr * 2;                     // B fragments (Xt)

  const unsigned nrec_y = (unsigned)((long)p.B * p.H * p.W * p.y_pitch * 4);
  // the X descriptor starts one row and one column BEFORE the tensor (offsets of halo pixels stay non-negative; the W + 1
  // pixels in front of the allocation are never requested: every padding position gets an out-of-range offset)
  const unsigned nrec_x = (unsigned)(((long)p.B * p.H * p.W + p.W + 1) * p.x_pitch * 4);
  auto rsrc = [&](const float* ptr, unsigned nrec) {
    const unsigned long long ab = reinterpret_cast<unsigned long long>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ab);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(ab >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(nrec), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsrc_y = rsrc(p.y, nrec_y);
  const __amdgpu_buffer_rsrc_t rsrc_x = rsrc(p.x - (long)(p.W + 1) * p.x_pitch, nrec_x);

  // group g -> (image, tile row, group of the row).  Decoded ONCE (three integer divisions cost ~60 instructions, and a phase
  // has only 18 MFMAs per wave to hide them under); every later group is the previous one advanced like an odometer.
  struct GPos {
    int gx, ty, b;
  };
  auto gdecode = [&](int g) {
    GPos q;
    q.gx = g % p.grow;
    const int r = g / p.grow;
    q.ty = r % p.trows;
    q.b = r / p.trows;
    return q;
  };
  auto gnext = [&](GPos& q) {
    if (++q.gx == p.grow) {
      q.gx = 0;
      if (++q.ty == p.trows) {
        q.ty = 0;
        ++q.b;
      }
    }
  };
  const int gh = p.sq ? 8 : 4, gw = p.sq ? 8 : 16;                          // output pixels of a group
  auto gpix = [&](const GPos& q) -> long { return ((long)q.b * p.H + gh * q.ty) * p.W + gw * q.gx; };   // the group's first pixel
  auto tile_row = [&](int tl) { return p.sq ? (tl >> 1) : 0; };             // tile position inside the group, in tiles
  auto tile_col = [&](int tl) { return p.sq ? (tl & 1) : tl; };

  f32x16 acc[9];
#pragma unroll
  for (int e = 0; e < 9; ++e)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;

  auto mfma_step = [&](const float* buf, int e, f32x2& af, f32x2& bf) {     // xi e with the fragments read a step ago
    f32x2 an = af, bn = bf;
    if (e < 8 && !(W4W_EXP & 8)) {
      an = *reinterpret_cast<const f32x2*>(buf + ard + (e + 1) * 256);
      bn = *reinterpret_cast<const f32x2*>(buf + brd + (e + 1) * 128);
    }
    acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc[e], 0, 0, 0);
    acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc[e], 0, 0, 0);
    af = an;
    bf = bn;
  };

  float bsum = 0.f;

  // =================================== waves 0-3: Yt = A dY A^T ===================================
  auto body_y = [&]() {
    const int tl = wid;                                  // tile of the group
    const unsigned ylane = (unsigned)((((long)(4 * tile_row(tl)) * p.W + 4 * tile_col(tl)) * p.y_pitch + n0 + lane) * 4);
    const int ywr = (tl >> 1) * 128 + lane * 2 + (tl & 1);
    // two register sets: group ph + 2 is requested at the START of phase ph into the set phase ph - 1 emptied - a whole
    // phase ahead of its use (requested at step 2 of the same set, six steps ahead, the next phase still opened with a wait)
    float dA[4][4], dB[4][4];
    GPos gq = gdecode(g_begin);                          // the group the next load() fetches
    auto load = [&](float (&d)[4][4], int ph) {
      const unsigned ok = ph < nph ? 0u : nrec_y;        // past the range: zeros
      const unsigned ybase = (unsigned)(gpix(gq) * p.y_pitch * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          d[r][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
              rsrc_y, ylane + ok, ybase + (unsigned)(((long)r * p.W + c) * p.y_pitch * 4), 0));
      if (ph + 1 < nph) gnext(gq);                       // (stays on the last group past the range: in-range addresses)
    };
    // A = [1 0 0 0; 1 1 1 1; 1 -1 1 -1; 1 2 4 8; 1 -2 4 -8; 0 0 0 1]
    float T[6][4];
    auto vertical2 = [&](const float (&d)[4][4], int cp) {   // columns 2 cp, 2 cp + 1 at once (packed fp32)
      const f32x2 d0 = {d[0][2 * cp], d[0][2 * cp + 1]}, d1 = {d[1][2 * cp], d[1][2 * cp + 1]};
      const f32x2 d2 = {d[2][2 * cp], d[2][2 * cp + 1]}, d3 = {d[3][2 * cp], d[3][2 * cp + 1]};
      const f32x2 s02 = d0 + d2, s13 = d1 + d3;
      const f32x2 e = fma2(4.f, d2, d0), f = fma2(4.f, d3, d1);
      const f32x2 t1 = s02 + s13, t2 = s02 - s13, t3 = fma2(2.f, f, e), t4 = fma2(-2.f, f, e);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int c = 2 * cp + k;
        T[0][c] = d0[k];
        T[1][c] = t1[k];
        T[2][c] = t2[k];
        T[3][c] = t3[k];
        T[4][c] = t4[k];
        T[5][c] = d3[k];
      }
      bsum += t1[0] + t1[1];                              // (written only when the layer has a bias)
    };
    auto horizontal = [&](float* ybuf, int i) {
      float* v = ybuf + ywr + i * (6 * 256);
      const float t0 = T[i][0], t1 = T[i][1], t2 = T[i][2], t3 = T[i][3];
      const float s02 = t0 + t2, s13 = t1 + t3;
      const float e = __builtin_fmaf(4.f, t2, t0), f = __builtin_fmaf(4.f, t3, t1);
      v[0 * 256] = t0;
      v[1 * 256] = s02 + s13;
      v[2 * 256] = s02 - s13;
      v[3 * 256] = __builtin_fmaf(2.f, f, e);
      v[4 * 256] = __builtin_fmaf(-2.f, f, e);
      v[5 * 256] = t3;
    };
    load(dA, 0);                                         // group 0 -> set A, transformed here
    load(dB, 1);                                         // group 1 -> set B, transformed during phase 0
    vertical2(dA, 0);
    vertical2(dA, 1);
#pragma unroll
    for (int i = 0; i < 6; ++i) horizontal(smem, i);
    __syncthreads();
    auto phase = [&](int ph, auto cur_c) {
      constexpr int cur = decltype(cur_c)::value;        // phase parity: group ph + 1 sits in set B (cur 0) / A (cur 1)
      const float* const ocur = smem + cur * OPB;
      float* const onxt = smem + (cur ^ 1) * OPB;
      f32x2 af = *reinterpret_cast<const f32x2*>(ocur + ard);
      f32x2 bf = *reinterpret_cast<const f32x2*>(ocur + brd);
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        mfma_step(ocur, e, af, bf);
        if (e == 0 && !(W4W_EXP & 4)) {                  // group ph + 2 into the set whose group was transformed last phase
          if (cur == 0) load(dA, ph + 2);
          else load(dB, ph + 2);
        }
        if (e < 2) {
          if (!(W4W_EXP & 1)) {
            if (cur == 0) vertical2(dB, e);
            else vertical2(dA, e);
          }
        } else if (e < 8) {
          if (!(W4W_EXP & 1)) horizontal(onxt, e - 2);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    };
    for (int ph = 0; ph < nph; ph += 2) {
      phase(ph, std::integral_constant<int, 0>{});
      if (ph + 1 < nph) phase(ph + 1, std::integral_constant<int, 1>{});
    }
  };

  // =================================== waves 4-7: Xt = B^T x B ===================================
  auto body_x = [&](auto half_c) {
    constexpr int HALF = decltype(half_c)::value;        // rows 0-2 / 3-5 of B^T x
    const int tl = 2 * ((wid >> 1) & 1) + lh;            // tile of the group
    const int c = lr;
    const int trow = tile_row(tl), tcol = tile_col(tl);  // (trow is wave-uniform: tl >> 1 comes from the wave id)
    const int tlast = p.sq ? 1 : 3;
    const unsigned xlane = (unsigned)((((long)(4 * trow) * p.W + 4 * tcol) * p.x_pitch + c0 + c) * 4);
    const int xwr = YB + (tl >> 1) * 64 + c * 2 + (tl & 1) + HALF * (18 * 128);
    constexpr int R0 = HALF ? 1 : 0;                     // the five raw rows this half needs: R0 .. R0 + 4
    float d[5][6];
    GPos gq = gdecode(g_begin);                          // the group the next load_rows() calls fetch
    auto load_rows = [&](int ph, int r_lo, int r_hi) {   // raw rows (index into d) r_lo .. r_hi - 1 of group ph
      const GPos q = gq;
      const unsigned xbase = (unsigned)(gpix(q) * p.x_pitch * 4);
      const bool live = ph < nph;
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        if (r < r_lo || r >= r_hi) continue;
        const int yy = gh * q.ty + 4 * trow - 1 + R0 + r;   // image row
        const bool rok = live && yy >= 0 && yy < p.H;    // wave-uniform
        // in the shifted descriptor pixel (4 ty - 1, 16 gx - 1) has the offset of pixel (4 ty, 16 gx)
        const unsigned soff = rok ? xbase + (unsigned)((long)(R0 + r) * p.W * p.x_pitch * 4) : nrec_x;
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
          unsigned vo = xlane + (unsigned)((long)cc * p.x_pitch * 4);
          if (cc == 0) vo = (q.gx == 0 && tcol == 0) ? nrec_x : vo;                     // column -1
          if (cc == 5) vo = (q.gx == p.grow - 1 && tcol == tlast) ? nrec_x : vo;        // column W
          d[r][cc] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_x, vo, soff, 0));
        }
      }
      if (r_hi == 5 && ph + 1 < nph) gnext(gq);          // the group's last row: on to the next group
    };
    f32x2 T[3][3];
    auto stage1 = [&](int cp) {
      auto D = [&](int rr) { return f32x2{d[rr - R0][2 * cp], d[rr - R0][2 * cp + 1]}; };
      if (HALF == 0) {
        T[0][cp] = fma2(4.f, D(0), fma2(-5.f, D(2), D(4)));
        const f32x2 a = fma2(-4.f, D(2), D(4)), b = fma2(-4.f, D(1), D(3));
        T[1][cp] = a + b;
        T[2][cp] = a - b;
      } else {
        const f32x2 cd = D(4) - D(2), f = D(3) - D(1);
        T[0][cp] = fma2(2.f, f, cd);
        T[1][cp] = fma2(-2.f, f, cd);
        T[2][cp] = fma2(4.f, D(1), fma2(-5.f, D(3), D(5)));
      }
    };
    auto stage2 = [&](float* obuf, int i, int part) {
      float* v = obuf + xwr + i * (6 * 128);
      const float t0 = T[i][0][0], t1 = T[i][0][1], t2 = T[i][1][0], t3 = T[i][1][1], t4 = T[i][2][0], t5 = T[i][2][1];
      if (part == 0) {
        const float a = __builtin_fmaf(-4.f, t2, t4), b = __builtin_fmaf(-4.f, t1, t3);
        v[0 * 128] = __builtin_fmaf(4.f, t0, __builtin_fmaf(-5.f, t2, t4));
        v[1 * 128] = a + b;
        v[2 * 128] = a - b;
      } else {
        const float cd = t4 - t2, f = t3 - t1;
        v[3 * 128] = __builtin_fmaf(2.f, f, cd);
        v[4 * 128] = __builtin_fmaf(-2.f, f, cd);
        v[5 * 128] = __builtin_fmaf(4.f, t1, __builtin_fmaf(-5.f, t3, t5));
      }
    };
    load_rows(0, 0, 5);
#pragma unroll
    for (int cp = 0; cp < 3; ++cp) stage1(cp);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      stage2(smem, i, 0);
      stage2(smem, i, 1);
    }
    load_rows(1, 0, 5);
    __syncthreads();
    auto phase = [&](int ph, auto cur_c) {
      constexpr int cur = decltype(cur_c)::value;
      const float* const ocur = smem + cur * OPB;
      float* const onxt = smem + (cur ^ 1) * OPB;
      f32x2 af = *reinterpret_cast<const f32x2*>(ocur + ard);
      f32x2 bf = *reinterpret_cast<const f32x2*>(ocur + brd);
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        mfma_step(ocur, e, af, bf);
        if (!(W4W_EXP & 2)) {
          if (e < 3) stage1(e);
          else stage2(onxt, (e - 3) >> 1, (e - 3) & 1);
        }
        // the registers of group ph + 1 are free after stage 1: group ph + 2 is requested right behind it (steps 3 and 4),
        // five steps ahead of its use
        if (e == 3 && !(W4W_EXP & 4)) load_rows(ph + 2, 0, 3);
        if (e == 4 && !(W4W_EXP & 4)) load_rows(ph + 2, 3, 5);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    };
    for (int ph = 0; ph < nph; ph += 2) {
      phase(ph, std::integral_constant<int, 0>{});
      if (ph + 1 < nph) phase(ph + 1, std::integral_constant<int, 1>{});
    }
  };

  if (wid < 4) body_y();
  else if (wid & 1) body_x(std::integral_constant<int, 1>{});
  else body_x(std::integral_constant<int, 0>{});

  // =================================== epilogue ===================================
  // bias sums: [tile 4][n 64] through LDS, summed in tile order by the first wave of the input-channel block 0
  // (the operand buffers are dead: the last phase ended with a barrier)
  float* const Bs = smem + 2 * MH;                         // behind the two exchange buffers (147 KB + 1 KB)
  if (p.bias && wid < 4) Bs[wid * 64 + lane] = bsum;
  const long n_w = (long)p.Nw * 9 * p.Cw;
  float* const slab = p.out + (long)split * p.slab;
  // the 36 xi of an (n, c) sit in four waves: accumulators through LDS, two rounds of 16 output channels per half
  float* const Mh = smem + nh * MH;
  const int th = tid & 255;
  const int ec = th & 31, eg = (th >> 5) & 3, erow = __builtin_amdgcn_readfirstlane(th >> 7);   // input channel, n group, tap rows
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    if (rd == 1) __syncthreads();
#pragma unroll
    for (int e = 0; e < 9; ++e)
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int g = 2 * rd + gg;
        const f32x4 v = {acc[e][4 * g], acc[e][4 * g + 1], acc[e][4 * g + 2], acc[e][4 * g + 3]};
        *reinterpret_cast<f32x4*>(Mh + (((xg * 9 + e) * 4 + (2 * gg + lh)) * 32 + lr) * 4) = v;
      }
    __syncthreads();
    if (rd == 0 && p.bias && tc == 0 && tid < 64)
      slab[n_w + n0 + tid] = ((Bs[tid] + Bs[64 + tid]) + Bs[128 + tid]) + Bs[192 + tid];
    // G^T = [1/4 -1/6 -1/6 1/24 1/24 0; 0 -1/6 1/6 1/12 -1/12 0; 0 -1/6 -1/6 1/6 1/6 1]: rows a of the 3x3 result.
    // This thread: input channel ec, output channels 4 eg .. 4 eg + 3 of the round (n on the vector lanes), tap rows
    // {0, 1} (erow 0) or {2} (erow 1).
    constexpr float k6 = 1.f / 6.f, k12 = 1.f / 12.f, k24 = 1.f / 24.f;
    auto m = [&](int i, int j) -> f32x4 { return *reinterpret_cast<const f32x4*>(Mh + (((i * 6 + j) * 4 + eg) * 32 + ec) * 4); };
    auto row3 = [&](const f32x4 (&q)[6], f32x4 (&o)[3]) {      // the same combination along the other axis
      const f32x4 s12 = add4(q[1], q[2]), d21 = sub4(q[2], q[1]), s34 = add4(q[3], q[4]), d34 = sub4(q[3], q[4]);
      o[0] = fma4(k24, s34, fma4(-k6, s12, q[0] * 0.25f));
      o[1] = fma4(k12, d34, d21 * k6);
      o[2] = add4(fma4(k6, s34, s12 * (-k6)), q[5]);
    };
    const int nbase = n0 + nh * 32 + rd * 16 + eg * 4;          // first of this thread's four output channels
    auto finish = [&](int a, const f32x4 (&X)[6]) {             // X[j]: row a of G^T m, per xi column j
      f32x4 o[3];
      row3(X, o);
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int k = 0; k < 4; ++k) slab[((long)(nbase + k) * 9 + a * 3 + b) * p.Cw + c0 + ec] = o[b][k];
    };
    if (erow == 0) {
      f32x4 X0[6], X1[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const f32x4 m1 = m(1, j), m2 = m(2, j), m3 = m(3, j), m4 = m(4, j);
        const f32x4 s12 = add4(m1, m2), s34 = add4(m3, m4);
        X0[j] = fma4(k24, s34, fma4(-k6, s12, m(0, j) * 0.25f));
        X1[j] = fma4(k12, sub4(m3, m4), sub4(m2, m1) * k6);
      }
      finish(0, X0);
      finish(1, X1);
    } else {
      f32x4 X2[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const f32x4 s12 = add4(m(1, j), m(2, j)), s34 = add4(m(3, j), m(4, j));
        X2[j] = add4(fma4(k6, s34, s12 * (-k6)), m(5, j));
      }
      finish(2, X2);
    }
  }
}

__global__ __launch_bounds__(512, 2) void wino4_wgrad_kernel(const WArgs p) { wino4_wgrad_body(p, (int)blockIdx.x); }

__global__ __launch_bounds__(512, 2) void wino4_wgrad2_kernel(const WArgs pa, const WArgs pb, const int na) {
  if ((int)blockIdx.x < na) wino4_wgrad_body(pa, (int)blockIdx.x);
  else wino4_wgrad_body(pb, (int)blockIdx.x - na);
}

// up to four layers in one launch: block ranges [0, e0), [e0, e1), [e1, e2), [e2, grid) (unused ranges are empty)
__global__ __launch_bounds__(512, 2) void wino4_wgrad4_kernel(const WArgs p0, const WArgs p1, const WArgs p2, const WArgs p3,
                                                               const int e0, const int e1, const int e2) {
  const int b = (int)blockIdx.x;
  if (b < e0) wino4_wgrad_body(p0, b);
  else if (b < e1) wino4_wgrad_body(p1, b - e0);
  else if (b < e2) wino4_wgrad_body(p2, b - e1);
  else wino4_wgrad_body(p3, b - e2);
}

// up to eight layers in one launch (the argument block is 8 x 104 bytes): range ends in e[0 .. 6]
struct WArgs8 {
  WArgs p[8];
  int e[7];
};
__global__ __launch_bounds__(512, 2) void wino4_wgrad8_kernel(const WArgs8 a) {
  const int b = (int)blockIdx.x;
  int k = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) k += b >= a.e[i] ? 1 : 0;           // block-uniform
  const int base = k == 0 ? 0 : a.e[k - 1];
  wino4_wgrad_body(a.p[k], b - base);
}

}  // namespace lgmwino4w

// ---- host ------------------------------------------------------------------------------------------------------------
bool lgm_wino4_wgrad_supported(const LgmConvGeom* g) {
  if (!(g->KH == 3 && g->KW == 3 && g->stride == 1 && g->pad == 1)) return false;
  if (g->Nw % 64 != 0 || g->Cw % 32 != 0) return false;
  const bool row_groups = g->W % 16 == 0 && g->H % 4 == 0, square_groups = g->W % 8 == 0 && g->H % 8 == 0;
  if (!row_groups && !square_groups) return false;
  const long groups = (long)g->B * g->H * g->W / 64;
  return groups >= 4;
}

// The layers that take this kernel: where the F(4x4) input gradient is preferred (lgm_conv3x3_wino4_preferred: the large
// maps at batches that fill the chip), so that a layer's two gradients switch together; LGM_NO_WINO4_WGRAD=1: never.
extern "C" int64_t lgm_conv3x3_wino4_preferred(const LgmConvGeom* g, int yx);
bool lgm_wino4_wgrad_use(const LgmConvGeom* g) {
  static const bool off = getenv("LGM_NO_WINO4_WGRAD") != nullptr || getenv("LGM_NO_WINO4") != nullptr;
  return !off && lgm_wino4_wgrad_supported(g) && lgm_conv3x3_wino4_preferred(g, 1) != 0;
}

// splits >= 2 always (the kernel only writes slabs); gps = tile groups per split; budget = workgroups of one round
void lgm_wino4_wgrad_plan(const LgmConvGeom* g, long budget, int* splits, int* gps, int* total_groups) {
  const long groups = (long)g->B * g->H * g->W / 64;
  const long blocks = (long)(g->Nw / 64) * (g->Cw / 32);
  long smax = groups / 2 < budget ? groups / 2 : budget;
  if (smax < 2) smax = 2;
  // rounds of `budget` workgroups x (phases per workgroup + ~6 phases of prologue / epilogue); ties go to fewer slabs
  long s = 2, best = -1;
  for (long c = 2; c <= smax; ++c) {
    const long rounds = (blocks * c + budget - 1) / budget;
    const long cost = rounds * ((groups + c - 1) / c + 6);
    if (best < 0 || cost < best) {
      best = cost;
      s = c;
    }
  }
  long per = (groups + s - 1) / s;
  s = (groups + per - 1) / per;
  *splits = (int)s;
  *gps = (int)per;
  *total_groups = (int)groups;
}

static void wino4_wgrad_prepare(const LgmConvGeom* g, const float* y, long y_pitch, const float* x, long x_pitch, float* out,
                                int bias, long slab, int splits, int gps, int total, lgmwino4w::WArgs& p) {
  p = lgmwino4w::WArgs{};
  p.y = y; p.x = x; p.out = out; p.bias = bias; p.slab = slab; p.y_pitch = y_pitch; p.x_pitch = x_pitch;
  p.B = g->B; p.H = g->H; p.W = g->W; p.Nw = g->Nw; p.Cw = g->Cw;
  p.tiles_c = g->Cw / 32;
  p.splits = splits; p.gps = gps; p.total_groups = total;
  p.sq = (g->W % 16 != 0) ? 1 : 0;
  static const int xcd_order = getenv("LGM_W4W_XCD") ? atoi(getenv("LGM_W4W_XCD")) : 1;
  p.xcd_order = xcd_order;
  p.grow = p.sq ? g->W / 8 : g->W / 16;
  p.trows = p.sq ? g->H / 8 : g->H / 4;
}

static constexpr size_t kW4Smem = (size_t)(2 * lgmwino4w::MH + 256) * sizeof(float);

int lgm_wino4_wgrad_launch(const LgmConvGeom* g, const float* y, long y_pitch, const float* x, long x_pitch, float* out,
                           int bias, long slab, int splits, int gps, int total, hipStream_t s) {
  using namespace lgmwino4w;
  WArgs p;
  wino4_wgrad_prepare(g, y, y_pitch, x, x_pitch, out, bias, slab, splits, gps, total, p);
  const unsigned nblocks = (unsigned)((g->Nw / 64) * (g->Cw / 32) * splits);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kW4Smem);
    attr = true;
  }
  lgm_note_kernel(LGM_KNAME("lgmwino4w::wino4_wgrad_kernel"));
  hipLaunchKernelGGL(wino4_wgrad_kernel, dim3(nblocks), dim3(512), kW4Smem, s, p);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

// two ... four layers in one launch: block ranges of one grid (as lgm_conv3x3_wino_wgradn does for the F(2x2) kernel)
int lgm_wino4_wgradn_launch(int n, const LgmConvGeom* const* gs, const float* const* ys, const long* yps,
                            const float* const* xs, const long* xps, float* const* outs, const int* biases, const long* slabs,
                            const int* splits, const int* gpss, const int* totals, hipStream_t s) {
  using namespace lgmwino4w;
  WArgs pp[8];
  unsigned nb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < n; ++k) {
    wino4_wgrad_prepare(gs[k], ys[k], yps[k], xs[k], xps[k], outs[k], biases[k], slabs[k], splits[k], gpss[k], totals[k], pp[k]);
    nb[k] = (unsigned)((gs[k]->Nw / 64) * (gs[k]->Cw / 32) * splits[k]);
  }
  for (int k = n; k < 8; ++k) pp[k] = pp[n - 1];            // never reached: its block range is empty
  if (n > 4) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_wgrad8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kW4Smem);
      attr = true;
    }
    WArgs8 a8;
    unsigned tot = 0;
    for (int k = 0; k < 8; ++k) {
      a8.p[k] = pp[k];
      tot += nb[k];
      if (k < 7) a8.e[k] = (int)tot;
    }
    lgm_note_kernel(LGM_KNAME("lgmwino4w::wino4_wgrad8_kernel"));
    hipLaunchKernelGGL(wino4_wgrad8_kernel, dim3(tot), dim3(512), kW4Smem, s, a8);
  } else if (n == 2) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_wgrad2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kW4Smem);
      attr = true;
    }
    lgm_note_kernel(LGM_KNAME("lgmwino4w::wino4_wgrad2_kernel"));
    hipLaunchKernelGGL(wino4_wgrad2_kernel, dim3(nb[0] + nb[1]), dim3(512), kW4Smem, s, pp[0], pp[1], (int)nb[0]);
  } else {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_wgrad4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kW4Smem);
      attr = true;
    }
    const int e0 = (int)nb[0], e1 = e0 + (int)nb[1], e2 = e1 + (int)nb[2];
    lgm_note_kernel(LGM_KNAME("lgmwino4w::wino4_wgrad4_kernel"));
    hipLaunchKernelGGL(wino4_wgrad4_kernel, dim3(nb[0] + nb[1] + nb[2] + nb[3]), dim3(512), kW4Smem, s, pp[0], pp[1], pp[2],
                       pp[3], e0, e1, e2);
  }
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

// ---- C-ABI (stand-alone form; the product reaches the kernel through lgm_conv_wgrad[_deferred] / lgm_conv3x3_wino_wgradn) ----
extern "C" int64_t lgm_conv3x3_wino4_wgrad_supported(const LgmConvGeom* g) {
  return (g && lgm_wino4_wgrad_supported(g)) ? 1 : 0;
}

extern "C" int64_t lgm_conv3x3_wino4_wgrad_workspace(const LgmConvGeom* g) {
  if (!g || !lgm_wino4_wgrad_supported(g)) return 0;
  int splits, gps, total;
  lgm_wino4_wgrad_plan(g, lgm_cu_budget(), &splits, &gps, &total);
  return (int64_t)splits * ((int64_t)g->Nw * 9 * g->Cw + g->Nw) * (int64_t)sizeof(float);
}

// slabs only: `workspace` receives `desc[6]` slabs [Nw][9][Cw] (+ Nw bias sums each); desc as lgm_conv_wgrad_deferred
extern "C" int lgm_conv3x3_wino4_wgrad(const LgmConvGeom* g, const float* y, int64_t y_pitch, const float* x, int64_t x_pitch,
                                       float* gw, float* gbias, float beta, void* workspace, int64_t workspace_bytes,
                                       int64_t* desc, void* stream) {
  LGM_REQUIRE(g && y && x && gw && workspace && desc, "conv3x3_wino4_wgrad: null pointer");
  LGM_REQUIRE(lgm_wino4_wgrad_supported(g), "conv3x3_wino4_wgrad: unsupported geometry");
  LGM_REQUIRE(y_pitch % 4 == 0 && x_pitch % 4 == 0 && y_pitch >= g->Nw && x_pitch >= g->Cw && lgm_aligned16(y) && lgm_aligned16(x) &&
              lgm_aligned16(gw) && lgm_aligned16(workspace) && (!gbias || lgm_aligned16(gbias)) &&
              ((long)g->B * g->H * g->W + g->W + 1) * x_pitch < (1L << 29) && (long)g->B * g->H * g->W * y_pitch < (1L << 29),
              "conv3x3_wino4_wgrad: 16-byte aligned operands with pitch %% 4 == 0 inside 32-bit offsets expected");
  int splits, gps, total;
  lgm_wino4_wgrad_plan(g, lgm_cu_budget(), &splits, &gps, &total);
  const long n_w = (long)g->Nw * 9 * g->Cw, slab = n_w + g->Nw;
  LGM_REQUIRE(workspace_bytes >= (int64_t)splits * slab * (int64_t)sizeof(float), "conv3x3_wino4_wgrad: workspace too small");
  if (int rc = lgm_wino4_wgrad_launch(g, y, y_pitch, x, x_pitch, (float*)workspace, gbias ? 1 : 0, slab, splits, gps, total,
                                      (hipStream_t)stream))
    return rc;
  union { float f; int64_t i; } bb;
  bb.i = 0;
  bb.f = beta;
  desc[0] = (int64_t)(uintptr_t)workspace; desc[1] = slab; desc[2] = (int64_t)(uintptr_t)gw; desc[3] = n_w;
  desc[4] = (int64_t)(uintptr_t)gbias; desc[5] = gbias ? g->Nw : 0; desc[6] = splits; desc[7] = bb.i;
  return LGM_OK;
}
