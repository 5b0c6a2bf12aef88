// Winograd F(4x4, 3x3) convolution, LIGHT workgroups: the same arithmetic and the same operands as winograd4.hip
// (reference Block.proj ddpm.py:160-171 and the 3x3 convolutions of the up path :93-97,377,413, forward and input
// gradient), decomposed so that a workgroup is HALF the size of that kernel's in every resource:
//
//   UNIT = 16 tiles (2 x 8 tiles = 8 x 32 output pixels of one image; the 4 x 4 tiles of one 16 x 16 image; 2 x 2 tiles of four
//   8 x 8 images) x 64 output channels x one split of the reduction; 256 threads = 4 waves, ONE per SIMD, 144 accumulator
//   registers each (wave = 9 of the 36 xi x 64 output channels x 16 tiles on v_mfma_f32_16x16x4_f32: D[channel][tile]), 74 KB
//   of LDS.
//
// Why: winograd4.hip's workgroup takes a CU whole (512 threads, 2 x 256 registers per SIMD
// lane, 115-147 KB of LDS).  (a) A launch with fewer than 256 units (per-rank batches under strong scaling) leaves CUs idle
// while the busy ones run two waves per SIMD that only share one MFMA pipe; with 16-tile units the same launch covers twice
// the CUs at one wave per SIMD.  (b) A CU that hosts one foreign resident workgroup (RCCL's all-reduce kernel, one
// 256-thread workgroup per channel) cannot take a whole-CU workgroup, so a chip-filling launch waits for a second round
// (+27...35 % on the step, tools/cu_hog_step.py); a light workgroup fits beside it, and because fp32 MFMA time is per SIMD a
// CU that holds ONE light workgroup finishes it in about half the time two co-resident ones take - the dispatcher's
// greedy assignment then loses almost nothing.  Two light workgroups per free CU = the occupancy of the big kernel.
//
// Same transformed weights U (layout of wino4_weights_kernel: [N/64][C/8][36][half][kq][32 rows][4]): lane group kk of the
// 16x16x4 MFMA contracts k = 2 kk, 2 kk + 1 (the two MFMAs of a phase), which are adjacent floats of that layout (one 8-byte
// load per 16-channel block and xi) and adjacent floats of V[xi][kq][tile][4] in LDS (one conflict-free ds_read_b64 per xi).
// Transform role: thread = (tile, reduction channel, half of the 6 rows of V) as in winograd4.hip; raw patches fetched two
// phases ahead, committed channel-planar, ONE barrier per phase, nine dealt steps of 8 MFMAs.
// Epilogue: accumulators through LDS in two rounds of 32 output channels ([xi][tile][8 quads XOR tile&7][4]: conflict-free
// 16-byte writes and reads), thread = (tile, channel quad, output row pair) applies A^T . A and stores 128-byte segments with
// bias / residual fused; split-K partial planes and the GroupNorm statistics rows as in winograd4.hip.
#include <stdlib.h>

#include <type_traits>

#include "lgm_common.h"

int lgm_splitk_reduce_launch(const float* ws, long ws_stride, int splits, const float* bias, const float* res,
                             long res_pitch, float* out, long out_pitch, long M, int N, hipStream_t s);

namespace lgmwino4l {
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int KC = 8;              // reduction channels per phase
constexpr int NXI = 36;
constexpr int NT = 16;             // tiles per unit
constexpr int VBUF = NXI * 128;    // floats per V buffer: [xi][kq 2][tile 16][4]
constexpr int MBUF = NXI * NT * 32;   // floats of the epilogue exchange: [xi][tile 16][8 quads (swizzled)][4]

template <int CLS>
struct Geo;
template <>
struct Geo<0> {   // maps with H % 8 == 0, W % 32 == 0: 2 x 8 tiles of one image
  static constexpr int NI = 1, TTH = 2, TTW = 8, PH = 10, PW = 34, RS = 34, IMG = PH * RS, PLANE = 353;
};
template <>
struct Geo<1> {   // 16 x 16 maps: the 4 x 4 tiles of one image; 4 RS = 16 (mod 32) spreads the two tile rows of a wave
  static constexpr int NI = 1, TTH = 4, TTW = 4, PH = 18, PW = 18, RS = 20, IMG = PH * RS, PLANE = 385;
};
template <>
struct Geo<2> {   // 8 x 8 maps: 2 x 2 tiles of four images; 4 RS = 8, IMG = 16 (mod 32): a wave's 2 x 2 x 2 (tx, ty, image)
  static constexpr int NI = 4, TTH = 2, TTW = 2, PH = 10, PW = 10, RS = 10, IMG = 112, PLANE = 449;   // tiles 8 banks apart
};

struct Args {
  const float* a;      // gathered activations, NHWC
  const float* u;      // transformed weights [N/64][C/8][36][2][2][32][4] (wino4_weights_kernel)
  const float* bias;
  const float* res;
  float* out;
  long a_pitch, res_pitch, out_pitch;
  int B, H, W;
  int C;               // reduction channels
  int N;               // produced channels
  int tb_h, tb_w, tiles_n, nbg;
  int splits, pps, units;
  int tn_slowest;      // unit order, see the kernel
  int xcd_ranges;      // 1: an XCD takes a contiguous unit range (default); 0: unit = blockIdx (LGM_WINO4_NO_XCD_RANGES=1, A/B)
  float* ws;
  long ws_stride;
  float* stats;        // STATS build only: [spatial unit][4 parts][2: sum y, sum y^2][N] of the PRE-BIAS outputs
};

__device__ __forceinline__ f32x4 add4(const f32x4 a, const f32x4 b) { return a + b; }
// hipcc emits four v_sub_f32 for a vector subtraction (the neg modifiers of v_pk_add_f32 are not selected)
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

template <int CLS, bool STATS = false>
__global__ __launch_bounds__(256, 2) void wino4l_conv_kernel(const Args p) {
  using GE = Geo<CLS>;
  constexpr int NI = GE::NI, PH = GE::PH, PW = GE::PW, RS = GE::RS, IMG = GE::IMG, PLANE = GE::PLANE;
  constexpr int RBUF = 8 * PLANE;
  constexpr int NPIX = NI * PH * PW;
  constexpr int NJ = (2 * NPIX + 255) / 256;
  static_assert(NJ >= 1 && NJ <= 4, "the commit is dealt out over the last NJ steps");
  static_assert(2 * RBUF + 2 * VBUF <= MBUF, "the epilogue exchange sets the LDS size");
  extern __shared__ __align__(16) float smem[];
  float* const Rb = smem;                   // [2][8 planes][PLANE]
  float* const Vb = smem + 2 * RBUF;        // [2][VBUF]
  float* const Mb = smem;                   // epilogue: aliases everything
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- unit ----
  // Hardware deals consecutive workgroup ids to the eight XCDs round-robin, each with its own 4 MB 16-way L2.  With
  // unit = blockIdx the 32 (64) workgroups an XCD runs together are units x, x + 8, ...: the same tile block of images
  // four apart, i.e. patches whose addresses differ by multiples of 1 MB and fall on the SAME L2 sets - they evict each
  // other between the four phases that share a 128-byte line (FETCH_SIZE 85 MB per launch for 33.5 MB of input at
  // 64 -> 64 @ 32 x 32, B = 128; the F(2x2) kernel, which always walked XCD-contiguous unit ranges: 38 MB).  An XCD takes a
  // CONTIGUOUS unit range instead: neighbouring tile blocks and images, addresses spread over all sets, halo rows and
  // the channel blocks of one tile block shared in one L2.
  int L = (p.xcd_ranges && (gridDim.x & 7) == 0) ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  // Two unit orders (host: wino4_unit_order): channel block fastest - the channel blocks and splits of one tile block sit in
  // one L2 and share its patch (large maps: the input is the big operand) - or channel block SLOWEST - an XCD works on
  // one or two (channel block, split) slices of U and streams the images past them (8 x 8 maps with hundreds of channels:
  // U is the big operand, 9 ... 19 MB, and every XCD would otherwise stream all of it)
  int tn, split, twi, thi, bg;
  if (p.tn_slowest) {
    twi = L % p.tb_w;
    L /= p.tb_w;
    thi = L % p.tb_h;
    L /= p.tb_h;
    bg = L % p.nbg;
    L /= p.nbg;
    split = L % p.splits;
    tn = L / p.splits;
  } else {
    tn = L % p.tiles_n;
    L /= p.tiles_n;
    split = L % p.splits;
    L /= p.splits;
    twi = L % p.tb_w;
    L /= p.tb_w;
    thi = L % p.tb_h;
    bg = L / p.tb_h;
  }
  const int n0 = tn * 64;
  const int h0 = thi * (4 * GE::TTH), w0 = twi * (4 * GE::TTW), b0 = bg * NI;
  const int ncc = p.C / KC;
  const int cc0 = split * p.pps;
  const int cc1 = min(ncc, cc0 + p.pps);
  const int nph = cc1 - cc0;

  // ---- raw patch slots: s = tid + 256 j -> pixel s >> 1 of the patch, channel quad s & 1 ----
  const unsigned nrec_a = (unsigned)((long)p.B * p.H * p.W * p.a_pitch * 4);
  __amdgpu_buffer_rsrc_t rsrc_a;
  {
    const unsigned long long ab = reinterpret_cast<unsigned long long>(p.a);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ab);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(ab >> 32));
    rsrc_a = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                               __builtin_amdgcn_readfirstlane(nrec_a), 0x00020000);
  }
  unsigned goff[NJ], plds[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int s = tid + 256 * j;
    const int pix = s >> 1, q = s & 1;
    unsigned g = nrec_a, l = (unsigned)(q * 4 * PLANE + NI * IMG);      // nothing -> zeros into the plane's pad
    if (pix < NPIX) {
      const int img = pix / (PH * PW);
      const int rem = pix - img * (PH * PW);
      const int py = rem / PW, px = rem - py * PW;
      const int gy = h0 - 1 + py, gx = w0 - 1 + px;
      l = (unsigned)(q * 4 * PLANE + img * IMG + py * RS + px);
      if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
        g = (unsigned)((((long)(b0 + img) * p.H + gy) * p.W + gx) * p.a_pitch + q * 4) * 4u;
    }
    goff[j] = g;
    plds[j] = l;
  }
  u32x4 rp[NJ];
  auto fetch1 = [&](int j, int ph) -> u32x4 {   // phase index relative to cc0; beyond the unit's range: zeros
    const unsigned soff = (unsigned)((cc0 + ph) * (KC * 4));
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ph < nph ? goff[j] : nrec_a, soff, 0);
  };
  auto commit_r = [&](float* rbuf, int j, const u32x4 r) {
    float* d = rbuf + plds[j];
    const f32x4 f = __builtin_bit_cast(f32x4, r);         // cast the whole vector first (hipcc, DESIGN finding 14)
    d[0] = f[0];
    d[PLANE] = f[1];
    d[2 * PLANE] = f[2];
    d[3 * PLANE] = f[3];
  };
  // the first three patches are requested before the rest of the set-up: a workgroup's first loads miss every cache
  u32x4 rq0[NJ], rq1[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) rq0[j] = fetch1(j, 0);
#pragma unroll
  for (int j = 0; j < NJ; ++j) rq1[j] = fetch1(j, 1);
#pragma unroll
  for (int j = 0; j < NJ; ++j) rp[j] = fetch1(j, 2);

  // ---- transform role: thread = (tile, reduction channel of the phase, half of the six rows of V) ----
  const int half = wid >> 1;                          // V rows 0-2 / 3-5; also the output row pair of the epilogue
  const int tk = (lane & 3) + 4 * (lane >> 5);        // reduction channel of the phase
  const int t_tile = 8 * (wid & 1) + ((lane >> 2) & 7);
  int t_img, t_ty, t_tx;
  if (CLS == 0) {
    t_img = 0;
    t_ty = t_tile >> 3;
    t_tx = t_tile & 7;
  } else if (CLS == 1) {
    t_img = 0;
    t_ty = t_tile >> 2;
    t_tx = t_tile & 3;
  } else {
    t_img = t_tile >> 2;
    t_ty = (t_tile >> 1) & 1;
    t_tx = t_tile & 1;
  }
  const int trd = tk * PLANE + t_img * IMG + 4 * t_ty * RS + 4 * t_tx;
  const int vwr = (tk >> 2) * 64 + t_tile * 4 + (tk & 3) + half * (18 * 128);

  // ---- MFMA role: wave = 9 xi x 64 output channels x 16 tiles; lane = (tile, k pair kk) ----
  const int xg = wid;
  const int mt = lane & 15, kk = lane >> 4;
  const int vrd = xg * (9 * 128) + (kk >> 1) * 64 + mt * 4 + 2 * (kk & 1);
  __amdgpu_buffer_rsrc_t rsrc_u;
  {
    const unsigned long long ub = reinterpret_cast<unsigned long long>(p.u);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ub);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(ub >> 32));
    rsrc_u = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                               __builtin_amdgcn_readfirstlane((unsigned)((long)p.N * p.C * NXI * 4)),
                                               0x00020000);
  }
  // channel block cb (16 channels) of the operand: half = cb & 1, rows 16 (cb >> 1) + channel; this lane's two k are the
  // floats 2 (kk & 1), 2 (kk & 1) + 1 of the row's k-quad kq = kk >> 1
  const unsigned ulane = (unsigned)((((kk >> 1) * 32 + mt) * 4 + 2 * (kk & 1)) * 4);
  const unsigned ubase = (unsigned)((tn * ncc + cc0) * NXI + xg * 9) * 2048u;
  // warm this XCD's L2 with the unit's operand range (as winograd4.hip: every launch inside a step starts cold, and one
  // wave per SIMD hides less of a miss): workgroups with blockIdx = x (mod 8) share an XCD, one 128-byte line per thread
  unsigned pf = 0;
  {
    // (an XCD's contiguous unit range holds gridDim / 8 / (tiles_n * splits) workgroups with this (channel block, split):
    // they share the range between them)
    const unsigned ubytes = min((unsigned)nph * (NXI * 2048u), 2u << 20);
    const unsigned nx = gridDim.x >> 3;
    const unsigned spatial = (unsigned)(p.nbg * p.tb_h * p.tb_w);          // workgroups per (channel block, split) slice
    const unsigned grp = p.tn_slowest ? 1u : (unsigned)(p.tiles_n * p.splits);
    const unsigned share = p.tn_slowest ? ((unsigned)blockIdx.x >> 3) % spatial : ((unsigned)blockIdx.x >> 3) / grp;
    const unsigned nshare = p.tn_slowest ? min(nx, spatial) : nx / grp;
    const unsigned off = (share * 256u + (unsigned)tid) * 128u;
    if (p.xcd_ranges && (gridDim.x & 7) == 0 && off < ubytes && nshare * 256u * 128u >= ubytes)
      pf = __builtin_amdgcn_raw_buffer_load_b32(rsrc_u, off, (unsigned)((tn * ncc + cc0) * NXI) * 2048u, 0);
  }
  struct UF {
    f32x2 c[4];
  };
  auto load_u = [&](int ph, int e) -> UF {           // phases past the unit's range are never consumed
    const unsigned soff = ubase + (unsigned)(ph * NXI + e) * 2048u;
    UF r;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc_u, ulane + (unsigned)(((cb & 1) * 64 + (cb >> 1) * 16) * 16), soff, 0);
      r.c[cb] = __builtin_bit_cast(f32x2, v);
    }
    return r;
  };

  // Everything from here on is instantiated twice, for the two halves of the workgroup (waves 0-1 / 2-3): the halves differ
  // in which three rows of V they build and which two output rows they finish, and a wave-uniform branch INSIDE a phase
  // would split its scheduling region (DESIGN finding 12) - so the branch is taken once, here.
  auto body = [&](auto half_c) {
    constexpr int HALF = decltype(half_c)::value;
    f32x4 acc[9][4];
#pragma unroll
    for (int e = 0; e < 9; ++e)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[e][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 T[3][3];                                    // T[i][cp] = rows (3 HALF + i) of B^T d, columns 2cp, 2cp + 1
    int roff0 = trd, roff1 = RBUF + trd;             // (offsets, not pointers: the LDS address space must survive)
    asm volatile("" : "+v"(roff0));
    asm volatile("" : "+v"(roff1));
    auto stage1 = [&](int nxt, int cp) {
      const float* r = smem + (nxt ? roff1 : roff0);
      f32x2 d[6];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr)
        if (rr != (HALF ? 0 : 5)) d[rr] = f32x2{r[rr * RS + 2 * cp], r[rr * RS + 2 * cp + 1]};
      if (HALF == 0) {
        // row0 = 4 d0 - 5 d2 + d4;  row1 = (d4 - 4 d2) + (d3 - 4 d1);  row2 = (d4 - 4 d2) - (d3 - 4 d1)
        T[0][cp] = fma2(4.f, d[0], fma2(-5.f, d[2], d[4]));
        const f32x2 a = fma2(-4.f, d[2], d[4]), b = fma2(-4.f, d[1], d[3]);
        T[1][cp] = a + b;
        T[2][cp] = a - b;
      } else {
        // row3 = (d4 - d2) + 2 (d3 - d1);  row4 = (d4 - d2) - 2 (d3 - d1);  row5 = 4 d1 - 5 d3 + d5
        const f32x2 c = d[4] - d[2], f = d[3] - d[1];
        T[0][cp] = fma2(2.f, f, c);
        T[1][cp] = fma2(-2.f, f, c);
        T[2][cp] = fma2(4.f, d[1], fma2(-5.f, d[3], d[5]));
      }
    };
    auto stage2 = [&](float* vbuf, int i, int part) {  // row i of the half, columns 0-2 (part 0) / 3-5 (part 1)
      float* v = vbuf + vwr + i * (6 * 128);
      const float t0 = T[i][0][0], t1 = T[i][0][1], t2 = T[i][1][0], t3 = T[i][1][1], t4 = T[i][2][0], t5 = T[i][2][1];
      if (part == 0) {
        const float a = __builtin_fmaf(-4.f, t2, t4), b = __builtin_fmaf(-4.f, t1, t3);
        v[0 * 128] = __builtin_fmaf(4.f, t0, __builtin_fmaf(-5.f, t2, t4));
        v[1 * 128] = a + b;
        v[2 * 128] = a - b;
      } else {
        const float c = t4 - t2, f = t3 - t1;
        v[3 * 128] = __builtin_fmaf(2.f, f, c);
        v[4 * 128] = __builtin_fmaf(-2.f, f, c);
        v[5 * 128] = __builtin_fmaf(4.f, t1, __builtin_fmaf(-5.f, t3, t5));
      }
    };

    // ---- prologue ----
    UF uq[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) uq[e] = load_u(0, e);
#pragma unroll
    for (int j = 0; j < NJ; ++j) commit_r(Rb, j, rq0[j]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) commit_r(Rb + RBUF, j, rq1[j]);
    __syncthreads();
    asm volatile("" ::"v"(pf));                      // the operand prefetch, issued before those patches, is complete
#pragma unroll
    for (int cp = 0; cp < 3; ++cp) stage1(0, cp);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      stage2(Vb, i, 0);
      stage2(Vb, i, 1);
    }
    __syncthreads();

    // ---- phase: nine steps of 8 MFMAs (one xi each), the side work dealt out over them; nothing crosses a step boundary
    // (sched_barrier), so the U fragments loaded at the end of step e for step e + 3 ARE three steps ahead.  Raw patches:
    // raw(ph + 1) is transformed (steps 0-8), raw(ph + 2) - in registers since the previous phase - is committed (last NJ
    // steps) and raw(ph + 3) requested in its place ----
    auto phase = [&](int ph, auto cur_c) {
      constexpr int cur = decltype(cur_c)::value;
      float* const rcur = Rb + cur * RBUF;
      float* const vcur = Vb + cur * VBUF;
      float* const vnxt = Vb + (cur ^ 1) * VBUF;
      f32x2 vf = *reinterpret_cast<const f32x2*>(vcur + vrd);
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        f32x2 vfn = vf;
        if (e < 8) vfn = *reinterpret_cast<const f32x2*>(vcur + vrd + (e + 1) * 128);
        const UF uf = uq[e % 3];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int cb = 0; cb < 4; ++cb)
            acc[e][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf.c[cb][g], vf[g], acc[e][cb], 0, 0, 0);
        uq[e % 3] = (e + 3 < 9) ? load_u(ph, e + 3) : load_u(ph + 1, e + 3 - 9);
        if (e < 3) stage1(cur ^ 1, e);
        else stage2(vnxt, (e - 3) >> 1, (e - 3) & 1);
        if (e >= 9 - NJ) {                               // raw(ph + 2), requested a whole phase ago, into the buffer
          commit_r(rcur, e - (9 - NJ), rp[e - (9 - NJ)]);   // phase ph - 1 finished reading; then the request for raw(ph + 3)
          rp[e - (9 - NJ)] = fetch1(e - (9 - NJ), ph + 3);
        }
        vf = vfn;
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    };
    for (int ph = 0; ph < nph; ph += 2) {
      phase(ph, std::integral_constant<int, 0>{});
      if (ph + 1 < nph) phase(ph + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue.  The 36 xi of a (tile, channel) sit in the four waves: the accumulators go through LDS, two ROUNDS of
    // 32 output channels (channel blocks 2 rd, 2 rd + 1).  Then thread = (tile, channel quad, output row pair): A^T . A in
    // registers, 128-byte segments per pixel out with bias / residual fused. ----
    const int eq = tid & 7, et = (tid >> 3) & 15, rpair = __builtin_amdgcn_readfirstlane(tid >> 7);
    int e_img, e_ty, e_tx;
    if (CLS == 0) {
      e_img = 0;
      e_ty = et >> 3;
      e_tx = et & 7;
    } else if (CLS == 1) {
      e_img = 0;
      e_ty = et >> 2;
      e_tx = et & 3;
    } else {
      e_img = et >> 2;
      e_ty = (et >> 1) & 1;
      e_tx = et & 1;
    }
    const long opix = ((long)(b0 + e_img) * p.H + h0 + 4 * e_ty) * p.W + w0 + 4 * e_tx;
    const bool partial = p.splits > 1;
    const int ncol0 = n0 + eq * 4;
    float* const obase = partial ? p.ws + (long)split * p.ws_stride + opix * p.N + ncol0 : p.out + opix * p.out_pitch + ncol0;
    const long opitch = partial ? (long)p.N : p.out_pitch;
    const bool has_res = !partial && p.res != nullptr;           // kernel argument: a scalar branch
    const float* const rbase = p.res + opix * p.res_pitch + ncol0;
    const int mrd = (et * 8 + (eq ^ (et & 7))) * 4;
    const int mwr = mt * 32;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      if (rd == 1) __syncthreads();                    // round 0's reads are done before its buffer is overwritten
#pragma unroll
      for (int e = 0; e < 9; ++e)
#pragma unroll
        for (int cbl = 0; cbl < 2; ++cbl)
          *reinterpret_cast<f32x4*>(Mb + (xg * 9 + e) * 512 + mwr + (((4 * cbl + kk) ^ (mt & 7)) * 4)) = acc[e][2 * rd + cbl];
      __syncthreads();
      // stage 1 (over the xi rows i, per xi column j): this thread's two rows of A^T m
      f32x4 X[2][6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        auto m = [&](int i) -> f32x4 { return *reinterpret_cast<const f32x4*>(Mb + (i * 6 + j) * 512 + mrd); };
        const f32x4 m1 = m(1), m2 = m(2), m3 = m(3), m4 = m(4);
        if (HALF == 0) {
          const f32x4 s1 = add4(m1, m2), s2 = add4(m3, m4);
          X[0][j] = add4(add4(m(0), s1), s2);          // row 0
          X[1][j] = fma4(4.f, s2, s1);                 // row 2
        } else {
          const f32x4 d1 = sub4(m1, m2), d2 = sub4(m3, m4);
          X[0][j] = fma4(2.f, d2, d1);                 // row 1
          X[1][j] = add4(fma4(8.f, d2, d1), m(5));     // row 3
        }
      }
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (!partial && p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + ncol0 + rd * 32);
      f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const long orow = (long)(2 * rr + rpair) * p.W;
        const f32x4 s1 = add4(X[rr][1], X[rr][2]), d1 = sub4(X[rr][1], X[rr][2]);
        const f32x4 s2 = add4(X[rr][3], X[rr][4]), d2 = sub4(X[rr][3], X[rr][4]);
        f32x4 y[4];
        y[0] = add4(add4(X[rr][0], s1), s2);
        y[1] = fma4(2.f, d2, d1);
        y[2] = fma4(4.f, s2, s1);
        y[3] = add4(fma4(8.f, d2, d1), X[rr][5]);
        if (STATS) {
#pragma unroll
          for (int oj = 0; oj < 4; ++oj) {
            st1 = add4(st1, y[oj]);
            st2 = __builtin_elementwise_fma(y[oj], y[oj], st2);
          }
        }
        if (has_res) {
#pragma unroll
          for (int oj = 0; oj < 4; ++oj)
            y[oj] = add4(y[oj], *reinterpret_cast<const f32x4*>(rbase + (orow + oj) * p.res_pitch + rd * 32));
        }
#pragma unroll
        for (int oj = 0; oj < 4; ++oj) *reinterpret_cast<f32x4*>(obase + (orow + oj) * opitch + rd * 32) = add4(y[oj], bv);
      }
      if (STATS) {
        // the 8 lanes of this wave that share the channel quad eq (lane = eq + 8 (tile & 7)): a fixed butterfly
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            st1[k] += __shfl_xor(st1[k], m, 64);
            st2[k] += __shfl_xor(st2[k], m, 64);
          }
        }
        if ((lane >> 3) == 0) {
          const int ublk = (bg * p.tb_h + thi) * p.tb_w + twi;
          float* sp = p.stats + ((long)(ublk * 4 + wid) * 2) * p.N + ncol0 + rd * 32;   // wave = 8 tiles x one row pair
          *reinterpret_cast<f32x4*>(sp) = st1;
          *reinterpret_cast<f32x4*>(sp + p.N) = st2;
        }
      }
    }
  };
  if (half == 0) body(std::integral_constant<int, 0>{});
  else body(std::integral_constant<int, 1>{});
}

static long unit_count(int cls, int B, int H, int W) {     // units per 64 produced channels, before split-K
  return cls == 1 ? B : cls == 2 ? B / 4 : (long)B * (H / 8) * (W / 32);
}

static int unit_class(int H, int W) {
  if (H == 8 && W == 8) return 2;
  if (H == 16 && W == 16) return 1;
  if (H >= 8 && W >= 32 && H % 8 == 0 && W % 32 == 0) return 0;
  return -1;
}

}  // namespace lgmwino4l

bool lgm_wino4l_supported(const LgmConvGeom* g, int gather_channels, int out_channels) {
  using namespace lgmwino4l;
  if (!(g->KH == 3 && g->KW == 3 && g->stride == 1 && g->pad == 1)) return false;
  if (gather_channels % 32 != 0 || out_channels % 64 != 0) return false;
  const int cls = unit_class(g->H, g->W);
  if (cls < 0) return false;
  const long pix = (long)g->B * g->H * g->W + g->W + 1;
  if (pix * gather_channels >= (1L << 29) || pix * out_channels >= (1L << 29)) return false;
  if ((long)gather_channels * out_channels * 36 >= (1L << 29)) return false;
  return g->B % (cls == 2 ? 4 : 1) == 0;
}

long lgm_wino4l_units(const LgmConvGeom* g, int out_channels) {
  using namespace lgmwino4l;
  const int cls = unit_class(g->H, g->W);
  return cls < 0 ? 0 : unit_count(cls, g->B, g->H, g->W) * (out_channels / 64);
}

// 0: maps with H % 8 == 0 and W % 32 == 0, 1: 16x16, 2: 8x8 (four images to a unit); -1: not taken
int lgm_wino4l_class(const LgmConvGeom* g) { return lgmwino4l::unit_class(g->H, g->W); }

// rows of GroupNorm statistics one image contributes per channel (STATS build: one row per wave of a unit)
int lgm_wino4l_stats_parts(const LgmConvGeom* g) {
  using namespace lgmwino4l;
  return unit_class(g->H, g->W) == 0 ? (g->H / 8) * (g->W / 32) * 4 : 0;
}

// Split-K plan.  Light workgroups come two to a CU (512 slots, two waves per SIMD - what hides a cold launch's memory
// latency): a launch is split until it fills them, while every split keeps >= min_pps phases.
int lgm_wino4l_splits(const LgmConvGeom* g, int gather_channels, int out_channels) {
  using namespace lgmwino4l;
  const long base = lgm_wino4l_units(g, out_channels);
  if (base <= 0) return 1;
  static const int forced = getenv("LGM_WINO4L_SPLITS") ? atoi(getenv("LGM_WINO4L_SPLITS")) : 0;
  const int phases = gather_channels / KC;
  int smax = phases / 2 < 16 ? phases / 2 : 16;
  if (smax < 1) smax = 1;
  if (forced > 0) return forced < smax ? forced : smax;
  // 256 = one light workgroup per CU: 6.89 vs 6.99 ms per step at B = 64 and 5.38 vs 5.41 at B = 32 against 512 (two per CU),
  // which only wins at B = 128 (10.22 vs 10.27), where the 32-tile kernel is the default anyway
  static const int target_env = getenv("LGM_WINO4L_TARGET") ? atoi(getenv("LGM_WINO4L_TARGET")) : 0;
  const int target = target_env > 0 ? target_env : lgm_cu_budget();
  if (base >= target * 3 / 4) return 1;
  static const int min_pps = getenv("LGM_WINO4L_MIN_PPS") ? atoi(getenv("LGM_WINO4L_MIN_PPS")) : 4;
  // (two light workgroups fit a CU: a grid a little above `target` does not run a second round, so the 32-tile kernel's
  // lgm_wino4_pick_splits does not apply - with it B = 64 on a rank's plans went 6.74 -> 6.84 ms)
  int s = (int)((target + base - 1) / base);
  if (s > smax) s = smax;
  while (s > 1 && phases / s < min_pps) --s;
  return s;
}

int lgm_wino4l_launch(const LgmConvGeom* g, int yx, const float* a, long a_pitch, const float* u, const float* bias,
                      const float* res, long res_pitch, float* out, long out_pitch, void* workspace, long workspace_bytes,
                      hipStream_t s, int64_t* partial, float* stats) {
  using namespace lgmwino4l;
  Args p{};
  p.stats = stats;
  p.a = a; p.u = u; p.bias = bias; p.res = res; p.out = out;
  p.a_pitch = a_pitch; p.res_pitch = res_pitch; p.out_pitch = out_pitch;
  p.B = g->B; p.H = g->H; p.W = g->W;
  p.C = yx ? g->Nw : g->Cw;
  p.N = yx ? g->Cw : g->Nw;
  const int cls = unit_class(g->H, g->W);
  p.tb_h = cls == 0 ? g->H / 8 : 1;
  p.tb_w = cls == 0 ? g->W / 32 : 1;
  p.nbg = cls == 2 ? g->B / 4 : g->B;
  p.tiles_n = p.N / 64;
  const long M = (long)g->B * g->H * g->W;
  p.splits = lgm_wino4l_splits(g, p.C, p.N);
  if (p.splits > 1) {
    const long need = (long)p.splits * M * p.N * (long)sizeof(float);
    if (!workspace || workspace_bytes < need || !lgm_aligned16(workspace)) p.splits = 1;
  }
  p.ws = (float*)workspace;
  p.ws_stride = M * p.N;
  p.pps = lgm_cdiv(p.C / KC, p.splits);
  p.splits = lgm_cdiv(p.C / KC, p.pps);
  p.units = (int)((long)p.nbg * p.tb_h * p.tb_w * p.tiles_n * p.splits);
  {
    // bytes the chip's eight L2s fetch under either order: patches once (x 1.2 halo) and all of U per XCD, or patches once per
    // channel block and U once
    static const bool no_ranges = getenv("LGM_WINO4_NO_XCD_RANGES") != nullptr;
    p.xcd_ranges = no_ranges ? 0 : 1;
    static const int forced = getenv("LGM_WINO4_TN_SLOWEST") ? atoi(getenv("LGM_WINO4_TN_SLOWEST")) : -1;
    const double in_b = 1.2 * (double)M * p.C * 4.0, u_b = 36.0 * p.C * p.N * 4.0;
    p.tn_slowest = forced >= 0 ? forced : ((in_b + 8.0 * u_b > in_b * p.tiles_n + u_b) ? 1 : 0);
  }
  const size_t smem = (size_t)MBUF * sizeof(float);
#define LGM_W4L(KERN)                                                                                             \
  do {                                                                                                            \
    auto kern = KERN;                                                                                             \
    static bool attr = false;                                                                                     \
    if (!attr) {                                                                                                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)smem);                                                                       \
      attr = true;                                                                                                \
    }                                                                                                             \
    hipLaunchKernelGGL(kern, dim3((unsigned)p.units), dim3(256), smem, s, p);                                     \
  } while (0)
  if (stats) {
    if (cls != 0 || p.splits != 1 || res || partial) {
      lgm_set_error("wino4l (stats): class-0 maps, an unsplit reduction and no residual expected");
      return LGM_ERR_UNSUPPORTED;
    }
    lgm_note_kernel(LGM_KNAME("lgmwino4l::wino4l_conv_kernel<0, true>"));
    LGM_W4L((wino4l_conv_kernel<0, true>));
  } else if (cls == 0) {
    lgm_note_kernel(LGM_KNAME("lgmwino4l::wino4l_conv_kernel<0, false>"));
    LGM_W4L((wino4l_conv_kernel<0, false>));
  } else if (cls == 1) {
    lgm_note_kernel(LGM_KNAME("lgmwino4l::wino4l_conv_kernel<1, false>"));
    LGM_W4L((wino4l_conv_kernel<1, false>));
  } else {
    lgm_note_kernel(LGM_KNAME("lgmwino4l::wino4l_conv_kernel<2, false>"));
    LGM_W4L((wino4l_conv_kernel<2, false>));
  }
#undef LGM_W4L
  if (partial) {
    partial[0] = p.splits;
    partial[1] = p.ws_stride;
    LGM_LAUNCH_CHECK();
    return LGM_OK;
  }
  if (p.splits > 1)
    return lgm_splitk_reduce_launch(p.ws, p.ws_stride, p.splits, bias, res, res_pitch, out, out_pitch, M, p.N, s);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

// ---- C-ABI (direct entry points: tests and tools; the product path reaches this kernel through lgm_conv3x3_wino4*, whose
// launcher picks the workgroup size - winograd4.hip) ----------------------------------------------------------------
extern "C" int64_t lgm_conv3x3_wino4l_supported(const LgmConvGeom* g, int yx) {
  if (!g) return 0;
  return lgm_wino4l_supported(g, yx ? g->Nw : g->Cw, yx ? g->Cw : g->Nw) ? 1 : 0;
}

extern "C" int64_t lgm_conv3x3_wino4l_workspace(const LgmConvGeom* g, int yx) {
  if (!g) return -1;
  const int gc = yx ? g->Nw : g->Cw, oc = yx ? g->Cw : g->Nw;
  if (!lgm_wino4l_supported(g, gc, oc)) return 0;
  const int s = lgm_wino4l_splits(g, gc, oc);
  return s > 1 ? (int64_t)s * g->B * g->H * g->W * oc * (int64_t)sizeof(float) : 0;
}

extern "C" int lgm_conv3x3_wino4l(int yx, const LgmConvGeom* g, const float* a, int64_t a_pitch, const float* u,
                                  const float* bias, const float* res, int64_t res_pitch, float* out, int64_t out_pitch,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
  LGM_REQUIRE(g && a && u && out, "conv3x3_wino4l: null pointer");
  const int gc = yx ? g->Nw : g->Cw, oc = yx ? g->Cw : g->Nw;
  LGM_REQUIRE(lgm_wino4l_supported(g, gc, oc), "conv3x3_wino4l: unsupported geometry (3x3/s1/p1, maps 8x8, 16x16 or H %% 8 == 0 "
              "and W %% 32 == 0, reduction channels %% 32, produced channels %% 64)");
  LGM_REQUIRE(a_pitch % 4 == 0 && a_pitch >= gc && lgm_aligned16(a) && lgm_aligned16(u) && lgm_aligned16(out) &&
              out_pitch % 4 == 0 && out_pitch >= oc && (!res || (lgm_aligned16(res) && res_pitch % 4 == 0 && res_pitch >= oc)) &&
              (!bias || lgm_aligned16(bias)), "conv3x3_wino4l: 16-byte aligned operands with pitch %% 4 == 0 expected");
  const long pix = (long)g->B * g->H * g->W + g->W + 1;
  LGM_REQUIRE(pix * a_pitch < (1L << 29) && pix * out_pitch < (1L << 29) && pix * (res ? res_pitch : 0) < (1L << 29),
              "conv3x3_wino4l: tensor too large for 32-bit offsets");
  return lgm_wino4l_launch(g, yx, a, a_pitch, u, bias, res, res_pitch, out, out_pitch, workspace, workspace_bytes,
                           (hipStream_t)stream, nullptr, nullptr);
}
