// Winograd F(4x4, 3x3) convolution in fp32 on v_mfma_f32_32x32x2_f32 — the 3x3 / stride 1 / pad 1 layers of the DDPM UNet
// on the LARGE feature maps (16x16, 32x32, 64x64; reference Block.proj ddpm.py:160-171 and the 3x3 convolutions of the
// up path :93-97,377,413), forward and input gradient.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 6x6 input tile d -> 4x4 output tile Y,  36 products per 16 outputs
//
// i.e. 2.25 multiplies per output instead of 4 (F(2x2,3x3), winograd.hip) or 9 (direct): 1.78x fewer MFMA FLOPs than the
// F(2x2) kernels for a comparable number of transform operations per output.  Interpolation points 0, +-1, +-2, inf; all
// arithmetic fp32 with fp32 accumulation (error against float64 a few 1e-6 of the output scale: tests/test_hip_winograd.py).
//
// Work decomposition.  UNIT = 32 tiles (4 x 8 tiles = 16 x 32 output pixels of one image; on 16 x 16 maps 4 x 4 tiles of two
// images) x 64 output channels x one split of the reduction; PHASE = 8 reduction channels.  512 threads = 8 waves, two
// per SIMD, 144 accumulator registers each:
//   * transform role: thread = (tile, reduction channel, half of the 6 rows of V): reads its 6x6 raw values from the
//     channel-planar LDS patch (plane stride = 1 mod 32 floats: conflict-free), computes 18 elements of V = B^T d B and
//     writes them to LDS as V[xi][k-half][tile][4] — every (tile, channel) is transformed ONCE per workgroup and serves
//     all 64 output channels;
//   * MFMA role: wave = (9 of the 36 xi) x (32 of the 64 output channels): per xi one ds_read_b128 of V (B operand,
//     tiles on the lanes), one 16-byte load of the transformed weights U straight from L2 (A operand, fragment order),
//     4 MFMAs into one 32x32 accumulator;
//   * the raw patch of phase p+2 is fetched into registers while phase p runs (out-of-range buffer offsets are the zero
//     padding), phase p+1's patch is transformed under phase p's MFMAs; ONE barrier per phase.
// Epilogue: the 36 xi of a (tile, channel) sit in four different waves, so the accumulators go through LDS (two passes
// of 32 output channels, 147 KB each, XOR-swizzled: conflict-free 16-byte writes and reads); thread = (tile, channel
// quad, output row pair) applies A^T . A and stores full 128-byte segments with bias / residual fused.  Split-K writes
// partial OUTPUT planes (the output transform is linear) for the fixed-order reducer or the plane-summing GroupNorm.
#include <atomic>
#include <stdlib.h>

#include <type_traits>

#include "lgm_common.h"

int lgm_splitk_reduce_launch(const float* ws, long ws_stride, int splits, const float* bias, const float* res,
                             long res_pitch, float* out, long out_pitch, long M, int N, hipStream_t s);

// diagnostic: when set (lgm_wino4_set_debug_buffer), the convolution runs its stamped build and writes, per workgroup,
// 32 int64: [0] stamp count, [1..] s_memtime (entry, set-up done, prologue done, every phase, both epilogue passes),
// [31] s_memrealtime at exit.  Never set on the product path.
static void* lgm_wino4_debug_buffer = nullptr;
static int lgm_wino4_debug_exp = 0;
static int lgm_wino4_debug_slots = 0;      // exp bits 16+: > 0 = every stamped launch takes the next slot of 4096 x 32 entries
static int lgm_wino4_debug_next = 0;       // (tools/wino4_step_stamps.py: the kernel's launches INSIDE a training step)
extern "C" int lgm_wino4_set_debug_buffer(void* buf, int exp) {
  lgm_wino4_debug_buffer = buf;
  lgm_wino4_debug_exp = exp & 0xffff;
  lgm_wino4_debug_slots = exp >> 16;
  lgm_wino4_debug_next = 0;
  return LGM_OK;
}

#ifndef LGM_WINO4_PFU
#define LGM_WINO4_PFU 1            // operand prefetch at kernel start (-DLGM_WINO4_PFU=0: A/B builds)
#endif

namespace lgmwino4 {
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int KC = 8;              // reduction channels per phase
constexpr int NXI = 36;
constexpr int VBUF = NXI * 256;    // floats per V buffer: [xi][k-half][tile 32][4]
constexpr int MBUF = NXI * 32 * 32;   // floats of the epilogue exchange: two halves of [xi][tile 32][4 quads (swizzled)][4]

template <int CLS>
struct Geo;
template <>
struct Geo<0> {   // maps >= 16 x 32: 4 x 8 tiles of one image
  static constexpr int NI = 1, TTH = 4, TTW = 8, PH = 18, PW = 34, RS = 34, IMG = PH * RS, PLANE = 641;
};
template <>
struct Geo<1> {   // 16 x 16 maps: 4 x 4 tiles of two images; 4 RS = 16 (mod 32) spreads the two tile rows of a wave
  static constexpr int NI = 2, TTH = 4, TTW = 4, PH = 18, PW = 18, RS = 20, IMG = PH * RS, PLANE = 737;
};
template <>
struct Geo<2> {   // 8 x 8 maps: 2 x 2 tiles of eight images; 4 RS = 8, IMG = 16 (mod 32): a wave's 2 x 2 x 2 (tx, ty, image)
  static constexpr int NI = 8, TTH = 2, TTW = 2, PH = 10, PW = 10, RS = 10, IMG = 112, PLANE = 897;   // tiles on 8 banks apart
};

struct Args {
  const float* a;      // gathered activations, NHWC
  const float* u;      // transformed weights [N/64][C/8][36][2][2][32][4]
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
  long long* dbg;      // diagnostic build only (wino4_conv_kernel<CLS, true>): per-workgroup cycle stamps
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

// EXP (diagnostic builds only): bit 0 drops the transform, bit 1 the patch fetch / commit, bit 2 the U loads, bit 3 the V
// reads of the phase loop -- wrong results, used to attribute the phase time.
// STATS (class 0, unsplit, no residual; the consumer is a GroupNorm whose slices do not fit a one-pass block): every wave
// also leaves, per produced channel, the sum and the sum of squares of its 128 output pixels BEFORE the bias is added
// (zero-mean-ish values: no cancellation in the squares) - lgm_gn_fwd_stats combines them in float64.
template <int CLS, bool DBG = false, int EXP = 0, bool STATS = false>
__global__ __launch_bounds__(512, 2) void wino4_conv_kernel(const Args p) {
  using GE = Geo<CLS>;
  constexpr bool PFU = LGM_WINO4_PFU != 0;
  int nstamp = 0;
  auto stamp = [&]() {
    if (DBG) {
      if (threadIdx.x == 0 && nstamp < 30) p.dbg[blockIdx.x * 32 + 1 + nstamp] = (long long)__builtin_amdgcn_s_memtime();
      ++nstamp;
    }
  };
  stamp();
  const long long rt_entry = DBG ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
  constexpr int NI = GE::NI, PH = GE::PH, PW = GE::PW, RS = GE::RS, IMG = GE::IMG, PLANE = GE::PLANE;
  constexpr int RBUF = 8 * PLANE;
  constexpr int NPIX = NI * PH * PW;
  constexpr int NJ = (2 * NPIX + 511) / 512;
  extern __shared__ __align__(16) float smem[];
  float* const Rb = smem;                   // [2][8 planes][PLANE]
  float* const Vb = smem + 2 * RBUF;        // [2][VBUF]
  float* const Mb = smem;                   // epilogue: aliases everything
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;

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

  // ---- raw patch slots: s = tid + 512 j -> pixel s >> 1 of the patch, channel quad s & 1 ----
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
    const int s = tid + 512 * j;
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
  // The first three patches are requested HERE, before the rest of the set-up (role addressing, descriptors, 144
  // accumulator zeros): a workgroup's first loads miss every cache, and ~2 k cycles of set-up fit under them.
  u32x4 rq0[NJ], rq1[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) rq0[j] = fetch1(j, 0);
#pragma unroll
  for (int j = 0; j < NJ; ++j) rq1[j] = fetch1(j, 1);
#pragma unroll
  for (int j = 0; j < NJ; ++j) rp[j] = fetch1(j, 2);

  // ---- transform role ----
  int t_tile, t_img, t_ty, t_tx;
  const int half = wid >> 2;                          // V rows 0-2 / 3-5; also the output row pair of the epilogue
  const int tk = (lane & 3) + 4 * (lane >> 5);        // reduction channel of the phase
  if (CLS == 0) {
    t_img = 0;
    t_ty = wid & 3;
    t_tx = (lane >> 2) & 7;
    t_tile = t_ty * 8 + t_tx;
  } else if (CLS == 1) {
    t_img = (wid >> 1) & 1;
    t_ty = 2 * (wid & 1) + ((lane >> 4) & 1);
    t_tx = (lane >> 2) & 3;
    t_tile = t_img * 16 + t_ty * 4 + t_tx;
  } else {
    t_img = 2 * (wid & 3) + ((lane >> 4) & 1);
    t_ty = (lane >> 3) & 1;
    t_tx = (lane >> 2) & 1;
    t_tile = t_img * 4 + t_ty * 2 + t_tx;
  }
  const int trd = tk * PLANE + t_img * IMG + 4 * t_ty * RS + 4 * t_tx;
  const int vwr = (tk >> 2) * 128 + t_tile * 4 + (tk & 3) + half * (18 * 256);

  // ---- MFMA role ----
  const int xg = wid & 3, ch = wid >> 2;
  const int vrd = xg * (9 * 256) + lh * 128 + lr * 4;
  __amdgpu_buffer_rsrc_t rsrc_u;
  {
    const unsigned long long ub = reinterpret_cast<unsigned long long>(p.u);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ub);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(ub >> 32));
    rsrc_u = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                               __builtin_amdgcn_readfirstlane((unsigned)((long)p.N * p.C * NXI * 4)),
                                               0x00020000);
  }
  const unsigned ulane = (unsigned)((ch * 64 + lane) * 16);
  const unsigned ubase = (unsigned)((tn * ncc + cc0) * NXI + xg * 9) * 2048u;
  // Warm this XCD's L2 with the unit's whole operand range.  Inside a training step every launch finds the caches cold
  // (tools/cold_launch.py: +5 ... 15 us per launch against back-to-back timing): the phase loop asks for U three xi ahead,
  // which hides an L2 hit, not a miss to memory - and every phase's 74 KB are new.  Workgroups with blockIdx = x (mod 8)
  // share an XCD; each touches its share of the range, one 128-byte line per thread (up to 2 MB: what stays in a 4 MB L2
  // beside the patches).  The value is consumed (nowhere) after the prologue's first barrier, by when it has long landed.
  unsigned pf = 0;
  if (PFU) {
    // (an XCD's contiguous unit range holds gridDim / 8 / (tiles_n * splits) workgroups with this (channel block, split):
    // they share the range between them)
    const unsigned ubytes = min((unsigned)nph * (NXI * 2048u), 2u << 20);
    const unsigned nx = gridDim.x >> 3;
    const unsigned spatial = (unsigned)(p.nbg * p.tb_h * p.tb_w);          // workgroups per (channel block, split) slice
    const unsigned grp = p.tn_slowest ? 1u : (unsigned)(p.tiles_n * p.splits);
    const unsigned share = p.tn_slowest ? ((unsigned)blockIdx.x >> 3) % spatial : ((unsigned)blockIdx.x >> 3) / grp;
    const unsigned nshare = p.tn_slowest ? min(nx, spatial) : nx / grp;
    const unsigned off = (share * 512u + (unsigned)tid) * 128u;
    if (p.xcd_ranges && (gridDim.x & 7) == 0 && off < ubytes && nshare * 512u * 128u >= ubytes)
      pf = __builtin_amdgcn_raw_buffer_load_b32(rsrc_u, off, (unsigned)((tn * ncc + cc0) * NXI) * 2048u, 0);
  }
  auto load_u = [&](int ph, int e) -> f32x4 {        // phases past the unit's range are never consumed
    const unsigned soff = ubase + (unsigned)(ph * NXI + e) * 2048u;
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, ulane, soff, 0);
    return __builtin_bit_cast(f32x4, v);
  };

  // Everything from here on is instantiated twice, for the two halves of the workgroup (waves 0-3 / 4-7): the halves
  // differ in which three rows of V they build and which two output rows they finish, and a wave-uniform branch INSIDE
  // a phase would split its scheduling region (DESIGN finding 12) - so the branch is taken once, here.
  auto body = [&](auto half_c) {
    constexpr int HALF = decltype(half_c)::value;
    f32x16 acc[9];
#pragma unroll
    for (int e = 0; e < 9; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
    f32x2 T[3][3];                                    // T[i][cp] = rows (3 HALF + i) of B^T d, columns 2cp, 2cp + 1
    // the two raw buffers' per-thread read bases, opaque to constant folding: a raw read is then base + immediate
    // (ds_read2_b32 offsets are 8-bit dword counts; folded into one base + a large constant the compiler spent a
    // v_add per read)
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
      float* v = vbuf + vwr + i * (6 * 256);
      const float t0 = T[i][0][0], t1 = T[i][0][1], t2 = T[i][1][0], t3 = T[i][1][1], t4 = T[i][2][0], t5 = T[i][2][1];
      if (part == 0) {
        const float a = __builtin_fmaf(-4.f, t2, t4), b = __builtin_fmaf(-4.f, t1, t3);
        v[0 * 256] = __builtin_fmaf(4.f, t0, __builtin_fmaf(-5.f, t2, t4));
        v[1 * 256] = a + b;
        v[2 * 256] = a - b;
      } else {
        const float c = t4 - t2, f = t3 - t1;
        v[3 * 256] = __builtin_fmaf(2.f, f, c);
        v[4 * 256] = __builtin_fmaf(-2.f, f, c);
        v[5 * 256] = __builtin_fmaf(4.f, t1, __builtin_fmaf(-5.f, t3, t5));
      }
    };

    // ---- prologue ----
    stamp();
    f32x4 uq[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) uq[e] = load_u(0, e);
    if (DBG) __builtin_amdgcn_sched_barrier(0);
    stamp();                                         // (diagnostic) set-up and accumulator zeros done, loads in flight
#pragma unroll
    for (int j = 0; j < NJ; ++j) commit_r(Rb, j, rq0[j]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) commit_r(Rb + RBUF, j, rq1[j]);
    if (DBG) __builtin_amdgcn_sched_barrier(0);
    stamp();                                         // first two patches landed and committed
    __syncthreads();
    if (PFU) asm volatile("" ::"v"(pf));             // the operand prefetch, issued before those patches, is complete
    stamp();
#pragma unroll
    for (int cp = 0; cp < 3; ++cp) stage1(0, cp);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      stage2(Vb, i, 0);
      stage2(Vb, i, 1);
    }
    if (DBG) __builtin_amdgcn_sched_barrier(0);
    stamp();                                         // first transform
    __syncthreads();
    stamp();

    // ---- phase: nine steps of 4 MFMAs (one xi each), the side work dealt out over them; nothing crosses a step
    // boundary (sched_barrier), so the U fragment loaded at the end of step e for step e + 3 IS three steps ahead.
    // Raw patches: raw(ph + 1) is transformed (steps 0-8), raw(ph + 2) - in registers since the previous phase - is
    // committed (steps 6-8) and raw(ph + 3) requested in its place ----
    auto phase = [&](int ph, auto cur_c) {
      constexpr int cur = decltype(cur_c)::value;
      float* const rcur = Rb + cur * RBUF;
      float* const vcur = Vb + cur * VBUF;
      float* const vnxt = Vb + (cur ^ 1) * VBUF;
      f32x4 vf = *reinterpret_cast<const f32x4*>(vcur + vrd);
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        f32x4 vfn = vf;
        if (e < 8 && !(EXP & 8)) vfn = *reinterpret_cast<const f32x4*>(vcur + vrd + (e + 1) * 256);
        const f32x4 uf = uq[e % 3];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf[s], vf[s], acc[e], 0, 0, 0);
        if (!(EXP & 4)) uq[e % 3] = (e + 3 < 9) ? load_u(ph, e + 3) : load_u(ph + 1, e + 3 - 9);
        if (!(EXP & 1)) {
          if (e < 3) stage1(cur ^ 1, e);
          else stage2(vnxt, (e - 3) >> 1, (e - 3) & 1);
        }
        if (e >= 9 - NJ && !(EXP & 2)) {               // raw(ph + 2), requested a whole phase ago, into the buffer
          commit_r(rcur, e - (9 - NJ), rp[e - (9 - NJ)]);   // phase ph - 1 finished reading; then the request for raw(ph + 3)
          rp[e - (9 - NJ)] = fetch1(e - (9 - NJ), ph + 3);
        }
        vf = vfn;
        __builtin_amdgcn_sched_barrier(0);
      }
      static_assert(NJ >= 1 && NJ <= 4, "the commit is dealt out over the last NJ steps");
      __syncthreads();
      stamp();
    };
    for (int ph = 0; ph < nph; ph += 2) {
      phase(ph, std::integral_constant<int, 0>{});
      if (ph + 1 < nph) phase(ph + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue.  The 36 xi of a (tile, channel) sit in four waves: the accumulators go through LDS.  Two ROUNDS of
    // 16 output channels per half: each half (waves 0-3 / 4-7 = output channels 0-31 / 32-63) hands two channel quads per
    // lane group to ITS OWN 72 KB buffer ([xi][tile][4 quads, XOR-swizzled][4]: conflict-free 16-byte writes and reads),
    // then finishes them: thread = (tile, quad, output row pair), A^T . A in registers, 64-byte segments out with bias /
    // residual fused.  All eight waves work in both rounds (two per SIMD: one's LDS / store latency under the other's
    // arithmetic); after the first round's hand-over half of the accumulator registers are free for the transform. ----
    const int th = tid & 255;
    const int eq = th & 3, et = (th >> 2) & 31, rpair = __builtin_amdgcn_readfirstlane(th >> 7);
    int e_img, e_ty, e_tx;
    if (CLS == 0) {
      e_img = 0;
      e_ty = et >> 3;
      e_tx = et & 7;
    } else if (CLS == 1) {
      e_img = et >> 4;
      e_ty = (et >> 2) & 3;
      e_tx = et & 3;
    } else {
      e_img = et >> 2;
      e_ty = (et >> 1) & 1;
      e_tx = et & 1;
    }
    const long opix = ((long)(b0 + e_img) * p.H + h0 + 4 * e_ty) * p.W + w0 + 4 * e_tx;
    const bool partial = p.splits > 1;
    // accumulator row i of half HALF = produced channel 32 (i >> 4) + 16 HALF + (i & 15) (the operand's row order, see
    // wino4_weights_kernel): round rd of the two halves covers channels [32 rd, 32 rd + 32) = one 128-byte line per pixel
    const int ncol0 = n0 + HALF * 16 + eq * 4;
    float* const obase = partial ? p.ws + (long)split * p.ws_stride + opix * p.N + ncol0 : p.out + opix * p.out_pitch + ncol0;
    const long opitch = partial ? (long)p.N : p.out_pitch;
    const bool has_res = !partial && p.res != nullptr;           // kernel argument: a scalar branch
    const float* const rbase = p.res + opix * p.res_pitch + ncol0;
    float* const Mh = Mb + HALF * (MBUF / 2);
    const int mrd = (et * 4 + (eq ^ ((et >> 1) & 3))) * 4;
    const int mwr = (lr * 4) * 4;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      if (rd == 1) __syncthreads();                    // round 0's reads are done before its buffer is overwritten
#pragma unroll
      for (int e = 0; e < 9; ++e)
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int g = 2 * rd + gg;
          const f32x4 v = {acc[e][4 * g], acc[e][4 * g + 1], acc[e][4 * g + 2], acc[e][4 * g + 3]};
          *reinterpret_cast<f32x4*>(Mh + (xg * 9 + e) * 512 + mwr + (((2 * gg + lh) ^ ((lr >> 1) & 3)) * 4)) = v;
        }
      __syncthreads();
      // stage 1 (over the xi rows i, per xi column j): this thread's two rows of A^T m
      f32x4 X[2][6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        auto m = [&](int i) -> f32x4 { return *reinterpret_cast<const f32x4*>(Mh + (i * 6 + j) * 512 + mrd); };
        const f32x4 m1 = m(1), m2 = m(2), m3 = m(3), m4 = m(4);
        if (rpair == 0) {
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
        // the 16 lanes of this wave that share the channel quad eq (lane = eq + 4 (tile & 15)): a fixed butterfly
#pragma unroll
        for (int m = 4; m < 64; m <<= 1) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            st1[k] += __shfl_xor(st1[k], m, 64);
            st2[k] += __shfl_xor(st2[k], m, 64);
          }
        }
        if ((lane >> 2) == 0) {
          const int ublk = (bg * p.tb_h + thi) * p.tb_w + twi;
          const int part = (tid >> 6) & 3;                                   // wave within the half: 16 tiles x one row pair
          float* sp = p.stats + ((long)(ublk * 4 + part) * 2) * p.N + ncol0 + rd * 32;
          *reinterpret_cast<f32x4*>(sp) = st1;
          *reinterpret_cast<f32x4*>(sp + p.N) = st2;
        }
      }
      stamp();
    }
    if (DBG && threadIdx.x == 0) {
      p.dbg[blockIdx.x * 32] = nstamp;
      p.dbg[blockIdx.x * 32 + 31] = (long long)__builtin_amdgcn_s_memrealtime();
      if (nstamp <= 29) p.dbg[blockIdx.x * 32 + 30] = rt_entry;       // 100 MHz, one clock for the chip
      if (blockIdx.x == 0) {                            // the launch's shape, in the slot's last row (never a workgroup's)
        long long* m = p.dbg + 4095 * 32;
        m[0] = p.C; m[1] = p.N; m[2] = p.H; m[3] = p.W; m[4] = p.B; m[5] = p.splits; m[6] = CLS; m[7] = gridDim.x;
        m[8] = p.res != nullptr; m[9] = p.tn_slowest;
      }
    }
  };
  if (half == 0) body(std::integral_constant<int, 0>{});
  else body(std::integral_constant<int, 1>{});
}

// ---------------------------------------------------------------------------------------------
// U = G g G^T for 3x3 weight slots of a flat buffer (weights [Np][9][Cp]), stored in the fragment
// order the convolution kernel reads.  table rows: [src offset, Np, Cp, dst_f offset, dst_b offset, first block]
//   forward operand  Uf[n/64][c/8][xi][(n%32)/16][(c%8)/4][16 ((n%64)/32) + n%16][c%4],  g[a][b] = w[n][3a+b][c]
//   input-gradient   Ub[c/64][n/8][xi][(c%32)/16][(n%8)/4][16 ((c%64)/32) + c%16][n%4],  g'[a][b] = w[n][3(2-a)+(2-b)][c]
// ---------------------------------------------------------------------------------------------
// fp32 throughout (fp64 made this kernel compute-bound: 150 us per call on the 64 x 64 configuration, 2 % of its step):
// three roundings per stage on constants that are not dyadic (1/6, 1/12, 1/24) - measured against the float64 definition
// in tests/test_hip_winograd.py (<= 3e-7 of the largest element), an order below the data transforms' own rounding
__device__ __forceinline__ void ggt36(const float (&g)[9], float (&u)[36]) {
  constexpr float k6 = 1.f / 6.f, k12 = 1.f / 12.f, k24 = 1.f / 24.f;
  float t[6][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const float g0 = g[b], g1 = g[3 + b], g2 = g[6 + b];
    const float s02 = g0 + g2, e = __builtin_fmaf(g0, k24, g2 * k6);
    t[0][b] = g0 * 0.25f;
    t[1][b] = -(s02 + g1) * k6;
    t[2][b] = -(s02 - g1) * k6;
    t[3][b] = __builtin_fmaf(g1, k12, e);
    t[4][b] = __builtin_fmaf(g1, -k12, e);
    t[5][b] = g2;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float g0 = t[i][0], g1 = t[i][1], g2 = t[i][2];
    const float s02 = g0 + g2, e = __builtin_fmaf(g0, k24, g2 * k6);
    u[i * 6 + 0] = g0 * 0.25f;
    u[i * 6 + 1] = -(s02 + g1) * k6;
    u[i * 6 + 2] = -(s02 - g1) * k6;
    u[i * 6 + 3] = __builtin_fmaf(g1, k12, e);
    u[i * 6 + 4] = __builtin_fmaf(g1, -k12, e);
    u[i * 6 + 5] = g2;
  }
}

__global__ __launch_bounds__(256) void wino4_weights_kernel(const float* __restrict__ src, float* __restrict__ dst_f,
                                                            float* __restrict__ dst_b, const long* __restrict__ table,
                                                            int n_slots) {
  __shared__ float wt[32 * 9 * 36];               // [n][tap][c (+4 pad)]
  int lo = 0, hi = n_slots - 1;
  const long bid = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid * 6 + 5] <= bid) lo = mid;
    else hi = mid - 1;
  }
  const long* row = table + lo * 6;
  const long soff = row[0];
  const int Np = (int)row[1], Cp = (int)row[2];
  const long lb = bid - row[5];
  const int cblocks = Cp / 32;
  const int nb = (int)(lb / cblocks), cb = (int)(lb % cblocks);
  const float* w = src + soff + ((long)nb * 32 * 9) * Cp + cb * 32;
  const int tid = threadIdx.x;
  for (int i = tid; i < 288 * 8; i += 256) {
    const int r = i >> 3, c4 = (i & 7) * 4;
    *reinterpret_cast<f32x4*>(wt + r * 36 + c4) = *reinterpret_cast<const f32x4*>(w + (long)r * Cp + c4);
  }
  __syncthreads();
  if (dst_f) {      // thread = (n, c quad)
    const int n = tid & 31, cq = tid >> 5;
    f32x4 uu[36];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float g[9], u[36];
#pragma unroll
      for (int t = 0; t < 9; ++t) g[t] = wt[(n * 9 + t) * 36 + cq * 4 + s];
      ggt36(g, u);
#pragma unroll
      for (int x = 0; x < 36; ++x) uu[x][s] = u[x];
    }
    // [(nb >> 1)][ph = cb * 4 + (cq >> 1)][xi][half = n >> 4][cq & 1][row = 16 (nb & 1) + (n & 15)][4]: the produced channels
    // of a 64-block are dealt to the two wave halves in groups of 16 (0-15 | 16-31 | 32-47 | 48-63 -> half 0 | 1 | 0 | 1), so
    // that one epilogue round of both halves completes a full 128-byte line per pixel
    float* d = dst_f + row[3] + (((long)(nb >> 1) * (Cp / 8) + cb * 4 + (cq >> 1)) * NXI) * 512 +
               (((n >> 4) * 2 + (cq & 1)) * 32 + 16 * (nb & 1) + (n & 15)) * 4;
#pragma unroll
    for (int x = 0; x < 36; ++x) *reinterpret_cast<f32x4*>(d + x * 512) = uu[x];
  }
  if (dst_b) {      // thread = (c, n quad)
    const int c = tid & 31, nq = tid >> 5;
    f32x4 uu[36];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float g[9], u[36];
#pragma unroll
      for (int t = 0; t < 9; ++t) g[8 - t] = wt[((nq * 4 + s) * 9 + t) * 36 + c];      // mirrored taps
      ggt36(g, u);
#pragma unroll
      for (int x = 0; x < 36; ++x) uu[x][s] = u[x];
    }
    float* d = dst_b + row[4] + (((long)(cb >> 1) * (Np / 8) + nb * 4 + (nq >> 1)) * NXI) * 512 +
               (((c >> 4) * 2 + (nq & 1)) * 32 + 16 * (cb & 1) + (c & 15)) * 4;
#pragma unroll
    for (int x = 0; x < 36; ++x) *reinterpret_cast<f32x4*>(d + x * 512) = uu[x];
  }
}

static long unit_count(int cls, int B, int H, int W) {     // units per 64 produced channels, before split-K
  return cls == 1 ? B / 2 : cls == 2 ? B / 8 : (long)B * (H / 16) * (W / 32);
}

static int unit_class(int H, int W) {
  if (H == 8 && W == 8) return 2;
  if (H == 16 && W == 16) return 1;
  if (H >= 16 && W >= 32 && H % 16 == 0 && W % 32 == 0) return 0;
  return -1;
}

}  // namespace lgmwino4

// ---- light workgroups (winograd4l.hip): same operands, 16-tile units, 256 threads ----
bool lgm_wino4l_supported(const LgmConvGeom* g, int gather_channels, int out_channels);
long lgm_wino4l_units(const LgmConvGeom* g, int out_channels);
int lgm_wino4l_stats_parts(const LgmConvGeom* g);
int lgm_wino4l_class(const LgmConvGeom* g);
int lgm_wino4l_splits(const LgmConvGeom* g, int gather_channels, int out_channels);
int lgm_wino4l_launch(const LgmConvGeom* g, int yx, const float* a, long a_pitch, const float* u, const float* bias,
                      const float* res, long res_pitch, float* out, long out_pitch, void* workspace, long workspace_bytes,
                      hipStream_t s, int64_t* partial, float* stats);
// LGM_WINO4_LIGHT: 1 = the light workgroups wherever they take the geometry, 0 = the 32-tile workgroups only.  Unset: launch
// by launch (below) unless lgm_wino4_set_light(1) was called - which lgm_hip.lightning.FlatGradSync does for a rank whose
// gradient exchange overlaps its backward pass on RCCL (a resident collective shares the chip: with one foreign
// workgroup resident the step costs +38 % on the 32-tile kernels and +30 % with the light ones, tools/cu_hog_step.py - and the
// per-rank batches are the ones they win at).  Launch by launch (the rule inside wino4_use_light): B = 128 keeps
// the 32-tile workgroups, 10.03 vs 10.13 ms all light; isolated and cache-warm the light kernel is 4-20 % faster at every
// batch, tools/wino4l_bench.py, but inside the step every launch starts cold and two waves per SIMD hide more of that).
static std::atomic<int> lgm_wino4_light_override{-1};   // lgm_wino4_set_light: read by whichever thread issues launches
extern "C" int lgm_wino4_set_light(int mode) {
  lgm_wino4_light_override.store(mode, std::memory_order_relaxed);
  return LGM_OK;
}
static bool wino4_use_light(const LgmConvGeom* g, int gather_channels, int out_channels) {
  // The environment is read ONCE.  No rule reads WORLD_SIZE any more (round 6): being one rank of several does
  // not by itself mean that a collective is resident beside the launches - lgm_hip.lightning.FlatGradSync switches the
  // light workgroups (and the CU margin) on when its exchange really overlaps the backward pass on RCCL.
  static const bool env_given = getenv("LGM_WINO4_LIGHT") != nullptr;
  static const int env_mode = env_given ? atoi(getenv("LGM_WINO4_LIGHT")) : 0;
  const int ov = lgm_wino4_light_override.load(std::memory_order_relaxed);
  const int mode = ov >= 0 ? ov : env_mode;
  if (!lgm_wino4l_supported(g, gather_channels, out_channels)) return false;
  if (mode != 0) return true;
  if (env_given || ov >= 0) return false;   // 0 asked for: 32-tile only
  // One GPU, nothing asked for: the light workgroups for the launches the 32-tile kernel cannot fill the chip with -
  // fewer 32-tile units than the class's bound (LGM_WINO4_LIGHT_BELOW="a,b,c": W % 32 == 0 maps, 16x16, 8x8; 0 = never).
  // Measured per step (tools/step_ab.sh): B = 64 6.94 -> 6.73 ms, B = 32 5.37 -> 5.21; at B = 128 (256 / 128 / 64 units)
  // the 16x16 launches are neutral and the 8x8 ones cost 0.06 ms light, so the bounds sit at or below those counts.
  static long below[3] = {-1, 0, 0};
  if (below[0] < 0) {
    long b0 = 200, b1 = 128, b2 = 64;
    if (const char* e = getenv("LGM_WINO4_LIGHT_BELOW")) {
      b1 = b2 = -1;
      sscanf(e, "%ld,%ld,%ld", &b0, &b1, &b2);
      if (b1 < 0) b1 = b0;
      if (b2 < 0) b2 = b1;
    }
    below[1] = b1, below[2] = b2, below[0] = b0 < 0 ? 0 : b0;
  }
  const int cls = lgm_wino4l_class(g);
  return cls >= 0 && lgm_wino4l_units(g, out_channels) / 2 < below[cls];
}

static bool wino4_big_supported(const LgmConvGeom* g, int gather_channels, int out_channels);
bool lgm_wino4_supported(const LgmConvGeom* g, int gather_channels, int out_channels) {
  return wino4_use_light(g, gather_channels, out_channels) || wino4_big_supported(g, gather_channels, out_channels);
}
static bool wino4_big_supported(const LgmConvGeom* g, int gather_channels, int out_channels) {
  using namespace lgmwino4;
  if (!(g->KH == 3 && g->KW == 3 && g->stride == 1 && g->pad == 1)) return false;
  if (gather_channels % 8 != 0 || out_channels % 64 != 0 || gather_channels % 32 != 0) return false;
  const int cls = unit_class(g->H, g->W);
  if (cls < 0) return false;
  const long pix = (long)g->B * g->H * g->W + g->W + 1;
  if (pix * gather_channels >= (1L << 29) || pix * out_channels >= (1L << 29)) return false;
  if ((long)gather_channels * out_channels * 36 >= (1L << 29)) return false;
  return g->B % (cls == 1 ? 2 : cls == 2 ? 8 : 1) == 0;
}

// Split count of an F(4x4) launch with `base` workgroups per split on `slots` workgroup slots.  A workgroup has the CU to
// itself, so a grid a little above the slot count costs a whole second round: ceil(slots / base) - the former rule - gave
// 256 -> 384 @ 8 x 8 (96 units) three splits = 288 workgroups = two rounds of 11 phases, 80 us, where two splits = 192
// workgroups run ONE round of 16 phases, 52 us.  Cost in phase units (2.45 us): rounds x (fixed part of a workgroup +
// its phases) + what the reducer (or the plane-summing GroupNorm) pays per plane - consulted only when the former rule's
// grid does not fit one round, so the measured plans of DESIGN section 3.5 (128 units -> 2 splits, 64 -> 4) stand.
int lgm_wino4_pick_splits(long base, long slots, int phases, int smax, int min_pps) {
  int s0 = (int)((slots + base - 1) / base);        // the former rule: fill the slots
  if (s0 > smax) s0 = smax;
  while (s0 > 1 && phases / s0 < min_pps) --s0;     // a split shorter than ~4 phases is mostly prologue and epilogue
  if (base * s0 <= slots) return s0;                // one round: the measured plans stand
  int best = s0;
  double best_cost = 1e30;
  for (int s = 1; s <= smax; ++s) {
    if (s > 1 && phases / s < min_pps) break;
    const long rounds = (base * s + slots - 1) / slots;
    const double pps = (double)((phases + s - 1) / s);
    const double cost = (double)rounds * (5.5 + pps) + (s > 1 ? 2.0 + 0.5 * s : 0.0);
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = s;
    }
  }
  return best;
}

int lgm_wino4_splits(const LgmConvGeom* g, int gather_channels, int out_channels) {
  using namespace lgmwino4;
  if (wino4_use_light(g, gather_channels, out_channels)) return lgm_wino4l_splits(g, gather_channels, out_channels);
  const int cls = unit_class(g->H, g->W);
  if (cls < 0) return 1;
  const long base = unit_count(cls, g->B, g->H, g->W) * (out_channels / 64);
  static const int forced = getenv("LGM_WINO4_SPLITS") ? atoi(getenv("LGM_WINO4_SPLITS")) : 0;
  const int phases = gather_channels / KC;
  int smax = phases / 2 < 16 ? phases / 2 : 16;
  if (smax < 1) smax = 1;
  if (forced > 0) return forced < smax ? forced : smax;
  const long slots = lgm_cu_budget();
  if (base >= slots * 3 / 4) return 1;
  static const int min_pps = getenv("LGM_WINO4_MIN_PPS") ? atoi(getenv("LGM_WINO4_MIN_PPS")) : 4;
  static const bool old_rule = getenv("LGM_WINO4_SPLITS_CEIL") != nullptr;      // A/B switch: the former rule
  if (old_rule) {
    int s = (int)((slots + base - 1) / base);
    if (s > smax) s = smax;
    while (s > 1 && phases / s < min_pps) --s;      // a split shorter than ~4 phases is mostly prologue and epilogue
    return s;
  }
  return lgm_wino4_pick_splits(base, slots, phases, smax, min_pps);
}

// partial (optional, int64 x 2): as lgm_wino_launch - the caller's consumer sums the split-K planes itself
int lgm_wino4_launch(const LgmConvGeom* g, int yx, const float* a, long a_pitch, const float* u, const float* bias,
                     const float* res, long res_pitch, float* out, long out_pitch, void* workspace, long workspace_bytes,
                     hipStream_t s, int64_t* partial = nullptr, float* stats = nullptr) {
  using namespace lgmwino4;
  if (wino4_use_light(g, yx ? g->Nw : g->Cw, yx ? g->Cw : g->Nw))
    return lgm_wino4l_launch(g, yx, a, a_pitch, u, bias, res, res_pitch, out, out_pitch, workspace, workspace_bytes, s, partial,
                             stats);
  Args p{};
  p.stats = stats;
  p.a = a; p.u = u; p.bias = bias; p.res = res; p.out = out;
  p.a_pitch = a_pitch; p.res_pitch = res_pitch; p.out_pitch = out_pitch;
  p.B = g->B; p.H = g->H; p.W = g->W;
  p.C = yx ? g->Nw : g->Cw;
  p.N = yx ? g->Cw : g->Nw;
  const int cls = unit_class(g->H, g->W);
  p.tb_h = cls == 0 ? g->H / 16 : 1;
  p.tb_w = cls == 0 ? g->W / 32 : 1;
  p.nbg = cls == 1 ? g->B / 2 : cls == 2 ? g->B / 8 : g->B;
  p.tiles_n = p.N / 64;
  const long M = (long)g->B * g->H * g->W;
  p.splits = lgm_wino4_splits(g, p.C, p.N);
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
  p.dbg = (long long*)lgm_wino4_debug_buffer;
  if (p.dbg && lgm_wino4_debug_slots > 0) {
    if (lgm_wino4_debug_next >= lgm_wino4_debug_slots || p.units > 4095) p.dbg = nullptr;     // out of slots: the plain build
    else p.dbg += (long)(lgm_wino4_debug_next++) * 4096 * 32;
  }
#define LGM_W4LAUNCH(CC, DD, EE)                                                                                \
  do {                                                                                                          \
    auto kern = wino4_conv_kernel<CC, DD, EE>;                                                                        \
    static bool attr = false;                                                                                   \
    if (!attr) {                                                                                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)smem);                                                                     \
      attr = true;                                                                                              \
    }                                                                                                           \
    hipLaunchKernelGGL(kern, dim3((unsigned)p.units), dim3(512), smem, s, p);                                   \
  } while (0)
  lgm_note_kernel(cls == 0 ? LGM_KNAME("lgmwino4::wino4_conv_kernel<0, false, 0, false>") : cls == 1 ? LGM_KNAME("lgmwino4::wino4_conv_kernel<1, false, 0, false>")
                                                                                    : LGM_KNAME("lgmwino4::wino4_conv_kernel<2, false, 0, false>"));
  if (p.dbg) {
    const int e = lgm_wino4_debug_exp;
    if (cls == 1) LGM_W4LAUNCH(1, true, 0);
    else if (cls == 2) LGM_W4LAUNCH(2, true, 0);
    else if (e == 0) LGM_W4LAUNCH(0, true, 0);
    else if (e == 1) LGM_W4LAUNCH(0, true, 1);
    else if (e == 2) LGM_W4LAUNCH(0, true, 2);
    else if (e == 3) LGM_W4LAUNCH(0, true, 3);
    else if (e == 4) LGM_W4LAUNCH(0, true, 4);
    else if (e == 7) LGM_W4LAUNCH(0, true, 7);
    else if (e == 8) LGM_W4LAUNCH(0, true, 8);
    else LGM_W4LAUNCH(0, true, 15);
  } else if (stats) {
    if (cls != 0 || p.splits != 1 || res || partial) {
      lgm_set_error("wino4 (stats): class-0 maps, an unsplit reduction and no residual expected");
      return LGM_ERR_UNSUPPORTED;
    }
    auto kern = wino4_conv_kernel<0, false, 0, true>;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr = true;
    }
    lgm_note_kernel(LGM_KNAME("lgmwino4::wino4_conv_kernel<0, false, 0, true>"));
    hipLaunchKernelGGL(kern, dim3((unsigned)p.units), dim3(512), smem, s, p);
  } else if (cls == 0) LGM_W4LAUNCH(0, false, 0);
  else if (cls == 1) LGM_W4LAUNCH(1, false, 0);
  else LGM_W4LAUNCH(2, false, 0);
#undef LGM_W4LAUNCH
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

// ---- C-ABI ------------------------------------------------------------------------------------
extern "C" int64_t lgm_conv3x3_wino4_supported(const LgmConvGeom* g, int yx) {
  if (!g) return 0;
  return lgm_wino4_supported(g, yx ? g->Nw : g->Cw, yx ? g->Cw : g->Nw) ? 1 : 0;
}

// Forward convolution that also leaves the GroupNorm statistics' raw material (lgm_conv3x3_wino4_stats): floats of the
// `stats` buffer it needs, or 0 when this geometry does not take it (class-0 maps, unsplit reduction, preferred at all);
// *parts_per_image = the partial (sum, sum of squares) rows every image contributes per channel.
extern "C" int64_t lgm_conv3x3_wino4_stats_floats(const LgmConvGeom* g, int* parts_per_image) {
  if (parts_per_image) *parts_per_image = 0;
  if (!g || !lgm_wino4_supported(g, g->Cw, g->Nw)) return 0;
  const bool light = wino4_use_light(g, g->Cw, g->Nw);
  if (light ? lgm_wino4l_stats_parts(g) == 0 : lgmwino4::unit_class(g->H, g->W) != 0) return 0;
  if (lgm_wino4_splits(g, g->Cw, g->Nw) != 1) return 0;
  static const bool off = getenv("LGM_NO_GN_EPI_STATS") != nullptr;      // A/B switch
  if (off) return 0;
  const int per = light ? lgm_wino4l_stats_parts(g) : (g->H / 16) * (g->W / 32) * 4;
  if (parts_per_image) *parts_per_image = per;
  return (int64_t)g->B * per * 2 * g->Nw;
}

extern "C" int64_t lgm_conv3x3_wino4_workspace(const LgmConvGeom* g, int yx) {
  if (!g) return -1;
  const int gc = yx ? g->Nw : g->Cw, oc = yx ? g->Cw : g->Nw;
  if (!lgm_wino4_supported(g, gc, oc)) return 0;
  const int s = lgm_wino4_splits(g, gc, oc);
  return s > 1 ? (int64_t)s * g->B * g->H * g->W * oc * (int64_t)sizeof(float) : 0;
}

// 1 when the F(4x4) kernel is expected to beat the F(2x2) kernel on this layer (measured on the MI355X, B = 128,
// tools/wino4_bench.py): enough units to fill the chip without splitting the reduction more than twice
extern "C" int64_t lgm_conv3x3_wino4_preferred(const LgmConvGeom* g, int yx) {
  if (!g) return 0;
  const int gc = yx ? g->Nw : g->Cw, oc = yx ? g->Cw : g->Nw;
  if (!lgm_wino4_supported(g, gc, oc)) return 0;
  static const int force = getenv("LGM_WINO4_FORCE") ? atoi(getenv("LGM_WINO4_FORCE")) : 0;
  if (force) return 1;
  // maps >= 16 x 32: from 128 units (B = 64 at 32 x 32: the reduction is split in two, 4 phases per workgroup: 7.28 vs
  // 7.35 ms per step against the F(2x2) pair on its joint plan; without the split 7.44); 256 units: 10.9 vs 11.4 ms;
  // 16 x 16 maps (more phases per unit) likewise from 128 units
  // with the light workgroups (16-tile units: twice the workgroups, one wave per SIMD each) already from 32 such units -
  // the per-rank batches 32 and 16 at 32 x 32: 5.35 -> 5.27 and 4.54 -> 4.52 ms per step, nothing changes at 64 and 128
  static const long min_units0_env = getenv("LGM_WINO4_MIN_UNITS") ? atol(getenv("LGM_WINO4_MIN_UNITS")) : -1;
  static const long min_units1_env = getenv("LGM_WINO4_MIN_UNITS1") ? atol(getenv("LGM_WINO4_MIN_UNITS1")) : -1;
  const long min_units0 = min_units0_env >= 0 ? min_units0_env : (wino4_use_light(g, gc, oc) ? 32 : 128);
  const long min_units1 = min_units1_env >= 0 ? min_units1_env : (wino4_use_light(g, gc, oc) ? 32 : 128);
  const bool light = wino4_use_light(g, gc, oc);
  const int cls = g->H == 8 && g->W == 8 ? 2 : g->H == 16 && g->W == 16 ? 1 : 0;
  // counted in 32-tile units whichever workgroup size runs: the thresholds keep their meaning
  const long base = light ? lgm_wino4l_units(g, oc) / 2 : lgmwino4::unit_count(cls, g->B, g->H, g->W) * (oc / 64);
  // forward with the light workgroups: LGM_WINO4_FWD_ALL=1 takes every layer they support (tuning knob)
  static const int fwd_all = getenv("LGM_WINO4_FWD_ALL") ? atoi(getenv("LGM_WINO4_FWD_ALL")) : 0;
  if (light && yx == 0 && fwd_all) return 1;
  static const int bwd_all = getenv("LGM_WINO4_BWD_ALL") ? atoi(getenv("LGM_WINO4_BWD_ALL")) : 0;
  if (light && yx == 1 && bwd_all) return 1;
  // 8 x 8 maps: few units (8 images each), so the reduction is split; worth it when a split still has >= 8 phases
  static const bool no8 = getenv("LGM_WINO4_NO8") != nullptr;           // A/B switch
  // Forward and backward: the heavy 8 x 8 layers leave the F(2x2) pair (input gradient + weight gradient in one launch) for
  // the F(4x4) input gradient + the F(4x4) weight gradient on 2 x 2-tile groups (winograd4_wgrad.hip), two layers per
  // launch: 10.28 -> 10.16 ms per step.  (With the F(2x2) weight gradient left standing alone the same move cost what it
  // gained: 10.97 vs 10.95, later 10.66 vs 10.66.)  LGM_WINO4_NOYX8=1: forward only.
  static const bool noyx8 = getenv("LGM_WINO4_NOYX8") != nullptr;
  static const int c2_minc = getenv("LGM_WINO4_C2_MINC") ? atoi(getenv("LGM_WINO4_C2_MINC")) : 256;
  if (cls == 2) return (!no8 && (yx == 0 || !noyx8) && base >= 32 && gc >= c2_minc) ? 1 : 0;
  static const int c1_minc = getenv("LGM_WINO4_C1_MINC") ? atoi(getenv("LGM_WINO4_C1_MINC")) : 0;
  if (cls == 1 && gc < c1_minc) return 0;
  // 16 x 16 maps with a reduction of >= 128 channels (>= 16 phases: two splits of >= 8) already from 64 units - the
  // per-rank batch of two GPUs: 6.98 -> 6.94 ms per step at B = 64, nothing changes at B = 128 (LGM_WINO4_MIN_UNITS1W)
  static const long min_units1w_env = getenv("LGM_WINO4_MIN_UNITS1W") ? atol(getenv("LGM_WINO4_MIN_UNITS1W")) : -1;
  const long min_units1w = min_units1w_env >= 0 ? min_units1w_env : (light ? 32 : 64);
  if (cls == 1 && gc >= 128 && base >= min_units1w) return 1;
  return base >= (cls == 1 ? min_units1 : min_units0) ? 1 : 0;
}

extern "C" int lgm_wino4_weights(const float* src, float* dst_f, float* dst_b, const int64_t* table, int n_slots,
                                 int64_t total_blocks, void* stream) {
  LGM_REQUIRE(src && table && n_slots > 0 && total_blocks > 0 && (dst_f || dst_b), "wino4_weights: bad arguments");
  hipLaunchKernelGGL(lgmwino4::wino4_weights_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     src, dst_f, dst_b, (const long*)table, n_slots);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_conv3x3_wino4(int yx, const LgmConvGeom* g, const float* a, int64_t a_pitch, const float* u,
                                 const float* bias, const float* res, int64_t res_pitch, float* out, int64_t out_pitch,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
  LGM_REQUIRE(g && a && u && out, "conv3x3_wino4: null pointer");
  const int gc = yx ? g->Nw : g->Cw, oc = yx ? g->Cw : g->Nw;
  LGM_REQUIRE(lgm_wino4_supported(g, gc, oc), "conv3x3_wino4: unsupported geometry (3x3/s1/p1, maps 16x16 or H %% 16 == 0 "
              "and W %% 32 == 0, reduction channels %% 32, produced channels %% 64)");
  LGM_REQUIRE(a_pitch % 4 == 0 && a_pitch >= gc && lgm_aligned16(a) && lgm_aligned16(u) && lgm_aligned16(out) &&
              out_pitch % 4 == 0 && out_pitch >= oc && (!res || (lgm_aligned16(res) && res_pitch % 4 == 0 && res_pitch >= oc)) &&
              (!bias || lgm_aligned16(bias)), "conv3x3_wino4: 16-byte aligned operands with pitch %% 4 == 0 expected");
  const long pix = (long)g->B * g->H * g->W + g->W + 1;
  LGM_REQUIRE(pix * a_pitch < (1L << 29) && pix * out_pitch < (1L << 29) && pix * (res ? res_pitch : 0) < (1L << 29),
              "conv3x3_wino4: tensor too large for 32-bit offsets");
  return lgm_wino4_launch(g, yx, a, a_pitch, u, bias, res, res_pitch, out, out_pitch, workspace, workspace_bytes,
                          (hipStream_t)stream);
}

extern "C" int lgm_conv3x3_wino4_stats(const LgmConvGeom* g, const float* x, int64_t x_pitch, const float* u,
                                       const float* bias, float* y, int64_t y_pitch, float* stats, int64_t stats_floats,
                                       void* stream) {
  LGM_REQUIRE(g && x && u && y && stats, "conv3x3_wino4_stats: null pointer");
  int per = 0;
  const int64_t need = lgm_conv3x3_wino4_stats_floats(g, &per);
  // == and not >=: the row count per image depends on the workgroup form (light: 8-row units, 32-tile: 16-row units).  A
  // caller that sized its buffer - and will tell lgm_gn_fwd_stats its rows per image - under another kernel selection would
  // otherwise read rows this launch never writes.
  LGM_REQUIRE(need > 0 && stats_floats == need && lgm_aligned16(stats) && g->Nw % 4 == 0,
              "conv3x3_wino4_stats: geometry not taken, or the statistics buffer (%ld floats) is not what THIS kernel selection "
              "writes (%ld: ask lgm_conv3x3_wino4_stats_floats again after lgm_wino4_set_light / lgm_set_cu_margin)",
              (long)stats_floats, (long)need);
  LGM_REQUIRE(x_pitch % 4 == 0 && x_pitch >= g->Cw && lgm_aligned16(x) && lgm_aligned16(u) && lgm_aligned16(y) &&
              y_pitch % 4 == 0 && y_pitch >= g->Nw && (!bias || lgm_aligned16(bias)),
              "conv3x3_wino4_stats: 16-byte aligned operands with pitch %% 4 == 0 expected");
  const long pix = (long)g->B * g->H * g->W + g->W + 1;
  LGM_REQUIRE(pix * x_pitch < (1L << 29) && pix * y_pitch < (1L << 29), "conv3x3_wino4_stats: tensor too large for 32-bit offsets");
  return lgm_wino4_launch(g, 0, x, x_pitch, u, bias, nullptr, 0, y, y_pitch, nullptr, 0, (hipStream_t)stream, nullptr, stats);
}

extern "C" int lgm_conv3x3_wino4_partial(int yx, const LgmConvGeom* g, const float* a, int64_t a_pitch, const float* u,
                                         const float* bias, float* out, int64_t out_pitch, void* workspace,
                                         int64_t workspace_bytes, int64_t* partial, void* stream) {
  LGM_REQUIRE(g && a && u && out && partial, "conv3x3_wino4_partial: null pointer");
  const int gc = yx ? g->Nw : g->Cw, oc = yx ? g->Cw : g->Nw;
  LGM_REQUIRE(lgm_wino4_supported(g, gc, oc), "conv3x3_wino4_partial: unsupported geometry");
  LGM_REQUIRE(a_pitch % 4 == 0 && a_pitch >= gc && lgm_aligned16(a) && lgm_aligned16(u) && lgm_aligned16(out) &&
              out_pitch % 4 == 0 && out_pitch >= oc && (!bias || lgm_aligned16(bias)),
              "conv3x3_wino4_partial: 16-byte aligned operands with pitch %% 4 == 0 expected");
  const long pix = (long)g->B * g->H * g->W + g->W + 1;
  LGM_REQUIRE(pix * a_pitch < (1L << 29) && pix * out_pitch < (1L << 29), "conv3x3_wino4_partial: tensor too large for 32-bit offsets");
  return lgm_wino4_launch(g, yx, a, a_pitch, u, bias, nullptr, 0, out, out_pitch, workspace, workspace_bytes,
                          (hipStream_t)stream, partial);
}
