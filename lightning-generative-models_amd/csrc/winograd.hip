// Winograd F(2x2, 3x3) convolution kernels in exact fp32 arithmetic on v_mfma_f32_32x32x2_f32 — the
// 3x3 / stride 1 / pad 1 layers of the DDPM UNet (reference Block.proj ddpm.py:160-171, Upsample's and
// the last stages' 3x3 convolutions :93-97,377,413) and everything autograd derives from them.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 4x4 input tile d -> 2x2 output tile Y
//
// turns the 9-tap convolution into 16 independent GEMMs  M[xi] = V[xi] (tiles x Cin) * U[xi] (Cin x Cout):
// 16 multiplies per 4 outputs instead of 36, i.e. 2.25x fewer MFMA FLOPs than the direct form, for a few
// fp32 additions per element in the input / output transforms (all constants are 0, +-1, +-1/2: the
// transforms are exact up to fp32 rounding of sums; measured error vs fp64 is the same order as the
// direct fp32 MFMA kernel's, see tests/test_hip_winograd.py).
//
//   wino_weights_kernel    U = G g G^T for every 3x3 weight slot of a flat parameter buffer, in ONE launch,
//                          in the fragment order the convolution kernel reads; forward operand
//                          (K = Cin, N = Cout) and input-gradient operand (K = Cout, N = Cin, taps mirrored)
//   wino_conv_kernel       forward / input gradient:  out[pix][n] = sum_k conv3x3(a)[pix][k->n] (+bias +res)
//
// Work decomposition of wino_conv_kernel: a UNIT = 64 tiles (8x8 tiles of one image = 16x16 output
// pixels; 4x4 tiles of 4 images at 8x8 maps; 2x2 tiles of 16 images at 4x4 maps) x 64 output channels
// x one split of the reduction; a unit is a sequence of PHASES of 8 reduction channels.  Per phase:
//   * the raw halo patch of the phase after next is fetched from global memory into registers (raw
//     buffer loads, out-of-range offsets return the zero padding), the patch of the next phase is
//     committed to LDS, and the 64 x 8 x (4x4) input transform of the next phase runs on the VALU
//     (thread = tile x channel quad x two transform rows), writing V[xi][k-half][tile][4] to LDS;
//   * each wave (32 tiles x 32 output channels, sixteen 32x32 accumulators = 256 AGPRs) issues
//     16 xi x 4 MFMAs: A fragments are one conflict-free ds_read_b128 per xi, B fragments (U) come
//     straight from global memory (L2) as one 16-byte load per xi, eight steps ahead in a register ring;
//   * one barrier per phase.  One persistent workgroup per CU walks a contiguous unit range.
// Epilogue per unit: the 4x4 -> 2x2 output transform is lane-local in the accumulator layout (every lane
// holds all 16 xi of its (tile, channel) pairs); the four output positions leave through a wave-private
// LDS transpose as 16-byte stores with bias / residual fused.  Split-K writes partial OUTPUTS (the output
// transform is linear) that the existing fixed-order reducer sums.
#include "lgm_common.h"

int lgm_splitk_reduce_launch(const float* ws, long ws_stride, int splits, const float* bias, const float* res,
                             long res_pitch, float* out, long out_pitch, long M, int N, hipStream_t s);

// diagnostic: when set (lgm_wino_set_debug_buffer), the convolution runs its stamped build and writes, per
// workgroup, 64 int64: [0] stamp count, [1] s_memrealtime at exit, [2..] s_memtime stamps (start, after the
// prologue, after every phase, after every epilogue).  Never set on the product path.
static void* lgm_wino_debug_buffer = nullptr;
static int lgm_wino_debug_mode = 0;
extern "C" int lgm_wino_set_debug_buffer(void* buf, int mode) {
  lgm_wino_debug_buffer = buf;
  lgm_wino_debug_mode = mode;
  return LGM_OK;
}

namespace lgmwino {
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Packed fp32 add / subtract on 16-byte values.  Beside fp32 MFMAs every VALU instruction costs ~7 cycles of MFMA
// issue whatever it computes (a v_pk_add_f32 ~9), so the transforms use the packed forms throughout; hipcc emits
// v_pk_add_f32 for vector additions but four v_sub_f32 for a vector subtraction (the neg_lo / neg_hi modifiers are
// not selected), hence the asm.  Plain VALU: no wait states are owed around it (VALU -> VALU is interlocked; the
// fragments these feed are consumed by MFMAs at least one fenced step later).
__device__ __forceinline__ f32x4 sub4(const f32x4 a, const f32x4 b) {
#ifdef LGM_WINO_NO_ASM
  return a - b;
#endif
  f32x2 lo, hi;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]"
      : "=v"(lo)
      : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(__builtin_shufflevector(b, b, 0, 1)));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]"
      : "=v"(hi)
      : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(__builtin_shufflevector(b, b, 2, 3)));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x2 sub2(const f32x2 a, const f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 add2(const f32x2 a, const f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x4 add4(const f32x4 a, const f32x4 b) {
#ifdef LGM_WINO_NO_ASM
  return a + b;
#endif
  f32x2 lo, hi;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(__builtin_shufflevector(b, b, 0, 1)));
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(__builtin_shufflevector(b, b, 2, 3)));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

constexpr int KC = 8;            // reduction channels per phase
constexpr int TB = 64;           // tiles per unit
constexpr int NWQ = 16;          // weight-fragment ring: xi steps of look-ahead (power of two, <= 16)
constexpr int VBUF = 16 * 2 * TB * 4;   // floats per V buffer: [xi][k-half][tile][4]

struct Args {
  const float* a;      // gathered activations, NHWC
  const float* u;      // transformed weights [N/32][C/8][16][2][32][4]
  const float* bias;
  const float* res;
  float* out;
  long a_pitch, res_pitch, out_pitch;
  int B, H, W;
  int C;               // reduction channels
  int N;               // produced channels
  int tb_h, tb_w, tiles_n;
  int units, per, splits, pps;
  float* ws;
  long ws_stride;
  int dbg_mode;        // 0: stamps per phase / epilogue, 1: per double step, 2: before and after every barrier
  long long* dbg;      // diagnostic build only (wino_conv_kernel<true>): per-workgroup cycle stamps
};

struct Phase {
  int L, cc, cc_end;
  int tn, split, twi, thi, bg;
  int n0, b0, h0, w0;
  unsigned border;
  long abase;
  bool valid;
};

__device__ __forceinline__ int xcd_swizzle(int bid, int nb) {
  return (nb % 8 == 0) ? (bid % 8) * (nb / 8) + bid / 8 : bid;
}

// tile of accumulator row / A-fragment lane l (0..31): the 16 lanes that one ds_read_b128 services together
// ({0-3,12-15,20-27} and {4-11,16-19,28-31}, MI355X LDS) get 16 CONSECUTIVE tiles, which the raw patch layout
// (column-parity planes, padded row stride) spreads over all 64 banks.  Any bijection works for the GEMM;
// the epilogue uses the same one.
__device__ __forceinline__ int tile_of_lane(int l) {
  return l < 4 ? l : l < 12 ? l + 12 : l < 16 ? l - 8 : l < 20 ? l + 8 : l < 28 ? l - 12 : l;
}

// EXP (diagnostic builds only): bit 1 drops the raw reads, bit 2 the weight-ring refills, bit 3 the raw commit /
// fetch of the phase loop -- wrong results, used to attribute the phase time.
// G = tiles per unit row (8: 8x8 tiles of one image; 4: 4x4 tiles of 4 images; 2: 2x2 tiles of 16 images): every
// LDS offset of the phase loop is then an instruction immediate (beside fp32 MFMAs EVERY instruction of the wave
// costs ~7 cycles of MFMA issue -- measured with this kernel's diagnostic builds -- so address arithmetic is not free).
template <int G>
struct Geo {
  static constexpr int TTW = G, TTH = G, NI = 64 / (G * G);
  static constexpr int lgTTW = G == 8 ? 3 : G == 4 ? 2 : 1, lgTT = 2 * lgTTW;
  static constexpr int PH = 2 * G + 2, PWr = 2 * G + 2;
  // row stride of a column-parity plane, in 16-byte positions: 2 * PWh = 8 (mod 16) makes the 16 lanes of one
  // ds_read_b128 group (two tile rows of 8 tiles) hit all 64 banks; 2 * PWh = 4 (mod 16) does it for the
  // 4 x 4-tile images of the 8 x 8 maps (four tile rows of 4 tiles)
  static constexpr int PWh = G == 8 ? 12 : G == 4 ? 10 : 3;
  static constexpr int NP = NI * PH * PWr;
  static constexpr int RPLANE = NI * PH * PWh * 4 + 16;
  static constexpr int NJ = (2 * NP + 255) / 256;
};

template <int G, bool DBG, int EXP = 0>
__global__ __launch_bounds__(256, 1) void wino_conv_kernel(const Args p) {
  using GE = Geo<G>;
  constexpr int TTW = GE::TTW, TTH = GE::TTH, NI = GE::NI, lgTTW = GE::lgTTW, lgTT = GE::lgTT, PH = GE::PH,
                PWr = GE::PWr, PWh = GE::PWh, NP = GE::NP, RPLANE = GE::RPLANE, NJ = GE::NJ;
  extern __shared__ __align__(16) float smem[];
  int nstamp = 0;
  auto stamp = [&]() {
    if (DBG) {
      if (threadIdx.x == 0 && nstamp < 62) p.dbg[blockIdx.x * 64 + 2 + nstamp] = (long long)__builtin_amdgcn_s_memtime();
      ++nstamp;
    }
  };
  stamp();
  float* Rb = smem;                              // 3 buffers x 4 planes ([k-half][column parity]) x RPLANE
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wtb = wid & 1, wcb = wid >> 1;       // MFMA role: tile half, output-channel half
  const int lr = lane & 31, lh = lane >> 5;

  const int Lb = xcd_swizzle(blockIdx.x, gridDim.x);
  const int L0 = Lb * p.per;
  const int L1 = min(p.units, L0 + p.per);
  if (L0 >= L1) return;
  const int ncc_total = p.C / KC;

  auto place = [&](Phase& ph) {
    ph.n0 = ph.tn * 64;
    ph.w0 = ph.twi * 2 * TTW;
    ph.h0 = ph.thi * 2 * TTH;
    ph.b0 = ph.bg * NI;
    ph.cc = ph.split * p.pps;
    ph.cc_end = min(ncc_total, ph.cc + p.pps);
    ph.border = 16u | (ph.h0 == 0 ? 1u : 0u) | (ph.h0 + 2 * TTH == p.H ? 2u : 0u) | (ph.w0 == 0 ? 4u : 0u) |
                (ph.w0 + 2 * TTW == p.W ? 8u : 0u);
    ph.abase = ((long)((ph.b0 * p.H + ph.h0) * p.W + ph.w0) * p.a_pitch) * 4;
  };
  auto decode = [&](Phase& ph, int L) {
    ph.L = L;
    ph.tn = L % p.tiles_n;
    int ts = L / p.tiles_n;
    ph.split = ts % p.splits;
    ts /= p.splits;
    ph.twi = ts % p.tb_w;
    ts /= p.tb_w;
    ph.thi = ts % p.tb_h;
    ph.bg = ts / p.tb_h;
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
          if (++ph.twi == p.tb_w) {
            ph.twi = 0;
            if (++ph.thi == p.tb_h) {
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

  // ---- raw patch bookkeeping: slot s = tid + 256 j -> position s >> 1, channel quad s & 1 ----
  // LDS image of a patch: plane (quad, column parity), [img][row][column >> 1][4 channels], row stride PWh.
  unsigned pdelta[NJ], plds[NJ];
  unsigned pflag = 0;                             // 5 bits per slot
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int s = tid + 256 * j;
    const int pos = s >> 1, q = s & 1;
    // a slot that does not exist fetches "nothing" (out-of-range offset) and commits it to the pad floats
    // at the end of a plane: no predication anywhere in the phase body
    unsigned d = 0, f = 16u, l = (unsigned)(q * 2 * RPLANE + NI * PH * PWh * 4);
    if (pos < NP) {
      const int img = pos / (PH * PWr);
      const int rem = pos - img * (PH * PWr);
      const int py = rem / PWr, px = rem - py * PWr;
      d = (unsigned)(((img * p.H + py) * p.W + px) * (int)p.a_pitch + q * 4) * 4u;
      f = (py == 0 ? 1u : 0u) | (py == PH - 1 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == PWr - 1 ? 8u : 0u);
      l = (unsigned)((q * 2 + (px & 1)) * RPLANE + ((img * PH + py) * PWh + (px >> 1)) * 4);
    }
    pdelta[j] = d;
    plds[j] = l;
    pflag |= f << (5 * j);
  }
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
  // per-slot fetch offsets of the unit a phase iterator stands on (zero padding = an offset past the descriptor's
  // range): recomputed when the iterator moves to another unit, not per phase
  auto slot_offsets = [&](const Phase& ph, unsigned (&po)[NJ]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const unsigned fl = (pflag >> (5 * j)) & ph.border & 31u;
      po[j] = fl == 0u ? pdelta[j] : nrec_a;
    }
  };
  auto fetch_at = [&](int j, const Phase& ph, const unsigned (&po)[NJ]) {
    const unsigned soff = (unsigned)(ph.abase + (long)ph.cc * (KC * 4));
    rp[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, po[j], soff, 0);
  };
  auto commit = [&](int j, float* rbuf) { *reinterpret_cast<u32x4*>(rbuf + plds[j]) = rp[j]; };

  // ---- input transform: every lane builds its OWN A fragments (tile = accumulator row lr, channel quad lh) in
  // registers straight from the raw patch -- the transformed tiles never go through LDS.  Raw element d[r][c] of
  // the lane's 4x4 input tile: plane lh*2 + (c & 1), row 2 ty + r, half-column tx + (c >> 1).
  int trd;
  {
    const int t = wtb * 32 + tile_of_lane(lr);
    const int img = t >> lgTT, rr = t & ((1 << lgTT) - 1);
    const int ty = rr >> lgTTW, tx = rr & (TTW - 1);
    trd = lh * 2 * RPLANE + ((img * PH + 2 * ty) * PWh + tx) * 4;
  }
  auto rd = [&](const float* rbuf, int r, int c) -> f32x4 {
    if (EXP & 2) return f32x4{1.f, 2.f, (float)r, (float)c};
    return *reinterpret_cast<const f32x4*>(rbuf + trd + (c & 1) * RPLANE + (r * PWh + (c >> 1)) * 4);
  };
  // rows (2 hrow, 2 hrow + 1) of V = B^T d B for this lane's tile: 8 fragments (xi = 8 hrow + 4 i + col)
  //   B^T d:  row0 = d0 - d2, row1 = d1 + d2, row2 = d2 - d1, row3 = d1 - d3;  then the same combination of columns
  f32x4 Af[2][8];
  f32x4 tq0[4], tq1[4], ec[4][3];
  auto tr_read = [&](const float* rbuf, int hrow, int c, int k) {       // raw rows hrow .. hrow + 2 of column c
#pragma unroll
    for (int r = 0; r < 3; ++r) ec[k][r] = rd(rbuf, hrow + r, c);
  };
  auto tr_rows = [&](int hrow, int c, int k) {
    if (hrow == 0) {
      tq0[c] = sub4(ec[k][0], ec[k][2]);
      tq1[c] = add4(ec[k][1], ec[k][2]);
    } else {
      tq0[c] = sub4(ec[k][1], ec[k][0]);
      tq1[c] = sub4(ec[k][0], ec[k][2]);
    }
  };
  auto tr_out = [&](int hrow, int o) {            // o = 0..7: row o >> 2 of the pair, column o & 3
    const f32x4* t = (o >> 2) ? tq1 : tq0;
    f32x4 v;
    if ((o & 3) == 0) v = sub4(t[0], t[2]);
    else if ((o & 3) == 1) v = add4(t[1], t[2]);
    else if ((o & 3) == 2) v = sub4(t[2], t[1]);
    else v = sub4(t[1], t[3]);
    Af[hrow][o] = v;
  };

  // ---- B fragments: U[n/32][c/8][xi][k-half][n%32][4]; lane (lr, lh) reads 16 bytes per xi.  Buffer loads:
  // constant per-lane offset + wave-uniform (SALU) block offset, no 64-bit VALU address arithmetic per load.
  const unsigned wlane = (unsigned)((lh * 32 + lr) * 4) * 4u;
  const int wcb_s = __builtin_amdgcn_readfirstlane(wcb);
  __amdgpu_buffer_rsrc_t rsrc_u;
  {
    const unsigned long long ub = reinterpret_cast<unsigned long long>(p.u);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ub);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(ub >> 32));
    rsrc_u = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                               __builtin_amdgcn_readfirstlane((unsigned)((long)p.N * p.C * 16 * 4)),
                                               0x00020000);
  }
  auto load_b = [&](const Phase& ph, int xi) -> f32x4 {
    const unsigned soff = (unsigned)((((ph.n0 >> 5) + wcb_s) * ncc_total + ph.cc) * 16 + xi) * 1024u;
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, wlane, soff, 0);
    return __builtin_bit_cast(f32x4, v);
  };

  // ---- epilogue addressing (kernel constants): byte offset of (this lane's tile, first channel of its register group 0)
  // relative to the unit's first pixel / first channel, for the output, the split-K partial buffer and the residual
  unsigned evoff, evoff_w, evoff_r;
  {
    const int t = wtb * 32 + tile_of_lane(lr);               // accumulator column = this lane's tile
    const int img = t >> lgTT, rr = t & ((1 << lgTT) - 1);
    const int ty = rr >> lgTTW, tx = rr & (TTW - 1);
    const long pix = ((long)img * p.H + 2 * ty) * p.W + 2 * tx;
    const int ch = wcb * 32 + 4 * lh;                        // first channel of register group 0
    evoff = (unsigned)((pix * p.out_pitch + ch) * 4);
    evoff_w = (unsigned)((pix * p.N + ch) * 4);
    evoff_r = (unsigned)((pix * p.res_pitch + ch) * 4);
  }
  auto make_rsrc = [&](const void* ptr, long bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane((unsigned)bytes), 0x00020000);
  };
  const long npix = (long)p.B * p.H * p.W;
  const __amdgpu_buffer_rsrc_t rsrc_o = make_rsrc(p.out, npix * p.out_pitch * 4);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(p.ws ? (const void*)p.ws : (const void*)p.out,
                                                  p.ws ? (long)p.splits * p.ws_stride * 4 : 4);
  const __amdgpu_buffer_rsrc_t rsrc_r = make_rsrc(p.res ? (const void*)p.res : (const void*)p.out,
                                                  p.res ? npix * p.res_pitch * 4 : 4);

  // ---- prologue ----
  Phase cur;
  cur.valid = true;
  decode(cur, L0);
  Phase nx1 = cur;
  advance(nx1);
  Phase nx2 = nx1;
  advance(nx2);
  Phase nx3 = nx2;
  advance(nx3);
  f32x4 wq[NWQ];
#pragma unroll
  for (int x = 0; x < NWQ; ++x) wq[x] = x < 16 ? load_b(cur, x) : load_b(nx1, x - 16);
  unsigned poff[NJ];
  {
    u32x4 r0[NJ];
    slot_offsets(cur, poff);
#pragma unroll
    for (int j = 0; j < NJ; ++j) fetch_at(j, cur, poff);
#pragma unroll
    for (int j = 0; j < NJ; ++j) r0[j] = rp[j];
    slot_offsets(nx1, poff);
#pragma unroll
    for (int j = 0; j < NJ; ++j) fetch_at(j, nx1, poff);
#pragma unroll
    for (int j = 0; j < NJ; ++j) *reinterpret_cast<u32x4*>(Rb + plds[j]) = r0[j];      // raw[cur] -> buffer 0
#pragma unroll
    for (int j = 0; j < NJ; ++j) commit(j, Rb + 4 * RPLANE);                            // raw[nx1] -> buffer 1
    slot_offsets(nx2, poff);
#pragma unroll
    for (int j = 0; j < NJ; ++j) fetch_at(j, nx2, poff);                                // raw[nx2] stays in registers
    slot_offsets(nx3, poff);
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    tr_read(Rb, 0, c, c);
    tr_rows(0, c, c);
  }
#pragma unroll
  for (int o = 0; o < 8; ++o) tr_out(0, o);

  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 dummy2[2] = {f32x2{0.f, 1.f}, f32x2{2.f, 3.f}};
  stamp();
  int rb0 = 0;                                     // raw ring: buffer rb0 = cur, rb0 + 1 = nx1, rb0 + 2 = commit target
  bool more = true;
  while (more) {
   // One iteration = one UNIT: the accumulators live only here (zeroed at the top, consumed by the epilogue
   // at the bottom), so that they stay in AGPRs across the phase loop; everything that is pipelined across
   // units (raw patch registers, LDS ring, weight ring, A fragments, phase iterators) lives outside.
   f32x16 acc[16];
#pragma unroll
   for (int x = 0; x < 16; ++x)
#pragma unroll
     for (int r = 0; r < 16; ++r) {
       // zero born in an AGPR: a VGPR-class zero makes the loop-carried accumulators VGPR-class, and the compiler
       // then copies all 256 of them into AGPRs before, and back after, every phase
       float z;
       asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(z));
       acc[x][r] = z;
     }
   asm volatile("s_nop 1");
   Phase fin = cur;
   bool last_of_unit;
   do {
    const int rb1 = rb0 == 2 ? 0 : rb0 + 1, rb2 = rb1 == 2 ? 0 : rb1 + 1;
    const float* R0 = Rb + rb0 * 4 * RPLANE;    // raw[cur]
    const float* R1 = Rb + rb1 * 4 * RPLANE;    // raw[nx1]
    float* Rw = Rb + rb2 * 4 * RPLANE;          // raw[nx2] is committed here during this phase
    last_of_unit = cur.cc + 1 >= cur.cc_end;

    // A phase = 8 double steps (xi = 2 y, 2 y + 1: two independent accumulation chains, 8 MFMAs).  Steps 0..3
    // multiply rows 0,1 of V (Af[0]) while rows 2,3 (Af[1]) are built from raw[cur]; steps 4..7 multiply Af[1]
    // while Af[0] of the NEXT phase is built from raw[nx1], nx2's raw patch is committed and nx3's fetched.
    //   build schedule over a half phase: step 0: read columns 0,1 | 1: read columns 2,3, row sums of 0,1 |
    //   2: row sums of 2,3, outputs 0..3 | 3: outputs 4..7
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      const int hb = y < 4 ? 1 : 0;               // fragment set being built
      const float* Rs = y < 4 ? R0 : R1;
      const int yy = y & 3;
      if (yy == 1) {
        tr_rows(hb, 0, 0);
        tr_rows(hb, 1, 1);
      }
      if (yy == 2) {
        tr_rows(hb, 2, 2);
        tr_rows(hb, 3, 3);
      }
      if (yy < 2) {
        tr_read(Rs, hb, 2 * yy, 2 * yy);
        tr_read(Rs, hb, 2 * yy + 1, 2 * yy + 1);
      }
      if (yy >= 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) tr_out(hb, 4 * (yy - 2) + k);
      }
      if (y >= 5 && !(EXP & 8)) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int j = 2 * (y - 5) + k;
          if (j < NJ) {
            commit(j, Rw);
            fetch_at(j, nx3, poff);
          }
        }
      }
      if (EXP & 1) {          // diagnostic: 8 extra independent VALU instructions per step (64 per phase)
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(dummy2[k & 1]) : "v"(dummy2[1 - (k & 1)]));
      }
      const int ha = y < 4 ? 0 : 1;
      const f32x4 a0 = Af[ha][(2 * y) & 7], a1 = Af[ha][(2 * y + 1) & 7];
      const f32x4 b0 = wq[(2 * y) & (NWQ - 1)], b1 = wq[(2 * y + 1) & (NWQ - 1)];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        // weights as the A operand, transformed tiles as B: D[channel][tile], so that a lane ends up with ONE tile
        // and four groups of 4 CONSECUTIVE channels (accumulator registers 4 g .. 4 g + 3) -> 16-byte stores
        acc[2 * y] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a0[s], acc[2 * y], 0, 0, 0);
        acc[2 * y + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[s], a1[s], acc[2 * y + 1], 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int x = 2 * y + k;       // refill the ring slot just consumed: NWQ steps ahead
        if (!(EXP & 4)) wq[x & (NWQ - 1)] = x + NWQ < 16 ? load_b(cur, x + NWQ) : load_b(nx1, x + NWQ - 16);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x096, 6, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (yy == 3) {
        // The fragments this half multiplied stay allocated to its end.  The transform's packed subtractions are
        // inline asm, whose register writes hipcc's hazard logic does not see: a fragment register that died with
        // its last MFMA could otherwise be handed to one of them while that MFMA is still fetching operands
        // (observed: wrong values in accumulator rows 1 mod 4 of columns 12-15 / 28-31).
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("" ::"v"(Af[ha][k]));
      }
      if (DBG && p.dbg_mode == 1) stamp();
    }
    if (DBG && p.dbg_mode == 2) stamp();
    __syncthreads();
    if (!(DBG && p.dbg_mode == 1)) stamp();
    fin = cur;
    more = nx1.valid;
    cur = nx1;
    nx1 = nx2;
    nx2 = nx3;
    {
      const int l3 = nx3.L;
      advance(nx3);
      if (nx3.L != l3) slot_offsets(nx3, poff);
    }
    rb0 = rb1;
   } while (!last_of_unit);

    {
      // ---- epilogue: lane-local output transform (every lane holds all 16 xi of its tile's 16 channels), then
      // 16-byte buffer stores straight from the accumulator layout (registers 4 g .. 4 g + 3 = channels
      // 8 g + 4 lh .. + 3 of the wave's 32).  Addressing: ONE per-lane offset (tile pixel, channel group 0) that is a
      // kernel constant, the channel group as an immediate, unit origin and output position (dy, dx) as scalars.
      asm volatile("s_nop 15\n\ts_nop 15");      // the last MFMAs' results must have landed before the explicit reads
      const bool direct = p.splits == 1;
      const long pit = direct ? p.out_pitch : (long)p.N;
      const unsigned sbase = (unsigned)((((long)(fin.b0 * p.H + fin.h0) * p.W + fin.w0) * pit + fin.n0) * 4) +
                             (direct ? 0u : (unsigned)((long)fin.split * p.ws_stride * 4));
      const unsigned rbase = (unsigned)((((long)(fin.b0 * p.H + fin.h0) * p.W + fin.w0) * p.res_pitch + fin.n0) * 4);
      const unsigned eo = direct ? evoff : evoff_w;
      const __amdgpu_buffer_rsrc_t rsrc_d = direct ? rsrc_o : rsrc_w;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (direct && p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + fin.n0 + wcb * 32 + 8 * g + 4 * lh);
        // S = A^T M for the group's 4 registers x 4 columns jx, as packed pairs over r
        f32x2 sv[2][4][2];
#pragma unroll
        for (int jx = 0; jx < 4; ++jx)
#pragma unroll
          for (int r2 = 0; r2 < 2; ++r2) {
            // explicit accumulator reads: left to itself the compiler copies all 256 accumulators to VGPRs at once
            float m[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int e = 0; e < 2; ++e)
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m[i][e]) : "a"(acc[4 * i + jx][4 * g + 2 * r2 + e]));
            const f32x2 m0 = {m[0][0], m[0][1]}, m1 = {m[1][0], m[1][1]}, m2 = {m[2][0], m[2][1]}, m3 = {m[3][0], m[3][1]};
            sv[0][jx][r2] = add2(m0, add2(m1, m2));
            sv[1][jx][r2] = sub2(sub2(m1, m2), m3);
          }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const unsigned spos = (unsigned)((dy * p.W + dx) * pit * 4);
            f32x2 y[2];
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
              y[r2] = dx == 0 ? add2(sv[dy][0][r2], add2(sv[dy][1][r2], sv[dy][2][r2]))
                              : sub2(sub2(sv[dy][1][r2], sv[dy][2][r2]), sv[dy][3][r2]);
            f32x4 v = add4(__builtin_shufflevector(y[0], y[1], 0, 1, 2, 3), bv);
            if (DBG && p.dbg_mode >= 3) {        // diagnostic: raw accumulators, position (dy, dx) <- xi = 4 (mode - 3) + 2 dy + dx
#pragma unroll
              for (int xs = 0; xs < 16; ++xs)
                if (xs == (p.dbg_mode >= 12 ? 3 : 4 * (p.dbg_mode - 3) + 2 * dy + dx)) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[e]) : "a"(acc[xs][4 * g + e]));
                }
            }
            if (direct && p.res) {
              const unsigned rpos = (unsigned)((dy * p.W + dx) * p.res_pitch * 4);
              v = add4(v, __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, evoff_r + g * 32, rbase + rpos, 0)));
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc_d, eo + g * 32, sbase + spos, 0);
            // A store of more than 8 bytes must not have its data registers overwritten in the next 2 wait states;
            // hipcc pads that for its own instructions but the next writers here are inline asm (accumulator reads,
            // packed adds), which it does not see.  Without the pad: wrong second dwords in lanes 12-15 / 28-31 of
            // every store but the last one of a channel group.
            asm volatile("s_nop 1" ::: "memory");
          }
      }
    }
    stamp();
  }
  if (EXP & 1) asm volatile("" ::"v"(dummy2[0]), "v"(dummy2[1]));
  if (DBG && threadIdx.x == 0) {
    p.dbg[blockIdx.x * 64] = nstamp;
    p.dbg[blockIdx.x * 64 + 1] = (long long)__builtin_amdgcn_s_memrealtime();
  }
}

// ---------------------------------------------------------------------------------------------
// Weight transform, table-driven over the 3x3 slots of a flat parameter buffer.  Table rows (int64 x 6):
//   src offset (floats) of w[Np][9][Cp], Np, Cp, dst offset of the forward operand, dst offset of the
//   input-gradient operand, first block.  One block = 32 n x 32 c.
// Forward operand   Uf[n/32][c/8][xi][(c%8)/4][n%32][c%4]   = (G g G^T)[xi],        g[a][b] = w[n][3a+b][c]
// Input-grad operand Ub[c/32][n/8][xi][(n%8)/4][c%32][n%4]  = (G g' G^T)[xi],       g'[a][b] = w[n][3(2-a)+(2-b)][c]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ggt(const f32x4 (&g)[9], f32x4 (&u)[16]) {
  // rows: G g  (4x3), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
  f32x4 t[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t[0][b] = g[b];
    t[1][b] = 0.5f * ((g[b] + g[6 + b]) + g[3 + b]);
    t[2][b] = 0.5f * ((g[b] + g[6 + b]) - g[3 + b]);
    t[3][b] = g[6 + b];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u[i * 4 + 0] = t[i][0];
    u[i * 4 + 1] = 0.5f * ((t[i][0] + t[i][2]) + t[i][1]);
    u[i * 4 + 2] = 0.5f * ((t[i][0] + t[i][2]) - t[i][1]);
    u[i * 4 + 3] = t[i][2];
  }
}

__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ src, float* __restrict__ dst_f,
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
  // stage w[32 n][9][32 c]: 288 rows of 128 bytes
  for (int i = tid; i < 288 * 8; i += 256) {
    const int r = i >> 3, c4 = (i & 7) * 4;
    *reinterpret_cast<f32x4*>(wt + r * 36 + c4) = *reinterpret_cast<const f32x4*>(w + (long)r * Cp + c4);
  }
  __syncthreads();
  if (dst_f) {      // thread = (n, c quad): 16 stores of 16 bytes, 512 contiguous bytes per (xi, k-half) and wave half
    const int n = tid & 31, cq = tid >> 5;        // cq 0..7 -> chunk cq >> 1, k-half cq & 1
    f32x4 g[9], u[16];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = *reinterpret_cast<const f32x4*>(wt + (n * 9 + t) * 36 + cq * 4);
    ggt(g, u);
    float* d = dst_f + row[3] + (((long)nb * (Cp / 8) + cb * 4 + (cq >> 1)) * 16) * 256 + ((cq & 1) * 32 + n) * 4;
#pragma unroll
    for (int x = 0; x < 16; ++x) *reinterpret_cast<f32x4*>(d + x * 256) = u[x];
  }
  if (dst_b) {      // thread = (c, n quad)
    const int c = tid & 31, nq = tid >> 5;
    f32x4 g[9], u[16];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) g[8 - t][s] = wt[((nq * 4 + s) * 9 + t) * 36 + c];      // mirrored taps
    ggt(g, u);
    float* d = dst_b + row[4] + (((long)cb * (Np / 8) + nb * 4 + (nq >> 1)) * 16) * 256 + ((nq & 1) * 32 + c) * 4;
#pragma unroll
    for (int x = 0; x < 16; ++x) *reinterpret_cast<f32x4*>(d + x * 256) = u[x];
  }
}

static inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// unit shape: 64 tiles = TTH x TTW tiles of NI images
static bool plan_unit(int H, int W, int* TTH, int* TTW, int* NI) {
  if (!pow2(H) || !pow2(W) || H < 4 || W < 4) return false;
  const int th = H / 2, tw = W / 2;               // tiles per image
  *TTW = tw < 8 ? tw : 8;
  int rows = TB / *TTW;
  if (rows > th) rows = th;
  *TTH = rows;
  *NI = TB / (rows * *TTW);
  if (*NI > 1 && (rows != th || *TTW != tw)) return false;
  return *NI * rows * *TTW == TB && th % rows == 0 && tw % *TTW == 0 && rows == *TTW;   // the three unit classes
}

}  // namespace lgmwino

bool lgm_wino_supported(const LgmConvGeom* g, int gather_channels, int out_channels) {
  using namespace lgmwino;
  int a, b, c;
  if (!(g->KH == 3 && g->KW == 3 && g->stride == 1 && g->pad == 1)) return false;
  if (gather_channels % 8 != 0 || out_channels % 64 != 0) return false;
  if (!plan_unit(g->H, g->W, &a, &b, &c)) return false;
  return g->B % c == 0;
}

int lgm_wino_splits(const LgmConvGeom* g, int gather_channels, int out_channels) {
  using namespace lgmwino;
  int TTH, TTW, NI;
  if (!plan_unit(g->H, g->W, &TTH, &TTW, &NI)) return 1;
  const long base = (long)(g->B / NI) * (g->H / (2 * TTH)) * (g->W / (2 * TTW)) * (out_channels / 64);
  if (base >= 1024) return 1;
  const int phases = gather_channels / KC;
  long smax = phases / 4 < 8 ? phases / 4 : 8;    // at least four phases (32 channels) per split
  if (smax < 1) smax = 1;
  long s = 1;
  double best = 1e30;
  for (long c = 1; c <= smax; ++c) {
    const long pps = (phases + c - 1) / c;
    if ((phases + pps - 1) / pps != c) continue;
    const double rounds = (double)((base * c + 255) / 256);
    // per unit: pps phases + ~1.5 phases of prologue / epilogue; partial-sum traffic penalty per split
    const double cost = rounds * ((double)pps + 1.5) / (double)phases + 0.03 * (double)(c - 1);
    if (cost < best - 1e-9) {
      best = cost;
      s = c;
    }
  }
  return (int)s;
}

int lgm_wino_launch(const LgmConvGeom* g, int yx, const float* a, long a_pitch, const float* u, const float* bias,
                    const float* res, long res_pitch, float* out, long out_pitch, void* workspace,
                    long workspace_bytes, hipStream_t s) {
  using namespace lgmwino;
  Args p{};
  p.a = a; p.u = u; p.bias = bias; p.res = res; p.out = out;
  p.a_pitch = a_pitch; p.res_pitch = res_pitch; p.out_pitch = out_pitch;
  p.B = g->B; p.H = g->H; p.W = g->W;
  p.C = yx ? g->Nw : g->Cw;
  p.N = yx ? g->Cw : g->Nw;
  int TTH, TTW, NI;
  plan_unit(g->H, g->W, &TTH, &TTW, &NI);
  p.tb_h = g->H / (2 * TTH);
  p.tb_w = g->W / (2 * TTW);
  p.tiles_n = p.N / 64;
  const long M = (long)g->B * g->H * g->W;
  p.splits = lgm_wino_splits(g, p.C, p.N);
  if (p.splits > 1) {
    const long need = (long)p.splits * M * p.N * (long)sizeof(float);
    if (!workspace || workspace_bytes < need || !lgm_aligned16(workspace)) p.splits = 1;
  }
  p.ws = (float*)workspace;
  p.ws_stride = M * p.N;
  p.pps = lgm_cdiv(p.C / KC, p.splits);
  p.splits = lgm_cdiv(p.C / KC, p.pps);
  p.units = (int)((long)(g->B / NI) * p.tb_h * p.tb_w * p.tiles_n * p.splits);
  p.per = lgm_cdiv(p.units, 256);
  const unsigned nblocks = (unsigned)lgm_cdiv(p.units, p.per);
  p.dbg = (long long*)lgm_wino_debug_buffer;
  p.dbg_mode = lgm_wino_debug_mode & 15;
  const int e = p.dbg ? (lgm_wino_debug_mode >> 4) : 0;
#define LGM_WLAUNCH(GG, D, E)                                                                                   \
  do {                                                                                                          \
    auto kern = wino_conv_kernel<GG, D, E>;                                                                     \
    const size_t smem = ((size_t)12 * Geo<GG>::RPLANE) * sizeof(float);                     \
    static bool attr = false;                                                                                   \
    if (!attr) {                                                                                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                          (int)smem);                                                                           \
      attr = true;                                                                                              \
    }                                                                                                           \
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), smem, s, p);                                             \
  } while (0)
  if (!p.dbg) {
    if (TTW == 8) LGM_WLAUNCH(8, false, 0);
    else if (TTW == 4) LGM_WLAUNCH(4, false, 0);
    else LGM_WLAUNCH(2, false, 0);
  } else if (TTW == 8) {                 // diagnostic builds (cycle stamps / attribution experiments): 8x8 units only
    if (e == 0) LGM_WLAUNCH(8, true, 0);
    else if (e == 1) LGM_WLAUNCH(8, true, 1);
    else if (e == 2) LGM_WLAUNCH(8, true, 2);
    else if (e == 4) LGM_WLAUNCH(8, true, 4);
    else if (e == 8) LGM_WLAUNCH(8, true, 8);
    else if (e == 14) LGM_WLAUNCH(8, true, 14);
    else LGM_WLAUNCH(8, true, 15);
  } else if (TTW == 4) {
    LGM_WLAUNCH(4, true, 0);
  } else {
    LGM_WLAUNCH(2, true, 0);
  }
#undef LGM_WLAUNCH
  if (p.splits > 1)
    return lgm_splitk_reduce_launch(p.ws, p.ws_stride, p.splits, bias, res, res_pitch, out, out_pitch, M, p.N, s);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

// ---- C-ABI ------------------------------------------------------------------------------------
extern "C" int64_t lgm_conv3x3_wino_supported(const LgmConvGeom* g, int yx) {
  if (!g) return 0;
  return lgm_wino_supported(g, yx ? g->Nw : g->Cw, yx ? g->Cw : g->Nw) ? 1 : 0;
}

extern "C" int64_t lgm_conv3x3_wino_workspace(const LgmConvGeom* g, int yx) {
  if (!g) return -1;
  const int gc = yx ? g->Nw : g->Cw, oc = yx ? g->Cw : g->Nw;
  if (!lgm_wino_supported(g, gc, oc)) return 0;
  const int s = lgm_wino_splits(g, gc, oc);
  return s > 1 ? (int64_t)s * g->B * g->H * g->W * oc * (int64_t)sizeof(float) : 0;
}

extern "C" int lgm_wino_weights(const float* src, float* dst_f, float* dst_b, const int64_t* table, int n_slots,
                                int64_t total_blocks, void* stream) {
  LGM_REQUIRE(src && table && n_slots > 0 && total_blocks > 0 && (dst_f || dst_b), "wino_weights: bad arguments");
  hipLaunchKernelGGL(lgmwino::wino_weights_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, src,
                     dst_f, dst_b, (const long*)table, n_slots);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_conv3x3_wino(int yx, const LgmConvGeom* g, const float* a, int64_t a_pitch, const float* u,
                                const float* bias, const float* res, int64_t res_pitch, float* out, int64_t out_pitch,
                                void* workspace, int64_t workspace_bytes, void* stream) {
  LGM_REQUIRE(g && a && u && out, "conv3x3_wino: null pointer");
  const int gc = yx ? g->Nw : g->Cw, oc = yx ? g->Cw : g->Nw;
  LGM_REQUIRE(lgm_wino_supported(g, gc, oc), "conv3x3_wino: unsupported geometry (3x3/s1/p1, H,W powers of two >= 4, "
              "reduction channels %% 8, produced channels %% 64, whole image groups)");
  LGM_REQUIRE(a_pitch % 4 == 0 && a_pitch >= gc && lgm_aligned16(a) && lgm_aligned16(u) && lgm_aligned16(out) &&
              out_pitch % 4 == 0 && out_pitch >= oc && (!res || (lgm_aligned16(res) && res_pitch % 4 == 0 && res_pitch >= oc)) &&
              (!bias || lgm_aligned16(bias)), "conv3x3_wino: 16-byte aligned operands with pitch %% 4 == 0 expected");
  LGM_REQUIRE(((long)g->B * g->H * g->W + g->W + 1) * a_pitch < (1L << 29), "conv3x3_wino: tensor too large for 32-bit offsets");
  return lgm_wino_launch(g, yx, a, a_pitch, u, bias, res, res_pitch, out, out_pitch, workspace, workspace_bytes,
                         (hipStream_t)stream);
}
