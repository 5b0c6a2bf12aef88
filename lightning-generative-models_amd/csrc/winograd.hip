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
#include <stdlib.h>

#include <type_traits>

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

// cache policy of the kernels' OUTPUT stores (experiment knob, -DLGM_STORE_AUX=n at build time): 0 default write-back,
// 2 = nt (streaming), 17 = sc0 | sc1 (write-through to memory)
#ifndef LGM_STORE_AUX
#define LGM_STORE_AUX 0
#endif

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
  int nbg;             // image groups (B / NI)
  int tile_fastest;    // unit order: 0 = channel block fastest; 1 (whole-image units only) = image group fastest, then
                       // split, channel block slowest - the workgroups of one XCD then share ONE channel block's weights
  int units, per, splits, pps;
  float* ws;
  long ws_stride;
  int dbg_mode;        // 0: stamps per phase / epilogue, 1: per double step, 2: before and after every barrier,
                       // 5: mode 0 + the prologue in four parts (setup | fetch issue | landed + barrier | transform)
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
  // positions per image of a plane.  G == 2 (16 images of 2 x 2 tiles): the 16 lanes one ds_read_b128 services together
  // are 4 images x 4 tiles at position offsets {0, 1, 6, 7} + image stride; 18 positions per image put two of the four
  // images on the same banks (conflict share 0.50, SQ_LDS_BANK_CONFLICT), 20 = 4 (mod 16) puts all 16 on different ones
  static constexpr int IMGS = PH * PWh + (G == 2 ? 2 : 0);
  static constexpr int NP = NI * PH * PWr;
  static constexpr int RPLANE = NI * IMGS * 4 + 16;
  static constexpr int NJ = (2 * NP + 255) / 256;
};

// The kernel body as a device function of (arguments, logical block id, logical grid size): it runs as its own
// launch (wino_conv_kernel) or as the first block range of the backward PAIR launch (wino_bwd_pair_kernel: input
// gradient and weight gradient of one layer side by side, see below).
template <int G, bool DBG, int EXP = 0>
__device__ __forceinline__ void wino_conv_body(const Args& p, const int bidx, const int nblk) {
  using GE = Geo<G>;
  constexpr int TTW = GE::TTW, TTH = GE::TTH, NI = GE::NI, lgTTW = GE::lgTTW, lgTT = GE::lgTT, PH = GE::PH,
                PWr = GE::PWr, PWh = GE::PWh, NP = GE::NP, RPLANE = GE::RPLANE, NJ = GE::NJ, IMGS = GE::IMGS;
  extern __shared__ __align__(16) float smem[];
  int nstamp = 0;
  auto stamp = [&]() {
    if (DBG) {
      if (threadIdx.x == 0 && nstamp < 62) p.dbg[bidx * 64 + 2 + nstamp] = (long long)__builtin_amdgcn_s_memtime();
      ++nstamp;
    }
  };
  stamp();
  float* Rb = smem;                              // 3 buffers x 4 planes ([k-half][column parity]) x RPLANE
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wtb = wid & 1, wcb = wid >> 1;       // MFMA role: tile half, output-channel half
  const int lr = lane & 31, lh = lane >> 5;

  const int Lb = xcd_swizzle(bidx, nblk);
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
    if (p.tile_fastest) {
      ph.bg = L % p.nbg;
      const int r = L / p.nbg;
      ph.split = r % p.splits;
      ph.tn = r / p.splits;
      ph.twi = ph.thi = 0;
      place(ph);
      return;
    }
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
    } else if (ph.L + 1 < L1 && p.tile_fastest) {
      ++ph.L;
      if (++ph.bg == p.nbg) {
        ph.bg = 0;
        if (++ph.split == p.splits) {
          ph.split = 0;
          ++ph.tn;
        }
      }
      place(ph);
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
    unsigned d = 0, f = 16u, l = (unsigned)(q * 2 * RPLANE + NI * IMGS * 4);
    if (pos < NP) {
      const int img = pos / (PH * PWr);
      const int rem = pos - img * (PH * PWr);
      const int py = rem / PWr, px = rem - py * PWr;
      d = (unsigned)(((img * p.H + py) * p.W + px) * (int)p.a_pitch + q * 4) * 4u;
      f = (py == 0 ? 1u : 0u) | (py == PH - 1 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == PWr - 1 ? 8u : 0u);
      l = (unsigned)((q * 2 + (px & 1)) * RPLANE + (img * IMGS + py * PWh + (px >> 1)) * 4);
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

  static_assert(NWQ == 16, "the prologue issues the whole weight ring for the first phase");
  // ---- prologue ----
  // Order matters here: ~850 scalar / vector instructions of set-up (four phase iterators with their integer divisions,
  // the epilogue's addressing, buffer descriptors) used to run BEFORE the first load was issued, and then the workgroup
  // waited ~1 us for that load (cycle stamps, mode 5: 2.0 k + 3.6 k cycles of set-up, 0.9 k of waiting).  Now the first
  // unit's weight fragments and raw patch are requested as soon as ITS iterator is decoded, and everything else - the
  // look-ahead iterators, the epilogue addressing - is computed under the shadow of those loads.
  if (DBG && p.dbg_mode == 5) stamp();             // mode 5: the prologue in four parts (setup | fetch issue | landed + barrier | first transform)
  Phase cur;
  cur.valid = true;
  decode(cur, L0);
  f32x4 wq[NWQ];
#pragma unroll
  for (int x = 0; x < NWQ; ++x) wq[x] = load_b(cur, x);          // NWQ == 16: all of them belong to the first phase
  unsigned poff[NJ];
  u32x4 r0[NJ];
  slot_offsets(cur, poff);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const unsigned soff0 = (unsigned)(cur.abase + (long)cur.cc * (KC * 4));
    r0[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, poff[j], soff0, 0);
  }
  Phase nx1 = cur;
  advance(nx1);
  slot_offsets(nx1, poff);
#pragma unroll
  for (int j = 0; j < NJ; ++j) fetch_at(j, nx1, poff);
  Phase nx2 = nx1;
  advance(nx2);
  Phase nx3 = nx2;
  advance(nx3);

  // ---- input transform: every lane builds its OWN A fragments (tile = accumulator row lr, channel quad lh) in
  // registers straight from the raw patch -- the transformed tiles never go through LDS.  Raw element d[r][c] of
  // the lane's 4x4 input tile: plane lh*2 + (c & 1), row 2 ty + r, half-column tx + (c >> 1).
  int trd;
  {
    const int t = wtb * 32 + tile_of_lane(lr);
    const int img = t >> lgTT, rr = t & ((1 << lgTT) - 1);
    const int ty = rr >> lgTTW, tx = rr & (TTW - 1);
    trd = lh * 2 * RPLANE + (img * IMGS + 2 * ty * PWh + tx) * 4;
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

  if (DBG && p.dbg_mode == 5) stamp();
#pragma unroll
  for (int j = 0; j < NJ; ++j) *reinterpret_cast<u32x4*>(Rb + plds[j]) = r0[j];        // raw[cur] -> buffer 0
#pragma unroll
  for (int j = 0; j < NJ; ++j) commit(j, Rb + 4 * RPLANE);                              // raw[nx1] -> buffer 1
  slot_offsets(nx2, poff);
#pragma unroll
  for (int j = 0; j < NJ; ++j) fetch_at(j, nx2, poff);                                  // raw[nx2] stays in registers
  slot_offsets(nx3, poff);
  __syncthreads();
  if (DBG && p.dbg_mode == 5) stamp();
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
   // No zero-fill: the first phase of a unit issues its first MFMA of every accumulator with the constant 0 as C
   // (256 v_accvgpr_write per unit cost ~2 k cycles of MFMA issue).  The accumulators are born as MFMA results, i.e.
   // in AGPRs: a VGPR-class zero would make the loop-carried accumulators VGPR-class, and the compiler would then
   // copy all 256 of them into AGPRs before, and back after, every phase.
   f32x16 acc[16];
   Phase fin = cur;
   bool last_of_unit;
   // Epilogue operands fetched HERE, a whole unit ahead: the four bias vectors and the residual rows of channel
   // group 0.  vmcnt counts loads and stores in issue order, so a load issued after a store can only be waited for
   // together with that store: every epilogue load is therefore issued BEFORE the stores it would otherwise queue
   // behind (groups 1..3 of the residual: one group ahead, inside the epilogue).  Before this the epilogue waited
   // for the write acknowledgement of the previous channel group four times per unit.
   const bool ep_res = p.splits == 1 && p.res != nullptr;
   const unsigned rbase_u = (unsigned)((((long)(cur.b0 * p.H + cur.h0) * p.W + cur.w0) * p.res_pitch + cur.n0) * 4);
   f32x4 rv[2][4];
   auto load_res = [&](int g, f32x4 (&dst)[4]) {   // the four output positions (dy, dx) of the lane's tile
#pragma unroll
     for (int q = 0; q < 4; ++q) {
       const unsigned rpos = (unsigned)(((q >> 1) * p.W + (q & 1)) * p.res_pitch * 4);
       dst[q] = f32x4{0.f, 0.f, 0.f, 0.f};
       if (ep_res)
         dst[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, evoff_r + g * 32, rbase_u + rpos, 0));
     }
   };
   load_res(0, rv[0]);
   f32x4 bvs[4];
#pragma unroll
   for (int g = 0; g < 4; ++g) {
     bvs[g] = f32x4{0.f, 0.f, 0.f, 0.f};
     if (p.splits == 1 && p.bias) bvs[g] = *reinterpret_cast<const f32x4*>(p.bias + cur.n0 + wcb * 32 + 8 * g + 4 * lh);
   }
   auto run_phase = [&](auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
        acc[2 * y] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a0[s], FIRST && s == 0 ? zero16 : acc[2 * y], 0, 0, 0);
        acc[2 * y + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[s], a1[s], FIRST && s == 0 ? zero16 : acc[2 * y + 1], 0, 0, 0);
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
   };
   run_phase(std::true_type{});
   while (!last_of_unit) run_phase(std::false_type{});

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
      const unsigned eo = direct ? evoff : evoff_w;
      const __amdgpu_buffer_rsrc_t rsrc_d = direct ? rsrc_o : rsrc_w;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = bvs[g];
        // next group's residual rows: ahead of this group's stores (see the unit top) and of its ~90 instructions
        if (g < 3) load_res(g + 1, rv[(g + 1) & 1]);
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
            v = add4(v, rv[g & 1][2 * dy + dx]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc_d, eo + g * 32, sbase + spos, LGM_STORE_AUX);
            // A store of more than 8 bytes must not have its data registers overwritten in the next 2 wait states;
            // hipcc pads that for its own instructions but the next writers here are inline asm (accumulator reads,
            // packed adds), which it does not see.  Without the pad: wrong second dwords in lanes 12-15 / 28-31 of
            // every store but the last one of a channel group.  The pad names the stored value as an operand: a
            // bare `s_nop` does not keep the scheduler from placing the next (non-volatile) packed add - whose result
            // may be allocated to the just-freed data registers - between the store and the pad.
            asm volatile("s_nop 1" : "+v"(v) : : "memory");
          }
      }
    }
    stamp();
  }
  if (EXP & 1) asm volatile("" ::"v"(dummy2[0]), "v"(dummy2[1]));
  if (DBG && threadIdx.x == 0) {
    p.dbg[bidx * 64] = nstamp;
    p.dbg[bidx * 64 + 1] = (long long)__builtin_amdgcn_s_memrealtime();
  }
}

template <int G, bool DBG, int EXP = 0>
__global__ __launch_bounds__(256, 1) void wino_conv_kernel(const Args p) {
  wino_conv_body<G, DBG, EXP>(p, (int)blockIdx.x, (int)gridDim.x);
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
  if (dst_f && row[3] >= 0) {      // thread = (n, c quad): 16 stores of 16 bytes, 512 contiguous bytes per (xi, k-half) and wave half
    const int n = tid & 31, cq = tid >> 5;        // cq 0..7 -> chunk cq >> 1, k-half cq & 1
    f32x4 g[9], u[16];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = *reinterpret_cast<const f32x4*>(wt + (n * 9 + t) * 36 + cq * 4);
    ggt(g, u);
    float* d = dst_f + row[3] + (((long)nb * (Cp / 8) + cb * 4 + (cq >> 1)) * 16) * 256 + ((cq & 1) * 32 + n) * 4;
#pragma unroll
    for (int x = 0; x < 16; ++x) *reinterpret_cast<f32x4*>(d + x * 256) = u[x];
  }
  if (dst_b && row[4] >= 0) {      // thread = (c, n quad)
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
  // 32-bit byte offsets inside the kernel: dense operands must stay below 2^31 bytes (callers with wider pitches
  // are checked again at the call: lgm_conv3x3_wino_fits)
  const long pix = (long)g->B * g->H * g->W + g->W + 1;
  if (pix * gather_channels >= (1L << 29) || pix * out_channels >= (1L << 29)) return false;
  return g->B % c == 0;
}

// fused = the consumer sums the partial planes itself (lgm_gn_fwd_planes / lgm_gn_bwd_planes): no reducer launch, and
// its read of the reduced tensor is replaced by a read of the planes
int lgm_wino_splits(const LgmConvGeom* g, int gather_channels, int out_channels, bool fused = false) {
  using namespace lgmwino;
  int TTH, TTW, NI;
  if (!plan_unit(g->H, g->W, &TTH, &TTW, &NI)) return 1;
  const long base = (long)(g->B / NI) * (g->H / (2 * TTH)) * (g->W / (2 * TTW)) * (out_channels / 64);
  if (base >= 1024) return 1;
  const int phases = gather_channels / KC;
  long smax = phases / 2 < 32 ? phases / 2 : 32;    // at least two phases (16 channels) per split
  if (smax < 1) smax = 1;
  // cost in phase times (~2.1 us): rounds of 256 workgroups x (phases per unit + ~1.5 of prologue / epilogue) + the
  // partial sums every extra split writes and the reducer reads back (8 bytes per output element at ~3 TB/s) -
  // 1.3 phases per split on the 4x4 maps at B = 128, 0.16 at B = 16, where splitting deeper is what fills the chip
  static const double k_ovh = getenv("LGM_FPLAN_OVH") ? atof(getenv("LGM_FPLAN_OVH")) : 1.5;       // tuning knobs (A/B runs)
  static const double k_spl = getenv("LGM_FPLAN_SPLIT") ? atof(getenv("LGM_FPLAN_SPLIT")) : 1.0;
  const double per_split = k_spl * (fused ? 6.0 : 8.0) * (double)g->B * g->H * g->W * out_channels / 3.0e12 / 2.1e-6;
  long s = 1;
  double best = 1e30;
  for (long c = 1; c <= smax; ++c) {
    const long pps = (phases + c - 1) / c;
    if ((phases + pps - 1) / pps != c) continue;
    const double rounds = (double)((base * c + lgm_cu_budget() - 1) / lgm_cu_budget());
    const double cost = rounds * ((double)pps + k_ovh) + (c > 1 ? (fused ? 0.2 : 2.4) : 0.0) + per_split * (double)(c - 1);
    if (cost < best - 1e-9) {
      best = cost;
      s = c;
    }
  }
  return (int)s;
}

// partial (optional, int64 x 2): the caller's consumer sums the split-K planes itself.  On return partial[0] = number
// of planes left in `workspace` (1: none, `out` is complete with bias / residual applied), partial[1] = plane stride
// in floats; with planes, `out`, `bias` and `res` are NOT touched / applied.
// arguments, grid and unit class of one forward / input-gradient launch
static void wino_prepare(const LgmConvGeom* g, int yx, const float* a, long a_pitch, const float* u, const float* bias,
                         const float* res, long res_pitch, float* out, long out_pitch, void* workspace,
                         long workspace_bytes, bool fused, lgmwino::Args& p, unsigned* nblocks_out, int* ttw_out,
                         int force_splits = 0) {
  using namespace lgmwino;
  p = Args{};
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
  p.splits = force_splits > 0 ? force_splits : lgm_wino_splits(g, p.C, p.N, fused);
  if (p.splits > 1) {
    const long need = (long)p.splits * M * p.N * (long)sizeof(float);
    if (!workspace || workspace_bytes < need || !lgm_aligned16(workspace)) p.splits = 1;
  }
  p.ws = (float*)workspace;
  p.ws_stride = M * p.N;
  p.pps = lgm_cdiv(p.C / KC, p.splits);
  p.splits = lgm_cdiv(p.C / KC, p.pps);
  p.units = (int)((long)(g->B / NI) * p.tb_h * p.tb_w * p.tiles_n * p.splits);
  p.per = lgm_cdiv(p.units, lgm_cu_budget());
  p.nbg = g->B / NI;
  // Small maps (a unit = whole images): image-group-fastest order, so that the workgroups one XCD runs together (a
  // contiguous unit range, xcd_swizzle) work on ONE channel block and fetch its transformed weights into that XCD's L2
  // once - channel-block-fastest made every XCD read the whole table (128.8 MB per launch for 512 -> 512 at 4x4 against
  // 16.8 MB of weights).  Time-neutral (those layers are not bound by the weight fetch); it is HBM-side traffic.
  static const bool tile_fastest = !(getenv("LGM_WINO_TN_FASTEST") != nullptr);   // A/B switch
  p.tile_fastest = (tile_fastest && p.tb_h == 1 && p.tb_w == 1 && p.nbg > 1) ? 1 : 0;
  *nblocks_out = (unsigned)lgm_cdiv(p.units, p.per);
  *ttw_out = TTW;
}

int lgm_wino_launch(const LgmConvGeom* g, int yx, const float* a, long a_pitch, const float* u, const float* bias,
                    const float* res, long res_pitch, float* out, long out_pitch, void* workspace,
                    long workspace_bytes, hipStream_t s, int64_t* partial = nullptr) {
  using namespace lgmwino;
  Args p;
  unsigned nblocks;
  int TTW;
  wino_prepare(g, yx, a, a_pitch, u, bias, res, res_pitch, out, out_pitch, workspace, workspace_bytes, partial != nullptr,
               p, &nblocks, &TTW);
  const long M = (long)g->B * g->H * g->W;
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
  lgm_note_kernel(TTW == 8 ? LGM_KNAME("lgmwino::wino_conv_kernel<8, false, 0>") : TTW == 4 ? LGM_KNAME("lgmwino::wino_conv_kernel<4, false, 0>")
                                                                                : LGM_KNAME("lgmwino::wino_conv_kernel<2, false, 0>"));
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
// every tensor the kernel addresses with 32-bit byte offsets fits (pitched operands: the caller's pitches)
extern "C" int64_t lgm_conv3x3_wino_fits(const LgmConvGeom* g, int64_t a_pitch, int64_t out_pitch, int64_t res_pitch) {
  if (!g) return 0;
  const long pix = (long)g->B * g->H * g->W + g->W + 1;
  return (pix * a_pitch < (1L << 29) && pix * out_pitch < (1L << 29) && pix * res_pitch < (1L << 29)) ? 1 : 0;
}

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

extern "C" int64_t lgm_conv3x3_wino_workspace_partial(const LgmConvGeom* g, int yx) {
  if (!g) return -1;
  const int gc = yx ? g->Nw : g->Cw, oc = yx ? g->Cw : g->Nw;
  if (!lgm_wino_supported(g, gc, oc)) return 0;
  const int s = lgm_wino_splits(g, gc, oc, true);
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
  LGM_REQUIRE(lgm_conv3x3_wino_fits(g, a_pitch, out_pitch, res ? res_pitch : 0), "conv3x3_wino: tensor too large for 32-bit offsets");
  return lgm_wino_launch(g, yx, a, a_pitch, u, bias, res, res_pitch, out, out_pitch, workspace, workspace_bytes,
                         (hipStream_t)stream);
}

extern "C" int lgm_conv3x3_wino_partial(int yx, const LgmConvGeom* g, const float* a, int64_t a_pitch, const float* u,
                                        const float* bias, float* out, int64_t out_pitch, void* workspace,
                                        int64_t workspace_bytes, int64_t* partial, void* stream) {
  LGM_REQUIRE(g && a && u && out && partial, "conv3x3_wino_partial: null pointer");
  const int gc = yx ? g->Nw : g->Cw, oc = yx ? g->Cw : g->Nw;
  LGM_REQUIRE(lgm_wino_supported(g, gc, oc), "conv3x3_wino_partial: unsupported geometry");
  LGM_REQUIRE(a_pitch % 4 == 0 && a_pitch >= gc && lgm_aligned16(a) && lgm_aligned16(u) && lgm_aligned16(out) &&
              out_pitch % 4 == 0 && out_pitch >= oc && (!bias || lgm_aligned16(bias)),
              "conv3x3_wino_partial: 16-byte aligned operands with pitch %% 4 == 0 expected");
  LGM_REQUIRE(lgm_conv3x3_wino_fits(g, a_pitch, out_pitch, 0), "conv3x3_wino_partial: tensor too large for 32-bit offsets");
  return lgm_wino_launch(g, yx, a, a_pitch, u, bias, nullptr, 0, out, out_pitch, workspace, workspace_bytes,
                         (hipStream_t)stream, partial);
}

// =====================================================================================================
// Winograd weight gradient:  dU[xi][n][c] = sum_tiles dYt[xi][tile][n] * Xt[xi][tile][c],  gW = G^T dU G
//   Xt = B^T d B of the 4x4 input tile (as in the forward kernel), dYt = A dY A^T of the 2x2 output-gradient tile:
// 16 MFMAs per pair of tiles (8 output pixels) instead of 9 per pair of pixels.  The contraction index is the
// TILE, so a lane's operand fragment is 4 tiles of ONE channel: the raw patches are staged in LDS channel-major
// with the four tiles of a fragment contiguous ([channel][row][column mod 4][tile slot], rows padded to 100 / 36
// floats: conflict-free ds_read_b128), and every lane transforms its own fragments in registers.
//
// A CHUNK = 8 tiles = 2 tile rows (the MFMA's two k values) x 4 slots: tiles tx = p, p+2, p+4, p+6 (parity p) of a
// 16-pixel column group on the large maps, 2 slots x 2 images at 8x8, 1 slot x 4 images at 4x4 -- taking every
// second tile keeps the four slots 4 pixel columns apart, so one 16-byte LDS row holds the same tile element of
// all four.  A workgroup = 64 output channels x 64 input channels x a range of chunks (split-K); it writes one
// slab [Nw][9][Cw] (+ bias partials) that the fixed-order slab reducer (lgm_wgrad_reduce_batch) sums.
// Signs: the kernel builds dYt with row / column 3 negated (saves the negations); (G^T dU G) absorbs them.
// =====================================================================================================
namespace lgmwino {

struct WGArgs {
  const float* y;
  const float* x;
  float* out;          // slabs: split k at out + k * slab
  int bias;            // 1: column sums of y into slab[n_w + n]
  long slab, y_pitch, x_pitch;
  int B, H, W, Nw, Cw;
  int tiles_c, splits, cps, total_chunks;
  int rpn, cgn;        // tile-row pairs per image (H / 4), 16-pixel column groups per row (W / 16, or 1)
  long long* dbg;      // diagnostic build only: cycle stamps (start, after the prologue, after every phase, end)
};

template <int G>
struct WGeo {
  static constexpr int IPC = G == 8 ? 1 : G == 4 ? 2 : 4;   // images per chunk
  static constexpr int XC = 16 / IPC;                        // patch columns per image
  static constexpr int SPI = 4 / IPC;                        // tile slots per image
};

constexpr int XLD = 100;    // floats per input channel in an X buffer  (6 rows x 4 x 4, padded)
constexpr int YLD = 36;     // floats per output channel in a Y buffer (4 rows x 2 x 4, padded)
constexpr int WXBUF = 64 * XLD, WYBUF = 64 * YLD, WBUF = WXBUF + WYBUF;

template <int G, bool DBG = false>
__device__ __forceinline__ void wino_wgrad_body(const WGArgs& p, const int bidx) {
  int nstamp = 0;
  auto stamp = [&]() {
    if (DBG) {
      if (threadIdx.x == 0 && nstamp < 62) p.dbg[bidx * 64 + 2 + nstamp] = (long long)__builtin_amdgcn_s_memtime();
      ++nstamp;
    }
  };
  stamp();
  using GE = WGeo<G>;
  constexpr int IPC = GE::IPC, XC = GE::XC, SPI = GE::SPI;
  extern __shared__ __align__(16) float smem[];    // 3 x (X buffer, Y buffer)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wa = wid & 1, wb = wid >> 1;            // input-channel half (A operand), output-channel half (B)
  const int lr = lane & 31, lh = lane >> 5;

  int bid = bidx;
  const int split = bid % p.splits;
  bid /= p.splits;
  const int tc = bid % p.tiles_c, tn = bid / p.tiles_c;
  const int n0 = tn * 64, c0 = tc * 64;
  const int ch_begin = split * p.cps;
  const int ch_end = min(p.total_chunks, ch_begin + p.cps);

  // ---- raw patch slots of this thread: X 6 x (position, channel quad), Y 2 x ----
  // A wave stages 16 POSITIONS x 4 channel quads (quad = lane & 3 + 4 * wave): the channel-major LDS address is
  // channel * XLD + f(position) with XLD a multiple of 4 (16-byte fragment reads), so 16 quads in one store hit only
  // two bank offsets (4 * XLD mod 32 = 16) and the 4 positions beside them four more - an 8-way conflict on every
  // commit (SQ_LDS_BANK_CONFLICT: 71 % of this kernel's LDS cycles).  16 consecutive positions give 16 distinct
  // offsets ((col & 3) * 4 + slot), times the two quad offsets = all 32 banks, two lanes each: the minimum.
  const int q4 = ((tid & 3) + 4 * (tid >> 6)) * 4;  // first channel of the thread's quad
  const int pslot = (tid >> 2) & 15;                // position of the thread inside a pass of 16
  unsigned xdelta[6], xlds[6], ydelta[2], ylds[2];
  unsigned xflag = 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int pos = pslot + 16 * j;                 // (il, rho, colidx), 96 positions
    const int il = pos / (6 * XC), rem = pos - il * 6 * XC;
    const int rho = rem / XC, ci = rem - rho * XC;
    xdelta[j] = (unsigned)(((il * p.H + rho) * p.W + ci) * (int)p.x_pitch + q4) * 4u;
    const int s = il * SPI + (ci >> 2);
    xlds[j] = (unsigned)(q4 * XLD + rho * 16 + (ci & 3) * 4 + s);
    const unsigned f = (rho == 0 ? 1u : 0u) | (rho == 5 ? 2u : 0u) | (ci == 0 ? 4u : 0u) | (ci == XC - 1 ? 8u : 0u);
    xflag |= f << (4 * j);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pos = pslot + 16 * j;                 // ((il * 4 + row) * SPI + sl) * 2 + dx, 32 positions
    const int dx = pos & 1, t = pos >> 1;
    const int sl = t % SPI, t2 = t / SPI;
    const int row = t2 & 3, il = t2 >> 2;
    ydelta[j] = (unsigned)(((il * p.H + row) * p.W + 4 * sl + dx) * (int)p.y_pitch + q4) * 4u;
    ylds[j] = (unsigned)(q4 * YLD + row * 8 + dx * 4 + il * SPI + sl);
  }
  const long pixels = (long)p.B * p.H * p.W;
  const unsigned nrec_y = (unsigned)((pixels * p.y_pitch - n0) * 4);
  const unsigned nrec_x = (unsigned)(((pixels + p.W + 1) * p.x_pitch - c0) * 4);
  auto make_rsrc = [](const float* base, unsigned nrec) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(nrec), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsrc_y = make_rsrc(p.y + n0, nrec_y);
  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(p.x + c0 - (long)(p.W + 1) * p.x_pitch, nrec_x);

  // chunk -> scalar offsets of its origin (image, row 4 rp, column c0px = 16 cg + 2 par) and its border mask
  struct Ch {
    unsigned sx, sy, border;
  };
  auto chunk = [&](int ch) -> Ch {
    Ch c;
    if (ch >= ch_end) {                             // past the range: every load falls out of both descriptors
      c.sx = c.sy = 0x80000000u;
      c.border = 0u;
      return c;
    }
    const int par = ch & 1;
    int t = ch >> 1;
    const int cg = t % p.cgn;
    t /= p.cgn;
    const int rp = t % p.rpn;
    const int img0 = (t / p.rpn) * IPC;
    const unsigned pix = (unsigned)((img0 * p.H + 4 * rp) * p.W + 16 * cg + 2 * par);
    c.sx = pix * (unsigned)p.x_pitch * 4u;
    c.sy = pix * (unsigned)p.y_pitch * 4u;
    c.border = (rp == 0 ? 1u : 0u) | (rp == p.rpn - 1 ? 2u : 0u) | (par == 0 && cg == 0 ? 4u : 0u) |
               (par == 1 && cg == p.cgn - 1 ? 8u : 0u);
    return c;
  };
  u32x4 rx[6], ry[2];
  auto fetch_x = [&](int j, const Ch& c) {
    const bool ok = ((xflag >> (4 * j)) & c.border & 15u) == 0u;
    rx[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, ok ? xdelta[j] : nrec_x, c.sx, 0);
  };
  auto fetch_y = [&](int j, const Ch& c) { ry[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, ydelta[j], c.sy, 0); };
  // (whole-vector bit_cast first: a per-element bit_cast of a loaded vector fed straight into a store wrote
  // element 0 four times, ROCm 7.2 hipcc)
  auto commit_x = [&](int j, float* buf) {
    const f32x4 f = __builtin_bit_cast(f32x4, rx[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i) buf[xlds[j] + i * XLD] = f[i];
  };
  auto commit_y = [&](int j, float* buf) {
    const f32x4 f = __builtin_bit_cast(f32x4, ry[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i) buf[WXBUF + ylds[j] + i * YLD] = f[i];
  };

  // ---- fragment builders: lane = (channel lr of its half, tile row lh); a fragment = the 4 tile slots ----
  const int xrd = (wa * 32 + lr) * XLD + lh * 32;            // + (r * 16 + c * 4): element d[r][c] of the tile
  const int yrd = WXBUF + (wb * 32 + lr) * YLD + lh * 16;    // + (dy * 8 + dx * 4)
  f32x4 Xf[2][8], Yf[2][8];
  f32x4 tq0[4], tq1[4], ec[2][3];
  auto xr_read = [&](const float* buf, int hrow, int c, int k) {
#pragma unroll
    for (int r = 0; r < 3; ++r) ec[k][r] = *reinterpret_cast<const f32x4*>(buf + xrd + ((hrow + r) * 16 + c * 4));
  };
  auto xr_rows = [&](int hrow, int c, int k) {
    if (hrow == 0) {
      tq0[c] = sub4(ec[k][0], ec[k][2]);
      tq1[c] = add4(ec[k][1], ec[k][2]);
    } else {
      tq0[c] = sub4(ec[k][1], ec[k][0]);
      tq1[c] = sub4(ec[k][0], ec[k][2]);
    }
  };
  auto xr_out = [&](int hrow, int o) {
    const f32x4* t = (o >> 2) ? tq1 : tq0;
    f32x4 v;
    if ((o & 3) == 0) v = sub4(t[0], t[2]);
    else if ((o & 3) == 1) v = add4(t[1], t[2]);
    else if ((o & 3) == 2) v = sub4(t[2], t[1]);
    else v = sub4(t[1], t[3]);
    Xf[hrow][o] = v;
  };
  // dYt' (row / column 3 not negated): rows 0,1 = (y0, y0 + y1), rows 2,3' = (y0 - y1, y1); each row (a, b) -> a, a+b, a-b, b
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  auto y_build = [&](const float* buf, int hrow) {
    const f32x4 y00 = *reinterpret_cast<const f32x4*>(buf + yrd), y01 = *reinterpret_cast<const f32x4*>(buf + yrd + 4),
                y10 = *reinterpret_cast<const f32x4*>(buf + yrd + 8), y11 = *reinterpret_cast<const f32x4*>(buf + yrd + 12);
    f32x4 a0, b0, a1, b1;
    if (hrow == 0) {
      a0 = y00; b0 = y01;
      a1 = add4(y00, y10); b1 = add4(y01, y11);
    } else {
      a0 = sub4(y00, y10); b0 = sub4(y01, y11);
      a1 = y10; b1 = y11;
    }
    Yf[hrow][0] = a0; Yf[hrow][1] = add4(a0, b0); Yf[hrow][2] = sub4(a0, b0); Yf[hrow][3] = b0;
    Yf[hrow][4] = a1; Yf[hrow][5] = add4(a1, b1); Yf[hrow][6] = sub4(a1, b1); Yf[hrow][7] = b1;
    // column sums of y (bias gradient): y00 + y01 is fragment 1 of rows 0,1; y10 + y11 fragment 5 of rows 2,3'
    bsum = add4(bsum, hrow == 0 ? Yf[0][1] : Yf[1][5]);
  };

  // ---- prologue: raw[ch0] -> buffer 0, raw[ch0 + 1] -> buffer 1, raw[ch0 + 2] in registers ----
  {
    const Ch c0c = chunk(ch_begin), c1c = chunk(ch_begin + 1), c2c = chunk(ch_begin + 2);
    u32x4 tx[6], ty[2];
#pragma unroll
    for (int j = 0; j < 6; ++j) fetch_x(j, c0c);
#pragma unroll
    for (int j = 0; j < 2; ++j) fetch_y(j, c0c);
#pragma unroll
    for (int j = 0; j < 6; ++j) tx[j] = rx[j];
#pragma unroll
    for (int j = 0; j < 2; ++j) ty[j] = ry[j];
#pragma unroll
    for (int j = 0; j < 6; ++j) fetch_x(j, c1c);
#pragma unroll
    for (int j = 0; j < 2; ++j) fetch_y(j, c1c);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const f32x4 f = __builtin_bit_cast(f32x4, tx[j]);
#pragma unroll
      for (int i = 0; i < 4; ++i) smem[xlds[j] + i * XLD] = f[i];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 f = __builtin_bit_cast(f32x4, ty[j]);
#pragma unroll
      for (int i = 0; i < 4; ++i) smem[WXBUF + ylds[j] + i * YLD] = f[i];
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) commit_x(j, smem + WBUF);
#pragma unroll
    for (int j = 0; j < 2; ++j) commit_y(j, smem + WBUF);
#pragma unroll
    for (int j = 0; j < 6; ++j) fetch_x(j, c2c);
#pragma unroll
    for (int j = 0; j < 2; ++j) fetch_y(j, c2c);
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    xr_read(smem, 0, c, c & 1);
    xr_rows(0, c, c & 1);
  }
#pragma unroll
  for (int o = 0; o < 8; ++o) xr_out(0, o);
  y_build(smem, 0);

  f32x16 acc[16];
#pragma unroll
  for (int x = 0; x < 16; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float z;
      asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(z));     // AGPR-born zero (see wino_conv_kernel)
      acc[x][r] = z;
    }
  asm volatile("s_nop 1");
  stamp();

  int rb0 = 0;
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const int rb1 = rb0 == 2 ? 0 : rb0 + 1, rb2 = rb1 == 2 ? 0 : rb1 + 1;
    const float* R0 = smem + rb0 * WBUF;      // raw[ch]
    const float* R1 = smem + rb1 * WBUF;      // raw[ch + 1]
    float* Rw = smem + rb2 * WBUF;            // raw[ch + 2] (in registers) is committed here during this phase
    const Ch c3 = chunk(ch + 3);
    // 8 double steps as in wino_conv_kernel: steps 0..3 multiply rows 0,1 (fragment sets 0) while rows 2,3 are
    // built from raw[ch]; steps 4..7 multiply rows 2,3 while rows 0,1 of the NEXT chunk are built from
    // raw[ch + 1]; the raw patch of chunk ch + 2 is committed (4 dword LDS stores per 16-byte load: the LDS
    // image is channel-major) and that of chunk ch + 3 fetched, two slots per step.
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      const int hb = y < 4 ? 1 : 0;
      const float* Rs = y < 4 ? R0 : R1;
      const int yy = y & 3;
      if (yy == 1) {
        xr_rows(hb, 0, 0);
        xr_rows(hb, 1, 1);
      }
      if (yy == 2) {
        xr_rows(hb, 2, 0);
        xr_rows(hb, 3, 1);
      }
      if (yy < 2) {
        xr_read(Rs, hb, 2 * yy, 0);
        xr_read(Rs, hb, 2 * yy + 1, 1);
      }
      if (yy >= 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) xr_out(hb, 4 * (yy - 2) + k);
      }
      if (yy == 1) y_build(Rs, hb);
      {                                         // raw slots: X 0..5, Y 0..1 -> one per step
        if (y < 6) {
          commit_x(y, Rw);
          fetch_x(y, c3);
        } else {
          commit_y(y - 6, Rw);
          fetch_y(y - 6, c3);
        }
      }
      const int ha = y < 4 ? 0 : 1;
      const f32x4 a0 = Xf[ha][(2 * y) & 7], a1 = Xf[ha][(2 * y + 1) & 7];
      const f32x4 b0 = Yf[ha][(2 * y) & 7], b1 = Yf[ha][(2 * y + 1) & 7];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        // D[output channel][input channel]: a lane ends up with ONE input channel (lr) and 16 output channels, so a
        // dword store of one register covers 32 consecutive input channels of a gw[n][tap][.] row per half wave
        // (two full 128-byte segments); the transposed orientation (16-byte stores, one row per lane) scatters
        // every store over 32 rows and ran the epilogue 10x slower
        acc[2 * y] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a0[s], acc[2 * y], 0, 0, 0);
        acc[2 * y + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[s], a1[s], acc[2 * y + 1], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x096, 6, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    stamp();
    rb0 = rb1;
  }

  // ---- epilogue: gW = G'^T dU' G' per (lane, register), G' = G with row 3 negated; 16-byte stores ----
  asm volatile("s_nop 15\n\ts_nop 15");
  float* out = p.out + (long)split * p.slab;
  __amdgpu_buffer_rsrc_t rsrc_o;
  {
    const unsigned long long ob = reinterpret_cast<unsigned long long>(out);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ob);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(ob >> 32));
    rsrc_o = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                               __builtin_amdgcn_readfirstlane((unsigned)(p.Nw * 9 * p.Cw) * 4u), 0x00020000);
  }
  // lane: input channel c0 + wa*32 + lr; register r: output channel n0 + wb*32 + (r & 3) + 8 (r >> 2) + 4 lh
  const unsigned row_bytes = 9u * (unsigned)p.Cw * 4u;
  const unsigned vo = (unsigned)(n0 + wb * 32 + 4 * lh) * row_bytes + (unsigned)(c0 + wa * 32 + lr) * 4u;
  const f32x2 half2 = {0.5f, 0.5f};
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      // R[a][j] = sum_i G'[i][a] dU'[i][j]:  R0 = d0 + (d1 + d2)/2,  R1 = (d1 - d2)/2,  R2 = (d1 + d2)/2 - d3
      f32x2 R[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float m[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 2; ++e)
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m[i][e]) : "a"(acc[4 * i + j][4 * g + 2 * r2 + e]));
        const f32x2 d0 = {m[0][0], m[0][1]}, d1 = {m[1][0], m[1][1]}, d2 = {m[2][0], m[2][1]}, d3 = {m[3][0], m[3][1]};
        const f32x2 hs = add2(d1, d2) * half2, hd = sub2(d1, d2) * half2;
        R[0][j] = add2(d0, hs);
        R[1][j] = hd;
        R[2][j] = sub2(hs, d3);
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const f32x2 hs = add2(R[a][1], R[a][2]) * half2, hd = sub2(R[a][1], R[a][2]) * half2;
        const f32x2 w[3] = {add2(R[a][0], hs), hd, sub2(hs, R[a][3])};
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int r = 4 * g + 2 * r2 + e;
            float v = w[b][e];
            asm volatile("" : "+v"(v));
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_o, vo,
                                                  (unsigned)((r & 3) + 8 * (r >> 2)) * row_bytes + (unsigned)((a * 3 + b) * p.Cw) * 4u, LGM_STORE_AUX);
          }
      }
    }
  }
  if (p.bias && tc == 0 && wa == 0) {
    // this lane's tiles: sum the four slots, then the two tile rows (lanes lr and lr + 32)
    float v = (bsum[0] + bsum[1]) + (bsum[2] + bsum[3]);
    v += __shfl_xor(v, 32, 64);
    if (lh == 0) out[(long)p.Nw * 9 * p.Cw + n0 + wb * 32 + lr] = v;
  }
  stamp();
  if (DBG && threadIdx.x == 0) p.dbg[bidx * 64] = nstamp;
}

template <int G, bool DBG = false>
__global__ __launch_bounds__(256, 1) void wino_wgrad_kernel(const WGArgs p) {
  wino_wgrad_body<G, DBG>(p, (int)blockIdx.x);
}

// Two layers' weight gradients in ONE launch (blocks [0, na): layer a, the rest: layer b).  On the large maps a layer's
// weights are small (64 x 64 x 9) and its pixels many: standing alone the kernel splits the pixels 256 ways to fill the
// chip, so every workgroup pays its prologue and a whole 147 KB slab for ~16 chunks of work.  Side by side, two layers
// take 128 workgroups each with twice the chunks: the same chip-wide work with half the prologues, epilogues and slabs
// (measured: 64 -> 64 at 32 x 32, B = 128: 65 us alone, 55 us per layer at twice the work per workgroup).  Same code and
// summation order per slab as the single-layer kernel; only the split count differs, as it does between batches.
template <int G>
__global__ __launch_bounds__(256, 1) void wino_wgrad2_kernel(const WGArgs pa, const WGArgs pb, const int na) {
  if ((int)blockIdx.x < na) wino_wgrad_body<G, false>(pa, (int)blockIdx.x);
  else wino_wgrad_body<G, false>(pb, (int)blockIdx.x - na);
}
// ... and of up to four layers (e0 / e1 / e2 = first block of layers 1 / 2 / 3; an unused layer starts at the grid's end)
template <int G>
__global__ __launch_bounds__(256, 1) void wino_wgrad4_kernel(const WGArgs p0, const WGArgs p1, const WGArgs p2, const WGArgs p3,
                                                             const int e0, const int e1, const int e2) {
  const int b = (int)blockIdx.x;
  if (b < e0) wino_wgrad_body<G, false>(p0, b);
  else if (b < e1) wino_wgrad_body<G, false>(p1, b - e0);
  else if (b < e2) wino_wgrad_body<G, false>(p2, b - e1);
  else wino_wgrad_body<G, false>(p3, b - e2);
}

// Backward PAIR: the input gradient (blocks [0, nconv)) and the weight gradient (the rest) of ONE 3x3 layer in ONE
// launch.  Both read the same output gradient and neither reads what the other writes.  At the per-GPU batches of a
// strong-scaled run (16 ... 64 images) each of them fills a quarter to a half of the chip's 256 CUs and is bounded
// by its own prologue / epilogue latency: side by side they take the time of one.  (Two streams or parallel graph
// branches do not achieve this on this runtime - measured slower at every batch, DESIGN.md section 4 - a single
// launch has no fork / join to pay for.)  Same code, same arithmetic as the separate kernels: bit-identical results.
template <int G>
__global__ __launch_bounds__(256, 1) void wino_bwd_pair_kernel(const Args pc, const WGArgs pw, const int nconv) {
  if ((int)blockIdx.x < nconv) wino_conv_body<G, false, 0>(pc, (int)blockIdx.x, nconv);
  else wino_wgrad_body<G, false>(pw, (int)blockIdx.x - nconv);
}

static bool wgrad_class(int H, int W, int* G, int* ipc) {
  if (H != W || !pow2(H) || H < 4) return false;
  *G = W >= 16 ? 8 : W == 8 ? 4 : 2;
  *ipc = W >= 16 ? 1 : W == 8 ? 2 : 4;
  return true;
}

}  // namespace lgmwino

bool lgm_wino_wgrad_supported(const LgmConvGeom* g) {
  using namespace lgmwino;
  int G, ipc;
  if (!(g->KH == 3 && g->KW == 3 && g->stride == 1 && g->pad == 1)) return false;
  if (g->Nw % 64 != 0 || g->Cw % 64 != 0) return false;
  if (!wgrad_class(g->H, g->W, &G, &ipc)) return false;
  if (g->B % ipc != 0) return false;
  const long chunks = (long)(g->B / ipc) * (g->H / 4) * (g->W >= 16 ? g->W / 16 : 1) * 2;
  return chunks >= 2;
}

// splits >= 2 always (the kernel only writes slabs); cps = chunks per split; budget = workgroups of one round
static void wino_wgrad_plan_budget(const LgmConvGeom* g, long budget, int* splits, int* cps, int* total_chunks);
void lgm_wino_wgrad_plan(const LgmConvGeom* g, int* splits, int* cps, int* total_chunks) {
  wino_wgrad_plan_budget(g, lgm_cu_budget(), splits, cps, total_chunks);
}
static void wino_wgrad_plan_budget(const LgmConvGeom* g, long budget, int* splits, int* cps, int* total_chunks) {
  using namespace lgmwino;
  int G, ipc;
  wgrad_class(g->H, g->W, &G, &ipc);
  const long chunks = (long)(g->B / ipc) * (g->H / 4) * (g->W >= 16 ? g->W / 16 : 1) * 2;
  const long blocks = (long)(g->Nw / 64) * (g->Cw / 64);
  // One workgroup per CU (4 waves = the CU's four SIMDs): 258 workgroups take twice as long as 256.  Pick the split
  // count that minimises  rounds of 256 workgroups x (chunks per workgroup + ~3 chunks of prologue / epilogue);
  // at least two chunks per workgroup and two slabs; ties go to fewer slabs.
  long smax = chunks / 2 < budget ? chunks / 2 : budget;
  if (smax < 2) smax = 2;
  long s = 2, best = -1;
  for (long c = 2; c <= smax; ++c) {
    const long rounds = (blocks * c + budget - 1) / budget;
    const long cost = rounds * ((chunks + c - 1) / c + 3);
    if (best < 0 || cost < best) {
      best = cost;
      s = c;
    }
  }
  long per = (chunks + s - 1) / s;
  s = (chunks + per - 1) / per;
  *splits = (int)s;
  *cps = (int)per;
  *total_chunks = (int)chunks;
}

static void wino_wgrad_prepare(const LgmConvGeom* g, const float* y, long y_pitch, const float* x, long x_pitch, float* out,
                               int bias, long slab, int splits, int cps, int total_chunks, lgmwino::WGArgs& p) {
  p = lgmwino::WGArgs{};
  p.y = y; p.x = x; p.out = out; p.bias = bias; p.slab = slab; p.y_pitch = y_pitch; p.x_pitch = x_pitch;
  p.B = g->B; p.H = g->H; p.W = g->W; p.Nw = g->Nw; p.Cw = g->Cw;
  p.tiles_c = g->Cw / 64;
  p.splits = splits; p.cps = cps; p.total_chunks = total_chunks;
  p.rpn = g->H / 4;
  p.cgn = g->W >= 16 ? g->W / 16 : 1;
}

int lgm_wino_wgrad_launch(const LgmConvGeom* g, const float* y, long y_pitch, const float* x, long x_pitch, float* out,
                          int bias, long slab, int splits, int cps, int total_chunks, hipStream_t s) {
  using namespace lgmwino;
  WGArgs p;
  wino_wgrad_prepare(g, y, y_pitch, x, x_pitch, out, bias, slab, splits, cps, total_chunks, p);
  int G, ipc;
  wgrad_class(g->H, g->W, &G, &ipc);
  const unsigned nblocks = (unsigned)((g->Nw / 64) * (g->Cw / 64) * splits);
  const size_t smem = (size_t)3 * WBUF * sizeof(float);
#define LGM_WGL(GG)                                                                                              \
  do {                                                                                                           \
    auto kern = wino_wgrad_kernel<GG>;                                                                           \
    static bool attr = false;                                                                                    \
    if (!attr) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)smem);                                                                      \
      attr = true;                                                                                               \
    }                                                                                                            \
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), smem, s, p);                                              \
  } while (0)
  lgm_note_kernel(G == 8 ? LGM_KNAME("lgmwino::wino_wgrad_kernel<8, false>") : G == 4 ? LGM_KNAME("lgmwino::wino_wgrad_kernel<4, false>")
                                                                          : LGM_KNAME("lgmwino::wino_wgrad_kernel<2, false>"));
  p.dbg = (long long*)lgm_wino_debug_buffer;
  if (p.dbg && G == 8) {
    auto kern = wino_wgrad_kernel<8, true>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), smem, s, p);
  } else if (G == 8) LGM_WGL(8);
  else if (G == 4) LGM_WGL(4);
  else LGM_WGL(2);
#undef LGM_WGL
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

// ---- two layers' weight gradients in one launch (deferred slab reduction only) -------------------------------------
static bool wgrad2_operands_ok(const LgmConvGeom* g, const float* y, long y_pitch, const float* x, long x_pitch,
                               const float* gw, const float* gbias) {
  return y && x && gw && y_pitch % 4 == 0 && x_pitch % 4 == 0 && y_pitch >= g->Nw && x_pitch >= g->Cw && lgm_aligned16(y) &&
         lgm_aligned16(x) && lgm_aligned16(gw) && (!gbias || lgm_aligned16(gbias)) &&
         ((long)g->B * g->H * g->W + g->W + 1) * x_pitch < (1L << 29) && (long)g->B * g->H * g->W * y_pitch < (1L << 29);
}

// F(4x4,3x3) weight gradient (csrc/winograd4_wgrad.hip): pairs of layers that both take it share a launch of THAT kernel
bool lgm_wino4_wgrad_use(const LgmConvGeom* g);
void lgm_wino4_wgrad_plan(const LgmConvGeom* g, long budget, int* splits, int* gps, int* total_groups);
int lgm_wino4_wgradn_launch(int n, const LgmConvGeom* const* gs, const float* const* ys, const long* yps,
                            const float* const* xs, const long* xps, float* const* outs, const int* biases, const long* slabs,
                            const int* splits, const int* gpss, const int* totals, hipStream_t s);
// F(4x4) weight gradients of 2 ... 4 layers in one launch (LGM_W4W_GROUP bounds n, default 4): every layer takes the F(4x4)
// kernel and still gets at least two slabs' worth of its 64 x 32-channel blocks out of the 256 workgroups (the 512-channel
// layers of the 64 x 64 configuration do not fit side by side: they take the F(2x2) grouped launch or go alone)
static bool wgradn_use4(int n, const LgmConvGeom* const* gs) {
  static const int nmax = getenv("LGM_W4W_GROUP") ? atoi(getenv("LGM_W4W_GROUP")) : 8;
  if (n < 2 || n > 8 || n > nmax) return false;
  long need = 0;
  double phases = 0.0;                 // tile groups x channel blocks of the whole group = phases the chip's workgroups share
  for (int k = 0; k < n; ++k) {
    if (!lgm_wino4_wgrad_use(gs[k])) return false;
    const long blocks = (long)(gs[k]->Nw / 64) * (gs[k]->Cw / 32);
    need += 2L * blocks;
    phases += (double)gs[k]->B * gs[k]->H * gs[k]->W / 64.0 * (double)blocks;
  }
  // More than two layers share a launch only while a workgroup's run stays short: past ~80 phases per workgroup the
  // prologue / epilogue / slab savings are used up and the longer tail costs more (32 x 32 maps at B = 128: four layers =
  // 64 phases, 10.03 -> 10.01 ms per step; 64 x 64 maps at B = 64: four layers = 128 phases, 16.78 -> 17.26 ms; eight layers
  // at 32 x 32 = 128 phases: 10.21 ms).  LGM_W4W_MAX_PHASES: tuning knob.
  static const double max_phases = getenv("LGM_W4W_MAX_PHASES") ? atof(getenv("LGM_W4W_MAX_PHASES")) : 80.0;
  if (n > 2 && phases / (double)lgm_cu_budget() > max_phases) return false;
  return need <= lgm_cu_budget();
}
static void wgrad4_budgets(int n, const LgmConvGeom* const* gs, long* budget) {     // blocks of 64 x 32 channels, shares by work
  double w[8], tot = 0;
  long mn[8], left = lgm_cu_budget();
  for (int k = 0; k < n; ++k) {
    w[k] = (double)gs[k]->B * gs[k]->H * gs[k]->W * gs[k]->Nw * gs[k]->Cw;
    tot += w[k];
    mn[k] = 2L * (gs[k]->Nw / 64) * (gs[k]->Cw / 32);
  }
  for (int k = 0; k < n; ++k) {
    long b = (long)((double)lgm_cu_budget() * w[k] / tot + (n == 2 && k == 0 ? 0.5 : 0.0));
    if (b < mn[k]) b = mn[k];
    budget[k] = b;
    left -= b;
  }
  int big = 0;
  for (int k = 1; k < n; ++k)
    if (budget[k] - mn[k] > budget[big] - mn[big]) big = k;
  budget[big] += left;
  if (budget[big] < mn[big]) budget[big] = mn[big];
}

// 1 when the grouped launch takes these n (2 ... 4) layers: all run the Winograd weight-gradient kernel of the same map
// class, and each still gets at least two slabs out of its share of the chip
static bool wgradn_supported(int n, const LgmConvGeom* const* gs) {
  using namespace lgmwino;
  if (n < 2 || n > 8) return false;
  if (n > 4) {                       // five ... eight layers: the F(4x4) kernel only
    for (int k = 0; k < n; ++k)
      if (!gs[k]) return false;
    return wgradn_use4(n, gs);
  }
  int G0 = 0;
  long need = 0;
  // layers that would each take the F(4x4) kernel but do not fit side by side (wgradn_use4: the 512-channel layers of the
  // 64 x 64 configuration) go alone on that kernel rather than together on the F(2x2) one: 17.24 -> 17.11 ms per step
  // (A/B: LGM_W4W_ALONE=0)
  static const bool alone4 = getenv("LGM_W4W_ALONE") ? atoi(getenv("LGM_W4W_ALONE")) != 0 : true;
  if (alone4 && !wgradn_use4(n, gs)) {
    bool all4 = true;
    for (int k = 0; k < n; ++k) all4 = all4 && gs[k] && lgm_wino4_wgrad_use(gs[k]);
    if (all4) return false;
  }
  for (int k = 0; k < n; ++k) {
    if (!gs[k] || !lgm_wino_wgrad_supported(gs[k])) return false;
    int G, ipc;
    wgrad_class(gs[k]->H, gs[k]->W, &G, &ipc);
    if (k == 0) G0 = G;
    else if (G != G0) return false;
    need += 2L * (gs[k]->Nw / 64) * (gs[k]->Cw / 64);
  }
  return need <= lgm_cu_budget();
}

// the chip's 256 workgroups are shared in proportion to the layers' MFMA work, at least two slabs' worth each
static void wgradn_budgets(int n, const LgmConvGeom* const* gs, long* budget) {
  double w[4], tot = 0;
  long mn[4], left = lgm_cu_budget();
  for (int k = 0; k < n; ++k) {
    w[k] = (double)gs[k]->B * gs[k]->H * gs[k]->W * gs[k]->Nw * gs[k]->Cw;
    tot += w[k];
    mn[k] = 2L * (gs[k]->Nw / 64) * (gs[k]->Cw / 64);
  }
  for (int k = 0; k < n; ++k) {
    long b = (long)((double)lgm_cu_budget() * w[k] / tot);
    if (b < mn[k]) b = mn[k];
    budget[k] = b;
    left -= b;
  }
  // rounding leftovers (or a deficit from the minimums) go to / come from the largest share
  int big = 0;
  for (int k = 1; k < n; ++k)
    if (budget[k] - mn[k] > budget[big] - mn[big]) big = k;
  budget[big] += left;
  if (budget[big] < mn[big]) budget[big] = mn[big];
}

extern "C" int64_t lgm_conv3x3_wino_wgradn_supported(int n, const LgmConvGeom* const* geoms) {
  return (geoms && wgradn_supported(n, geoms)) ? 1 : 0;
}
extern "C" int64_t lgm_conv3x3_wino_wgrad2_supported(const LgmConvGeom* ga, const LgmConvGeom* gb) {
  const LgmConvGeom* gs[2] = {ga, gb};
  return wgradn_supported(2, gs) ? 1 : 0;
}

// out[k] = bytes of slab workspace layer k needs in the grouped launch (its split count is planned against a share of
// the chip, so it differs from lgm_conv_wgrad_workspace's)
extern "C" int lgm_conv3x3_wino_wgradn_workspaces(int n, const LgmConvGeom* const* geoms, int64_t* out) {
  LGM_REQUIRE(geoms && out && wgradn_supported(n, geoms), "conv3x3_wino_wgradn_workspaces: unsupported group of layers");
  if (wgradn_use4(n, geoms)) {
    long b4[8];
    wgrad4_budgets(n, geoms, b4);
    for (int k = 0; k < n; ++k) {
      int splits, gps, total;
      lgm_wino4_wgrad_plan(geoms[k], b4[k], &splits, &gps, &total);
      out[k] = (int64_t)splits * ((int64_t)geoms[k]->Nw * 9 * geoms[k]->Cw + geoms[k]->Nw) * (int64_t)sizeof(float);
    }
    return LGM_OK;
  }
  long bud[4];
  wgradn_budgets(n, geoms, bud);
  for (int k = 0; k < n; ++k) {
    int splits, cps, total;
    wino_wgrad_plan_budget(geoms[k], bud[k], &splits, &cps, &total);
    out[k] = (int64_t)splits * ((int64_t)geoms[k]->Nw * 9 * geoms[k]->Cw + geoms[k]->Nw) * (int64_t)sizeof(float);
  }
  return LGM_OK;
}
extern "C" int lgm_conv3x3_wino_wgrad2_workspaces(const LgmConvGeom* ga, const LgmConvGeom* gb, int64_t* out) {
  const LgmConvGeom* gs[2] = {ga, gb};
  return lgm_conv3x3_wino_wgradn_workspaces(2, gs, out);
}

extern "C" int lgm_conv3x3_wino_wgradn(int n, const LgmWgradItem* it, void* stream) {
  using namespace lgmwino;
  LGM_REQUIRE(it && n >= 2 && n <= 8, "conv3x3_wino_wgradn: 2 ... 8 layers expected");
  const LgmConvGeom* gs[8];
  for (int k = 0; k < n; ++k) gs[k] = it[k].g;
  LGM_REQUIRE(wgradn_supported(n, gs), "conv3x3_wino_wgradn: unsupported group of layers");
  for (int k = 0; k < n; ++k)
    LGM_REQUIRE(it[k].desc && it[k].ws && lgm_aligned16(it[k].ws) &&
                wgrad2_operands_ok(it[k].g, it[k].y, it[k].y_pitch, it[k].x, it[k].x_pitch, it[k].gw, it[k].gbias),
                "conv3x3_wino_wgradn: layer %d: 16-byte aligned operands with pitch %% 4 == 0 inside 32-bit offsets expected", k);
  if (wgradn_use4(n, gs)) {                       // all layers on the F(4x4) kernel: one launch of that kernel
    long b4[8];
    wgrad4_budgets(n, gs, b4);
    const float* ys[8];
    const float* xs[8];
    long yps[8], xps[8], slabs[8];
    float* outs[8];
    int biases[8], splits[8], gpss[8], totals[8];
    for (int k = 0; k < n; ++k) {
      const LgmConvGeom* g = gs[k];
      lgm_wino4_wgrad_plan(g, b4[k], &splits[k], &gpss[k], &totals[k]);
      const long n_w = (long)g->Nw * 9 * g->Cw;
      slabs[k] = n_w + g->Nw;
      LGM_REQUIRE(it[k].ws_bytes >= (int64_t)splits[k] * slabs[k] * (int64_t)sizeof(float),
                  "conv3x3_wino_wgradn: workspace %d too small", k);
      ys[k] = it[k].y; xs[k] = it[k].x; yps[k] = it[k].y_pitch; xps[k] = it[k].x_pitch;
      outs[k] = (float*)it[k].ws; biases[k] = it[k].gbias ? 1 : 0;
      union { float f; int64_t i; } bbits;
      bbits.i = 0;
      bbits.f = it[k].beta;
      int64_t* d = it[k].desc;
      d[0] = (int64_t)(uintptr_t)it[k].ws; d[1] = slabs[k]; d[2] = (int64_t)(uintptr_t)it[k].gw; d[3] = n_w;
      d[4] = (int64_t)(uintptr_t)it[k].gbias; d[5] = it[k].gbias ? g->Nw : 0; d[6] = splits[k]; d[7] = bbits.i;
    }
    return lgm_wino4_wgradn_launch(n, gs, ys, yps, xs, xps, outs, biases, slabs, splits, gpss, totals, (hipStream_t)stream);
  }
  int G, ipc;
  wgrad_class(gs[0]->H, gs[0]->W, &G, &ipc);
  long bud[4];
  wgradn_budgets(n, gs, bud);
  WGArgs pp[4];
  unsigned nb[4] = {0, 0, 0, 0};
  for (int k = 0; k < n; ++k) {
    const LgmConvGeom* g = gs[k];
    int splits, cps, total;
    wino_wgrad_plan_budget(g, bud[k], &splits, &cps, &total);
    const long n_w = (long)g->Nw * 9 * g->Cw, slab = n_w + g->Nw;
    LGM_REQUIRE(it[k].ws_bytes >= (int64_t)splits * slab * (int64_t)sizeof(float), "conv3x3_wino_wgradn: workspace %d too small", k);
    wino_wgrad_prepare(g, it[k].y, it[k].y_pitch, it[k].x, it[k].x_pitch, (float*)it[k].ws, it[k].gbias ? 1 : 0, slab, splits,
                       cps, total, pp[k]);
    nb[k] = (unsigned)((g->Nw / 64) * (g->Cw / 64) * splits);
    union { float f; int64_t i; } bbits;
    bbits.i = 0;
    bbits.f = it[k].beta;
    int64_t* d = it[k].desc;
    d[0] = (int64_t)(uintptr_t)it[k].ws; d[1] = slab; d[2] = (int64_t)(uintptr_t)it[k].gw; d[3] = n_w;
    d[4] = (int64_t)(uintptr_t)it[k].gbias; d[5] = it[k].gbias ? g->Nw : 0; d[6] = splits; d[7] = bbits.i;
  }
  for (int k = n; k < 4; ++k) pp[k] = pp[n - 1];           // never reached: its block range is empty
  const size_t smem = (size_t)3 * WBUF * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  const int e0 = (int)nb[0], e1 = e0 + (int)nb[1], e2 = e1 + (int)nb[2];
  const unsigned grid = nb[0] + nb[1] + nb[2] + nb[3];
#define LGM_WGN(GG)                                                                                              \
  do {                                                                                                           \
    static bool attr2 = false, attr4 = false;                                                                    \
    if (n == 2) {                                                                                                \
      auto kern = wino_wgrad2_kernel<GG>;                                                                        \
      if (!attr2) {                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        attr2 = true;                                                                                            \
      }                                                                                                          \
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, pp[0], pp[1], e0);                                \
    } else {                                                                                                     \
      auto kern = wino_wgrad4_kernel<GG>;                                                                        \
      if (!attr4) {                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        attr4 = true;                                                                                            \
      }                                                                                                          \
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, pp[0], pp[1], pp[2], pp[3], e0, e1, e2);          \
    }                                                                                                            \
  } while (0)
  if (n == 2)
    lgm_note_kernel(G == 8 ? LGM_KNAME("lgmwino::wino_wgrad2_kernel<8>") : G == 4 ? LGM_KNAME("lgmwino::wino_wgrad2_kernel<4>") : LGM_KNAME("lgmwino::wino_wgrad2_kernel<2>"));
  else
    lgm_note_kernel(G == 8 ? LGM_KNAME("lgmwino::wino_wgrad4_kernel<8>") : G == 4 ? LGM_KNAME("lgmwino::wino_wgrad4_kernel<4>") : LGM_KNAME("lgmwino::wino_wgrad4_kernel<2>"));
  if (G == 8) LGM_WGN(8);
  else if (G == 4) LGM_WGN(4);
  else LGM_WGN(2);
#undef LGM_WGN
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_conv3x3_wino_wgrad2(const LgmConvGeom* ga, const float* ya, int64_t ya_pitch, const float* xa,
                                       int64_t xa_pitch, float* gwa, float* gba, float beta_a, void* wsa, int64_t wsa_bytes,
                                       int64_t* desca, const LgmConvGeom* gb, const float* yb, int64_t yb_pitch,
                                       const float* xb, int64_t xb_pitch, float* gwb, float* gbb, float beta_b, void* wsb,
                                       int64_t wsb_bytes, int64_t* descb, void* stream) {
  LgmWgradItem it[2] = {{ga, ya, ya_pitch, xa, xa_pitch, gwa, gba, beta_a, wsa, wsa_bytes, desca},
                        {gb, yb, yb_pitch, xb, xb_pitch, gwb, gbb, beta_b, wsb, wsb_bytes, descb}};
  return lgm_conv3x3_wino_wgradn(2, it, stream);
}

// Joint plan of the backward pair.  Standing alone, each kernel splits its reduction until ITS grid fills the 256 CUs;
// side by side that is two rounds again.  Here the two split counts are chosen together: the launch takes about
//   max( (sum over both grids of blocks x time per block) / 256 CUs,  the longest single block )
// in units of one phase / chunk (64 MFMAs per wave, ~2.1 us), plus what the splits cost afterwards: the input
// gradient's partial planes (written, then read by the reducer or the consuming GroupNorm) and the weight gradient's
// slabs (written by the kernel, read by the batched slab reducer).
struct WinoPairPlan {
  int csplits, wsplits, cps, total_chunks;
};
static WinoPairPlan wino_pair_plan_search(const LgmConvGeom* g, bool fused);
WinoPairPlan lgm_wino_pair_plan(const LgmConvGeom* g, bool fused) {
  // the search below costs ~1 ms: once per geometry and thread
  struct Key {
    int B, H, W, C, N, f;
    bool operator==(const Key& o) const { return B == o.B && H == o.H && W == o.W && C == o.C && N == o.N && f == o.f; }
  };
  thread_local Key keys[64];
  thread_local WinoPairPlan plans[64];
  thread_local int n = 0;
  const Key k{g->B, g->H, g->W, g->Cw, g->Nw, fused ? 1 : 0};
  for (int i = 0; i < n; ++i)
    if (keys[i] == k) return plans[i];
  const WinoPairPlan p = wino_pair_plan_search(g, fused);
  if (n < 64) {
    keys[n] = k;
    plans[n] = p;
    ++n;
  }
  return p;
}
static WinoPairPlan wino_pair_plan_search(const LgmConvGeom* g, bool fused) {
  using namespace lgmwino;
  WinoPairPlan best{1, 2, 1, 2};
  int TTH, TTW, NI, G, ipc;
  plan_unit(g->H, g->W, &TTH, &TTW, &NI);
  wgrad_class(g->H, g->W, &G, &ipc);
  const int gc = g->Nw, oc = g->Cw;                       // input gradient: reduces over Nw, produces Cw
  const long base = (long)(g->B / NI) * (g->H / (2 * TTH)) * (g->W / (2 * TTW)) * (oc / 64);
  const int phases = gc / KC;
  long cmax = phases / 2 < 32 ? phases / 2 : 32;
  if (cmax < 1 || base >= 1024) cmax = 1;
  const double per_split = (fused ? 6.0 : 8.0) * (double)g->B * g->H * g->W * oc / 3.0e12 / 2.1e-6;
  const long chunks = (long)(g->B / ipc) * (g->H / 4) * (g->W >= 16 ? g->W / 16 : 1) * 2;
  const long blocks = (long)(g->Nw / 64) * (g->Cw / 64);
  long smax = chunks / 2 < 256 ? chunks / 2 : 256;
  if (smax < 2) smax = 2;
  // per slab: read back by the batched slab reducer at HBM rate (its write rides in the kernel's epilogue)
  static const double k_slab = getenv("LGM_PLAN_SLAB") ? atof(getenv("LGM_PLAN_SLAB")) : 1.0;     // tuning knobs (A/B runs)
  static const double k_tc = getenv("LGM_PLAN_TC") ? atof(getenv("LGM_PLAN_TC")) : 1.5;
  static const double k_tw = getenv("LGM_PLAN_TW") ? atof(getenv("LGM_PLAN_TW")) : 3.0;
  static const double k_wph = getenv("LGM_PLAN_WPH") ? atof(getenv("LGM_PLAN_WPH")) : 1.0;
  const double slab_cost = k_slab * 4.0 * ((double)g->Nw * 9 * g->Cw + g->Nw) / 4.5e12 / 2.1e-6;
  // makespan of n1 blocks of t1 followed by n2 blocks of t2 on 256 CUs, one block per CU, dispatched in order
  auto makespan = [](long n1, double t1, long n2, double t2) {
    const long P = lgm_cu_budget();
    const long r1 = n1 / P, m1 = n1 % P;
    double ta = (double)r1 * t1;                 // the P - m1 CUs without a block of the last conv round
    double tb = (double)(r1 + (m1 ? 1 : 0)) * t1;
    const long ca = P - m1, cb = m1;
    double end = tb > ta ? tb : ta;
    if (n1 == 0) end = 0.0;
    long left = n2;
    while (left > 0) {                           // next free group takes blocks; merge of two arithmetic sequences
      if (cb == 0 || ta <= tb) {
        const long take = left < ca ? left : ca;
        ta += t2;
        left -= take;
        if (ta > end) end = ta;
      } else {
        const long take = left < cb ? left : cb;
        tb += t2;
        left -= take;
        if (tb > end) end = tb;
      }
    }
    return end;
  };
  double bestc = 1e30;
  for (long c = 1; c <= cmax; ++c) {
    const long pps = (phases + c - 1) / c;
    if ((phases + pps - 1) / pps != c) continue;
    const double tc = (double)pps + k_tc;
    const double conv_after = (c > 1 ? (fused ? 0.2 : 2.4) : 0.0) + per_split * (double)(c - 1);
    for (long w = 2; w <= smax; ++w) {
      const long per = (chunks + w - 1) / w;
      if ((chunks + per - 1) / per != w) continue;
      const double tw = k_wph * (double)per + k_tw;
      const double cost = makespan(base * c, tc, blocks * w, tw) + conv_after + slab_cost * (double)w;
      if (cost < bestc - 1e-9) {
        bestc = cost;
        best = WinoPairPlan{(int)c, (int)w, (int)per, (int)chunks};
      }
    }
  }
  return best;
}

// One launch for the input gradient and the weight gradient of a 3x3 layer (wino_bwd_pair_kernel).  The split-K
// reducer of the input gradient (when it split and the consumer does not sum the planes itself) follows as usual.
// slab_stride = floats between two slabs; the slabs (lgm_wino_pair_plan(...).wsplits of them) go to `slabs`.
int lgm_wino_pair_launch(const LgmConvGeom* g, const float* gy, long gy_pitch, const float* x, long x_pitch,
                         const float* u_b, const float* res, long res_pitch, float* gx, long gx_pitch, void* dws,
                         long dws_bytes, int64_t* partial, float* slabs, int bias, long slab, hipStream_t s) {
  using namespace lgmwino;
  const WinoPairPlan pl = lgm_wino_pair_plan(g, partial != nullptr);
  Args pc;
  unsigned nconv;
  int TTW;
  wino_prepare(g, 1, gy, gy_pitch, u_b, nullptr, res, res_pitch, gx, gx_pitch, dws, dws_bytes, partial != nullptr, pc,
               &nconv, &TTW, pl.csplits);
  pc.dbg = nullptr;
  pc.dbg_mode = 0;
  WGArgs pw;
  wino_wgrad_prepare(g, gy, gy_pitch, x, x_pitch, slabs, bias, slab, pl.wsplits, pl.cps, pl.total_chunks, pw);
  pw.dbg = nullptr;
  int G, ipc;
  wgrad_class(g->H, g->W, &G, &ipc);
  if (G != TTW) {
    lgm_set_error("wino_pair: map classes of the two kernels differ (%d vs %d)", G, TTW);
    return LGM_ERR_UNSUPPORTED;
  }
  const unsigned nw = (unsigned)((g->Nw / 64) * (g->Cw / 64) * pl.wsplits);
  const size_t smem_w = (size_t)3 * WBUF * sizeof(float);
#define LGM_PAIR(GG)                                                                                             \
  do {                                                                                                           \
    auto kern = wino_bwd_pair_kernel<GG>;                                                                        \
    size_t smem = ((size_t)12 * Geo<GG>::RPLANE) * sizeof(float);                                                \
    if (smem_w > smem) smem = smem_w;                                                                            \
    static bool attr = false;                                                                                    \
    if (!attr) {                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)smem);                                                                      \
      attr = true;                                                                                               \
    }                                                                                                            \
    hipLaunchKernelGGL(kern, dim3(nconv + nw), dim3(256), smem, s, pc, pw, (int)nconv);                          \
  } while (0)
  lgm_note_kernel(G == 8 ? LGM_KNAME("lgmwino::wino_bwd_pair_kernel<8>") : G == 4 ? LGM_KNAME("lgmwino::wino_bwd_pair_kernel<4>")
                                                                        : LGM_KNAME("lgmwino::wino_bwd_pair_kernel<2>"));
  if (G == 8) LGM_PAIR(8);
  else if (G == 4) LGM_PAIR(4);
  else LGM_PAIR(2);
#undef LGM_PAIR
  const long M = (long)g->B * g->H * g->W;
  if (partial) {
    partial[0] = pc.splits;
    partial[1] = pc.ws_stride;
    LGM_LAUNCH_CHECK();
    return LGM_OK;
  }
  if (pc.splits > 1)
    return lgm_splitk_reduce_launch(pc.ws, pc.ws_stride, pc.splits, nullptr, res, res_pitch, gx, gx_pitch, M, pc.N, s);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
