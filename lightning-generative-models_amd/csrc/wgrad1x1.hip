// Weight gradient of 1x1 convolutions (to_qkv / to_out / res_conv / Downsample convs of the UNet,
// reference ddpm.py:103,187,213,215,252,253):
//     gw[n][c] = sum_p y[p][n] x[p][c],     gbias[n] = sum_p y[p][n]
// A streaming reduction over the pixels with a tiny output, so the floor is reading x and y once.
// One persistent workgroup per CU owns an (NB x KB) block of gw (NB, KB = 64 or 128: usually ALL
// of gw, so nothing is read twice) and a contiguous range of pixel rows, walked in 64-row chunks:
//   * chunk c+1 is fetched with raw buffer loads (32-bit lane offset + scalar chunk offset; rows
//     past the tensor fall out of the descriptor's range and fetch nothing) issued one per second
//     MFMA step of chunk c, held in registers, committed to the other LDS buffer at the chunk's
//     end: one barrier per 128 MFMAs of a wave;
//   * both operands are pixel-major in LDS, so every MFMA fragment (k = pixel) is a ds_read_b32 of
//     32 consecutive floats, read one step ahead of its use;
//   * partial results go to per-split slabs in the caller's workspace; the deterministic slab
//     reduction is the shared one (conv_igemm.hip), immediate or batched over many layers.
#include "lgm_common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct PArgs {
  const float* y;   // [P, Nw] rows
  const float* x;   // [P, Cw] rows
  float* out;       // gw or workspace slabs
  float* bias_out;  // or null
  float beta;
  long slab, y_pitch, x_pitch;
  int P, Nw, Cw;
  int tiles_n, tiles_k, splits, chunks_per_split, total_chunks;
};

// The kernel body as a device function of (arguments, logical block id): its own launch (wgrad1x1_kernel) or one block
// range of wgrad1x1_group_kernel (several layers' weight gradients in ONE launch).
template <int TN, int TK>   // wave tile (32 TN) x (32 TK); waves 2 (n) x 2 (k); block tile NB x KB
__device__ __forceinline__ void wgrad1x1_body(const PArgs& p, const int bidx) {
  constexpr int NB = 64 * TN, KB = 64 * TK;
  constexpr int R = 64;                         // pixel rows per chunk
  constexpr int NLY = NB / 16, NLX = KB / 16;   // 16-byte loads per thread and chunk
  constexpr int YROWS = 1024 / NB, XROWS = 1024 / KB;   // rows covered by one pass of 256 threads
  extern __shared__ __align__(16) float smem[];
  float* Ys = smem;                 // [2][R][NB]
  float* Xs = smem + 2 * R * NB;    // [2][R][KB]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wn = wid >> 1, wk = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;

  int bid = bidx;
  const int split = bid % p.splits;
  bid /= p.splits;
  const int tk = bid % p.tiles_k, tn = bid / p.tiles_k;
  const int n0 = tn * NB, c0 = tk * KB;
  const int ch_begin = split * p.chunks_per_split;
  const int ch_end = min(p.total_chunks, ch_begin + p.chunks_per_split);

  f32x16 acc[TN][TK];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const bool do_bias = p.bias_out != nullptr && tk == 0;
  float bsum = 0.f;

  // descriptors pinned to scalar registers; offsets stay below 2^31 (checked by the host)
  auto make_rsrc = [](const float* base, unsigned nrec) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(nrec), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsrc_y = make_rsrc(p.y + n0, (unsigned)(((long)p.P * p.y_pitch - n0) * 4));
  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(p.x + c0, (unsigned)(((long)p.P * p.x_pitch - c0) * 4));
  const int yrow = tid / (NB / 4), ycol = (tid % (NB / 4)) * 4;
  const int xrow = tid / (KB / 4), xcol = (tid % (KB / 4)) * 4;
  const unsigned voff_y = (unsigned)(yrow * (int)p.y_pitch + ycol) * 4u;
  const unsigned voff_x = (unsigned)(xrow * (int)p.x_pitch + xcol) * 4u;
  const unsigned ypass = (unsigned)(YROWS * (int)p.y_pitch) * 4u, xpass = (unsigned)(XROWS * (int)p.x_pitch) * 4u;
  u32x4 ry[NLY], rx[NLX];
  unsigned soff_y = 0, soff_x = 0;
  auto chunk_base = [&](int ch, bool exists) {
    // a chunk past this block's range: offsets beyond both descriptors, the loads fetch nothing
    soff_y = exists ? (unsigned)(ch * R) * (unsigned)p.y_pitch * 4u : 0x80000000u;
    soff_x = exists ? (unsigned)(ch * R) * (unsigned)p.x_pitch * 4u : 0x80000000u;
  };
  auto load_one = [&](int idx) {   // idx is a constant after unrolling
    if (idx < NLY) ry[idx] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, voff_y, soff_y + (unsigned)idx * ypass, 0);
    else rx[idx - NLY] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, voff_x, soff_x + (unsigned)(idx - NLY) * xpass, 0);
  };
  auto commit = [&](int buf) {
    float* yd = Ys + buf * R * NB + yrow * NB + ycol;
    float* xd = Xs + buf * R * KB + xrow * KB + xcol;
#pragma unroll
    for (int i = 0; i < NLY; ++i) *reinterpret_cast<u32x4*>(yd + i * YROWS * NB) = ry[i];
#pragma unroll
    for (int i = 0; i < NLX; ++i) *reinterpret_cast<u32x4*>(xd + i * XROWS * KB) = rx[i];
  };

  if (ch_begin < ch_end) {
    chunk_base(ch_begin, true);
#pragma unroll
    for (int idx = 0; idx < NLY + NLX; ++idx) load_one(idx);
    commit(0);
  }
  __syncthreads();

  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const int buf = (ch - ch_begin) & 1;
    chunk_base(ch + 1, ch + 1 < ch_end);
    const float* ap = Ys + buf * R * NB + lh * NB + wn * 32 * TN + lr;
    const float* bp = Xs + buf * R * KB + lh * KB + wk * 32 * TK + lr;
    float a_cur[TN], b_cur[TK], a_nxt[TN], b_nxt[TK];
    auto read_step = [&](int s, float (&a)[TN], float (&b)[TK]) {
#pragma unroll
      for (int i = 0; i < TN; ++i) a[i] = ap[2 * s * NB + i * 32];
#pragma unroll
      for (int j = 0; j < TK; ++j) b[j] = bp[2 * s * KB + j * 32];
    };
    read_step(0, a_cur, b_cur);
#pragma unroll
    for (int s = 0; s < R / 2; ++s) {
      const bool ld = (s & 1) == 0 && s / 2 < NLY + NLX;
      read_step(s + 1 < R / 2 ? s + 1 : s, a_nxt, b_nxt);
      if (ld) load_one(s / 2);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i], b_cur[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, TN + TK, 0);   // next step's LDS reads first ...
      if (ld) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (TN * TK > 1) __builtin_amdgcn_sched_group_barrier(0x008, TN * TK - 1, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, TN * TK, 0);  // ... then this step's MFMAs
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) a_cur[i] = a_nxt[i];
#pragma unroll
      for (int j = 0; j < TK; ++j) b_cur[j] = b_nxt[j];
      if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // keep the scheduler's read-ahead short
    }
    if (do_bias) {   // column sums of this chunk's Y rows, four reads in flight
      constexpr int LANES = 256 / NB;          // row lanes per column
      const float* col = Ys + buf * R * NB + (tid / NB) * NB + (tid % NB);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 4
      for (int r = 0; r < R / LANES; r += 4) {
        s0 += col[(r + 0) * LANES * NB];
        s1 += col[(r + 1) * LANES * NB];
        s2 += col[(r + 2) * LANES * NB];
        s3 += col[(r + 3) * LANES * NB];
      }
      bsum += (s0 + s1) + (s2 + s3);
    }
    commit(buf ^ 1);     // that buffer was last read in the previous chunk, before its closing barrier
    __syncthreads();
  }

  // ---- epilogue: wave-private LDS transpose, 16-byte stores (8 full rows of 32 per instruction)
  float* out = p.out + (p.splits > 1 ? (long)split * p.slab : 0L);
  float* Ts = smem + wid * LGM_TS_FLOATS;          // the operand buffers are dead (barrier above)
  const bool acc_out = p.splits == 1 && p.beta != 0.f;
  // (two copies of the loop: a read-modify-write of gw inside it would make every tile wait for its
  // loads and, the vector-memory queue retiring in order, for the previous tile's stores)
  if (acc_out) {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TK; ++j) {
        const int cc = c0 + wk * 32 * TK + j * 32 + (lane & 7) * 4;
        f32x4 prev[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 32 * TN + i * 32 + (lane >> 3) + 8 * q;
          prev[q] = *reinterpret_cast<const f32x4*>(out + (long)n * p.Cw + cc);
        }
        lgm_wave_lds_sync();
        lgm_tile_to_lds(acc[i][j], Ts, lane);
        lgm_wave_lds_sync();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 32 * TN + i * 32 + (lane >> 3) + 8 * q;
          *reinterpret_cast<f32x4*>(out + (long)n * p.Cw + cc) = lgm_tile_row4(Ts, lane, q) + p.beta * prev[q];
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TK; ++j) {
        const int cc = c0 + wk * 32 * TK + j * 32 + (lane & 7) * 4;
        lgm_wave_lds_sync();
        lgm_tile_to_lds(acc[i][j], Ts, lane);
        lgm_wave_lds_sync();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 32 * TN + i * 32 + (lane >> 3) + 8 * q;
          *reinterpret_cast<f32x4*>(out + (long)n * p.Cw + cc) = lgm_tile_row4(Ts, lane, q);
        }
      }
  }
  if (do_bias) {
    constexpr int LANES = 256 / NB;
    __syncthreads();                 // every wave is done with its transpose scratch
    float* red = smem + 4 * LGM_TS_FLOATS;
    red[tid] = bsum;                 // [row lane][column]
    __syncthreads();
    if (tid < NB) {
      float v = red[tid];
#pragma unroll
      for (int l = 1; l < LANES; ++l) v += red[l * NB + tid];
      float* bo = p.bias_out + (p.splits > 1 ? (long)split * p.slab : 0L) + n0 + tid;
      if (acc_out) v += p.beta * bo[0];
      bo[0] = v;
    }
  }
}

template <int TN, int TK>
__global__ __launch_bounds__(256) void wgrad1x1_kernel(const PArgs p) {
  wgrad1x1_body<TN, TK>(p, (int)blockIdx.x);
}

// Two ... four layers with the same block tile in one grid: blocks [0, e0) run layer a, [e0, e1) layer b, [e1, e2) layer c,
// the rest layer d.  A stand-alone launch of this kernel is ~13 us of prologue, epilogue and ramp around a K loop that runs
// at the MFMA rate (tools/step listing: 27 us for 14 us of MFMAs at 128 -> 64 @ 32 x 32, B = 128); the weight gradient has no
// reader before the optimizer, so up to four layers wait for each other (GradCtx.queue_wgrad1x1) and share ONE of them -
// each on its share of the chip with a proportionally longer pixel range per workgroup (fewer slabs as well).
template <int TN, int TK>
__global__ __launch_bounds__(256) void wgrad1x1_group_kernel(const PArgs a, const PArgs b, const PArgs c, const PArgs d,
                                                             const int e0, const int e1, const int e2) {
  const int bid = (int)blockIdx.x;
  if (bid < e0) wgrad1x1_body<TN, TK>(a, bid);
  else if (bid < e1) wgrad1x1_body<TN, TK>(b, bid - e0);
  else if (bid < e2) wgrad1x1_body<TN, TK>(c, bid - e1);
  else wgrad1x1_body<TN, TK>(d, bid - e2);
}

inline int tile_of(int dim) { return dim % 128 == 0 ? 128 : 64; }

}  // namespace

void lgm_wgrad1x1_plan(const LgmConvGeom* g, int* splits, int* chunks_per_split);

// 1x1, stride 1, unpadded, channel counts in whole 64-blocks, whole 64-pixel chunks, 32-bit byte offsets
bool lgm_wgrad1x1_supported(const LgmConvGeom* g, long y_pitch, long x_pitch) {
  if (!(g->KH == 1 && g->KW == 1 && g->stride == 1 && g->pad == 0)) return false;
  if (g->Nw % 64 != 0 || g->Cw % 64 != 0) return false;
  const long P = (long)g->B * g->H * g->W;
  if (P % 64 != 0) return false;
  if (!((P + 64) * y_pitch < (1L << 29) && (P + 64) * x_pitch < (1L << 29))) return false;
  int splits, per;
  lgm_wgrad1x1_plan(g, &splits, &per);
  return per >= 4;     // shorter pixel ranges are all prologue: the tiled kernel does those better
}

void lgm_wgrad1x1_plan(const LgmConvGeom* g, int* splits, int* chunks_per_split) {
  const long P = (long)g->B * g->H * g->W;
  const int total = (int)(P / 64);
  if (total < 1) {      // not a case for this kernel (lgm_wgrad1x1_supported says no)
    *splits = 1;
    *chunks_per_split = 1;
    return;
  }
  const int units = (g->Nw / tile_of(g->Nw)) * (g->Cw / tile_of(g->Cw));
  int s = lgm_cu_budget() / units;
  if (s > total) s = total;
  if (s < 1) s = 1;
  const int per = lgm_cdiv(total, s);
  *chunks_per_split = per;
  *splits = lgm_cdiv(total, per);
}

int lgm_wgrad1x1_launch(const LgmConvGeom* g, const float* y, long y_pitch, const float* x, long x_pitch, float* out,
                        float* bias_out, float beta, long slab, int splits, int chunks_per_split, hipStream_t s) {
  PArgs p{};
  p.y = y; p.x = x; p.out = out; p.bias_out = bias_out; p.beta = beta; p.slab = slab;
  p.y_pitch = y_pitch; p.x_pitch = x_pitch;
  p.P = g->B * g->H * g->W; p.Nw = g->Nw; p.Cw = g->Cw;
  const int NB = tile_of(g->Nw), KB = tile_of(g->Cw);
  p.tiles_n = g->Nw / NB; p.tiles_k = g->Cw / KB;
  p.splits = splits; p.chunks_per_split = chunks_per_split; p.total_chunks = p.P / 64;
  const size_t smem = (size_t)2 * 64 * (NB + KB) * sizeof(float);
  const unsigned nblocks = (unsigned)(p.tiles_n * p.tiles_k * splits);
#define LGM_W1_LAUNCH(TNV, TKV)                                                                                        \
  do {                                                                                                                 \
    auto kern = wgrad1x1_kernel<TNV, TKV>;                                                                             \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    lgm_note_kernel(LGM_KNAME("wgrad1x1_kernel<" #TNV ", " #TKV ">"));                                                            \
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), smem, s, p);                                                    \
  } while (0)
  if (NB == 128 && KB == 128) LGM_W1_LAUNCH(2, 2);
  else if (NB == 128) LGM_W1_LAUNCH(2, 1);
  else if (KB == 128) LGM_W1_LAUNCH(1, 2);
  else LGM_W1_LAUNCH(1, 1);
#undef LGM_W1_LAUNCH
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

// ---- several layers in one launch --------------------------------------------------------------------------------------
// Plan: the chip's workgroup slots are dealt out in proportion to the layers' work (64-row chunks x weight blocks); a layer's
// share / its weight blocks = its split count.  false: not a group this kernel takes (2 ... 4 layers, each supported on its
// own, one block tile for all, every split at least four chunks long).
static bool wgrad1x1_group_plan(int n, const LgmConvGeom* const* gs, int* splits, int* per) {
  if (n < 2 || n > 4) return false;
  double work[4], tot = 0.0;
  int units[4], total[4];
  const int nb0 = tile_of(gs[0]->Nw), kb0 = tile_of(gs[0]->Cw);
  for (int k = 0; k < n; ++k) {
    const LgmConvGeom* g = gs[k];
    if (!g || !lgm_wgrad1x1_supported(g, g->Nw, g->Cw)) return false;
    if (tile_of(g->Nw) != nb0 || tile_of(g->Cw) != kb0) return false;
    units[k] = (g->Nw / nb0) * (g->Cw / kb0);
    total[k] = (int)(((long)g->B * g->H * g->W) / 64);
    work[k] = (double)total[k] * units[k];
    tot += work[k];
  }
  const int budget = lgm_cu_budget();
  long used = 0;
  for (int k = 0; k < n; ++k) {
    int s = (int)((double)budget * work[k] / tot) / units[k];
    if (s < 1) s = 1;
    if (s > total[k]) s = total[k];
    per[k] = lgm_cdiv(total[k], s);
    if (per[k] < 4) return false;
    splits[k] = lgm_cdiv(total[k], per[k]);
    used += (long)splits[k] * units[k];
  }
  return used <= budget;
}

extern "C" int64_t lgm_wgrad1x1_group_supported(int n, const LgmConvGeom* const* geoms) {
  if (!geoms || n < 2 || n > 4) return 0;
  int splits[4], per[4];
  return wgrad1x1_group_plan(n, geoms, splits, per) ? 1 : 0;
}

extern "C" int lgm_wgrad1x1_group_workspaces(int n, const LgmConvGeom* const* geoms, int64_t* out) {
  LGM_REQUIRE(geoms && out && n >= 2 && n <= 4, "wgrad1x1_group_workspaces: 2 ... 4 layers expected");
  int splits[4], per[4];
  LGM_REQUIRE(wgrad1x1_group_plan(n, geoms, splits, per), "wgrad1x1_group_workspaces: unsupported group of layers");
  for (int k = 0; k < n; ++k) {
    const int64_t slab = (int64_t)geoms[k]->Nw * geoms[k]->Cw + geoms[k]->Nw;
    out[k] = splits[k] > 1 ? (int64_t)splits[k] * slab * (int64_t)sizeof(float) : 16;
  }
  return LGM_OK;
}

extern "C" int lgm_wgrad1x1_group(int n, const LgmWgradItem* it, void* stream) {
  LGM_REQUIRE(it && n >= 2 && n <= 4, "wgrad1x1_group: 2 ... 4 layers expected");
  const LgmConvGeom* gs[4];
  for (int k = 0; k < n; ++k) gs[k] = it[k].g;
  int splits[4], per[4];
  LGM_REQUIRE(wgrad1x1_group_plan(n, gs, splits, per), "wgrad1x1_group: unsupported group of layers");
  PArgs pp[4];
  int nb[4] = {0, 0, 0, 0};
  const int NB = tile_of(gs[0]->Nw), KB = tile_of(gs[0]->Cw);
  for (int k = 0; k < n; ++k) {
    const LgmConvGeom* g = gs[k];
    LGM_REQUIRE(it[k].desc && it[k].y && it[k].x && it[k].gw && it[k].y_pitch % 4 == 0 && it[k].x_pitch % 4 == 0 &&
                    it[k].y_pitch >= g->Nw && it[k].x_pitch >= g->Cw && lgm_aligned16(it[k].y) && lgm_aligned16(it[k].x) &&
                    lgm_aligned16(it[k].gw) && (!it[k].gbias || lgm_aligned16(it[k].gbias)) &&
                    lgm_wgrad1x1_supported(g, it[k].y_pitch, it[k].x_pitch),
                "wgrad1x1_group: layer %d: 16-byte aligned operands with pitch %% 4 == 0 inside 32-bit offsets expected", k);
    const long n_w = (long)g->Nw * g->Cw, slab = n_w + g->Nw;
    PArgs& p = pp[k];
    p = PArgs{};
    p.y = it[k].y; p.x = it[k].x; p.beta = it[k].beta; p.slab = slab;
    p.y_pitch = it[k].y_pitch; p.x_pitch = it[k].x_pitch;
    p.P = g->B * g->H * g->W; p.Nw = g->Nw; p.Cw = g->Cw;
    p.tiles_n = g->Nw / NB; p.tiles_k = g->Cw / KB;
    p.splits = splits[k]; p.chunks_per_split = per[k]; p.total_chunks = p.P / 64;
    if (splits[k] > 1) {
      LGM_REQUIRE(it[k].ws && lgm_aligned16(it[k].ws) && it[k].ws_bytes >= (int64_t)splits[k] * slab * (int64_t)sizeof(float),
                  "wgrad1x1_group: workspace %d too small", k);
      p.out = (float*)it[k].ws;
      p.bias_out = it[k].gbias ? (float*)it[k].ws + n_w : nullptr;
    } else {
      p.out = it[k].gw;
      p.bias_out = it[k].gbias;
    }
    nb[k] = p.tiles_n * p.tiles_k * splits[k];
    union { float f; int64_t i; } bbits;
    bbits.i = 0;
    bbits.f = it[k].beta;
    int64_t* d = it[k].desc;
    d[0] = (int64_t)(uintptr_t)it[k].ws; d[1] = slab; d[2] = (int64_t)(uintptr_t)it[k].gw; d[3] = n_w;
    d[4] = (int64_t)(uintptr_t)it[k].gbias; d[5] = it[k].gbias ? g->Nw : 0; d[6] = splits[k]; d[7] = bbits.i;
  }
  for (int k = n; k < 4; ++k) pp[k] = pp[n - 1];           // never reached: its block range is empty
  const int e0 = nb[0], e1 = e0 + nb[1], e2 = e1 + nb[2];
  const unsigned nblocks = (unsigned)(e2 + nb[3]);
  const size_t smem = (size_t)2 * 64 * (NB + KB) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define LGM_W1G_LAUNCH(TNV, TKV)                                                                                       \
  do {                                                                                                                 \
    auto kern = wgrad1x1_group_kernel<TNV, TKV>;                                                                       \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    lgm_note_kernel(LGM_KNAME("wgrad1x1_group_kernel<" #TNV ", " #TKV ">"));                                            \
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), smem, s, pp[0], pp[1], pp[2], pp[3], e0, e1, e2);               \
  } while (0)
  if (NB == 128 && KB == 128) LGM_W1G_LAUNCH(2, 2);
  else if (NB == 128) LGM_W1G_LAUNCH(2, 1);
  else if (KB == 128) LGM_W1G_LAUNCH(1, 2);
  else LGM_W1G_LAUNCH(1, 1);
#undef LGM_W1G_LAUNCH
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
