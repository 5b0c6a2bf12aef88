// 1x1 convolution / linear layer with resident weights:  Y[M,N] = X[M,K] W[N,K]^T (+bias, +res),
// K = 64 / 128 / 192 / 256 (to_qkv / to_out / res_conv / Downsample convs of the UNet, reference
// ddpm.py:103,187,213,215,252,253, forward and -- through the transposed weight copy -- input gradient).
//
// One persistent workgroup per CU keeps an NB-column slice of W in LDS for its whole life (NB = 128
// for K <= 128, else 64) and streams 64-row tiles of X through a single LDS buffer:
//   * tile t+1 is fetched with raw buffer loads (32-bit lane offset + scalar tile offset; rows past
//     the block's range fall out of the descriptor and fetch nothing), one load per second group of
//     MFMAs of tile t, held in registers and committed once tile t's MFMAs are done;
//   * the weights are read from L2 once per workgroup (not once per row tile), X once per column
//     slice; Y of tile t-1 is written straight from the accumulator layout (full 128-byte row
//     segments per store) between the MFMAs of tile t.
#include "lgm_common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct SArgs {
  const float* x;     // [M, K] rows, pitch x_pitch
  const float* w;     // [N][K]
  const float* bias;  // [N] or null
  const float* res;   // [M, N] or null
  float* out;         // [M, N]
  long x_pitch, res_pitch, out_pitch;
  int M, N, K;
  int n_slices, rsplit, tiles_per_block, total_tiles;
};

template <int KQ, int TN, bool HAS_RES>   // K = 32 KQ; block tile 64 rows x (64 TN) columns; waves 2 (m) x 2 (n)
__global__ __launch_bounds__(256) void gemm_stream_kernel(const SArgs p) {
  constexpr int K = 32 * KQ, LD = K + 4, NB = 64 * TN, BM = 64;
  constexpr int NLX = K / 16;                 // 16-byte loads per thread and X tile
  extern __shared__ __align__(16) float smem[];
  float* Wsl = smem;                 // [NB][LD]
  float* Xs = smem + NB * LD;        // [1 or 2][BM][LD]
  constexpr bool DB = KQ <= 6;       // double-buffered X tiles where they fit (K = 256: single buffer, two barriers)
  constexpr int XBUF = BM * LD;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int slice = blockIdx.x % p.n_slices, rs = blockIdx.x / p.n_slices;
  const int n0 = slice * NB;
  const int t_begin = rs * p.tiles_per_block;
  const int t_end = min(p.total_tiles, t_begin + p.tiles_per_block);

  // X tile loader: the tile is a (64 x K/4) grid of 16-byte elements, element e = tid + 256 u
  constexpr int TPR = K / 4;                  // threads per row
  auto make_rsrc = [](const float* base, unsigned nrec) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(nrec), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(p.x, (unsigned)((long)p.M * p.x_pitch * 4));
  unsigned voff[NLX];
  int lds_off[NLX];
#pragma unroll
  for (int u = 0; u < NLX; ++u) {
    const int e = tid + 256 * u;
    const int row = e / TPR, c4 = (e % TPR) * 4;
    voff[u] = (unsigned)(row * (int)p.x_pitch + c4) * 4u;
    lds_off[u] = row * LD + c4;
  }
  u32x4 rx[NLX];
  unsigned soff = 0;
  auto tile_base = [&](int t, bool exists) {
    soff = exists ? (unsigned)(t * BM) * (unsigned)p.x_pitch * 4u : 0x80000000u;
  };
  auto load_one = [&](int u) { rx[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, voff[u], soff, 0); };
  auto commit = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NLX; ++u) *reinterpret_cast<u32x4*>(Xs + buf * XBUF + lds_off[u]) = rx[u];
  };

  // first X tile on its way, then the resident weight slice
  tile_base(t_begin, t_begin < t_end);
#pragma unroll
  for (int u = 0; u < NLX; ++u) load_one(u);
  for (int e = tid; e < NB * TPR; e += 256) {
    const int row = e / TPR, c4 = (e % TPR) * 4;
    *reinterpret_cast<f32x4*>(Wsl + row * LD + c4) = *reinterpret_cast<const f32x4*>(p.w + (long)(n0 + row) * K + c4);
  }
  commit(0);
  __syncthreads();

  const float* b_base = Wsl + (wn * 32 * TN + lr) * LD + lh * 4;

  // Output / residual addressing in the accumulator layout: register r of a 32x32 tile is row
  // (r & 3) + 8 (r >> 2) + 4 lh, column lr, so one dword store writes two full 128-byte row segments.
  // Raw buffer accesses again: lane offset + scalar (tile, row) offset; the tile "before the first"
  // gets an offset past the descriptor, which drops its stores and zero-fills its loads, and a
  // residual is a compile-time variant -- the steady-state loop has no branches.
  const long out_bytes = ((long)p.M * p.out_pitch) * 4, res_bytes = p.res ? ((long)p.M * p.res_pitch) * 4 : 0;
  const __amdgpu_buffer_rsrc_t rsrc_o = make_rsrc(p.out, (unsigned)out_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_r = make_rsrc(p.res ? p.res : p.out, (unsigned)res_bytes);
  const int ncol = n0 + wn * 32 * TN + lr;
  const unsigned vo_out = (unsigned)((wm * 32 + 4 * lh) * (int)p.out_pitch + ncol) * 4u;
  const unsigned vo_res = (unsigned)((wm * 32 + 4 * lh) * (int)p.res_pitch + ncol) * 4u;
  float bvs[TN];
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) bvs[jn] = p.bias ? p.bias[ncol + jn * 32] : 0.f;
  // Have the bias values in hand before the tile loop: a load still pending at the loop head makes
  // the compiler wait for vmcnt(0) at its first use in EVERY iteration, i.e. for the previous tile's
  // stores and the prefetch loads as well.
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) asm volatile("" : "+v"(bvs[jn]));

  // Two accumulator sets alternate between consecutive tiles (no copies): while tile t is multiplied
  // into one, the finished rows of tile t-1 leave from the other.
  f32x16 accA[TN], accB[TN];
  float rvA[TN][16], rvB[TN][16];
  unsigned so_prev_out = 0x80000000u;      // no previous tile yet: its stores fall out of range
  constexpr int NST = 16 * TN, ITERS = 4 * KQ, SPI = (NST + ITERS - 1) / ITERS;   // stores per MFMA group
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto store_one = [&](const f32x16 (&pa)[TN], const float (&pr)[TN][16], int k) {   // k constant after unrolling
    const int jn = k / 16, r = k % 16;
    const int rowc = (r & 3) + 8 * (r >> 2);
    float v = pa[jn][r] + bvs[jn];
    if (HAS_RES) v += pr[jn][r];
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_o, vo_out + (unsigned)(jn * 32 * 4),
                                          so_prev_out + (unsigned)rowc * (unsigned)p.out_pitch * 4u, 0);
  };
  auto run_tile = [&](f32x16 (&acc)[TN], float (&rv)[TN][16], const f32x16 (&pa)[TN], const float (&pr)[TN][16],
                      int t, int buf) {
    tile_base(t + 1, t + 1 < t_end);
    const unsigned so_res = (unsigned)(t * BM) * (unsigned)p.res_pitch * 4u;
    const float* a_base = Xs + buf * XBUF + (wm * 32 + lr) * LD + lh * 4;
    // operand fragments one group ahead of their MFMAs (an LDS read issued right before its use
    // costs its whole latency once per group: the MFMA queue of an in-order wave runs dry)
    f32x4 fa, fb[TN], fa_n, fb_n[TN];
    auto read_frag = [&](int it, f32x4& a, f32x4 (&b)[TN]) {
      a = *reinterpret_cast<const f32x4*>(a_base + it * 8);
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) b[jn] = *reinterpret_cast<const f32x4*>(b_base + jn * 32 * LD + it * 8);
    };
    read_frag(0, fa, fb);
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const int it = q * 4 + kc;
        read_frag(it + 1 < ITERS ? it + 1 : it, fa_n, fb_n);
        // next X tile: two loads per group, all of them early so they have most of the tile to land
        if (2 * it < NLX) load_one(2 * it);
        if (2 * it + 1 < NLX) load_one(2 * it + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn)   // the tile's first MFMA starts from a constant-zero accumulator
            acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[jn][s], (it == 0 && s == 0) ? zero16 : acc[jn], 0, 0, 0);
        // the previous tile's output rows leave while this tile is multiplied
#pragma unroll
        for (int k = it * SPI; k < (it + 1) * SPI && k < NST; ++k) store_one(pa, pr, k);
        // this tile's residual rows arrive
#pragma unroll
        for (int k = it * SPI; HAS_RES && k < (it + 1) * SPI && k < NST; ++k) {
          const int jn = k / 16, r = k % 16;
          const int rowc = (r & 3) + 8 * (r >> 2);
          rv[jn][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                    rsrc_r, vo_res + (unsigned)(jn * 32 * 4),
                                                    so_res + (unsigned)rowc * (unsigned)p.res_pitch * 4u, 0));
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 1 + TN, 0);       // next group's LDS reads first ...
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * TN, 0);       // ... then this group's MFMAs ...
        __builtin_amdgcn_sched_group_barrier(0x010, 2 + 2 * SPI, 0);  // ... then its global loads / stores
        fa = fa_n;
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) fb[jn] = fb_n[jn];
      }
    }
    so_prev_out = (unsigned)(t * BM) * (unsigned)p.out_pitch * 4u;
    if (DB) {
      commit(buf ^ 1);          // the other buffer was last read before the previous closing barrier
      __syncthreads();
    } else {
      __syncthreads();          // every wave is done reading this X tile
      commit(0);                // the next one (fetched during the MFMAs above)
      __syncthreads();
    }
  };
  for (int t = t_begin; t < t_end; t += 2) {
    run_tile(accA, rvA, accB, rvB, t, 0);
    if (t + 1 < t_end) run_tile(accB, rvB, accA, rvA, t + 1, DB ? 1 : 0);
  }
  // the last tile's rows
  if ((t_end - t_begin) & 1) {
#pragma unroll
    for (int k = 0; k < NST; ++k) store_one(accA, rvA, k);
  } else if (t_end > t_begin) {
#pragma unroll
    for (int k = 0; k < NST; ++k) store_one(accB, rvB, k);
  }
}

struct Plan {
  int kq, tn, n_slices, rsplit, tiles_per_block, total_tiles;
  size_t smem;
};

bool make_plan(long M, int N, int K, Plan* pl) {
  if (!(K == 64 || K == 128 || K == 192 || K == 256)) return false;
  if (M < 64 || M % 64 != 0 || M > (1L << 30)) return false;
  const int nb = (K <= 128 && N % 128 == 0) ? 128 : 64;   // 128-column slices where they divide N and fit
  if (N % nb != 0) return false;
  pl->kq = K / 32;
  pl->tn = nb / 64;
  pl->n_slices = N / nb;
  pl->total_tiles = (int)(M / 64);
  int rs = lgm_cu_budget() / pl->n_slices;
  if (rs < 1) rs = 1;
  if (rs > pl->total_tiles) rs = pl->total_tiles;
  pl->tiles_per_block = lgm_cdiv(pl->total_tiles, rs);
  pl->rsplit = lgm_cdiv(pl->total_tiles, pl->tiles_per_block);
  pl->smem = (size_t)(nb + (K <= 192 ? 128 : 64)) * (K + 4) * sizeof(float);
  return true;
}

}  // namespace

// whole 64-row tiles, at least four per workgroup (the resident weight slice must pay for itself),
// 32-bit byte offsets into X
bool lgm_gemm_stream_supported(long M, int N, int K, long x_pitch, long out_pitch, long res_pitch) {
  Plan pl;
  if (!make_plan(M, N, K, &pl)) return false;
  if ((M + 64) * x_pitch >= (1L << 29) || (M + 64) * out_pitch >= (1L << 29) || (M + 64) * res_pitch >= (1L << 29))
    return false;
  return pl.tiles_per_block >= 4;
}

int lgm_gemm_stream_launch(const float* x, long x_pitch, const float* w, const float* bias, const float* res,
                           long res_pitch, float* out, long out_pitch, long M, int N, int K, hipStream_t s) {
  Plan pl;
  if (!make_plan(M, N, K, &pl)) {
    lgm_set_error("gemm_stream: unsupported shape M=%ld N=%d K=%d", M, N, K);
    return LGM_ERR_UNSUPPORTED;
  }
  SArgs p{};
  p.x = x; p.w = w; p.bias = bias; p.res = res; p.out = out;
  p.x_pitch = x_pitch; p.res_pitch = res_pitch; p.out_pitch = out_pitch;
  p.M = (int)M; p.N = N; p.K = K;
  p.n_slices = pl.n_slices; p.rsplit = pl.rsplit; p.tiles_per_block = pl.tiles_per_block; p.total_tiles = pl.total_tiles;
  const unsigned nblocks = (unsigned)(pl.n_slices * pl.rsplit);
  const size_t smem = pl.smem;
#define LGM_GS_LAUNCH1(KQV, TNV, RESV)                                                                                 \
  do {                                                                                                                 \
    auto kern = gemm_stream_kernel<KQV, TNV, RESV>;                                                                    \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    lgm_note_kernel(LGM_KNAME("gemm_stream_kernel<" #KQV ", " #TNV ", " #RESV ">"));                                              \
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), smem, s, p);                                                    \
  } while (0)
#define LGM_GS_LAUNCH(KQV, TNV)                    \
  do {                                             \
    if (res) LGM_GS_LAUNCH1(KQV, TNV, true);       \
    else LGM_GS_LAUNCH1(KQV, TNV, false);          \
  } while (0)
  switch (pl.kq) {
    case 2:
      if (pl.tn == 2) LGM_GS_LAUNCH(2, 2);
      else LGM_GS_LAUNCH(2, 1);
      break;
    case 4:
      if (pl.tn == 2) LGM_GS_LAUNCH(4, 2);
      else LGM_GS_LAUNCH(4, 1);
      break;
    case 6: LGM_GS_LAUNCH(6, 1); break;
    default: LGM_GS_LAUNCH(8, 1); break;
  }
#undef LGM_GS_LAUNCH
#undef LGM_GS_LAUNCH1
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
