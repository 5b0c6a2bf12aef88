// 1x1 convolution / linear layers with a short reduction (K <= 256):  Y[M,N] = X[M,K] W[N,K]^T (+bias,+res)
//
// With K this small a tile-per-block GEMM is all prologue and epilogue.  Here a workgroup keeps a
// BM-row tile of X resident in LDS (loaded once, full K) and walks over ALL output columns in
// groups whose weight rows are staged in LDS as well.  Inside a group the loop issues NO global
// loads — only ds_read_b128 + MFMA + the epilogue's stores — so the (in-order) vector-memory queue
// never makes an MFMA wait for a store to drain; stores of tile t overlap the MFMAs of tile t+1.
// X is read from HBM exactly once.  LDS is sized for two workgroups per CU where K allows it.
// Used for to_qkv / to_out / res_conv / Downsample convs (ddpm.py:103,187,213,215,252,253) forward,
// and for their input gradients through the transposed weight copy.
#include "lgm_common.h"

namespace {

struct RArgs {
  const float* x;     // [M, K] rows, pitch x_pitch
  const float* w;     // [N][K]
  const float* bias;  // [N] or null
  const float* res;   // [M, N] or null
  float* out;         // [M, N]
  long x_pitch, res_pitch, out_pitch;
  int M, N, K;
  int wcols;          // weight rows (output columns) staged per group, multiple of 64
  int ncb;            // output columns per workgroup (blockIdx.y walks the column ranges), multiple of 64
};

template <int TM>   // BM = 64 * TM rows per workgroup; waves 2 (m) x 2 (n), wave tile (32*TM) x 32
__global__ __launch_bounds__(256) void gemm_rows_kernel(const RArgs p) {
  constexpr int BM = 64 * TM;
  extern __shared__ __align__(16) float smem[];
  const int LD = p.K + 4;
  float* Xs = smem;
  float* Ws = smem + BM * LD;
  float* Ts = smem + BM * LD + p.wcols * LD + (threadIdx.x >> 6) * LGM_TS_FLOATS;   // wave-private
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * BM;
  const int k4 = p.K / 4;

  auto stage = [&](float* dst, const float* src, long src_pitch, int row0, int nrows, int row_limit) {
    const int total = nrows * k4;
    for (int base = tid; base < total; base += 256 * 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256;
        const int row = idx / k4, c4 = idx - row * k4;
        const bool ok = idx < total && (row0 + row) < row_limit;
        v[u] = ok ? *reinterpret_cast<const f32x4*>(src + (long)(row0 + row) * src_pitch + c4 * 4)
                  : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256;
        if (idx < total) {
          const int row = idx / k4, c4 = idx - row * k4;
          *reinterpret_cast<f32x4*>(dst + row * LD + c4 * 4) = v[u];
        }
      }
    }
  };

  stage(Xs, p.x, p.x_pitch, m0, BM, p.M);
  const int KQ = p.K / 32;
  const float* a_base = Xs + (wm * 32 * TM + lr) * LD + lh * 4;
  const float* b_lane = Ws + (wn * 32 + lr) * LD + lh * 4;

  // few row tiles (small maps): the columns are dealt out over blockIdx.y as well, so that the launch still fills the chip -
  // with one workgroup per 64 rows a 8192-row layer ran on 128 CUs, each working through ALL columns (24.9 us for
  // 128 -> 384 at 8 x 8, B = 128, of which 11.7 are the MFMAs of one workgroup)
  const int n_lo = (int)blockIdx.y * p.ncb, n_hi = min(p.N, n_lo + p.ncb);
  for (int g0 = n_lo; g0 < n_hi; g0 += p.wcols) {
    const int gcols = min(p.wcols, n_hi - g0);
    if (g0 > n_lo) __syncthreads();       // previous group's weight rows fully consumed
    stage(Ws, p.w, p.K, g0, gcols, p.N);
    __syncthreads();
    for (int nt = 0; nt < gcols / 64; ++nt) {
      f32x16 acc[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      const float* bp = b_lane + nt * 64 * LD;
      for (int q = 0; q < KQ; ++q) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          f32x4 fa[TM];
#pragma unroll
          for (int i = 0; i < TM; ++i)
            fa[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * LD + q * 32 + kc * 8);
          const f32x4 fb = *reinterpret_cast<const f32x4*>(bp + q * 32 + kc * 8);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[s], acc[i], 0, 0, 0);
        }
      }
      // ---- epilogue: wave-private LDS transpose, then 16-byte stores (8 full rows per instruction).
      // Residual rows are loaded up front and the stores are unconditional (M is a multiple of the
      // row tile): a conditional store inside the row loop makes the compiler wait for the previous
      // store to complete (s_waitcnt vmcnt(0)) before every row.
      const int nc = g0 + nt * 64 + wn * 32 + (lane & 7) * 4;
      const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 rv[TM][4];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) rv[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.res) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const long m = m0 + wm * 32 * TM + i * 32 + (lane >> 3) + 8 * j;
            rv[i][j] = *reinterpret_cast<const f32x4*>(p.res + m * p.res_pitch + nc);
          }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        lgm_wave_lds_sync();                 // earlier read-back of this scratch is complete
        lgm_tile_to_lds(acc[i], Ts, lane);
        lgm_wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long m = m0 + wm * 32 * TM + i * 32 + (lane >> 3) + 8 * j;
          *reinterpret_cast<f32x4*>(p.out + m * p.out_pitch + nc) = lgm_tile_row4(Ts, lane, j) + bv + rv[i][j];
        }
      }
    }
  }
}

// column ranges per row tile: enough workgroups for the chip's slots, whole 64-column tiles, the X tile re-read per range
int col_ranges(long row_tiles, int N) {
  static const int off = getenv("LGM_GR_NO_COLSPLIT") != nullptr;        // A/B switch
  if (off) return 1;
  int r = (int)((lgm_cu_budget() + row_tiles - 1) / row_tiles);
  const int tiles_n = N / 64;
  if (r > tiles_n) r = tiles_n;
  if (r < 1) r = 1;
  while (tiles_n % r != 0) --r;                                          // equal ranges
  return r;
}

// LDS plan: rows per workgroup and weight columns per group
void plan(int N, int K, int* bm, int* wcols, size_t* smem) {
  const long row = (long)(K + 4) * 4;
  *bm = K <= 64 ? 128 : 64;
  const long xs = (long)*bm * row + 4L * LGM_TS_FLOATS * 4;   // X tile + the four epilogue scratches
  long budget = (getenv("LGM_GR_KB") ? atoi(getenv("LGM_GR_KB")) : 78) * 1024L;   // two workgroups per CU
  if (xs + 64 * row > budget) budget = 150 * 1024;
  long cols = (budget - xs) / (64 * row) * 64;
  if (cols > N) cols = N;
  *wcols = (int)cols;
  *smem = (size_t)(xs + cols * row);
}

}  // namespace

bool lgm_gemm_rows_supported(long M, int N, int K) {
  if (K % 32 != 0 || N % 64 != 0 || N < 128 || K > 256) return false;   // N >= 128: the X tile is reused
  int bm, wcols;
  size_t smem;
  plan(N, K, &bm, &wcols, &smem);
  static const long min_tiles = getenv("LGM_GR_MIN_TILES") ? atol(getenv("LGM_GR_MIN_TILES")) : 96;   // tuning knob
  // whole row tiles; enough workgroups to fill the chip once the columns are dealt out as well (col_ranges)
  return wcols >= 64 && M % bm == 0 && M / bm >= min_tiles;
}

int lgm_gemm_rows_launch(const float* x, long x_pitch, const float* w, const float* bias, const float* res,
                         long res_pitch, float* out, long out_pitch, long M, int N, int K, hipStream_t s) {
  RArgs p{};
  p.x = x; p.w = w; p.bias = bias; p.res = res; p.out = out;
  p.x_pitch = x_pitch; p.res_pitch = res_pitch; p.out_pitch = out_pitch;
  p.M = (int)M; p.N = N; p.K = K;
  int bm;
  size_t smem;
  plan(N, K, &bm, &p.wcols, &smem);
  const unsigned nblocks = (unsigned)lgm_cdiv(M, bm);
  const int ranges = col_ranges((long)nblocks, N);
  p.ncb = N / ranges;
  if (p.wcols > p.ncb) p.wcols = p.ncb;
  if (bm == 128) {
    static size_t attr = 0;
    if (smem > attr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rows_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr = smem;
    }
    lgm_note_kernel(LGM_KNAME("gemm_rows_kernel<2>"));
    hipLaunchKernelGGL(gemm_rows_kernel<2>, dim3(nblocks, ranges), dim3(256), smem, s, p);
  } else {
    static size_t attr = 0;
    if (smem > attr) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rows_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr = smem;
    }
    lgm_note_kernel(LGM_KNAME("gemm_rows_kernel<1>"));
    hipLaunchKernelGGL(gemm_rows_kernel<1>, dim3(nblocks, ranges), dim3(256), smem, s, p);
  }
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
