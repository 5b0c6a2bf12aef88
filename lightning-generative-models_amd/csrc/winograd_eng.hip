// Non-fused Winograd engine (round 6): transform launch -> batched fp32 MFMA GEMM -> transform launch.
//
//   V[xi][t][k]  = input transform of tile t (B^T d B)                lgm_weng_*_in*      light, HBM-bound
//   M[xi][t][n]  = sum_k V[xi][t][k] * U[xi][n][k]                    lgm_weng_gemm       ONE batched NT GEMM, weight-stationary
//   y            = output transform (A^T M A) (+ bias)                lgm_weng_*_out*     light, HBM-bound
//
// The fused Winograd kernels of winograd.hip / winograd4.hip keep V and M on chip; they pay for it with a workgroup shape
// that the SMALL feature maps (4 x 4: one F(4x4) tile per image, so a workgroup's tile count is the batch) and the STRIDED
// layers cannot fill.  Here a workgroup of the GEMM owns a (rows x 128 columns) block of ONE xi and streams U[xi] once for all
// the tiles of the batch; V and M (tens of MB) live in L2 / Infinity Cache between the three launches.
//
// Flavours (all fp32, transform constants exact in fp32 or rounded once; U is prepared by the caller):
//   f43      F(4x4, 3x3), stride 1, pad 1 (reference Block.proj ddpm.py:160 on the 4 x 4 ... maps): 36 xi, 6 x 6 input tiles.
//   f42 xy   4x4 / stride-2 / pad-1 convolution X -> Y (reference dcgan.py:150-158 Discriminator) as F(4x4, 2x2) on the four pixel
//            phases of the input: y[oh] = sum_p sum_a X_p[oh + a] w[2a + p], X_p[i] = x[2i + p - 1]; the four phases are summed
//            INSIDE the GEMM (k = (phase, channel)): 25 xi, 100 products per 16 outputs instead of 256.
//   f42 yx   the same layer Y -> X (input gradient; ConvTranspose2d forward, reference dcgan.py:79-87 Generator): phase (p, q)
//            of the output is a 2x2 full correlation of the (padded) Y side with the flipped taps of that phase; the four
//            phases are four independent Winograd problems over windows one pixel apart: batch = 4 * 25 GEMMs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "lgm_common.h"

namespace lgmweng {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// ---------------------------------------------------------------------------------------------------------------------
// Batched NT GEMM  C[b][m][n] = sum_k A[b][m][k] * Bm[b][n][k]   (fp32, v_mfma_f32_32x32x2_f32)
//   workgroup = 256 threads = 4 waves as 2 x 2; tile BM x 128 (BM = 128: wave = 64 x 64 = four 32x32 accumulators; BM = 64:
//   wave = 32 x 64), K chunk 32, LDS row stride 36 floats (conflict-free ds_read_b128 for the 16-lane groups of gfx950), two
//   LDS stages, the next chunk's global loads in flight during the current chunk's MFMAs.
//   The reduction index is dealt out as k = 16 * (lane >> 5) + step (any bijection of k serves a sum): a lane's sixteen operand
//   values of a chunk are consecutive floats of a tile row = four ds_read_b128.
//   Operands through raw buffer loads: rows >= M / N and k >= K read as zeros (range check), so ragged shapes need no branches;
//   stores are guarded.  K % 4 == 0, lda / ldb % 4 == 0, 16-byte aligned bases.
// ---------------------------------------------------------------------------------------------------------------------
struct GemmArgs {
  const float* A;
  const float* Bm;
  float* C;
  int M, N, K, lda, ldb, ldc;
  long a_batch, b_batch, c_batch;      // strides in floats between consecutive batch entries
  int tiles_m, tiles_n;
  const float* bias;                   // [N] added per column, or NULL   (batch == 1 callers: the 1x1 convolutions)
  const float* res;                    // [M][ldr] added elementwise, or NULL (may alias C)
  int ldr;
};

constexpr int BK = 32, LD = 36;

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void weng_gemm_kernel(GemmArgs p) {
  constexpr int RT = BM / 64, CT = BN / 64;          // 32 x 32 accumulator tiles per wave along m / n (waves are 2 x 2)
  constexpr int A_ST = BM * LD, B_ST = BN * LD;      // floats per LDS stage
  extern __shared__ float smem[];
  float* As = smem;                                  // [2][BM][LD]
  float* Bs = smem + 2 * A_ST;                       // [2][BN][LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  // Workgroup order.  Hardware deals consecutive workgroup ids out to the eight XCDs round-robin; with (tile, xi) = the
  // launch order every XCD's L2 would stream every U[xi].  An XCD takes a CONTIGUOUS range of the (xi, tile) list instead
  // (xi slowest, column tile fastest: the workgroups an XCD runs together share one or two U[xi] and V[xi] slices).
  const int per_batch = p.tiles_m * p.tiles_n;
  const int total = per_batch * (int)gridDim.y;
  int L = blockIdx.x + blockIdx.y * per_batch;
  if ((total & 7) == 0) L = (L & 7) * (total >> 3) + (L >> 3);
  const int batch = L / per_batch, tile = L % per_batch;
  const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
  const float* A = p.A + batch * p.a_batch + (long)tm * BM * p.lda;
  const float* Bm = p.Bm + batch * p.b_batch + (long)tn * BN * p.ldb;
  const int rows_a = min(BM, p.M - tm * BM), rows_b = min(BN, p.N - tn * BN);
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, (unsigned)(((long)(rows_a - 1) * p.lda + p.K) * 4));
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(Bm, (unsigned)(((long)(rows_b - 1) * p.ldb + p.K) * 4));
  // loaders: a tile row holds BK / 4 = 8 sixteen-byte pieces; thread -> (row = e / 8, piece = e % 8), e = tid + 256 u
  constexpr int NA = BM * 8 / 256, NB = BN * 8 / 256;
  unsigned voa[NA], vob[NB];
  int la[NA], lb[NB];
  const int pc = tid & 7;                            // the same piece for every u (256 % 8 == 0)
#pragma unroll
  for (int u = 0; u < NA; ++u) {
    const int row = (tid + 256 * u) >> 3;
    // a row past the block's last row must not alias the next rows' bytes: send it out of range
    voa[u] = row < rows_a ? (unsigned)(row * p.lda + pc * 4) * 4u : 0x80000000u;
    la[u] = row * LD + pc * 4;
  }
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int row = (tid + 256 * u) >> 3;
    vob[u] = row < rows_b ? (unsigned)(row * p.ldb + pc * 4) * 4u : 0x80000000u;
    lb[u] = row * LD + pc * 4;
  }
  struct Stage {
    u32x4 a[NA], b[NB];
  };
  const int nchunks = (p.K + BK - 1) / BK;
  auto fetch = [&](Stage& R, int c) {
    // within a row the k range is checked by hand (the descriptor only knows the block's last byte); a chunk past the last
    // one is fetched as "nothing" (every offset out of range), so the pipeline below has no branches around its loads
    const bool in = c < nchunks && c * BK + pc * 4 < p.K;
    const unsigned so = in ? (unsigned)c * BK * 4u : 0x80000000u;
#pragma unroll
    for (int u = 0; u < NA; ++u) R.a[u] = __builtin_amdgcn_raw_buffer_load_b128(ra, voa[u], so, 0);
#pragma unroll
    for (int u = 0; u < NB; ++u) R.b[u] = __builtin_amdgcn_raw_buffer_load_b128(rb, vob[u], so, 0);
  };
  auto commit = [&](const Stage& R, int st) {
#pragma unroll
    for (int u = 0; u < NA; ++u) *reinterpret_cast<u32x4*>(As + st * A_ST + la[u]) = R.a[u];
#pragma unroll
    for (int u = 0; u < NB; ++u) *reinterpret_cast<u32x4*>(Bs + st * B_ST + lb[u]) = R.b[u];
  };
  f32x16 acc[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto compute = [&](int st) {
    const float* a0 = As + st * A_ST + (wm * (RT * 32) + lr) * LD + 16 * lh;
    const float* b0 = Bs + st * B_ST + (wn * (CT * 32) + lr) * LD + 16 * lh;
#pragma unroll
    for (int h = 0; h < 2; ++h) {                    // two halves of the lane's sixteen k: operand registers stay at 32
      f32x4 av[RT][2], bv[CT][2];
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) av[i][q] = *reinterpret_cast<const f32x4*>(a0 + i * 32 * LD + 8 * h + 4 * q);
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) bv[j][q] = *reinterpret_cast<const f32x4*>(b0 + j * 32 * LD + 8 * h + 4 * q);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][q][s], bv[j][q][s], acc[i][j], 0, 0, 0);
    }
  };
  // Two chunks of global loads in flight (register sets R0 / R1), two LDS stages: chunk c + 2 is requested before chunk c is
  // multiplied and committed to LDS behind chunk c + 1's MFMAs - a load has two chunks (>= 4 k cycles) to land; with ONE chunk
  // of distance the 64-row tiles (2 k cycles of MFMAs per chunk) waited for every fetch (first version: 0.35 - 0.57 of peak).
  Stage R0, R1;
  fetch(R0, 0);
  fetch(R1, 1);
  commit(R0, 0);
  __syncthreads();
  for (int c = 0; c < nchunks; c += 2) {
    fetch(R0, c + 2);
    compute(0);
    commit(R1, 1);                                   // chunk c + 1 (an empty fetch past the end: zeros, never multiplied)
    __syncthreads();
    if (c + 1 >= nchunks) break;
    fetch(R1, c + 3);
    compute(1);
    commit(R0, 0);                                   // chunk c + 2
    __syncthreads();
  }
  // epilogue: straight from the accumulator layout (lane = column, 16 rows in registers): one dword store instruction covers
  // two full 128-byte row segments
  float* C = p.C + batch * p.c_batch;
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int col = tn * BN + wn * (CT * 32) + j * 32 + lr;
      const int row0 = tm * BM + wm * (RT * 32) + i * 32;
      const float bv = (p.bias && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.M && col < p.N) {
          float v = acc[i][j][r] + bv;
          if (p.res) v += p.res[(long)row * p.ldr + col];
          C[(long)row * p.ldc + col] = v;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Transforms.  One thread per (tile, 4 channels): 16-byte loads / stores, channels fastest (coalesced).
// ---------------------------------------------------------------------------------------------------------------------
// F(4, 3): B^T (6 x 6), A^T (4 x 6).  F(4, 2): B^T (5 x 5), A^T (4 x 5).  (tools/weng_matrices.py derives and checks them.)
__device__ __forceinline__ void bt6(const f32x4* d, f32x4* o) {     // o = B^T d   (one column of six)
  o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  o[1] = -4.f * d[1] - 4.f * d[2] + d[3] + d[4];
  o[2] = 4.f * d[1] - 4.f * d[2] - d[3] + d[4];
  o[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
  o[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
  o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
__device__ __forceinline__ void at6(const f32x4* m, f32x4* o) {     // o = A^T m   (six -> four)
  o[0] = m[0] + m[1] + m[2] + m[3] + m[4];
  o[1] = m[1] - m[2] + 2.f * m[3] - 2.f * m[4];
  o[2] = m[1] + m[2] + 4.f * m[3] + 4.f * m[4];
  o[3] = m[1] - m[2] + 8.f * m[3] - 8.f * m[4] + m[5];
}
__device__ __forceinline__ void bt5(const f32x4* d, f32x4* o) {     // F(4, 2): B^T d (five)
  o[0] = 2.f * d[0] - d[1] - 2.f * d[2] + d[3];
  o[1] = -2.f * d[1] - d[2] + d[3];
  o[2] = 2.f * d[1] - 3.f * d[2] + d[3];
  o[3] = -d[1] + d[3];
  o[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
}
__device__ __forceinline__ void at5(const f32x4* m, f32x4* o) {     // F(4, 2): A^T m (five -> four)
  o[0] = m[0] + m[1] + m[2] + m[3];
  o[1] = m[1] - m[2] + 2.f * m[3];
  o[2] = m[1] + m[2] + 4.f * m[3];
  o[3] = m[1] - m[2] + 8.f * m[3] + m[4];
}

// v = B^T d B with d read column by column through load(r, c) and v handed out row by row through store(r, c, value): only the
// half-transformed tile (NT x NT x 4 floats) is ever live (holding d, t and v at once spilled past 256 registers)
template <int NT, class Load, class Store>
__device__ __forceinline__ void tile_in(Load load, Store store) {
  f32x4 t[NT][NT];
#pragma unroll
  for (int c = 0; c < NT; ++c) {                     // columns: t = B^T d
    f32x4 col[NT], o[NT];
#pragma unroll
    for (int r = 0; r < NT; ++r) col[r] = load(r, c);
    if constexpr (NT == 6) bt6(col, o); else bt5(col, o);
#pragma unroll
    for (int r = 0; r < NT; ++r) t[r][c] = o[r];
  }
#pragma unroll
  for (int r = 0; r < NT; ++r) {                     // rows: v = t B
    f32x4 o[NT];
    if constexpr (NT == 6) bt6(t[r], o); else bt5(t[r], o);
#pragma unroll
    for (int c = 0; c < NT; ++c) store(r, c, o[c]);
  }
}
// y = A^T m A, same scheme (4 x NT half-transformed values live)
template <int NT, class Load, class Store>
__device__ __forceinline__ void tile_out(Load load, Store store) {
  f32x4 t[4][NT];
#pragma unroll
  for (int c = 0; c < NT; ++c) {
    f32x4 col[NT], o[4];
#pragma unroll
    for (int r = 0; r < NT; ++r) col[r] = load(r, c);
    if constexpr (NT == 6) at6(col, o); else at5(col, o);
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r][c] = o[r];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    f32x4 o[4];
    if constexpr (NT == 6) at6(t[r], o); else at5(t[r], o);
#pragma unroll
    for (int c = 0; c < 4; ++c) store(r, c, o[c]);
  }
}

// f43 input: x [B][H][W][C] (pitch) -> V[36][T][C], T = B * (H/4) * (W/4), tile origin (4 ty - 1, 4 tx - 1)
__global__ __launch_bounds__(256) void f43_in_kernel(const float* __restrict__ x, long pitch, int B, int H, int W, int C,
                                                     float* __restrict__ V) {
  const int C4 = C / 4, TY = H / 4, TX = W / 4;
  const long T = (long)B * TY * TX;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * C4) return;
  const int c4 = (int)(i % C4);
  const long t = i / C4;
  const int tx = (int)(t % TX), ty = (int)((t / TX) % TY), b = (int)(t / ((long)TX * TY));
  tile_in<6>(
      [&](int r, int s) {
        const int h = 4 * ty - 1 + r, w = 4 * tx - 1 + s;
        return (h >= 0 && h < H && w >= 0 && w < W)
                   ? *reinterpret_cast<const f32x4*>(x + (((long)b * H + h) * W + w) * pitch + 4 * c4)
                   : f32x4{0.f, 0.f, 0.f, 0.f};
      },
      [&](int r, int s, f32x4 v) { *reinterpret_cast<f32x4*>(V + ((long)(r * 6 + s) * T + t) * C + 4 * c4) = v; });
}
// f43 output: M[36][T][N] -> y [B][H][W][N] (pitch) + bias
__global__ __launch_bounds__(256) void f43_out_kernel(const float* __restrict__ M, int B, int H, int W, int N,
                                                      const float* __restrict__ bias, float* __restrict__ y, long pitch) {
  const int N4 = N / 4, TY = H / 4, TX = W / 4;
  const long T = (long)B * TY * TX;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * N4) return;
  const int n4 = (int)(i % N4);
  const long t = i / N4;
  const int tx = (int)(t % TX), ty = (int)((t / TX) % TY), b = (int)(t / ((long)TX * TY));
  const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + 4 * n4) : f32x4{0.f, 0.f, 0.f, 0.f};
  tile_out<6>([&](int r, int s) { return *reinterpret_cast<const f32x4*>(M + ((long)(r * 6 + s) * T + t) * N + 4 * n4); },
              [&](int r, int s, f32x4 v) {
                *reinterpret_cast<f32x4*>(y + (((long)b * H + 4 * ty + r) * W + 4 * tx + s) * pitch + 4 * n4) = v + bv;
              });
}

// f42 xy input: x [B][H][W][C] -> V[25][T][4 C], T = B * (Ho/4) * (Wo/4), Ho = H/2; k = (2p + q) * C + c;
// X_pq[u][v] = x[2 (4 ty + u) + p - 1][2 (4 tx + v) + q - 1], u, v in 0..4
__global__ __launch_bounds__(256) void f42_in_xy_kernel(const float* __restrict__ x, long pitch, int B, int H, int W, int C,
                                                        float* __restrict__ V) {
  const int C4 = C / 4, TY = H / 8, TX = W / 8;
  const long T = (long)B * TY * TX;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * 4 * C4) return;
  const int c4 = (int)(i % C4);
  const int pq = (int)((i / C4) % 4), pp = pq >> 1, qq = pq & 1;
  const long t = i / (4 * C4);
  const int tx = (int)(t % TX), ty = (int)((t / TX) % TY), b = (int)(t / ((long)TX * TY));
  const long K = 4L * C;
  tile_in<5>(
      [&](int u, int s) {
        const int h = 2 * (4 * ty + u) + pp - 1, w = 2 * (4 * tx + s) + qq - 1;
        return (h >= 0 && h < H && w >= 0 && w < W)
                   ? *reinterpret_cast<const f32x4*>(x + (((long)b * H + h) * W + w) * pitch + 4 * c4)
                   : f32x4{0.f, 0.f, 0.f, 0.f};
      },
      [&](int r, int s, f32x4 v) {
        *reinterpret_cast<f32x4*>(V + ((long)(r * 5 + s) * T + t) * K + (long)pq * C + 4 * c4) = v;
      });
}
// What the implicit-GEMM kernels' LgmPostOp does in their epilogue, here in the output transform (the tensor is being written
// anyway): out = act(v + bias) then * (mask > 0 ? 1 : mask_slope) - an activation behind a forward convolution, or the
// derivative of the activation in front of the layer applied to an input gradient (mask = that activation's saved output).
struct Epi {
  int act;                 // 0, 3 = ReLU, 4 = LeakyReLU (elementwise.hip's codes)
  float slope;
  const float* mask;       // same pixel layout as the output; NULL = none
  long mask_pitch;
  float mask_slope;
};
__device__ __forceinline__ f32x4 epi_apply(f32x4 v, const Epi& e, long pix, int c) {
  if (e.act == 3) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
  } else if (e.act == 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * e.slope;
  }
  if (e.mask) {
    const f32x4 m = *reinterpret_cast<const f32x4*>(e.mask + pix * e.mask_pitch + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = m[j] > 0.f ? v[j] : v[j] * e.mask_slope;
  }
  return v;
}

// f42 xy output: M[25][T][N] -> y [B][Ho][Wo][N] + bias
__global__ __launch_bounds__(256) void f42_out_xy_kernel(const float* __restrict__ M, int B, int Ho, int Wo, int N,
                                                         const float* __restrict__ bias, float* __restrict__ y, long pitch,
                                                         Epi epi) {
  const int N4 = N / 4, TY = Ho / 4, TX = Wo / 4;
  const long T = (long)B * TY * TX;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * N4) return;
  const int n4 = (int)(i % N4);
  const long t = i / N4;
  const int tx = (int)(t % TX), ty = (int)((t / TX) % TY), b = (int)(t / ((long)TX * TY));
  const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + 4 * n4) : f32x4{0.f, 0.f, 0.f, 0.f};
  tile_out<5>([&](int r, int s) { return *reinterpret_cast<const f32x4*>(M + ((long)(r * 5 + s) * T + t) * N + 4 * n4); },
              [&](int r, int s, f32x4 v) {
                const long pix = ((long)b * Ho + 4 * ty + r) * Wo + 4 * tx + s;
                *reinterpret_cast<f32x4*>(y + pix * pitch + 4 * n4) = epi_apply(v + bv, epi, pix, 4 * n4);
              });
}
// f42 yx input: dy [B][Ho][Wo][Ny] -> V[4][25][T][Ny], T = B * (Ho/4) * (Wo/4); phase (p, q) reads the 5 x 5 window whose first
// pixel is (4 ty - p, 4 tx - q) (zeros outside)
__global__ __launch_bounds__(256) void f42_in_yx_kernel(const float* __restrict__ dy, long pitch, int B, int Ho, int Wo, int Ny,
                                                        float* __restrict__ V) {
  const int C4 = Ny / 4, TY = Ho / 4, TX = Wo / 4;
  const long T = (long)B * TY * TX;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * 4 * C4) return;
  const int c4 = (int)(i % C4);
  const long t = (i / C4) % T;
  const int pq = (int)(i / ((long)C4 * T)), pp = pq >> 1, qq = pq & 1;
  const int tx = (int)(t % TX), ty = (int)((t / TX) % TY), b = (int)(t / ((long)TX * TY));
  tile_in<5>(
      [&](int u, int s) {
        const int h = 4 * ty - pp + u, w = 4 * tx - qq + s;
        return (h >= 0 && h < Ho && w >= 0 && w < Wo)
                   ? *reinterpret_cast<const f32x4*>(dy + (((long)b * Ho + h) * Wo + w) * pitch + 4 * c4)
                   : f32x4{0.f, 0.f, 0.f, 0.f};
      },
      [&](int r, int s, f32x4 v) {
        *reinterpret_cast<f32x4*>(V + (((long)pq * 25 + r * 5 + s) * T + t) * Ny + 4 * c4) = v;
      });
}
// f42 yx output: M[4][25][T][C] -> dx [B][2 Ho][2 Wo][C]: phase (p, q), tile output (u, v) -> pixel (8 ty + 2 u + 1 - p, 8 tx + 2 v + 1 - q)
__global__ __launch_bounds__(256) void f42_out_yx_kernel(const float* __restrict__ M, int B, int Ho, int Wo, int C,
                                                         const float* __restrict__ bias, float* __restrict__ dx, long pitch,
                                                         Epi epi) {
  const int C4 = C / 4, TY = Ho / 4, TX = Wo / 4;
  const long T = (long)B * TY * TX;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * 4 * C4) return;
  const int c4 = (int)(i % C4);
  const int pq = (int)((i / C4) % 4), pp = pq >> 1, qq = pq & 1;
  const long t = i / (4 * C4);
  const int tx = (int)(t % TX), ty = (int)((t / TX) % TY), b = (int)(t / ((long)TX * TY));
  const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
  const int H = 2 * Ho, W = 2 * Wo;
  tile_out<5>(
      [&](int r, int s) { return *reinterpret_cast<const f32x4*>(M + (((long)pq * 25 + r * 5 + s) * T + t) * C + 4 * c4); },
      [&](int u, int s, f32x4 v) {
        const int h = 8 * ty + 2 * u + 1 - pp, w = 8 * tx + 2 * s + 1 - qq;
        const long pix = ((long)b * H + h) * W + w;
        *reinterpret_cast<f32x4*>(dx + pix * pitch + 4 * c4) = epi_apply(v + bv, epi, pix, 4 * c4);
      });
}

// U of one 4x4 / stride-2 layer for both directions from its weights w [Nw][16][Cw] (tap = 4 kh + kw), float64 arithmetic,
// rounded once:  xy: U[xi][n][(2p + q) Cw + c] = (G g_pq G^T)[xi], g_pq[a][b] = w[2a + p][2b + q]
//                yx: U[2p + q][xi][c][n]       = (G f_pq G^T)[xi], f_pq[a][b] = w[2 (1 - a) + p][2 (1 - b) + q]
// G = [[1/2, 0], [-1/2, -1/2], [-1/6, 1/6], [1/6, 1/3], [0, 1]].  One instantiation per direction: xy (thread = (n, 4 channels):
// 16-byte loads and stores along c), yx (thread = (4 channels, n), n fastest: the stores run along n).
__device__ __forceinline__ void g42(const double (&g)[2][2], double (&u)[5][5]) {
  double t[5][2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    t[0][b] = 0.5 * g[0][b];
    t[1][b] = -0.5 * (g[0][b] + g[1][b]);
    t[2][b] = (g[1][b] - g[0][b]) / 6.0;
    t[3][b] = g[0][b] / 6.0 + g[1][b] / 3.0;
    t[4][b] = g[1][b];
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    u[i][0] = 0.5 * t[i][0];
    u[i][1] = -0.5 * (t[i][0] + t[i][1]);
    u[i][2] = (t[i][1] - t[i][0]) / 6.0;
    u[i][3] = t[i][0] / 6.0 + t[i][1] / 3.0;
    u[i][4] = t[i][1];
  }
}
template <bool yx>
__global__ __launch_bounds__(256) void f42_weights_kernel(const float* __restrict__ w, int Nw, int Cw,
                                                          float* __restrict__ Uxy, float* __restrict__ Uyx) {
  const int C4 = Cw / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)Nw * C4) return;
  const int n = yx ? (int)(i % Nw) : (int)(i / C4);
  const int c4 = yx ? (int)(i / Nw) : (int)(i % C4);
  f32x4 tap[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) tap[k] = *reinterpret_cast<const f32x4*>(w + ((long)n * 16 + k) * Cw + 4 * c4);
#pragma unroll
  for (int pq = 0; pq < 4; ++pq) {
    const int pp = pq >> 1, qq = pq & 1;
    double u[4][5][5];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double g[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int aa = yx ? 1 - a : a, bb = yx ? 1 - b : b;
          g[a][b] = (double)tap[(2 * aa + pp) * 4 + 2 * bb + qq][j];
        }
      g42(g, u[j]);
    }
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
      for (int s2 = 0; s2 < 5; ++s2) {
        const int xi = r * 5 + s2;
        if (!yx) {
          const f32x4 v = {(float)u[0][r][s2], (float)u[1][r][s2], (float)u[2][r][s2], (float)u[3][r][s2]};
          *reinterpret_cast<f32x4*>(Uxy + ((long)xi * Nw + n) * (4L * Cw) + (long)pq * Cw + 4 * c4) = v;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) Uyx[(((long)pq * 25 + xi) * Cw + 4 * c4 + j) * Nw + n] = (float)u[j][r][s2];
        }
      }
  }
}

}  // namespace lgmweng

using namespace lgmweng;

static int weng_gemm_launch(const float* A, const float* Bm, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                            int batch, int64_t a_batch, int64_t b_batch, int64_t c_batch, const float* bias,
                            const float* res, int ldr, void* stream);
extern "C" int lgm_weng_gemm(const float* A, const float* Bm, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                             int batch, int64_t a_batch, int64_t b_batch, int64_t c_batch, void* stream) {
  LGM_REQUIRE(A && Bm && C && M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535, "weng_gemm: bad arguments");
  LGM_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && lda >= K && ldb >= K && ldc >= N && lgm_aligned16(A) &&
                  lgm_aligned16(Bm) && a_batch % 4 == 0 && b_batch % 4 == 0,
              "weng_gemm: K, lda, ldb and the batch strides must be multiples of 4 floats, operands 16-byte aligned");
  LGM_REQUIRE(((long)M * lda + K) * 4 < (1L << 31) && ((long)N * ldb + K) * 4 < (1L << 31), "weng_gemm: operand block too large");
  return weng_gemm_launch(A, Bm, C, M, N, K, lda, ldb, ldc, batch, a_batch, b_batch, c_batch, nullptr, nullptr, 0, stream);
}
/* y[m][n] = bias[n] + res[m][n] + sum_k x[m][k] w[n][k]: a 1x1 convolution / linear layer (Conv2d ddpm.py:96-103, 187, 213-215,
 * 252-253) as ONE un-split GEMM launch - no split-K planes, no reducer launch */
extern "C" int lgm_weng_gemm_epi(const float* A, const float* Bm, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                                 const float* bias, const float* res, int ldr, void* stream) {
  LGM_REQUIRE(A && Bm && C && M > 0 && N > 0 && K > 0, "weng_gemm_epi: bad arguments");
  LGM_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && lda >= K && ldb >= K && ldc >= N && lgm_aligned16(A) &&
                  lgm_aligned16(Bm) && (!res || ldr >= N),
              "weng_gemm_epi: K, lda, ldb must be multiples of 4 floats, operands 16-byte aligned");
  LGM_REQUIRE(((long)M * lda + K) * 4 < (1L << 31) && ((long)N * ldb + K) * 4 < (1L << 31), "weng_gemm_epi: operand block too large");
  return weng_gemm_launch(A, Bm, C, M, N, K, lda, ldb, ldc, 1, 0, 0, 0, bias, res, ldr, stream);
}
static int weng_gemm_launch(const float* A, const float* Bm, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                            int batch, int64_t a_batch, int64_t b_batch, int64_t c_batch, const float* bias,
                            const float* res, int ldr, void* stream) {
  GemmArgs p{A, Bm, C, M, N, K, lda, ldb, ldc, (long)a_batch, (long)b_batch, (long)c_batch, 0, 0, bias, res, ldr};
  // Tile choice, measured (tools/weng_gemm_bench.py, fraction of the 157.3 TFLOP/s fp32 MFMA peak, warm): the 128 x 128 tile
  // reaches 0.83 on a 4096^3 product (64 x 64: 0.80), but every shape the engine meets (M = tiles of a batch: 128 ... 2048,
  // batch 25 ... 100) is a few hundred to a few thousand workgroups of 8 ... 64 K-chunks, where start / drain and the
  // power-of-two workgroup counts decide: 64 x 64 is the fastest tile on all thirteen of them (0.49 - 0.72 against
  // 0.27 - 0.62 for 128 x 128).  So: 128 x 128 only when it still gives every CU four workgroups; LGM_WENG_TILE=<bm>x<bn> pins.
  struct Cand { int bm, bn; };
  const Cand cands[4] = {{128, 128}, {64, 128}, {128, 64}, {64, 64}};
  static const char* pin = getenv("LGM_WENG_TILE");
  int best = (long)lgm_cdiv(M, 128) * lgm_cdiv(N, 128) * batch >= 4L * lgm_cu_budget() ? 0 : 3;
  if (pin) {
    int a = 0, b = 0;
    if (sscanf(pin, "%dx%d", &a, &b) == 2)
      for (int i = 0; i < 4; ++i)
        if (a == cands[i].bm && b == cands[i].bn) best = i;
  }
  const int bm = cands[best].bm, bn = cands[best].bn;
  p.tiles_m = lgm_cdiv(M, bm);
  p.tiles_n = lgm_cdiv(N, bn);
  const size_t sm = sizeof(float) * 2 * (size_t)(bm + bn) * LD;
  const dim3 grid(p.tiles_m * p.tiles_n, batch);
  auto launch = [&](auto kern) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL(kern, grid, dim3(256), sm, (hipStream_t)stream, p);
  };
  if (bm == 128 && bn == 128) launch(weng_gemm_kernel<128, 128>), lgm_note_kernel(LGM_KNAME("lgmweng::weng_gemm_kernel<128, 128>"));
  else if (bm == 64 && bn == 128) launch(weng_gemm_kernel<64, 128>), lgm_note_kernel(LGM_KNAME("lgmweng::weng_gemm_kernel<64, 128>"));
  else if (bm == 128 && bn == 64) launch(weng_gemm_kernel<128, 64>), lgm_note_kernel(LGM_KNAME("lgmweng::weng_gemm_kernel<128, 64>"));
  else launch(weng_gemm_kernel<64, 64>), lgm_note_kernel(LGM_KNAME("lgmweng::weng_gemm_kernel<64, 64>"));
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

static int weng_check(const void* a, int64_t pitch, int C, const char* who) {
  LGM_REQUIRE(a && C > 0 && C % 4 == 0 && pitch % 4 == 0 && pitch >= C && lgm_aligned16(a),
              "%s: tensor must be non-null, 16-byte aligned, channels / pitch multiples of 4", who);
  return LGM_OK;
}
extern "C" int lgm_weng_f43_in(const float* x, int64_t x_pitch, int B, int H, int W, int C, float* V, void* stream) {
  if (int rc = weng_check(x, x_pitch, C, "weng_f43_in")) return rc;
  LGM_REQUIRE(V && lgm_aligned16(V) && B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, "weng_f43_in: H, W multiples of 4");
  const long n = (long)B * (H / 4) * (W / 4) * (C / 4);
  hipLaunchKernelGGL(f43_in_kernel, dim3(lgm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, (long)x_pitch, B, H, W, C, V);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
extern "C" int lgm_weng_f43_out(const float* M, int B, int H, int W, int N, const float* bias, float* y, int64_t y_pitch,
                                void* stream) {
  if (int rc = weng_check(y, y_pitch, N, "weng_f43_out")) return rc;
  LGM_REQUIRE(M && lgm_aligned16(M) && B > 0 && H % 4 == 0 && W % 4 == 0 && (!bias || lgm_aligned16(bias)), "weng_f43_out: bad arguments");
  const long n = (long)B * (H / 4) * (W / 4) * (N / 4);
  hipLaunchKernelGGL(f43_out_kernel, dim3(lgm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, M, B, H, W, N, bias, y, (long)y_pitch);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
extern "C" int lgm_weng_f42_in_xy(const float* x, int64_t x_pitch, int B, int H, int W, int C, float* V, void* stream) {
  if (int rc = weng_check(x, x_pitch, C, "weng_f42_in_xy")) return rc;
  LGM_REQUIRE(V && lgm_aligned16(V) && B > 0 && H % 8 == 0 && W % 8 == 0, "weng_f42_in_xy: H, W multiples of 8");
  const long n = (long)B * (H / 8) * (W / 8) * C;
  hipLaunchKernelGGL(f42_in_xy_kernel, dim3(lgm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, (long)x_pitch, B, H, W, C, V);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
extern "C" int lgm_weng_f42_out_xy(const float* M, int B, int Ho, int Wo, int N, const float* bias, float* y, int64_t y_pitch,
                                   void* stream) {
  if (int rc = weng_check(y, y_pitch, N, "weng_f42_out_xy")) return rc;
  LGM_REQUIRE(M && lgm_aligned16(M) && B > 0 && Ho % 4 == 0 && Wo % 4 == 0 && (!bias || lgm_aligned16(bias)), "weng_f42_out_xy: bad arguments");
  const long n = (long)B * (Ho / 4) * (Wo / 4) * (N / 4);
  hipLaunchKernelGGL(f42_out_xy_kernel, dim3(lgm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, M, B, Ho, Wo, N, bias, y, (long)y_pitch,
                     Epi{0, 0.f, nullptr, 0, 0.f});
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
static int weng_epi(Epi& e, int act, float slope, const float* mask, int64_t mask_pitch, float mask_slope, int C, const char* who) {
  LGM_REQUIRE(act == 0 || act == 3 || act == 4, "%s: activation %d not taken (0, 3 = ReLU, 4 = LeakyReLU)", who, act);
  LGM_REQUIRE(!mask || (lgm_aligned16(mask) && mask_pitch % 4 == 0 && mask_pitch >= C), "%s: mask must be 16-byte aligned, pitch %% 4", who);
  e = Epi{act, slope, mask, (long)mask_pitch, mask_slope};
  return LGM_OK;
}
extern "C" int lgm_weng_f42_out_xy_post(const float* M, int B, int Ho, int Wo, int N, const float* bias, float* y,
                                        int64_t y_pitch, int act, float slope, const float* mask, int64_t mask_pitch,
                                        float mask_slope, void* stream) {
  if (int rc = weng_check(y, y_pitch, N, "weng_f42_out_xy_post")) return rc;
  LGM_REQUIRE(M && lgm_aligned16(M) && B > 0 && Ho % 4 == 0 && Wo % 4 == 0 && (!bias || lgm_aligned16(bias)), "weng_f42_out_xy_post: bad arguments");
  Epi e;
  if (int rc = weng_epi(e, act, slope, mask, mask_pitch, mask_slope, N, "weng_f42_out_xy_post")) return rc;
  const long n = (long)B * (Ho / 4) * (Wo / 4) * (N / 4);
  hipLaunchKernelGGL(f42_out_xy_kernel, dim3(lgm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, M, B, Ho, Wo, N, bias, y, (long)y_pitch, e);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
extern "C" int lgm_weng_f42_in_yx(const float* dy, int64_t pitch, int B, int Ho, int Wo, int Ny, float* V, void* stream) {
  if (int rc = weng_check(dy, pitch, Ny, "weng_f42_in_yx")) return rc;
  LGM_REQUIRE(V && lgm_aligned16(V) && B > 0 && Ho % 4 == 0 && Wo % 4 == 0, "weng_f42_in_yx: Ho, Wo multiples of 4");
  const long n = (long)B * (Ho / 4) * (Wo / 4) * Ny;
  hipLaunchKernelGGL(f42_in_yx_kernel, dim3(lgm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, (long)pitch, B, Ho, Wo, Ny, V);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
extern "C" int lgm_weng_f42_out_yx(const float* M, int B, int Ho, int Wo, int C, const float* bias, float* dx, int64_t pitch,
                                   void* stream) {
  if (int rc = weng_check(dx, pitch, C, "weng_f42_out_yx")) return rc;
  LGM_REQUIRE(M && lgm_aligned16(M) && B > 0 && Ho % 4 == 0 && Wo % 4 == 0 && (!bias || lgm_aligned16(bias)), "weng_f42_out_yx: bad arguments");
  const long n = (long)B * (Ho / 4) * (Wo / 4) * C;
  hipLaunchKernelGGL(f42_out_yx_kernel, dim3(lgm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, M, B, Ho, Wo, C, bias, dx, (long)pitch,
                     Epi{0, 0.f, nullptr, 0, 0.f});
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
extern "C" int lgm_weng_f42_out_yx_post(const float* M, int B, int Ho, int Wo, int C, const float* bias, float* dx,
                                        int64_t pitch, int act, float slope, const float* mask, int64_t mask_pitch,
                                        float mask_slope, void* stream) {
  if (int rc = weng_check(dx, pitch, C, "weng_f42_out_yx_post")) return rc;
  LGM_REQUIRE(M && lgm_aligned16(M) && B > 0 && Ho % 4 == 0 && Wo % 4 == 0 && (!bias || lgm_aligned16(bias)), "weng_f42_out_yx_post: bad arguments");
  Epi e;
  if (int rc = weng_epi(e, act, slope, mask, mask_pitch, mask_slope, C, "weng_f42_out_yx_post")) return rc;
  const long n = (long)B * (Ho / 4) * (Wo / 4) * C;
  hipLaunchKernelGGL(f42_out_yx_kernel, dim3(lgm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, M, B, Ho, Wo, C, bias, dx, (long)pitch, e);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
/* U of both directions from the layer's weights (either output may be NULL) */
extern "C" int lgm_weng_f42_weights(const float* w, int Nw, int Cw, float* Uxy, float* Uyx, void* stream) {
  LGM_REQUIRE(w && lgm_aligned16(w) && Nw > 0 && Cw > 0 && Cw % 4 == 0 && (Uxy || Uyx) && (!Uxy || lgm_aligned16(Uxy)),
              "weng_f42_weights: bad arguments (Cw %% 4 == 0, 16-byte aligned)");
  const long n = (long)Nw * (Cw / 4);
  if (Uxy) hipLaunchKernelGGL(f42_weights_kernel<false>, dim3(lgm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, Nw, Cw, Uxy, Uyx);
  if (Uyx) hipLaunchKernelGGL(f42_weights_kernel<true>, dim3(lgm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, Nw, Cw, Uxy, Uyx);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
