// The VQ-VAE's ResidualStack forward in ONE launch (reference models/modules/residual.py:5-43; used by the encoder and
// the decoder, vqvae.py:45-47, :71-73):   per layer  y = relu(conv3x3(cur)),  cur' = relu(conv1x1(y) + cur)
// (`cur` arrives with the first in-place ReLU applied; the last cur' carries the stack's final ReLU).
//
// On the 4 x 4 maps of the 32 x 32 configuration a layer is two GEMMs of 4096 rows: six launches of 5-12 us per stack,
// all latency.  Here a workgroup owns TWO images (32 pixels) for the whole stack; the activations never leave LDS
// between layers (the tape copies the backward pass needs are written on the way).  Conventions of linattn_fused.hip: a
// lane owns one pixel, products are D[channel][pixel], the reduction index is dealt out in the order the operand
// registers hold it.  The 3x3 convolution's reduction (9 taps x 128 channels) is split over the four waves by channel
// quarter - partial sums through LDS, summed in wave order - and the 1x1 convolution's four 32-channel output tiles are
// one per wave.  Fixed summation orders: run-to-run identical.
#include "lgm_common.h"

namespace {

constexpr int C = 128;      // stack width
constexpr int R = 32;       // residual hidden width
constexpr int XL = C + 4;   // row stride of the activation tile [32 pixels][C]: 132 = 4 (mod 64), conflict-free 16-byte rows
constexpr int PL = R + 4;   // row stride of the partial-sum tiles
constexpr int MAXL = 4;

struct RArgs {
  const float* x;
  long x_pitch;
  const float* w3[MAXL];    // [R][9][C]
  const float* w1[MAXL];    // [C][1][R]
  float* y[MAXL];           // [B 16][R]
  float* z[MAXL];           // [B 16][C]
  long rows;                // B * 16
  int layers;
};

__global__ __launch_bounds__(256, 1) void resstack_fwd_kernel(const RArgs p) {
  __shared__ __align__(16) float Xs[32 * XL];
  __shared__ __align__(16) float Ps[4][32 * PL];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const long row0 = (long)blockIdx.x * 32;
  const long row = row0 + lr;
  const bool live = row < p.rows;
  const int py = (lr >> 2) & 3, px = lr & 3;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // the block's 32 pixel rows: 32 x 128 floats = 1024 16-byte pieces
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + 256 * u;
    const int r = e >> 5, c4 = (e & 31) * 4;
    f32x4 v = zero4;
    if (row0 + r < p.rows) v = *reinterpret_cast<const f32x4*>(p.x + (row0 + r) * p.x_pitch + c4);
    *reinterpret_cast<f32x4*>(Xs + r * XL + c4) = v;
  }
  __syncthreads();
  for (int l = 0; l < p.layers; ++l) {
    // ---- 3x3 convolution, this wave's channel quarter: partial y^T[n][pixel] ----
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* w3 = p.w3[l] + (long)lr * 9 * C + wid * 32 + 4 * lh;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ny = py + tap / 3 - 1, nx = px + tap % 3 - 1;
      const bool in = ny >= 0 && ny < 4 && nx >= 0 && nx < 4;
      const float* xp = Xs + ((lr & 16) + (in ? ny * 4 + nx : 0)) * XL + wid * 32 + 4 * lh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w3 + tap * C + 8 * g);
        f32x4 xv = *reinterpret_cast<const f32x4*>(xp + 8 * g);
        if (!in) xv = zero4;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j], xv[j], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<f32x4*>(&Ps[wid][lr * PL + 8 * g + 4 * lh]) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    __syncthreads();
    // ---- y = relu(sum of the four partials), in wave order; the lane's 16 values in operand order ----
    f32x4 yv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int o = lr * PL + 8 * g + 4 * lh;
      f32x4 s = (*reinterpret_cast<const f32x4*>(&Ps[0][o]) + *reinterpret_cast<const f32x4*>(&Ps[1][o])) +
                (*reinterpret_cast<const f32x4*>(&Ps[2][o]) + *reinterpret_cast<const f32x4*>(&Ps[3][o]));
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k] = fmaxf(s[k], 0.f);
      yv[g] = s;
      if (wid == 0 && live) *reinterpret_cast<f32x4*>(p.y[l] + row * R + 8 * g + 4 * lh) = s;
    }
    // ---- 1x1 convolution, this wave's 32 output channels; + cur, ReLU; back into the tile in place ----
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* w1 = p.w1[l] + (long)(wid * 32 + lr) * R + 4 * lh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w1 + 8 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j], yv[g][j], acc, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float* xo = Xs + lr * XL + wid * 32 + 8 * g + 4 * lh;
      f32x4 v = *reinterpret_cast<const f32x4*>(xo);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k] + acc[4 * g + k], 0.f);
      *reinterpret_cast<f32x4*>(xo) = v;
      if (live) *reinterpret_cast<f32x4*>(p.z[l] + row * C + wid * 32 + 8 * g + 4 * lh) = v;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int64_t lgm_resstack_fwd_supported(int H, int W, int Cin, int hidden, int Rh, int layers) {
  return H == 4 && W == 4 && Cin == C && hidden == C && Rh == R && layers >= 1 && layers <= MAXL ? 1 : 0;
}

// x: [B, 4, 4, 128] with the stack's first (in-place) ReLU applied; w3[l]: [32][9][128], w1[l]: [128][1][32] (the flat
// parameter layouts); y[l] [B,4,4,32] and z[l] [B,4,4,128] (contiguous) receive every layer's two activations.
extern "C" int lgm_resstack_fwd(const float* x, int64_t x_pitch, int B, int H, int W, int Cin, int hidden, int Rh,
                                int layers, const float* const* w3, const float* const* w1, float* const* y,
                                float* const* z, void* stream) {
  LGM_REQUIRE(lgm_resstack_fwd_supported(H, W, Cin, hidden, Rh, layers), "resstack_fwd: geometry %dx%d C=%d R=%d layers=%d unsupported",
              H, W, Cin, Rh, layers);
  LGM_REQUIRE(x && w3 && w1 && y && z && B > 0 && x_pitch % 4 == 0 && lgm_aligned16(x), "resstack_fwd: bad arguments");
  RArgs a{};
  a.x = x; a.x_pitch = x_pitch; a.rows = (long)B * 16; a.layers = layers;
  for (int l = 0; l < layers; ++l) {
    LGM_REQUIRE(w3[l] && w1[l] && y[l] && z[l] && lgm_aligned16(w3[l]) && lgm_aligned16(w1[l]) && lgm_aligned16(y[l]) &&
                    lgm_aligned16(z[l]), "resstack_fwd: layer %d: null / unaligned pointer", l);
    a.w3[l] = w3[l]; a.w1[l] = w1[l]; a.y[l] = y[l]; a.z[l] = z[l];
  }
  lgm_note_kernel(LGM_KNAME("resstack_fwd_kernel"));
  hipLaunchKernelGGL(resstack_fwd_kernel, dim3((unsigned)lgm_cdiv(a.rows, 32)), dim3(256), 0, (hipStream_t)stream, a);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
