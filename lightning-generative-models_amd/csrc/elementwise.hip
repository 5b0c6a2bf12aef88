// Small HBM-bound kernels: column sums, time embedding, activations, resampling copies,
// NCHW<->NHWC conversion, diffusion q_sample / v-target / weighted-MSE loss.
#include <atomic>
#include <stdarg.h>

#include "lgm_common.h"

// ---------------------------------------------------------------------------------------
// error string + ABI version
// ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void lgm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* lgm_last_error(void) { return g_err; }
static thread_local const char* g_kernel = "";
void lgm_note_kernel(const char* name) { g_kernel = name; }
extern "C" const char* lgm_last_kernel(void) { return g_kernel; }
// the name registry: one pointer per LGM_KNAME site of the whole library (section bounds from the linker)
extern "C" const char* const __start_lgm_knames[];
extern "C" const char* const __stop_lgm_knames[];
// CU margin: lgm_set_cu_margin(margin) (-1: back to the default), else LGM_CU_MARGIN, else 0.  The library no longer derives
// it from WORLD_SIZE (round 6): lgm_hip.lightning.FlatGradSync sets 16 for a rank whose exchange overlaps its backward pass
// on RCCL - the one case in which a collective's workgroups hold CUs beside the launches.  Measured with 16
// foreign 256-thread workgroups resident (tools/cu_hog_step.py, light F(4x4) workgroups, ms per step, margin 0 / 16 / 32):
// B = 64: 8.15 / 7.48 / 7.49 (alone 6.75 / 6.86 / 6.92); B = 16: 4.90 / 4.70 / 4.69 (alone 4.48 / 4.49 / 4.50).
static std::atomic<int> lgm_cu_margin_override{-1};
extern "C" int lgm_cu_margin(void) { return 256 - lgm_cu_budget(); }
extern "C" int lgm_set_cu_margin(int margin) {
  LGM_REQUIRE(margin <= 128, "lgm_set_cu_margin: margin %d leaves less than half of the chip", margin);
  lgm_cu_margin_override.store(margin, std::memory_order_relaxed);
  return LGM_OK;
}
int lgm_cu_budget() {
  static const int env_margin = getenv("LGM_CU_MARGIN") ? atoi(getenv("LGM_CU_MARGIN")) : 0;
  const int ov = lgm_cu_margin_override.load(std::memory_order_relaxed);
  int m = ov >= 0 ? ov : env_margin;
  if (m < 0) m = 0;
  if (m > 128) m = 128;
  return 256 - m;
}

extern "C" int lgm_kernel_name_count(void) { return (int)(__stop_lgm_knames - __start_lgm_knames); }
extern "C" const char* lgm_kernel_name(int i) {
  return (i >= 0 && i < lgm_kernel_name_count()) ? __start_lgm_knames[i] : nullptr;
}
extern "C" int lgm_abi_version(void) { return LGM_ABI_VERSION; }

namespace {

// ---------------------------------------------------------------------------------------
// colsum: out[c] = beta*out[c] + sum_r a[r, c]
// ---------------------------------------------------------------------------------------
// rows per stage-1 block: at least 64, and at most 128 splits overall
static inline long cs_rows(long rows) {
  long r = (rows + 127) / 128;
  if (r < 64) r = 64;
  return (r + 3) / 4 * 4;
}

// block = CL columns x (256 / CL) row lanes, CL = the power of two >= min(cols, 64) (a 4-column matrix - the bias
// gradient of an image-end layer - used 4 of 64 column lanes and ran 512 dependent loads per thread: 63 us for
// 4 MB); sums rows [r0, r1) of `a` (fixed order => deterministic)
__global__ __launch_bounds__(256) void colsum_stage(const float* __restrict__ a, long pitch, long rows, long cols,
                                                    long rows_per_block, float* __restrict__ out, long out_pitch,
                                                    float beta, int lg_cl) {
  __shared__ float sh[256];
  const int CL = 1 << lg_cl, RL = 256 >> lg_cl;
  const int cl = threadIdx.x & (CL - 1), rl = threadIdx.x >> lg_cl;
  const long c = (long)blockIdx.x * CL + cl;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float s0 = 0.f, s1 = 0.f;
  if (c < cols) {
    long r = r0 + rl;
    for (; r + RL < r1; r += 2 * RL) {   // two independent chains keep more loads in flight
      s0 += a[r * pitch + c];
      s1 += a[(r + RL) * pitch + c];
    }
    if (r < r1) s0 += a[r * pitch + c];
  }
  sh[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (rl == 0 && c < cols) {
    float v = 0.f;
    for (int k = 0; k < RL; ++k) v += sh[(k << lg_cl) + cl];
    float* o = out + (long)blockIdx.y * out_pitch + c;
    if (beta != 0.f) v += beta * o[0];
    o[0] = v;
  }
}

}  // namespace

extern "C" int64_t lgm_colsum_workspace(int64_t rows, int64_t cols) {
  return (int64_t)lgm_cdiv(rows, cs_rows(rows)) * cols * (int64_t)sizeof(float) + 16;
}

extern "C" int lgm_colsum(const float* a, int64_t pitch, int64_t rows, int64_t cols, float* out, float beta,
                          void* workspace, void* stream) {
  LGM_REQUIRE(a && out && workspace && rows > 0 && cols > 0, "colsum: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const long rpb = cs_rows(rows);
  const int ns = lgm_cdiv(rows, rpb);
  int lg = 2;                                    // column lanes: 4 ... 64
  while (lg < 6 && (1L << lg) < cols) ++lg;
  const int cl = 1 << lg;
  if (ns == 1) {
    hipLaunchKernelGGL(colsum_stage, dim3(lgm_cdiv(cols, cl), 1), dim3(256), 0, s, a, (long)pitch, (long)rows,
                       (long)cols, rpb, out, 0L, beta, lg);
  } else {
    hipLaunchKernelGGL(colsum_stage, dim3(lgm_cdiv(cols, cl), ns), dim3(256), 0, s, a, (long)pitch, (long)rows,
                       (long)cols, rpb, (float*)workspace, (long)cols, 0.f, lg);
    hipLaunchKernelGGL(colsum_stage, dim3(lgm_cdiv(cols, cl), 1), dim3(256), 0, s, (const float*)workspace,
                       (long)cols, (long)ns, (long)cols, (long)ns, out, 0L, beta, lg);
  }
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_colsum_deferred(const float* a, int64_t pitch, int64_t rows, int64_t cols, float* out, float beta,
                                   void* workspace, int64_t* desc, void* stream) {
  LGM_REQUIRE(a && out && workspace && desc && rows > 0 && cols > 0, "colsum_deferred: bad arguments");
  LGM_REQUIRE(cols % 4 == 0 && lgm_aligned16(out) && lgm_aligned16(workspace),
              "colsum_deferred: cols %% 4 == 0 and 16-byte aligned out / workspace required");
  const long rpb = cs_rows(rows);
  const int ns = lgm_cdiv(rows, rpb);
  int lg = 2;
  while (lg < 6 && (1L << lg) < cols) ++lg;
  hipLaunchKernelGGL(colsum_stage, dim3(lgm_cdiv(cols, 1 << lg), ns), dim3(256), 0, (hipStream_t)stream, a, (long)pitch,
                     (long)rows, (long)cols, rpb, (float*)workspace, (long)cols, 0.f, lg);
  LGM_LAUNCH_CHECK();
  union { float f; int64_t i; } bb;
  bb.i = 0; bb.f = beta;
  desc[0] = (int64_t)(uintptr_t)workspace; desc[1] = cols; desc[2] = (int64_t)(uintptr_t)out; desc[3] = cols;
  desc[4] = 0; desc[5] = 0; desc[6] = ns; desc[7] = bb.i;
  return LGM_OK;
}

// ---------------------------------------------------------------------------------------
// Sinusoidal position embedding  (ddpm.py:125-132): emb[b] = cat(sin(t*f), cos(t*f)).  The frequency
// table f[i] = exp(i * -(ln(theta)/(half-1))) is computed by the CALLER on the host exactly as the
// reference computes it (torch.exp on the CPU): a device expf differs by an ulp, which times t = 999 is
// 1e-4 in the argument.
// ---------------------------------------------------------------------------------------
namespace {
__global__ void posemb_kernel(const int64_t* __restrict__ t, int B, int dim, const float* __restrict__ freqs,
                              float* __restrict__ out, long pitch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (i >= B * half) return;
  const int b = i / half, j = i % half;
  const float arg = (float)t[b] * freqs[j];
  out[(long)b * pitch + j] = sinf(arg);
  out[(long)b * pitch + half + j] = cosf(arg);
}

enum { ACT_SILU = 1, ACT_GELU = 2, ACT_RELU = 3, ACT_LRELU = 4, ACT_TANH = 5 };

__device__ __forceinline__ float act_fwd(float x, int act, float slope) {
  switch (act) {
    case ACT_SILU: return x / (1.f + expf(-x));
    case ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
    case ACT_RELU: return x > 0.f ? x : 0.f;
    case ACT_LRELU: return x > 0.f ? x : x * slope;
    case ACT_TANH: return tanhf(x);
  }
  return x;
}
// derivative w.r.t. the pre-activation x
__device__ __forceinline__ float act_bwd(float x, int act, float slope) {
  switch (act) {
    case ACT_SILU: {
      const float s = 1.f / (1.f + expf(-x));
      return s * (1.f + x * (1.f - s));
    }
    case ACT_GELU: {
      const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
      return cdf + x * pdf;
    }
    case ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case ACT_LRELU: return x > 0.f ? 1.f : slope;
    case ACT_TANH: {
      const float t = tanhf(x);
      return 1.f - t * t;
    }
  }
  return 1.f;
}

// y[r, c] = act(x[r, c] + bias[c]) (+ res[r,c])   rows x cols with pitches; cols % 4 == 0
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, long x_pitch,
                                                      const float* __restrict__ bias, const float* __restrict__ res,
                                                      long res_pitch, float* __restrict__ y, long y_pitch, long rows,
                                                      int cols, int act, float slope) {
  const int c4n = cols / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * c4n) return;
  const long r = i / c4n;
  const int c = (int)(i % c4n) * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(x + r * x_pitch + c);
  if (bias) v += *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = act_fwd(v[k], act, slope);
  if (res) v += *reinterpret_cast<const f32x4*>(res + r * res_pitch + c);
  *reinterpret_cast<f32x4*>(y + r * y_pitch + c) = v;
}

// gx = gy * act'(x + bias)  (optionally accumulated)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x, long x_pitch,
                                                      const float* __restrict__ bias, const float* __restrict__ gy,
                                                      long gy_pitch, float* __restrict__ gx, long gx_pitch,
                                                      int accumulate, long rows, int cols, int act, float slope) {
  const int c4n = cols / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * c4n) return;
  const long r = i / c4n;
  const int c = (int)(i % c4n) * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(x + r * x_pitch + c);
  if (bias) v += *reinterpret_cast<const f32x4*>(bias + c);
  f32x4 g = *reinterpret_cast<const f32x4*>(gy + r * gy_pitch + c);
#pragma unroll
  for (int k = 0; k < 4; ++k) g[k] *= act_bwd(v[k], act, slope);
  float* o = gx + r * gx_pitch + c;
  if (accumulate) g += *reinterpret_cast<const f32x4*>(o);
  *reinterpret_cast<f32x4*>(o) = g;
}

// y = alpha*a + beta*b  on strided [rows, cols] matrices (b optional)
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ a, long a_pitch, float alpha,
                                                    const float* __restrict__ b, long b_pitch, float beta,
                                                    float* __restrict__ y, long y_pitch, long rows, int cols) {
  const int c4n = cols / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * c4n) return;
  const long r = i / c4n;
  const int c = (int)(i % c4n) * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(a + r * a_pitch + c) * alpha;
  if (b) v += *reinterpret_cast<const f32x4*>(b + r * b_pitch + c) * beta;
  *reinterpret_cast<f32x4*>(y + r * y_pitch + c) = v;
}

// nearest-neighbour x2 upsample, NHWC: y[b, 2h+i, 2w+j, :] = x[b, h, w, :]
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const float* __restrict__ x, long x_pitch,
                                                             float* __restrict__ y, long y_pitch, int B, int H, int W,
                                                             int C) {
  const int c4n = C / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * 2 * H * 2 * W * c4n;
  if (i >= total) return;
  const int c = (int)(i % c4n) * 4;
  long pix = i / c4n;
  const int ow = (int)(pix % (2 * W));
  pix /= 2 * W;
  const int oh = (int)(pix % (2 * H));
  const int b = (int)(pix / (2 * H));
  const long src = ((long)(b * H + (oh >> 1)) * W + (ow >> 1)) * x_pitch + c;
  const long dst = ((long)(b * 2 * H + oh) * 2 * W + ow) * y_pitch + c;
  *reinterpret_cast<f32x4*>(y + dst) = *reinterpret_cast<const f32x4*>(x + src);
}
// gx[b,h,w,:] = sum of the 4 gy children (fixed order)
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ gy, long gy_pitch,
                                                             float* __restrict__ gx, long gx_pitch, int B, int H, int W,
                                                             int C, int accumulate) {
  const int c4n = C / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * H * W * c4n;
  if (i >= total) return;
  const int c = (int)(i % c4n) * 4;
  long pix = i / c4n;
  const int w = (int)(pix % W);
  pix /= W;
  const int h = (int)(pix % H);
  const int b = (int)(pix / H);
  const long base = ((long)(b * 2 * H + 2 * h) * 2 * W + 2 * w) * gy_pitch + c;
  const long rowp = (long)2 * W * gy_pitch;
  f32x4 s = (*reinterpret_cast<const f32x4*>(gy + base) + *reinterpret_cast<const f32x4*>(gy + base + gy_pitch)) +
            (*reinterpret_cast<const f32x4*>(gy + base + rowp) +
             *reinterpret_cast<const f32x4*>(gy + base + rowp + gy_pitch));
  float* o = gx + ((long)(b * H + h) * W + w) * gx_pitch + c;
  if (accumulate) s += *reinterpret_cast<const f32x4*>(o);
  *reinterpret_cast<f32x4*>(o) = s;
}

// pixel-unshuffle "b c (h p1) (w p2) -> b (c p1 p2) h w" (ddpm.py:102) in NHWC:
// y[b, h, w, c*4 + p1*2 + p2] = x[b, 2h+p1, 2w+p2, c].  dir = 0 forward, 1 = inverse (gradient).
__global__ __launch_bounds__(256) void unshuffle_kernel(const float* __restrict__ src, long src_pitch,
                                                        float* __restrict__ dst, long dst_pitch, int B, int H, int W,
                                                        int C, int dir, int accumulate) {
  // H, W are the LOW-resolution sizes; C the high-resolution channel count.
  // one thread per (low-res pixel, channel c): moves the 4 (p1,p2) values -> float4 on the 4C side.
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * H * W * C;
  if (i >= total) return;
  const int c = (int)(i % C);
  long pix = i / C;
  const int w = (int)(pix % W);
  pix /= W;
  const int h = (int)(pix % H);
  const int b = (int)(pix / H);
  const long hi00 = ((long)(b * 2 * H + 2 * h) * 2 * W + 2 * w);
  const long lo = ((long)(b * H + h) * W + w);
  if (dir == 0) {
    f32x4 v;
    v[0] = src[(hi00)*src_pitch + c];
    v[1] = src[(hi00 + 1) * src_pitch + c];
    v[2] = src[(hi00 + 2 * W) * src_pitch + c];
    v[3] = src[(hi00 + 2 * W + 1) * src_pitch + c];
    *reinterpret_cast<f32x4*>(dst + lo * dst_pitch + c * 4) = v;
  } else {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + lo * src_pitch + c * 4);
    float* d = dst;
    if (accumulate) {
      d[(hi00)*dst_pitch + c] += v[0];
      d[(hi00 + 1) * dst_pitch + c] += v[1];
      d[(hi00 + 2 * W) * dst_pitch + c] += v[2];
      d[(hi00 + 2 * W + 1) * dst_pitch + c] += v[3];
    } else {
      d[(hi00)*dst_pitch + c] = v[0];
      d[(hi00 + 1) * dst_pitch + c] = v[1];
      d[(hi00 + 2 * W) * dst_pitch + c] = v[2];
      d[(hi00 + 2 * W + 1) * dst_pitch + c] = v[3];
    }
  }
}

// NCHW [B,C,H,W] dense  <->  NHWC with pitch (pad channels written as zero on the way in)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           long dst_pitch, int B, int C, int HW, int Cpad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * HW * Cpad;
  if (i >= total) return;
  const int c = (int)(i % Cpad);
  const long pix = i / Cpad;
  const int b = (int)(pix / HW);
  const int p = (int)(pix % HW);
  dst[pix * dst_pitch + c] = c < C ? src[((long)b * C + c) * HW + p] : 0.f;
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, long src_pitch,
                                                           float* __restrict__ dst, int B, int C, int HW) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * C * HW;
  if (i >= total) return;
  const int p = (int)(i % HW);
  const long bc = i / HW;
  const int c = (int)(bc % C);
  const int b = (int)(bc / C);
  dst[i] = src[((long)b * HW + p) * src_pitch + c];
}

// ---------------------------------------------------------------------------------------
// Diffusion training elementwise (ddpm.py:869-876, 684-688, 945):
//   x0 = img*2-1 (auto_normalize) ; x_t = sa[t]*x0 + sb[t]*noise ; v = sa[t]*noise - sb[t]*x0
// img/noise NCHW dense [B,C,HW]; outputs NHWC with pitch (pad channels zeroed).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qsample_kernel(const float* __restrict__ img, const float* __restrict__ noise,
                                                      const int64_t* __restrict__ t, const float* __restrict__ sa,
                                                      const float* __restrict__ sb, int normalize,
                                                      float* __restrict__ xt, float* __restrict__ target, long pitch,
                                                      int B, int C, int HW, int Cpad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * HW * Cpad;
  if (i >= total) return;
  const int c = (int)(i % Cpad);
  const long pix = i / Cpad;
  const int b = (int)(pix / HW);
  const int p = (int)(pix % HW);
  float xv = 0.f, tv = 0.f;
  if (c < C) {
    const long s = ((long)b * C + c) * HW + p;
    float x0 = img[s];
    if (normalize) x0 = x0 * 2.f - 1.f;
    const float n = noise[s];
    const float a = sa[t[b]], bb = sb[t[b]];
    xv = a * x0 + bb * n;
    tv = a * n - bb * x0;
  }
  xt[pix * pitch + c] = xv;
  if (target) target[pix * pitch + c] = tv;
}

// per-sample weighted MSE (ddpm.py:921-925): loss = mean_b( w[t_b] * mean_{chw} (out-target)^2 )
// stage 1: one block per sample -> per-sample value; stage 2: one block -> scalar.  Also emits
// gout = gscale * 2 * w[t_b] * (out - target) / (C*HW*B) when gout != null (gscale read from device).
__global__ __launch_bounds__(256) void mse_sample_kernel(const float* __restrict__ out, const float* __restrict__ target,
                                                         long pitch, const int64_t* __restrict__ t,
                                                         const float* __restrict__ lw, int C, int HW, int Cpad,
                                                         float* __restrict__ per_sample) {
  __shared__ float sh[16];
  const int b = blockIdx.x;
  const long n = (long)HW * Cpad;
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    const int c = (int)(i % Cpad);
    const long pix = (long)b * HW + i / Cpad;
    if (c < C) {
      const float d = out[pix * pitch + c] - target[pix * pitch + c];
      s += d * d;
    }
  }
  s = lgm_block_sum(s, sh);
  if (threadIdx.x == 0) per_sample[b] = s / ((float)C * (float)HW) * (lw ? lw[t[b]] : 1.f);
}
// the same with one 16-byte load per pixel and operand (Cpad == 4: the three image channels + one lane of padding): the scalar
// form above spends 12.8 us per launch at every batch on sixteen dependent rounds of 64-bit divisions and 4-byte loads
__global__ __launch_bounds__(256) void mse_sample4_kernel(const float* __restrict__ out, const float* __restrict__ target,
                                                          long pitch, const int64_t* __restrict__ t,
                                                          const float* __restrict__ lw, int C, int HW,
                                                          float* __restrict__ per_sample) {
  __shared__ float sh[16];
  const int b = blockIdx.x;
  float s = 0.f;
  for (int p0 = 0; p0 < HW; p0 += 1024) {                 // four pixels per thread and round: eight loads in flight
    f32x4 o[4], g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int px = p0 + threadIdx.x + 256 * u;
      const long off = ((long)b * HW + (px < HW ? px : HW - 1)) * pitch;
      o[u] = *reinterpret_cast<const f32x4*>(out + off);
      g[u] = *reinterpret_cast<const f32x4*>(target + off);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool live = p0 + threadIdx.x + 256 * u < HW;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float d = o[u][c] - g[u][c];
        s += (live && c < C) ? d * d : 0.f;
      }
    }
  }
  s = lgm_block_sum(s, sh);
  if (threadIdx.x == 0) per_sample[b] = s / ((float)C * (float)HW) * (lw ? lw[t[b]] : 1.f);
}
__global__ void mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += v[i];
  s = lgm_block_sum(s, sh);
  if (threadIdx.x == 0) out[0] = s / (float)n;
}
__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ out, const float* __restrict__ target,
                                                      long pitch, const int64_t* __restrict__ t,
                                                      const float* __restrict__ lw, const float* __restrict__ gloss,
                                                      int B, int C, int HW, int Cpad, float* __restrict__ gout) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * HW * Cpad;
  if (i >= total) return;
  const int c = (int)(i % Cpad);
  const long pix = i / Cpad;
  const int b = (int)(pix / HW);
  float g = 0.f;
  if (c < C) {
    const float w = lw ? lw[t[b]] : 1.f;
    const float scale = gloss[0] * 2.f * w / ((float)C * (float)HW * (float)B);
    g = scale * (out[pix * pitch + c] - target[pix * pitch + c]);
  }
  gout[pix * pitch + c] = g;
}


// One reverse-diffusion update for a whole batch at a shared timestep (ddpm.py:707-757, 805-829):
//   x0  = clamp(A*x + Bv*v, -1, 1)            (predict_start_from_v + clip)
//   eps = (R*x - x0) / Rm1                    (predict_noise_from_start)
//   out = C0*x0 + C1*x + C2*eps + C3*noise
// x, v: NHWC pitch Cpad; noise: NCHW dense (or null); out: NHWC pitch Cpad; x0_out optional.
// one update of one element; contraction off so that the by-value and the table-driven kernel round identically
__device__ __forceinline__ void sample_update(float xv, float vv, float nz, float A, float Bv, int clip, float R,
                                              float Rm1, float C0, float C1, float C2, float C3, float& o, float& x0) {
#pragma clang fp contract(off)
  x0 = A * xv + Bv * vv;
  if (clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
  const float eps = (R * xv - x0) / Rm1;
  o = C0 * x0 + C1 * xv + C2 * eps;
  if (C3 != 0.f) o += C3 * nz;
}

__global__ __launch_bounds__(256) void sample_step_kernel(const float* __restrict__ x, const float* __restrict__ v,
                                                          const float* __restrict__ noise, float* __restrict__ out,
                                                          float* __restrict__ x0_out, int B, int C, int HW, int Cpad,
                                                          float A, float Bv, int clip, float R, float Rm1, float C0,
                                                          float C1, float C2, float C3) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * HW * Cpad;
  if (i >= total) return;
  const int c = (int)(i % Cpad);
  const long pix = i / Cpad;
  float o = 0.f, x0 = 0.f;
  if (c < C) {
    float nz = 0.f;
    if (noise && C3 != 0.f) {
      const int b = (int)(pix / HW), p = (int)(pix % HW);
      nz = noise[((long)b * C + c) * HW + p];
    }
    sample_update(x[i], v[i], nz, A, Bv, clip, R, Rm1, C0, C1, C2, noise ? C3 : 0.f, o, x0);
  }
  out[i] = o;
  if (x0_out) x0_out[i] = x0;
}

// GaussianDiffusion's `extract(table, t, shape) * tensor` algebra with a PER-SAMPLE timestep (ddpm.py:673-705, 869-876) on
// dense NCHW tensors: one thread per element, the three table values of the sample are wave-uniform loads.  Contraction is
// off: the reference rounds both products and the sum separately.  A timestep outside the table is clamped to it (the
// reference's gather raises; a device kernel must not fault).
__device__ __forceinline__ float extract_axpby_one(float a, float bb, float d, float xv, float yv, int clip) {
#pragma clang fp contract(off)
  float o = a * xv + bb * yv;
  if (d != 1.f) o = o / d;
  if (clip) o = fminf(fmaxf(o, -1.f), 1.f);
  return o;
}
__global__ __launch_bounds__(256) void extract_axpby_kernel(const float* __restrict__ ta, const float* __restrict__ tb,
                                                            const float* __restrict__ td, const long* __restrict__ t,
                                                            const float* __restrict__ x, const float* __restrict__ y,
                                                            float sb, int clip, float* __restrict__ out, long per,
                                                            int n_table) {
#pragma clang fp contract(off)
  const int b = blockIdx.y;
  long ti = t[b];
  ti = ti < 0 ? 0 : (ti >= n_table ? n_table - 1 : ti);
  const float a = ta ? ta[ti] : 1.f, bb = sb * (tb ? tb[ti] : 1.f), d = td ? td[ti] : 1.f;
  const long base = (long)b * per;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x)
    out[base + i] = extract_axpby_one(a, bb, d, x[base + i], y ? y[base + i] : 0.f, clip);
}
__global__ __launch_bounds__(256) void model_predictions_kernel(const float* __restrict__ x, const float* __restrict__ v,
                                                                const long* __restrict__ t, const float* __restrict__ sa,
                                                                const float* __restrict__ s1, const float* __restrict__ r,
                                                                const float* __restrict__ rm1, int clip,
                                                                float* __restrict__ pn, float* __restrict__ xs, long per,
                                                                int n_table) {
#pragma clang fp contract(off)
  const int b = blockIdx.y;
  long ti = t[b];
  ti = ti < 0 ? 0 : (ti >= n_table ? n_table - 1 : ti);
  const float A = sa[ti], S = s1[ti], R = r[ti], Rm1 = rm1[ti];
  const long base = (long)b * per;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
    const float xv = x[base + i];
    float x0 = A * xv - S * v[base + i];
    if (clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
    xs[base + i] = x0;
    pn[base + i] = (R * xv - x0) / Rm1;
  }
}

// Graph-replayed sampling (lgm_hip/sampler.py): the per-step scalars come from a device table indexed by a
// device-side step counter, so ONE captured graph serves every step of a chain.
//   sampler_time_kernel : t[b] = ttable[counter]                              (before the UNet forward)
//   sample_step_table   : sample_step with row `counter` of table[n][8] = (A, Bv, R, Rm1, C0, C1, C2, C3), in place
//   sampler_advance     : counter += 1                                        (last node of the graph)
__global__ void sampler_time_kernel(const long* __restrict__ ttable, const int* __restrict__ counter,
                                    long* __restrict__ t, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) t[i] = ttable[counter[0]];
}
__global__ __launch_bounds__(256) void sample_step_table_kernel(float* __restrict__ x, const float* __restrict__ v,
                                                                const float* __restrict__ noise,
                                                                float* __restrict__ x0_out, int B, int C, int HW,
                                                                int Cpad, const float* __restrict__ table,
                                                                const int* __restrict__ counter, int clip) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * HW * Cpad;
  if (i >= total) return;
  const float* row = table + 8 * counter[0];
  const float A = row[0], Bv = row[1], R = row[2], Rm1 = row[3], C0 = row[4], C1 = row[5], C2 = row[6], C3 = row[7];
  const int c = (int)(i % Cpad);
  const long pix = i / Cpad;
  float o = 0.f, x0 = 0.f;
  if (c < C) {
    float nz = 0.f;
    if (noise && C3 != 0.f) {
      const int b = (int)(pix / HW), p = (int)(pix % HW);
      nz = noise[((long)b * C + c) * HW + p];
    }
    sample_update(x[i], v[i], nz, A, Bv, clip, R, Rm1, C0, C1, C2, noise ? C3 : 0.f, o, x0);
  }
  x[i] = o;
  if (x0_out) x0_out[i] = x0;
}
__global__ void sampler_advance_kernel(int* counter) { counter[0] += 1; }

// ---------------------------------------------------------------------------------------
// The UNet's time embedding in ONE launch (reference ddpm.py:119-132 SinusoidalPosEmb, :328-333 time_mlp = Linear ->
// GELU -> Linear, and the SiLU in front of every ResnetBlock.mlp's Linear :181-183).  It was six launches of 4 - 7 us that
// no batch size shrinks (posemb, GEMM, GELU, split-K GEMM, reducer, SiLU): ~32 us of every training step and of every
// sampling step.  A workgroup owns RB rows of the batch for the whole chain; a row's vectors live in LDS / registers; a
// weight row is read by lpr = min(64, K / 4) lanes together (16 bytes each: one coalesced instruction per row, or per
// 64 / lpr rows) and summed by a fixed xor-butterfly, so a row's result does not depend on the batch it arrives in
// (2 ranks x B/2 rows == 1 rank x B rows bit for bit).
// ---------------------------------------------------------------------------------------
// One linear layer of the chain for the RB rows a workgroup owns: out[r][n] = bias[n] + sum_k xs[r][k] W[n][k], K <= 256.
// lpr = K / 4 lanes share a weight row (16 bytes each; k order per lane k = 4 sub + j, then a fixed xor-butterfly); a wave
// owns the rows n = wave * rpp + grp + u * stride.  EVERY global load of a layer (U weight rows + their bias per lane) is
// issued by tm_fetch() before anything is computed - for both layers at kernel entry - because a load that misses costs
// 1 - 2 us here (every launch starts cold): the first version (one weight row per loop trip, bias read in the epilogue) spent
// 140 us in ~70 dependent round trips, the second (eight rows per trip) 50 us.
constexpr int TM_RB = 4;
constexpr int TM_THREADS = 512;    // 8 waves, two per SIMD: 256 registers per lane for the prefetched weight rows
template <int U>
struct TmRows {
  f32x4 w[U];
  float b[U];
};
typedef unsigned tm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tm_rsrc(const float* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// Raw buffer loads: ONE lane offset per operand (row wave * rpp + grp, column 4 sub) and a compile-time scalar offset per u
// (rows u * stride apart: stride * K = nwaves * 256 floats whatever K is), so the U loads of a wave need no address
// registers (64-bit flat addresses for 32 rows spilled 340 registers); a row >= N is outside the descriptor's range and
// reads as zeros.
template <int U>
__device__ __forceinline__ void tm_fetch(TmRows<U>& R, const float* __restrict__ W, const float* __restrict__ bias, int K,
                                         int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NW = TM_THREADS / 64;
  const int lpr = K / 4, rpp = 64 / lpr, sub = lane % lpr, grp = lane / lpr;
  const __amdgpu_buffer_rsrc_t rw = tm_rsrc(W, (unsigned)N * (unsigned)K * 4u);
  const __amdgpu_buffer_rsrc_t rb = tm_rsrc(bias, (unsigned)N * 4u);
  const unsigned vw = ((unsigned)(wave * rpp + grp) * (unsigned)K + 4u * (unsigned)sub) * 4u;
  const unsigned vb = (unsigned)(wave * rpp + grp) * 4u;
  const unsigned sb = (unsigned)(NW * rpp) * 4u;      // bias: rows are stride apart (wave-uniform, runtime)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    R.w[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, vw, (unsigned)u * (NW * 256u * 4u), 0));
    R.b[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, vb, (unsigned)u * sb, 0));
  }
}
template <int U, class Emit>
__device__ __forceinline__ void tm_rows(const TmRows<U>& R, const float* __restrict__ xs, int K, int N, Emit emit) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int lpr = K / 4, rpp = 64 / lpr, sub = lane % lpr, grp = lane / lpr, stride = nwaves * rpp;
  float xin[TM_RB][4];
#pragma unroll
  for (int r = 0; r < TM_RB; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) xin[r][j] = xs[r * K + 4 * sub + j];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int nb = wave * rpp + u * stride;           // wave-uniform: the butterfly below is never divergent
    if (nb >= N) break;
    float acc[TM_RB];
#pragma unroll
    for (int r = 0; r < TM_RB; ++r) {
      acc[r] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[r] = fmaf(xin[r][j], R.w[u][j], acc[r]);
    }
    for (int off = lpr >> 1; off > 0; off >>= 1)
#pragma unroll
      for (int r = 0; r < TM_RB; ++r) acc[r] += __shfl_xor(acc[r], off, 64);
    if (sub == 0 && nb + grp < N) emit(nb + grp, R.b[u], acc);
  }
}

__global__ __launch_bounds__(TM_THREADS) void time_mlp_fwd_kernel(const long* __restrict__ t, int B, int dim,
                                                                  const float* __restrict__ freqs,
                                                                  const float* __restrict__ W1, const float* __restrict__ b1,
                                                                  const float* __restrict__ W2, const float* __restrict__ b2,
                                                                  int td, float* __restrict__ pe, float* __restrict__ a1,
                                                                  float* __restrict__ h, float* __restrict__ temb,
                                                                  float* __restrict__ st) {
  extern __shared__ float tm_sm[];
  float* spe = tm_sm;                    // [RB][dim]
  float* sh = tm_sm + TM_RB * dim;       // [RB][td]  activations (the next layer's input)
  float* sa = sh + TM_RB * td;           // [RB][td]  pre-activations
  const int r0 = blockIdx.x * TM_RB, half = dim / 2;
  // 8 waves.  Layer 2 (K = td <= 256): at most 32 weight rows per wave (rpp = 1 at td = 256); layer 1: td * dim / 2048 <= 8
  // (host check: time_dim * dim <= 16384).
  TmRows<8> R1;
  TmRows<32> R2;
  tm_fetch(R1, W1, b1, dim, td);
  for (int i = threadIdx.x; i < TM_RB * half; i += blockDim.x) {
    const int r = i / half, j = i % half, b = r0 + r;
    float sv = 0.f, cv = 0.f;
    if (b < B) {                         // posemb_kernel's arithmetic
      const float arg = (float)t[b] * freqs[j];
      sv = sinf(arg), cv = cosf(arg);
      pe[(long)b * dim + j] = sv;
      pe[(long)b * dim + half + j] = cv;
    }
    spe[r * dim + j] = sv;
    spe[r * dim + half + j] = cv;
  }
  tm_fetch(R2, W2, b2, td, td);          // in flight while layer 1 is computed
  __syncthreads();
  // the activations run in their own elementwise passes: inside the unrolled row loop 32 inlined erff / expf bodies cost
  // 310 spilled registers
  tm_rows(R1, spe, dim, td, [&](int n, float bias, const float* acc) {
#pragma unroll
    for (int r = 0; r < TM_RB; ++r) sa[r * td + n] = acc[r] + bias;
  });
  __syncthreads();
  for (int i = threadIdx.x; i < TM_RB * td; i += blockDim.x) {
    const int r = i / td, n = i % td;
    const float a = sa[i], g = act_fwd(a, ACT_GELU, 0.f);
    sh[i] = g;
    if (r0 + r < B) a1[(long)(r0 + r) * td + n] = a, h[(long)(r0 + r) * td + n] = g;
  }
  __syncthreads();
  tm_rows(R2, sh, td, td, [&](int n, float bias, const float* acc) {
#pragma unroll
    for (int r = 0; r < TM_RB; ++r) sa[r * td + n] = acc[r] + bias;
  });
  __syncthreads();
  for (int i = threadIdx.x; i < TM_RB * td; i += blockDim.x) {
    const int r = i / td, n = i % td;
    if (r0 + r < B) {
      const float e = sa[i];
      temb[(long)(r0 + r) * td + n] = e;
      st[(long)(r0 + r) * td + n] = act_fwd(e, ACT_SILU, 0.f);
    }
  }
}

// backward, row-local half: gtemb = gst * silu'(temb); gh = gtemb W2 (thread = (k, quarter of the n range): coalesced weight
// rows, eight of them requested per trip, LDS-broadcast gtemb; the four quarters are added in order); ga1 = gh * gelu'(a1)
__global__ __launch_bounds__(TM_THREADS) void time_mlp_bwd_rows_kernel(const float* __restrict__ gst,
                                                                       const float* __restrict__ a1,
                                                                       const float* __restrict__ temb,
                                                                       const float* __restrict__ W2, int B, int td,
                                                                       float* __restrict__ gtemb, float* __restrict__ ga1) {
  extern __shared__ float tm_sm[];       // [td][RB] (one 16-byte broadcast read per n), then [4][td][RB] partial sums
  float* part = tm_sm + td * TM_RB;
  const int r0 = blockIdx.x * TM_RB;
  for (int i = threadIdx.x; i < TM_RB * td; i += blockDim.x) {
    const int r = i / td, n = i % td, b = r0 + r;
    float g = 0.f;
    if (b < B) {
      g = gst[(long)b * td + n] * act_bwd(temb[(long)b * td + n], ACT_SILU, 0.f);
      gtemb[(long)b * td + n] = g;
    }
    tm_sm[n * TM_RB + r] = g;
  }
  __syncthreads();
  const int kq = blockDim.x / 4;                      // threads along k per quarter (256)
  const int q = threadIdx.x / kq, kt = threadIdx.x % kq;
  const int nq = td / 4;                              // n range of a quarter (td % 16 == 0)
  for (int k = kt; k < td; k += kq) {
    float acc[TM_RB];
#pragma unroll
    for (int r = 0; r < TM_RB; ++r) acc[r] = 0.f;
    for (int n = q * nq; n < (q + 1) * nq; n += 8) {
      float w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = W2[(long)(n + u) * td + k];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(tm_sm + (n + u) * TM_RB);
#pragma unroll
        for (int r = 0; r < TM_RB; ++r) acc[r] = fmaf(g[r], w[u], acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < TM_RB; ++r) part[(q * td + k) * TM_RB + r] = acc[r];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TM_RB * td; i += blockDim.x) {
    const int k = i / TM_RB, r = i % TM_RB;
    if (r0 + r < B) {
      const float gh = ((part[(0 * td + k) * TM_RB + r] + part[(1 * td + k) * TM_RB + r]) + part[(2 * td + k) * TM_RB + r]) +
                       part[(3 * td + k) * TM_RB + r];
      ga1[(long)(r0 + r) * td + k] = gh * act_bwd(a1[(long)(r0 + r) * td + k], ACT_GELU, 0.f);
    }
  }
}

// backward, batch-reducing half: gW[n][k] = beta gW[n][k] + sum_r gy[r][n] x[r][k], gb[n] = beta gb[n] + sum_r gy[r][n], rows in
// order (deterministic).  A workgroup owns TN consecutive n; 32 rows of x and gy at a time are staged in LDS by all threads
// (many loads in flight: the row-by-row version spent its 120 us waiting for one load per trip), thread = (k, n slice).
// blockIdx.y = 0: the second linear (x = h, K = td), 1: the first (x = pe, K = dim).  K <= 256.
__global__ __launch_bounds__(256) void time_mlp_bwd_wgrad_kernel(const float* __restrict__ gtemb, const float* __restrict__ h,
                                                                 const float* __restrict__ ga1, const float* __restrict__ pe,
                                                                 int B, int dim, int td, float* __restrict__ gw2,
                                                                 float* __restrict__ gb2, float* __restrict__ gw1,
                                                                 float* __restrict__ gb1, float beta) {
  constexpr int TN = 16, RC = 32;
  const bool first = blockIdx.y == 1;
  const float* gy = first ? ga1 : gtemb;
  const float* x = first ? pe : h;
  const int K = first ? dim : td;
  float* gw = first ? gw1 : gw2;
  float* gb = first ? gb1 : gb2;
  const int n0 = blockIdx.x * TN;
  __shared__ float sgy[RC][TN];
  __shared__ float sx[RC * 256];
  const int kthreads = K;                            // threads along k (K <= 256, a power of two >= 16)
  const int nsl = 256 / kthreads;                    // n slices
  const int per = TN / nsl;                          // outputs per thread along n
  const int kk = threadIdx.x % kthreads, sl = threadIdx.x / kthreads;
  float acc[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) acc[i] = 0.f;
  float bsum = 0.f;
  for (int rb = 0; rb < B; rb += RC) {
    const int rows = B - rb < RC ? B - rb : RC;
    __syncthreads();
    {   // all of a chunk's loads first, then the LDS writes (a load -> store loop is one round trip per trip)
      float tg[RC * TN / 256], tx[RC];
#pragma unroll
      for (int q = 0; q < RC * TN / 256; ++q) {
        const int i = threadIdx.x + 256 * q, r = i / TN, j = i % TN;
        tg[q] = r < rows ? gy[(long)(rb + r) * td + n0 + j] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < RC; ++q) {
        const int i = threadIdx.x + 256 * q;          // rows are dense: one flat copy of rows * K floats
        tx[q] = i < rows * K ? x[(long)rb * K + i] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < RC * TN / 256; ++q) {
        const int i = threadIdx.x + 256 * q;
        sgy[i / TN][i % TN] = tg[q];
      }
#pragma unroll
      for (int q = 0; q < RC; ++q)
        if (threadIdx.x + 256 * q < RC * K) sx[threadIdx.x + 256 * q] = tx[q];
    }
    __syncthreads();
    for (int r = 0; r < rows; ++r) {
      const float xv = sx[r * K + kk];
#pragma unroll
      for (int i = 0; i < TN; ++i)
        if (i < per) acc[i] = fmaf(sgy[r][sl * per + i], xv, acc[i]);
    }
    if (threadIdx.x < TN)
      for (int r = 0; r < rows; ++r) bsum += sgy[r][threadIdx.x];
  }
#pragma unroll
  for (int i = 0; i < TN; ++i)
    if (i < per) {
      float* d = gw + (long)(n0 + sl * per + i) * K + kk;
      *d = beta != 0.f ? beta * *d + acc[i] : acc[i];
    }
  if (threadIdx.x < TN) {
    float* d = gb + n0 + threadIdx.x;
    *d = beta != 0.f ? beta * *d + bsum : bsum;
  }
}
}  // namespace

extern "C" int lgm_extract_axpby(const float* ta, const float* tb, const float* td, const int64_t* t, const float* x,
                                 const float* y, float sb, int clip, float* out, int B, int64_t per_sample, int n_table,
                                 void* stream) {
  LGM_REQUIRE(t && x && out && B > 0 && B <= 65535 && per_sample > 0 && n_table > 0, "extract_axpby: bad arguments");
  LGM_REQUIRE(y || !tb, "extract_axpby: a second table without a second tensor");
  const int gx = (int)(per_sample < 256L * 4096 ? lgm_cdiv(per_sample, 256) : 4096);
  hipLaunchKernelGGL(extract_axpby_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, ta, tb, td, (const long*)t, x,
                     y, sb, clip, out, (long)per_sample, n_table);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
extern "C" int lgm_model_predictions(const float* x, const float* v, const int64_t* t, const float* sqrt_ac,
                                     const float* sqrt_1mac, const float* sqrt_recip, const float* sqrt_recipm1, int clip,
                                     float* pred_noise, float* x_start, int B, int64_t per_sample, int n_table,
                                     void* stream) {
  LGM_REQUIRE(x && v && t && sqrt_ac && sqrt_1mac && sqrt_recip && sqrt_recipm1 && pred_noise && x_start && B > 0 &&
                  B <= 65535 && per_sample > 0 && n_table > 0,
              "model_predictions: bad arguments");
  const int gx = (int)(per_sample < 256L * 4096 ? lgm_cdiv(per_sample, 256) : 4096);
  hipLaunchKernelGGL(model_predictions_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x, v, (const long*)t,
                     sqrt_ac, sqrt_1mac, sqrt_recip, sqrt_recipm1, clip, pred_noise, x_start, (long)per_sample, n_table);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
static bool tm_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static bool tm_dims_ok(int dim, int td) {
  // K / 4 lanes per weight row must be a power of two <= 64, or K a multiple of 256 up to 1024
  auto ok = [](int K) { return K >= 16 && K <= 256 && K % 4 == 0 && tm_pow2(K / 4); };   // K / 4 lanes share a weight row
  return ok(dim) && ok(td) && td % 32 == 0 && dim % 2 == 0 && (long)td * dim <= 16384;
}
extern "C" int64_t lgm_time_mlp_supported(int dim, int time_dim) { return tm_dims_ok(dim, time_dim) ? 1 : 0; }
extern "C" int lgm_time_mlp_fwd(const int64_t* t, int B, int dim, const float* freqs, const float* w1, const float* b1,
                                const float* w2, const float* b2, int time_dim, float* pe, float* a1, float* h,
                                float* temb, float* st, void* stream) {
  LGM_REQUIRE(t && freqs && w1 && b1 && w2 && b2 && pe && a1 && h && temb && st && B > 0, "time_mlp_fwd: bad arguments");
  LGM_REQUIRE(tm_dims_ok(dim, time_dim), "time_mlp_fwd: dim %d / time_dim %d not taken (lgm_time_mlp_supported)", dim, time_dim);
  LGM_REQUIRE(lgm_aligned16(w1) && lgm_aligned16(w2), "time_mlp_fwd: weights must be 16-byte aligned");
  const size_t sm = sizeof(float) * TM_RB * (size_t)(dim + 2 * time_dim);
  hipLaunchKernelGGL(time_mlp_fwd_kernel, dim3(lgm_cdiv(B, TM_RB)), dim3(TM_THREADS), sm, (hipStream_t)stream, (const long*)t, B,
                     dim, freqs, w1, b1, w2, b2, time_dim, pe, a1, h, temb, st);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
extern "C" int lgm_time_mlp_bwd(const float* gst, const float* pe, const float* a1, const float* h, const float* temb,
                                const float* w2, int B, int dim, int time_dim, float* gtemb, float* ga1, float* gw1,
                                float* gb1, float* gw2, float* gb2, float beta, void* stream) {
  LGM_REQUIRE(gst && pe && a1 && h && temb && w2 && gtemb && ga1 && gw1 && gb1 && gw2 && gb2 && B > 0,
              "time_mlp_bwd: bad arguments");
  LGM_REQUIRE(tm_dims_ok(dim, time_dim), "time_mlp_bwd: dim %d / time_dim %d not taken (lgm_time_mlp_supported)", dim, time_dim);
  hipLaunchKernelGGL(time_mlp_bwd_rows_kernel, dim3(lgm_cdiv(B, TM_RB)), dim3(TM_THREADS), sizeof(float) * TM_RB * time_dim * 5,
                     (hipStream_t)stream, gst, a1, temb, w2, B, time_dim, gtemb, ga1);
  hipLaunchKernelGGL(time_mlp_bwd_wgrad_kernel, dim3(time_dim / 16, 2), dim3(256), 0, (hipStream_t)stream, gtemb, h, ga1, pe,
                     B, dim, time_dim, gw2, gb2, gw1, gb1, beta);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
extern "C" int lgm_sampler_time(const int64_t* ttable, const int32_t* counter, int64_t* t, int B, void* stream) {
  LGM_REQUIRE(ttable && counter && t && B > 0, "sampler_time: bad arguments");
  hipLaunchKernelGGL(sampler_time_kernel, dim3(lgm_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const long*)ttable, (const int*)counter, (long*)t, B);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
extern "C" int lgm_sample_step_table(float* x, const float* v, const float* noise, float* x0_out, int B, int C, int HW,
                                     int Cpad, const float* table, const int32_t* counter, int clip, int advance,
                                     void* stream) {
  LGM_REQUIRE(x && v && table && counter && B > 0 && C > 0 && HW > 0 && Cpad >= C, "sample_step_table: bad arguments");
  hipLaunchKernelGGL(sample_step_table_kernel, dim3(lgm_cdiv((long)B * HW * Cpad, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, v, noise, x0_out, B, C, HW, Cpad, table, (const int*)counter, clip);
  if (advance) hipLaunchKernelGGL(sampler_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (int*)counter);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_posemb(const int64_t* t, int B, int dim, const float* freqs, float* out, int64_t pitch,
                          void* stream) {
  LGM_REQUIRE(t && out && freqs && B > 0 && dim >= 4 && dim % 2 == 0 && pitch >= dim, "posemb: bad arguments");
  hipLaunchKernelGGL(posemb_kernel, dim3(lgm_cdiv((long)B * dim / 2, 256)), dim3(256), 0, (hipStream_t)stream, t, B,
                     dim, freqs, out, (long)pitch);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

static int check_mat(const void* a, int64_t pitch, int64_t rows, int cols, const char* who) {
  LGM_REQUIRE(a && rows > 0 && cols > 0 && cols % 4 == 0 && pitch % 4 == 0 && pitch >= cols && lgm_aligned16(a),
              "%s: matrix must be non-null, 16B aligned, cols/pitch multiples of 4", who);
  return LGM_OK;
}

extern "C" int lgm_act_fwd(const float* x, int64_t x_pitch, const float* bias, const float* res, int64_t res_pitch,
                           float* y, int64_t y_pitch, int64_t rows, int cols, int act, float slope, void* stream) {
  if (int rc = check_mat(x, x_pitch, rows, cols, "act_fwd(x)")) return rc;
  if (int rc = check_mat(y, y_pitch, rows, cols, "act_fwd(y)")) return rc;
  if (res) if (int rc = check_mat(res, res_pitch, rows, cols, "act_fwd(res)")) return rc;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(lgm_cdiv(rows * (cols / 4), 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (long)x_pitch, bias, res, (long)res_pitch, y, (long)y_pitch, (long)rows, cols, act, slope);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_act_bwd(const float* x, int64_t x_pitch, const float* bias, const float* gy, int64_t gy_pitch,
                           float* gx, int64_t gx_pitch, int accumulate, int64_t rows, int cols, int act, float slope,
                           void* stream) {
  if (int rc = check_mat(x, x_pitch, rows, cols, "act_bwd(x)")) return rc;
  if (int rc = check_mat(gy, gy_pitch, rows, cols, "act_bwd(gy)")) return rc;
  if (int rc = check_mat(gx, gx_pitch, rows, cols, "act_bwd(gx)")) return rc;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(lgm_cdiv(rows * (cols / 4), 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (long)x_pitch, bias, gy, (long)gy_pitch, gx, (long)gx_pitch, accumulate, (long)rows, cols, act, slope);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_axpby(const float* a, int64_t a_pitch, float alpha, const float* b, int64_t b_pitch, float beta,
                         float* y, int64_t y_pitch, int64_t rows, int cols, void* stream) {
  if (int rc = check_mat(a, a_pitch, rows, cols, "axpby(a)")) return rc;
  if (int rc = check_mat(y, y_pitch, rows, cols, "axpby(y)")) return rc;
  if (b) if (int rc = check_mat(b, b_pitch, rows, cols, "axpby(b)")) return rc;
  hipLaunchKernelGGL(axpby_kernel, dim3(lgm_cdiv(rows * (cols / 4), 256)), dim3(256), 0, (hipStream_t)stream, a,
                     (long)a_pitch, alpha, b, (long)b_pitch, beta, y, (long)y_pitch, (long)rows, cols);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_upsample2x_fwd(const float* x, int64_t x_pitch, float* y, int64_t y_pitch, int B, int H, int W,
                                  int C, void* stream) {
  if (int rc = check_mat(x, x_pitch, (long)B * H * W, C, "upsample2x_fwd(x)")) return rc;
  if (int rc = check_mat(y, y_pitch, (long)B * H * W * 4, C, "upsample2x_fwd(y)")) return rc;
  const long total = (long)B * 4 * H * W * (C / 4);
  hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(lgm_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (long)x_pitch, y, (long)y_pitch, B, H, W, C);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_upsample2x_bwd(const float* gy, int64_t gy_pitch, float* gx, int64_t gx_pitch, int B, int H, int W,
                                  int C, int accumulate, void* stream) {
  if (int rc = check_mat(gx, gx_pitch, (long)B * H * W, C, "upsample2x_bwd(gx)")) return rc;
  if (int rc = check_mat(gy, gy_pitch, (long)B * H * W * 4, C, "upsample2x_bwd(gy)")) return rc;
  const long total = (long)B * H * W * (C / 4);
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(lgm_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, gy,
                     (long)gy_pitch, gx, (long)gx_pitch, B, H, W, C, accumulate);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_pixel_unshuffle(const float* src, int64_t src_pitch, float* dst, int64_t dst_pitch, int B,
                                   int Hlo, int Wlo, int C, int inverse, int accumulate, void* stream) {
  LGM_REQUIRE(src && dst && B > 0 && Hlo > 0 && Wlo > 0 && C > 0, "pixel_unshuffle: bad arguments");
  LGM_REQUIRE((inverse ? src_pitch : dst_pitch) % 4 == 0, "pixel_unshuffle: 4C-side pitch %% 4 != 0");
  const long total = (long)B * Hlo * Wlo * C;
  hipLaunchKernelGGL(unshuffle_kernel, dim3(lgm_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (long)src_pitch, dst, (long)dst_pitch, B, Hlo, Wlo, C, inverse, accumulate);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_nchw_to_nhwc(const float* src, float* dst, int64_t dst_pitch, int B, int C, int HW, int Cpad,
                                void* stream) {
  LGM_REQUIRE(src && dst && B > 0 && C > 0 && HW > 0 && Cpad >= C && dst_pitch >= Cpad, "nchw_to_nhwc: bad arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(lgm_cdiv((long)B * HW * Cpad, 256)), dim3(256), 0, (hipStream_t)stream,
                     src, dst, (long)dst_pitch, B, C, HW, Cpad);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_nhwc_to_nchw(const float* src, int64_t src_pitch, float* dst, int B, int C, int HW, void* stream) {
  LGM_REQUIRE(src && dst && B > 0 && C > 0 && HW > 0 && src_pitch >= C, "nhwc_to_nchw: bad arguments");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(lgm_cdiv((long)B * C * HW, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (long)src_pitch, dst, B, C, HW);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_qsample_target(const float* img, const float* noise, const int64_t* t, const float* sqrt_ac,
                                  const float* sqrt_1mac, int normalize, float* xt, float* target, int64_t pitch,
                                  int B, int C, int HW, int Cpad, void* stream) {
  LGM_REQUIRE(img && noise && t && sqrt_ac && sqrt_1mac && xt && B > 0 && C > 0 && HW > 0 && Cpad >= C && pitch >= Cpad,
              "qsample_target: bad arguments");
  hipLaunchKernelGGL(qsample_kernel, dim3(lgm_cdiv((long)B * HW * Cpad, 256)), dim3(256), 0, (hipStream_t)stream, img,
                     noise, t, sqrt_ac, sqrt_1mac, normalize, xt, target, (long)pitch, B, C, HW, Cpad);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

namespace {
// x_hat = tanh(pre) and the per-sample reconstruction term mean_{chw} (x_hat - target)^2 in one pass (VQ-VAE decoder end,
// vqvae.py:85-88 + the recon loss :130): one block per sample, like mse_sample_kernel
__global__ __launch_bounds__(256) void tanh_mse_fwd_kernel(const float* __restrict__ pre, const float* __restrict__ target,
                                                           long pitch, int C, int HW, int Cpad, float* __restrict__ xh,
                                                           float* __restrict__ per_sample) {
  __shared__ float sh[16];
  const int b = blockIdx.x;
  const long n = (long)HW * Cpad;
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    const int c = (int)(i % Cpad);
    const long pix = (long)b * HW + i / Cpad;
    const float t = tanhf(pre[pix * pitch + c]);
    xh[pix * pitch + c] = t;
    if (c < C) {
      const float d = t - target[pix * pitch + c];
      s += d * d;
    }
  }
  s = lgm_block_sum(s, sh);
  if (threadIdx.x == 0) per_sample[b] = s / ((float)C * (float)HW);
}
// (one 16-byte load / store per pixel when Cpad == 4, as mse_sample4_kernel)
__global__ __launch_bounds__(256) void tanh_mse_fwd4_kernel(const float* __restrict__ pre, const float* __restrict__ target,
                                                            long pitch, int C, int HW, float* __restrict__ xh,
                                                            float* __restrict__ per_sample) {
  __shared__ float sh[16];
  const int b = blockIdx.x;
  float s = 0.f;
  for (int p0 = 0; p0 < HW; p0 += 1024) {
    f32x4 o[4], g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int px = p0 + threadIdx.x + 256 * u;
      const long off = ((long)b * HW + (px < HW ? px : HW - 1)) * pitch;
      o[u] = *reinterpret_cast<const f32x4*>(pre + off);
      g[u] = *reinterpret_cast<const f32x4*>(target + off);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int px = p0 + threadIdx.x + 256 * u;
      if (px < HW) {
        f32x4 tv;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          tv[c] = tanhf(o[u][c]);
          const float d = tv[c] - g[u][c];
          s += c < C ? d * d : 0.f;
        }
        *reinterpret_cast<f32x4*>(xh + ((long)b * HW + px) * pitch) = tv;
      }
    }
  }
  s = lgm_block_sum(s, sh);
  if (threadIdx.x == 0) per_sample[b] = s / ((float)C * (float)HW);
}
// its backward, with the loss weights folded in: gpre = (gloss w_recon) 2 (x_hat - target) / (C HW B) (1 - x_hat^2);
// g2[0] = gloss w_recon, g2[1] = gloss w_vq for the quantiser's backward later in the stream
__global__ __launch_bounds__(256) void tanh_mse_bwd_kernel(const float* __restrict__ xh, const float* __restrict__ target,
                                                           long pitch, const float* __restrict__ gloss, float w_recon,
                                                           float w_vq, int B, int C, int HW, int Cpad,
                                                           float* __restrict__ gpre, float* __restrict__ g2) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * HW * Cpad;
  const float gr = gloss[0] * w_recon;
  if (i == 0) {
    g2[0] = gr;
    g2[1] = gloss[0] * w_vq;
  }
  if (i >= total) return;
  const int c = (int)(i % Cpad);
  const long pix = i / Cpad;
  float g = 0.f;
  if (c < C) {
    const float t = xh[pix * pitch + c];
    const float scale = gr * 2.f * 1.f / ((float)C * (float)HW * (float)B);
    g = scale * (t - target[pix * pitch + c]);
    g = g * (1.f - t * t);
  }
  gpre[pix * pitch + c] = g;
}
}  // namespace

extern "C" int lgm_tanh_mse_fwd(const float* pre, const float* target, int64_t pitch, int B, int C, int HW, int Cpad,
                                float* xh, float* per_sample, void* stream) {
  LGM_REQUIRE(pre && target && xh && per_sample && B > 0, "tanh_mse_fwd: bad arguments");
  if (Cpad == 4 && pitch % 4 == 0 && lgm_aligned16(pre) && lgm_aligned16(target) && lgm_aligned16(xh))
    hipLaunchKernelGGL(tanh_mse_fwd4_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pre, target, (long)pitch, C, HW, xh,
                       per_sample);
  else
    hipLaunchKernelGGL(tanh_mse_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pre, target, (long)pitch, C, HW, Cpad,
                       xh, per_sample);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_tanh_mse_bwd(const float* xh, const float* target, int64_t pitch, const float* gloss, float w_recon,
                                float w_vq, int B, int C, int HW, int Cpad, float* gpre, float* g2, void* stream) {
  LGM_REQUIRE(xh && target && gloss && gpre && g2 && B > 0, "tanh_mse_bwd: bad arguments");
  hipLaunchKernelGGL(tanh_mse_bwd_kernel, dim3(lgm_cdiv((long)B * HW * Cpad, 256)), dim3(256), 0, (hipStream_t)stream, xh,
                     target, (long)pitch, gloss, w_recon, w_vq, B, C, HW, Cpad, gpre, g2);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_weighted_mse_fwd(const float* out, const float* target, int64_t pitch, const int64_t* t,
                                    const float* loss_weight, int B, int C, int HW, int Cpad, float* per_sample,
                                    float* loss, void* stream) {
  LGM_REQUIRE(out && target && per_sample && B > 0 && (!loss_weight || t), "weighted_mse_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (Cpad == 4 && pitch % 4 == 0 && lgm_aligned16(out) && lgm_aligned16(target))
    hipLaunchKernelGGL(mse_sample4_kernel, dim3(B), dim3(256), 0, s, out, target, (long)pitch, t, loss_weight, C, HW, per_sample);
  else
    hipLaunchKernelGGL(mse_sample_kernel, dim3(B), dim3(256), 0, s, out, target, (long)pitch, t, loss_weight, C, HW, Cpad,
                       per_sample);
  if (loss) hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, (const float*)per_sample, B, loss);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_weighted_mse_bwd(const float* out, const float* target, int64_t pitch, const int64_t* t,
                                    const float* loss_weight, const float* gloss, int B, int C, int HW, int Cpad,
                                    float* gout, void* stream) {
  LGM_REQUIRE(out && target && gloss && gout && B > 0 && (!loss_weight || t), "weighted_mse_bwd: bad arguments");
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(lgm_cdiv((long)B * HW * Cpad, 256)), dim3(256), 0, (hipStream_t)stream, out,
                     target, (long)pitch, t, loss_weight, gloss, B, C, HW, Cpad, gout);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_sample_step(const float* x, const float* v, const float* noise, float* out, float* x0_out, int B,
                               int C, int HW, int Cpad, float A, float Bv, int clip, float R, float Rm1, float C0,
                               float C1, float C2, float C3, void* stream) {
  LGM_REQUIRE(x && v && out && B > 0 && C > 0 && HW > 0 && Cpad >= C, "sample_step: bad arguments");
  hipLaunchKernelGGL(sample_step_kernel, dim3(lgm_cdiv((long)B * HW * Cpad, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     v, noise, out, x0_out, B, C, HW, Cpad, A, Bv, clip, R, Rm1, C0, C1, C2, C3);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

// ---------------------------------------------------------------------------------------
// WGAN-GP helpers (wgan.py:84-156)
// ---------------------------------------------------------------------------------------
namespace {

// interpolates: out[b] = alpha[b]*x[b] + (1-alpha[b])*y[b]   (dense rows of `rowlen` floats)
__global__ __launch_bounds__(256) void lerp_rows_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ alpha, float* __restrict__ out,
                                                        long B, long rowlen) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * rowlen) return;
  const float a = alpha[i / rowlen];
  out[i] = a * x[i] + (1.f - a) * y[i];
}

// gradient penalty with the reference's CHANNEL-ONLY norm (wgan.py:153-154):
//   r[p] = sqrt(sum_c g[p,c]^2);  partial[block] = sum_p (r-1)^2;  gbar[p,c] = coef*(r-1)/r * g[p,c],
//   coef = gscale * lambda * 2 / npix.  One thread per pixel, C <= 4 (padded to 4 floats).
__global__ __launch_bounds__(256) void gp_penalty_kernel(const float* __restrict__ g, long npix, int C, float lambda,
                                                         const float* __restrict__ gscale, float* __restrict__ partial,
                                                         float* __restrict__ gbar) {
  __shared__ float sh[16];
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float pen = 0.f;
  if (p < npix) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + p * 4);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < C) s += v[c] * v[c];
    const float r = sqrtf(s);
    pen = (r - 1.f) * (r - 1.f);
    if (gbar) {
      // r == 0 (a pixel whose channel gradient is exactly zero: dead / clipped critic): torch's backward of
      // norm(2, dim=1) uses the zero subgradient there (wgan.py:153-154), not (r-1)/r = -inf
      const float k = r > 0.f ? gscale[0] * lambda * 2.f / (float)npix * (r - 1.f) / r : 0.f;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < C) o[c] = k * v[c];
      *reinterpret_cast<f32x4*>(gbar + p * 4) = o;
    }
  }
  pen = lgm_block_sum(pen, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = pen;
}

// R1 penalty (r1gan.py:77): partial[block] = sum_p sum_c g[p,c]^2;  gbar = gscale / B * g.
__global__ __launch_bounds__(256) void r1_penalty_kernel(const float* __restrict__ g, long npix, float inv_b,
                                                         const float* __restrict__ gscale, float* __restrict__ partial,
                                                         float* __restrict__ gbar) {
  __shared__ float sh[16];
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float pen = 0.f;
  if (p < npix) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + p * 4);
    pen = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);   // the padding lane carries zeros
    if (gbar) *reinterpret_cast<f32x4*>(gbar + p * 4) = v * (gscale[0] * inv_b);
  }
  pen = lgm_block_sum(pen, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = pen;
}

// vals[slot] = scale * sum_i partial[i]   (single block, fixed order)
__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ partial, long n, long stride,
                                                        float scale, float* __restrict__ out) {
  __shared__ float sh[16];
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += blockDim.x) s += partial[i * stride];
  s = lgm_block_sum(s, sh);
  if (threadIdx.x == 0) out[0] = s * scale;
}

// out[r][c] = (c == col) ? scale * (vptr ? vptr[0] : 1) : 0
__global__ void fill_col_kernel(float* __restrict__ out, long pitch, long n, int ncols, int col, float scale,
                                const float* __restrict__ vptr) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * ncols) return;
  const long r = i / ncols;
  const int c = (int)(i % ncols);
  out[r * pitch + c] = (c == col) ? scale * (vptr ? vptr[0] : 1.f) : 0.f;
}

// vals = (real, fake, gp, d_loss): d_loss = fake - real + gp
__global__ void wgan_dloss_kernel(float* __restrict__ vals) { vals[3] = vals[1] - vals[0] + vals[2]; }


// Device-side input pipeline of the reference DataModule (data/datamodule.py:41-53): ToTensor (u8 -> [0,1]),
// Normalize(0.5, 0.5), CenterCropMinXY (data/utils.py:7-35), Resize(S, bilinear, antialias=True),
// RandomHorizontalFlip (flags drawn by the caller).  The antialiased resize is the separable triangle
// filter of torch's _upsample_bilinear2d_aa: support = max(scale, 1), window [int(c - support + .5),
// int(c + support + .5)), weights max(0, 1 - |(j + .5 - c) / max(scale, 1)|) normalised to 1.
// One thread per output pixel, all (<= 4) channels; u8 HWC in, fp32 NCHW out.
__global__ __launch_bounds__(256) void image_transform_kernel(const unsigned char* __restrict__ src, int H, int W, int C,
                                                              const unsigned char* __restrict__ flip,
                                                              float* __restrict__ dst, int S, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ox = (int)(i % S), oy = (int)((i / S) % S);
  const long b = i / ((long)S * S);
  const int D = H < W ? H : W, top = (H - D) / 2, left = (W - D) / 2;
  const float scale = (float)D / (float)S;
  const float support = scale >= 1.f ? scale : 1.f;
  const float inv = 1.f / support;
  const int sx = (flip && flip[b]) ? S - 1 - ox : ox;     // flip after the resize == mirrored source column
  const float cy = scale * (oy + 0.5f), cx = scale * (sx + 0.5f);
  const int ymin = max(0, (int)(cy - support + 0.5f)), ymax = min(D, (int)(cy + support + 0.5f));
  const int xmin = max(0, (int)(cx - support + 0.5f)), xmax = min(D, (int)(cx + support + 0.5f));
  float wys = 0.f, wxs = 0.f;
  for (int y = ymin; y < ymax; ++y) wys += fmaxf(0.f, 1.f - fabsf((y - cy + 0.5f) * inv));
  for (int x = xmin; x < xmax; ++x) wxs += fmaxf(0.f, 1.f - fabsf((x - cx + 0.5f) * inv));
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const unsigned char* img = src + b * (long)H * W * C;
  for (int y = ymin; y < ymax; ++y) {
    const float wy = fmaxf(0.f, 1.f - fabsf((y - cy + 0.5f) * inv)) / wys;
    float row[4] = {0.f, 0.f, 0.f, 0.f};
    for (int x = xmin; x < xmax; ++x) {
      const float wx = fmaxf(0.f, 1.f - fabsf((x - cx + 0.5f) * inv)) / wxs;
      const unsigned char* px = img + ((long)(top + y) * W + left + x) * C;
      for (int c = 0; c < C; ++c) row[c] += wx * (float)px[c];
    }
    for (int c = 0; c < C; ++c) acc[c] += wy * row[c];
  }
  for (int c = 0; c < C; ++c)
    dst[((b * C + c) * S + oy) * (long)S + ox] = acc[c] * (2.f / 255.f) - 1.f;   // /255, (x - .5) / .5
}

}  // namespace

extern "C" int lgm_image_transform(const unsigned char* src, int64_t B, int H, int W, int C, const unsigned char* flip,
                                   float* dst, int S, void* stream) {
  LGM_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C >= 1 && C <= 4 && S > 0, "image_transform: bad arguments");
  const long total = (long)B * S * S;
  hipLaunchKernelGGL(image_transform_kernel, dim3(lgm_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src, H, W, C,
                     flip, dst, S, total);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_lerp_rows(const float* x, const float* y, const float* alpha, float* out, int64_t B,
                             int64_t rowlen, void* stream) {
  LGM_REQUIRE(x && y && alpha && out && B > 0 && rowlen > 0, "lerp_rows: bad arguments");
  hipLaunchKernelGGL(lerp_rows_kernel, dim3(lgm_cdiv(B * rowlen, 256)), dim3(256), 0, (hipStream_t)stream, x, y, alpha,
                     out, (long)B, (long)rowlen);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int64_t lgm_gp_penalty_workspace(int64_t npix) { return (int64_t)lgm_cdiv(npix, 256) * 4 + 16; }

extern "C" int lgm_gp_penalty(const float* g, int64_t npix, int C, float lambda, const float* gscale, float* loss_out,
                              float* gbar, void* workspace, void* stream) {
  LGM_REQUIRE(g && loss_out && workspace && npix > 0 && C >= 1 && C <= 4 && (!gbar || gscale) && lgm_aligned16(g),
              "gp_penalty: bad arguments (dense NHWC4 gradient expected)");
  hipStream_t s = (hipStream_t)stream;
  const int nb = lgm_cdiv(npix, 256);
  hipLaunchKernelGGL(gp_penalty_kernel, dim3(nb), dim3(256), 0, s, g, (long)npix, C, lambda, gscale, (float*)workspace,
                     gbar);
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, (long)nb, 1L,
                     lambda / (float)npix, loss_out);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_r1_penalty(const float* g, int64_t npix, int64_t B, const float* gscale, float* loss_out,
                              float* gbar, void* workspace, void* stream) {
  LGM_REQUIRE(g && loss_out && workspace && npix > 0 && B > 0 && (!gbar || gscale) && lgm_aligned16(g),
              "r1_penalty: bad arguments (dense NHWC4 gradient expected)");
  hipStream_t s = (hipStream_t)stream;
  const int nb = lgm_cdiv(npix, 256);
  hipLaunchKernelGGL(r1_penalty_kernel, dim3(nb), dim3(256), 0, s, g, (long)npix, 1.f / (float)B, gscale,
                     (float*)workspace, gbar);
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, (long)nb, 1L, 0.5f / (float)B,
                     loss_out);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_mean_col(const float* v, int64_t pitch, int64_t n, float scale, float* out, void* stream) {
  LGM_REQUIRE(v && out && n > 0, "mean_col: bad arguments");
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, v, (long)n, (long)pitch,
                     scale / (float)n, out);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_fill_col(float* out, int64_t pitch, int64_t n, int ncols, int col, float scale, const float* vptr,
                            void* stream) {
  LGM_REQUIRE(out && n > 0 && ncols > 0 && col < ncols && pitch >= ncols, "fill_col: bad arguments");
  hipLaunchKernelGGL(fill_col_kernel, dim3(lgm_cdiv(n * ncols, 256)), dim3(256), 0, (hipStream_t)stream, out,
                     (long)pitch, (long)n, ncols, col, scale, vptr);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

namespace {
__global__ void vqvae_loss_kernel(const float* recon, const float* out3, float w_recon, float w_vq, float* vals4) {
  const float r = recon[0], v = out3[0];
  vals4[0] = r * w_recon + v * w_vq;
  vals4[1] = r;
  vals4[2] = v;
  vals4[3] = out3[1];
}
// the same with the reconstruction term still as per-sample values: their mean (mean_kernel's arithmetic) is taken here
__global__ __launch_bounds__(256) void vqvae_loss_mean_kernel(const float* per_sample, int n, const float* out3, float w_recon,
                                                              float w_vq, float* vals4) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += per_sample[i];
  s = lgm_block_sum(s, sh);
  if (threadIdx.x == 0) {
    const float r = s / (float)n, v = out3[0];
    vals4[0] = r * w_recon + v * w_vq;
    vals4[1] = r;
    vals4[2] = v;
    vals4[3] = out3[1];
  }
}
__global__ void scale_pair_kernel(const float* g, float w0, float w1, float* out2) {
  out2[0] = g[0] * w0;
  out2[1] = g[0] * w1;
}
}  // namespace

extern "C" int lgm_vqvae_loss(const float* recon, const float* out3, float w_recon, float w_vq, float* vals4,
                              void* stream) {
  LGM_REQUIRE(recon && out3 && vals4, "vqvae_loss: null pointer");
  hipLaunchKernelGGL(vqvae_loss_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, recon, out3, w_recon, w_vq, vals4);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_vqvae_loss_samples(const float* per_sample, int n, const float* out3, float w_recon, float w_vq,
                                      float* vals4, void* stream) {
  LGM_REQUIRE(per_sample && out3 && vals4 && n > 0, "vqvae_loss_samples: bad arguments");
  hipLaunchKernelGGL(vqvae_loss_mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, per_sample, n, out3, w_recon, w_vq,
                     vals4);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_scale_pair(const float* g, float w0, float w1, float* out2, void* stream) {
  LGM_REQUIRE(g && out2, "scale_pair: null pointer");
  hipLaunchKernelGGL(scale_pair_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, g, w0, w1, out2);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}

extern "C" int lgm_wgan_dloss(float* vals4, void* stream) {
  LGM_REQUIRE(vals4, "wgan_dloss: null pointer");
  hipLaunchKernelGGL(wgan_dloss_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, vals4);
  LGM_LAUNCH_CHECK();
  return LGM_OK;
}
