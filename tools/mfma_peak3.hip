// Which ingredient of the conv3x3 inner loop costs MFMA issue rate?  One wave per SIMD.
//  bit0: A operands come from LDS (2 ds_read_b128 per 8 MFMAs, one group ahead)
//  bit1: B operands come from global memory (4 x 16 B per lane per 32 MFMAs, one step ahead)
//  bit2: one ds_write_b128 per 32 MFMAs
//  bit3: a workgroup barrier every 288 MFMAs
//  bit4: one 16 B/lane HBM fetch per step (unique addresses, conv-style address arithmetic)
//  bit5: masked commit (4 v_cndmask) of the fetched value before the ds_write
//  bit6: an epilogue (LDS transpose + 8 x 16 B stores per lane) every second phase
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(256, 1) void burn(float* out, const float* w, int phases, const float* act, float* dst, int H, int W, long pitch) {
  __shared__ __align__(16) float lds[2 * 288 * 36 + 4 * 1152];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 2 * 288 * 36; i += 256) lds[i] = 1e-3f * (i & 15);
  __syncthreads();
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const float* wl = w + (long)lr * 576 + lh * 4;
  f32x4 cb[4], nb[4];
  for (int kc = 0; kc < 4; ++kc) cb[kc] = *reinterpret_cast<const f32x4*>(wl + kc * 8);
  const float* a0 = lds + lr * 36 + lh * 4;
  const float* a1 = lds + (lr + 34) * 36 + lh * 4;
  f32x4 fa[2][2];
  fa[0][0] = *reinterpret_cast<const f32x4*>(a0);
  fa[0][1] = *reinterpret_cast<const f32x4*>(a1);
  fa[1][0] = fa[0][0]; fa[1][1] = fa[0][1];
  const long long mt0 = __builtin_amdgcn_s_memtime(), rt0 = wall_clock64();
  f32x4 rp[9];
  int ppos[9];
  for (int j = 0; j < 9; ++j) {
    rp[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pos = tid / 8 + 32 * j;
    ppos[j] = pos < 204 ? ((pos / 34) << 10) | (pos % 34) : -1;
  }
  const int c4 = (tid % 8) * 4;
  unsigned mrp = 0x155;
  float* Ts = lds + 288 * 36 + (tid >> 6) * 1152;
  for (int ph = 0; ph < phases; ++ph) {
    const int tile = (blockIdx.x * 61 + ph * 7) % 1024;       // some (image, row-block) far apart in memory
    const int b0 = tile >> 3, h0 = (tile & 7) * 4;
    unsigned mnew = 0;
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      if (V & 32) {
        *reinterpret_cast<f32x4*>(lds + 288 * 36 + (tid / 8 + 32 * u) * 36 + c4) =
            ((mrp >> u) & 1u) ? rp[u] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (V & 16) {
        const int pk = ppos[u];
        const int py = pk >> 10, px = pk & 1023;
        const int ih = h0 + py - 1, iw = px - 1;
        const bool ok = pk >= 0 && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        const unsigned off = ok ? (unsigned)((b0 * H + ih) * W + iw) * (unsigned)pitch + (unsigned)c4 : 0u;
        rp[u] = *reinterpret_cast<const f32x4*>(act + (ph & 1) * 32 + off);
        mnew |= (ok ? 1u : 0u) << u;
      }
      if (V & 2) {
        const float* src = wl + ((u + 1) % 9) * 32 + (ph & 1) * 288;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) nb[kc] = *reinterpret_cast<const f32x4*>(src + kc * 8);
      }
      if (V & 4) *reinterpret_cast<f32x4*>(lds + 288 * 36 + (tid / 8 + 32 * u) * 36 + (tid % 8) * 4) = cb[0];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const int g = u * 4 + kc;
        if (V & 1) {
          const int off = ((g + 1) % 36) / 4 * 36 * 2 + ((g + 1) % 4) * 8;
          fa[(g + 1) & 1][0] = *reinterpret_cast<const f32x4*>(a0 + off);
          fa[(g + 1) & 1][1] = *reinterpret_cast<const f32x4*>(a1 + off);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][0][s], cb[kc][s], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][1][s], cb[kc][s], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (V & 2) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) cb[kc] = nb[kc];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    mrp = (V & 16) ? mnew : mrp;
    if (V & 8) __syncthreads();
    if ((V & 64) && (ph & 1)) {
      const int nc = (lane & 7) * 4 + ((tid >> 6) & 1) * 32;
      for (int i = 0; i < 2; ++i) {
        f32x16& a = i ? acc1 : acc0;
        for (int r = 0; r < 16; ++r) Ts[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + lr] = a[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int j = 0; j < 4; ++j) {
          const long m = (long)((b0 * H + h0 + (tid >> 7) * 2 + i) * W + (lane >> 3) + 8 * j);
          *reinterpret_cast<f32x4*>(dst + m * 64 + nc) = *reinterpret_cast<const f32x4*>(Ts + ((lane >> 3) + 8 * j) * 36 + (lane & 7) * 4);
        }
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
      }
    }
  }
  if (rp[3][0] == 3.25f && mrp == 77) out[1] = 1.f;
  if (blockIdx.x == 0 && tid == 0 && phases > 100) {
    const long long mt1 = __builtin_amdgcn_s_memtime(), rt1 = wall_clock64();
    out[2] = (float)((double)(mt1 - mt0) / (double)(rt1 - rt0) * 0.1);   // memtime ticks per ns
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  if (s == 123.456f) out[0] = s;
}

template <int V>
void run(const char* name) {
  float *d, *w, *act, *dst;
  (void)hipMalloc(&d, 16);
  const int H = 32, W = 32; const long pitch = 64;
  (void)hipMalloc(&act, 128L * H * W * pitch * 4);
  (void)hipMemset(act, 0, 128L * H * W * pitch * 4);
  (void)hipMalloc(&dst, 128L * H * W * 64 * 4);
  (void)hipMalloc(&w, 64 * 576 * 4 * 2);
  (void)hipMemset(w, 0, 64 * 576 * 4 * 2);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int phases = 2000;
  burn<V><<<256, 256>>>(d, w, 10, act, dst, H, W, pitch);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  burn<V><<<256, 256>>>(d, w, phases, act, dst, H, W, pitch);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flop = 256.0 * 4 * phases * 288.0 * 4096.0;
  float hres[4] = {0, 0, 0, 0};
  (void)hipMemcpy(hres, d, 16, hipMemcpyDeviceToHost);
  printf("V=%3d %-44s %.3f ms  %.1f TFLOP/s   s_memtime %.3f ticks/ns\n", V, name, ms, flop / (ms * 1e-3) / 1e12, hres[2]);
  (void)hipFree(d);
  (void)hipFree(w);
  (void)hipFree(act);
  (void)hipFree(dst);
}

int main() {
  run<0>("registers only");
  run<1>("A from LDS");
  run<2>("B from global");
  run<3>("A from LDS + B from global");
  run<4>("ds_write per step");
  run<8>("barrier per phase");
  run<15>("A LDS + B global + ds_write + barrier");
  run<31>("... + HBM fetch per step");
  run<63>("... + masked commit");
  run<127>("... + epilogue every 2nd phase");
  run<79>("15 + epilogue only");
  run<47>("15 + masked commit only");
  return 0;
}
