// Diagnostic: what fp32-MFMA rate and clock does this chip hold (a) on a register-only loop and (b) with an
// LDS B-operand read + global A-operand load per 16 MFMAs like the ensemble kernel?  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const f32x4 *w, float *out, unsigned long long *stamps, int iters) {
  __shared__ f32x4 lds[2048];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = f32x4{0.001f * i, 0.5f, -0.25f, 0.125f};
  __syncthreads();
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  f32x4 a[4], b;
  for (int t = 0; t < 4; ++t) a[t] = w[(t * 64 + lane)];
  b = lds[lane];
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int g = 0; g < iters; ++g) {
    f32x4 an[4], bn;
    if (MODE == 1) {
      for (int t = 0; t < 4; ++t) an[t] = w[((size_t)((g * 4 + t) & 1023) * 64 + lane)];
      bn = lds[(g * 64 + lane) & 2047];
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], b[s], acc[t], 0, 0, 0);
    if (MODE == 1) {
      __builtin_amdgcn_sched_barrier(0);
      for (int t = 0; t < 4; ++t) a[t] = an[t];
      b = bn;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  const int blocks = 512 * 8, iters = 2000;
  f32x4 *w; float *out; unsigned long long *st;
  hipMalloc(&w, 1024 * 64 * sizeof(f32x4)); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&st, blocks * 16);
  float *hw = (float *)malloc(1024 * 64 * 16);
  for (int i = 0; i < 1024 * 64 * 4; ++i) hw[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(w, hw, 1024 * 64 * 16, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, w, out, st, iters);
      else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, w, out, st, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2 * 64];
      hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
      double flop = (double)blocks * 4 * iters * 16 * 4096.0;
      double clk = 0; for (int i = 0; i < 64; ++i) clk += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0; clk /= 64;
      printf("mode %d rep %d: %.3f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz  cycles/wg %llu\n", mode, rep, ms,
             flop / ms / 1e9, clk, h[0]);
    }
  }
  return 0;
}
