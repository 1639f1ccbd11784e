// Diagnostic for DESIGN §8 (0): can the fp32 GEMMs of the ensemble forward run on the bf16 matrix cores?
//   a = a1 + a2 + a3, a_i = successive bf16 roundings of the remainder (3 x 8 = 24 mantissa bits);
//   a.b ~= sum_{i+j<=4} a_i b_j : six v_mfma_f32_32x32x16_bf16 with fp32 accumulation, every partial product exact.
// (1) accuracy: C[32x32] = A[32xK] B[Kx32], K = 512, against float64 -- fp32 MFMA, the six-term split and, for
//     scale, a three-term (i+j<=3) and a plain one-term bf16 product;
// (2) rate: register-only loops of {8 x mfma_f32_32x32x2f32} vs {6 x mfma_f32_32x32x16_bf16} per K = 16 slab and
//     accumulator, one wave per SIMD on every CU, same number of slabs.
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/split_bf16_probe.hip -o /tmp/split_probe && /tmp/split_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float a, __bf16 &p1, __bf16 &p2, __bf16 &p3) {
  p1 = (__bf16)a;
  const float r1 = a - (float)p1;     // exact
  p2 = (__bf16)r1;
  const float r2 = r1 - (float)p2;    // exact
  p3 = (__bf16)r2;
}

// One wave computes C = A B for A [32][K] row-major, B [K][32] row-major, four ways.  Lane l: r = l & 31, h = l >> 5.
// fp32 32x32x2: A operand = A[r][2s + h], B operand = B[2s + h][r].
// bf16 32x32x16: element j of the fragment = A[r][16s + 8h + j] / B[16s + 8h + j][r].
__global__ void accuracy_kernel(const float *A, const float *B, int K, float *C32, float *C6, float *C3, float *C1) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc32, acc6, acc3, acc1;
  for (int i = 0; i < 16; ++i) acc32[i] = acc6[i] = acc3[i] = acc1[i] = 0.0f;
  for (int s = 0; s < K / 2; ++s)
    acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + 2 * s + h], B[(2 * s + h) * 32 + r], acc32, 0, 0, 0);
  for (int s = 0; s < K / 16; ++s) {
    bf16x8 a1, a2, a3, b1, b2, b3;
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * s + 8 * h + j;
      __bf16 p1, p2, p3;
      split3(A[r * K + k], p1, p2, p3);
      a1[j] = p1; a2[j] = p2; a3[j] = p3;
      split3(B[k * 32 + r], p1, p2, p3);
      b1[j] = p1; b2[j] = p2; b3[j] = p3;
    }
    // smallest terms first
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc6, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc3, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc3, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc3, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc1, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    C32[row * 32 + r] = acc32[i]; C6[row * 32 + r] = acc6[i]; C3[row * 32 + r] = acc3[i]; C1[row * 32 + r] = acc1[i];
  }
}

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void rate_kernel(float *out, int slabs) {
  const int l = threadIdx.x & 63;
  f32x16 acc[NACC];
  for (int t = 0; t < NACC; ++t)
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
  float af[8], bf[8];
  bf16x8 a1, a2, a3, b1, b2, b3;
  for (int j = 0; j < 8; ++j) {
    af[j] = 0.01f * (float)(l + j) - 0.3f; bf[j] = 0.02f * (float)(l - j) + 0.1f;
    __bf16 p1, p2, p3;
    split3(af[j], p1, p2, p3); a1[j] = p1; a2[j] = p2; a3[j] = p3;
    split3(bf[j], p1, p2, p3); b1[j] = p1; b2[j] = p2; b3[j] = p3;
  }
  for (int g = 0; g < slabs; ++g) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      if (MODE == 0) {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc[t], 0, 0, 0);
      } else {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[t], 0, 0, 0);
      }
    }
  }
  float s = 0.0f;
  for (int t = 0; t < NACC; ++t)
    for (int i = 0; i < 16; ++i) s += acc[t][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// (3) what the slab loop of csrc/ens_split.hip pays next to its 48 MFMAs (4 n-tiles x 2 row halves x 6 terms), one wave
// per SIMD: VAR 0 = the MFMAs alone; 1 = + the 12 weight-fragment loads of the next slab (L2-resident buffer, ping-pong
// registers); 2 = + the split of 16 fresh float32 values into 3 x 2 bf16 fragments; 3 = both.
template <int VAR>
__global__ __launch_bounds__(256) void slab_kernel(const bf16x8 *w, float *out, int slabs) {
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[4][2];
  for (int t = 0; t < 4; ++t)
    for (int b = 0; b < 2; ++b)
      for (int i = 0; i < 16; ++i) acc[t][b][i] = 0.0f;
  bf16x8 A[2][4][3], BX[2][3], BY[2][3];
  float raw[16];
  for (int j = 0; j < 16; ++j) raw[j] = 0.01f * (float)(l + j) - 0.3f;
  const bf16x8 *wa = w + wave * 12 * 64 + l;
  for (int t = 0; t < 4; ++t)
    for (int i = 0; i < 3; ++i) { A[0][t][i] = wa[(t * 3 + i) * 64]; A[1][t][i] = A[0][t][i]; }
  for (int i = 0; i < 3; ++i) { BX[0][i] = A[0][0][i]; BY[0][i] = A[0][1][i]; BX[1][i] = A[0][2][i]; BY[1][i] = A[0][3][i]; }
  auto step = [&](int s, int cur) {
    const int nxt = cur ^ 1;
    if (VAR & 1) {
      const bf16x8 *q = wa + (size_t)((s + 1) & 63) * 48 * 64;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 3; ++i) A[nxt][t][i] = q[(t * 3 + i) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (VAR & 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __bf16 p1, p2, p3;
        split3(raw[j], p1, p2, p3); BX[nxt][0][j] = p1; BX[nxt][1][j] = p2; BX[nxt][2][j] = p3;
        split3(raw[8 + j], p1, p2, p3); BY[nxt][0][j] = p1; BY[nxt][1][j] = p2; BY[nxt][2][j] = p3;
        raw[j] += 0.125f; raw[8 + j] -= 0.0625f;     // fresh values every slab
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bf16x8 *a = A[cur][t];
      const bf16x8 *bx = (VAR & 2) ? BX[cur] : BX[0], *by = (VAR & 2) ? BY[cur] : BY[0];
      acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bx[0], acc[t][0], 0, 0, 0);
      acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bx[2], acc[t][0], 0, 0, 0);
      acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bx[1], acc[t][0], 0, 0, 0);
      acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bx[0], acc[t][0], 0, 0, 0);
      acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bx[1], acc[t][0], 0, 0, 0);
      acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bx[0], acc[t][0], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], by[0], acc[t][1], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], by[2], acc[t][1], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], by[1], acc[t][1], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], by[0], acc[t][1], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], by[1], acc[t][1], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], by[0], acc[t][1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int s = 0; s < slabs; s += 2) { step(s, 0); step(s + 1, 1); }
  float r = 0.0f;
  for (int t = 0; t < 4; ++t)
    for (int b = 0; b < 2; ++b)
      for (int i = 0; i < 16; ++i) r += acc[t][b][i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

static double max_rel(const float *c, const double *ref, double scale) {
  double m = 0.0;
  for (int i = 0; i < 1024; ++i) m = fmax(m, fabs((double)c[i] - ref[i]) / scale);
  return m;
}

int main() {
  const int K = 512;
  float *hA = (float *)malloc(32 * K * 4), *hB = (float *)malloc(K * 32 * 4);
  srand(7);
  auto rnd = []() {   // ~N(0,1) with a wide dynamic range of magnitudes
    double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
    return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
  };
  for (int i = 0; i < 32 * K; ++i) hA[i] = (float)(rnd() * exp(2.0 * rnd()));
  for (int i = 0; i < K * 32; ++i) hB[i] = (float)(rnd() * 0.05);
  double ref[1024], mag[1024], scale = 0.0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double s = 0.0, m = 0.0;
      for (int k = 0; k < K; ++k) { s += (double)hA[i * K + k] * hB[k * 32 + j]; m += fabs((double)hA[i * K + k] * hB[k * 32 + j]); }
      ref[i * 32 + j] = s; mag[i * 32 + j] = m; scale = fmax(scale, m);
    }
  float *dA, *dB, *dC;
  hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, K * 32 * 4); hipMalloc(&dC, 4 * 1024 * 4);
  hipMemcpy(dA, hA, 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, K * 32 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(accuracy_kernel, dim3(1), dim3(64), 0, 0, dA, dB, K, dC, dC + 1024, dC + 2048, dC + 3072);
  float hC[4096];
  hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
  // error relative to sum_k |a_k b_k| (the quantity a float32 dot product's error bound scales with)
  double e[4] = {0, 0, 0, 0};
  for (int v = 0; v < 4; ++v)
    for (int i = 0; i < 1024; ++i) e[v] = fmax(e[v], fabs((double)hC[v * 1024 + i] - ref[i]) / mag[i]);
  // numpy-style float32 accumulation in k order, for scale
  double e_seq = 0.0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      float s = 0.0f;
      for (int k = 0; k < K; ++k) s += hA[i * K + k] * hB[k * 32 + j];
      e_seq = fmax(e_seq, fabs((double)s - ref[i * 32 + j]) / mag[i * 32 + j]);
    }
  printf("{\"K\": %d, \"max_err_over_sum_abs\": {\"fp32_mfma\": %.3e, \"split_bf16_6\": %.3e, \"split_bf16_3\": %.3e, "
         "\"bf16_1\": %.3e, \"fp32_sequential_host\": %.3e}", K, e[0], e[1], e[2], e[3], e_seq);
  (void)max_rel; (void)scale;

  // ---- rate ------------------------------------------------------------------------------------------------
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount, slabs = 20000;
  float *out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  float ms[2][2];
  for (int mode = 0; mode < 2; ++mode)
    for (int na = 0; na < 2; ++na) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (mode == 0 && na == 0) hipLaunchKernelGGL((rate_kernel<0, 1>), dim3(blocks), dim3(256), 0, 0, out, slabs);
        if (mode == 0 && na == 1) hipLaunchKernelGGL((rate_kernel<0, 4>), dim3(blocks), dim3(256), 0, 0, out, slabs / 4);
        if (mode == 1 && na == 0) hipLaunchKernelGGL((rate_kernel<1, 1>), dim3(blocks), dim3(256), 0, 0, out, slabs);
        if (mode == 1 && na == 1) hipLaunchKernelGGL((rate_kernel<1, 4>), dim3(blocks), dim3(256), 0, 0, out, slabs / 4);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1);
        if (t < best) best = t;
      }
      ms[mode][na] = best;
    }
  // useful flops: each (slab, accumulator) is a 32 x 32 x 16 product = 32768 flop; 4 waves per workgroup
  const double flop = (double)blocks * 4 * slabs * 32768.0;
  printf(", \"cus\": %d, \"fp32_equivalent_tflops\": {\"fp32_mfma_1acc\": %.1f, \"fp32_mfma_4acc\": %.1f, "
         "\"split_bf16_6_1acc\": %.1f, \"split_bf16_6_4acc\": %.1f}, \"speedup_4acc\": %.2f}\n",
         blocks, flop / ms[0][0] * 1e-9, flop / ms[0][1] * 1e-9, flop / ms[1][0] * 1e-9, flop / ms[1][1] * 1e-9,
         ms[0][1] / ms[1][1]);
  // ---- slab loop variants ------------------------------------------------------------------------------------
  {
    bf16x8 *w; hipMalloc(&w, (size_t)64 * 48 * 64 * 16 + 4 * 12 * 64 * 16);
    hipMemset(w, 0, (size_t)64 * 48 * 64 * 16 + 4 * 12 * 64 * 16);
    const int sl = 4000;
    float t[4];
    for (int v = 0; v < 4; ++v) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (v == 0) hipLaunchKernelGGL(slab_kernel<0>, dim3(blocks), dim3(256), 0, 0, w, out, sl);
        if (v == 1) hipLaunchKernelGGL(slab_kernel<1>, dim3(blocks), dim3(256), 0, 0, w, out, sl);
        if (v == 2) hipLaunchKernelGGL(slab_kernel<2>, dim3(blocks), dim3(256), 0, 0, w, out, sl);
        if (v == 3) hipLaunchKernelGGL(slab_kernel<3>, dim3(blocks), dim3(256), 0, 0, w, out, sl);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float x; hipEventElapsedTime(&x, e0, e1);
        if (x < best) best = x;
      }
      t[v] = best;
    }
    printf("{\"slab_loop_ns_per_slab\": {\"mfma_only\": %.1f, \"plus_12_loads\": %.1f, \"plus_split_16\": %.1f, \"plus_both\": %.1f}}\n",
           t[0] * 1e6 / sl, t[1] * 1e6 / sl, t[2] * 1e6 / sl, t[3] * 1e6 / sl);
  }
  return 0;
}
