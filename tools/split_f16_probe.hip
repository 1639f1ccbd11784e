// Diagnostic for DESIGN §3 (round 2): a float32 product on the f16 matrix cores as THREE exact partial products.
//   a = a1 + a2, a1 = f16(a s), a2 = f16(a s - a1)  (s a power of two that lifts the operand into the top of the f16
//   range, so both pieces are normal numbers: 11 + 1 + 11 significant bits >= the 24 of a float32);
//   a.b ~= a2 b1 + a1 b2 + a1 b1 : three v_mfma_f32_32x32x16_f16 with fp32 accumulation, every partial product exact,
//   the dropped a2 b2 <= 2^-24 |ab|.
// (1) accuracy of C[32x32] = A[32xK] B[Kx32], K = 512, against float64: fp32 MFMA, six-term bf16 (round 1), the
//     three-term f16 split with per-matrix / per-column power-of-two scales, and a two-term f16 product for scale;
// (2) rate, register-only loops on random data, one or two waves per SIMD: 6 x bf16 32x32x16, 3 x f16 32x32x16,
//     3 x f16 16x16x32 (same output tile per wave), in float32-equivalent TFLOP/s;
// (3) the slab loop the kernel would run: 8 waves (two per SIMD), per 16-deep slab and wave 4 weight-fragment loads
//     (L2-resident image, 16 B per lane each), 4 LDS fragment reads and 12 MFMAs on a 64 x 64 wave tile.
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/split_f16_probe.hip -o /tmp/f16_probe && /tmp/f16_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float a, __bf16 &p1, __bf16 &p2, __bf16 &p3) {
  p1 = (__bf16)a;
  const float r1 = a - (float)p1;
  p2 = (__bf16)r1;
  const float r2 = r1 - (float)p2;
  p3 = (__bf16)r2;
}
__device__ __forceinline__ void split2h(float a, _Float16 &p1, _Float16 &p2) {
  p1 = (_Float16)a;
  p2 = (_Float16)(a - (float)p1);   // the difference is exact in float32
}

// sa: scale of A (one power of two); sb[32]: per-column scales of B
__global__ void accuracy_kernel(const float *A, const float *B, int K, float sa, const float *sb, float *C32, float *C6,
                                float *C3h, float *C2h) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc32, acc6, acc3, acc2;
  for (int i = 0; i < 16; ++i) acc32[i] = acc6[i] = acc3[i] = acc2[i] = 0.0f;
  for (int s = 0; s < K / 2; ++s)
    acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + 2 * s + h], B[(2 * s + h) * 32 + r], acc32, 0, 0, 0);
  const float tb = sb[r];
  for (int s = 0; s < K / 16; ++s) {
    bf16x8 a1, a2, a3, b1, b2, b3;
    f16x8 ha1, ha2, hb1, hb2;
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * s + 8 * h + j;
      __bf16 p1, p2, p3;
      split3(A[r * K + k], p1, p2, p3);
      a1[j] = p1; a2[j] = p2; a3[j] = p3;
      split3(B[k * 32 + r], p1, p2, p3);
      b1[j] = p1; b2[j] = p2; b3[j] = p3;
      _Float16 q1, q2;
      split2h(A[r * K + k] * sa, q1, q2);
      ha1[j] = q1; ha2[j] = q2;
      split2h(B[k * 32 + r] * tb, q1, q2);
      hb1[j] = q1; hb2[j] = q2;
    }
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc6, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha2, hb1, acc3, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha1, hb2, acc3, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha1, hb1, acc3, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha1, hb2, acc2, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha1, hb1, acc2, 0, 0, 0);
  }
  const float inv = 1.0f / (sa * tb);   // powers of two: exact
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    C32[row * 32 + r] = acc32[i]; C6[row * 32 + r] = acc6[i]; C3h[row * 32 + r] = acc3[i] * inv; C2h[row * 32 + r] = acc2[i] * inv;
  }
}

// MODE 0: 6 x bf16 32x32x16; 1: 3 x f16 32x32x16; 2: 3 x f16 16x16x32 on four 16x16 tiles x two k halves (the same
// 32 x 32 x 16... per accumulator group: 32x32 outputs, K = 32 -> counted as two slabs)
template <int MODE, int WPS>
__global__ __launch_bounds__(256 * WPS) void rate_kernel(const float *seed, float *out, int slabs) {
  const int l = threadIdx.x & 63;
  constexpr int NACC = 4;
  f32x16 acc[NACC];
  for (int t = 0; t < NACC; ++t)
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
  bf16x8 a1, a2, a3, b1, b2, b3;
  f16x8 ha1, ha2, hb1, hb2;
  for (int j = 0; j < 8; ++j) {
    const float af = seed[(threadIdx.x * 8 + j) & 4095], bf = seed[(threadIdx.x * 8 + j + 2048) & 4095];
    __bf16 p1, p2, p3;
    split3(af, p1, p2, p3); a1[j] = p1; a2[j] = p2; a3[j] = p3;
    split3(bf, p1, p2, p3); b1[j] = p1; b2[j] = p2; b3[j] = p3;
    _Float16 q1, q2;
    split2h(af * 64.0f, q1, q2); ha1[j] = q1; ha2[j] = q2;
    split2h(bf * 64.0f, q1, q2); hb1[j] = q1; hb2[j] = q2;
  }
  for (int g = 0; g < slabs; ++g) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      if (MODE == 0) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[t], 0, 0, 0);
      } else if (MODE == 1) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha2, hb1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha1, hb2, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha1, hb1, acc[t], 0, 0, 0);
      } else {
        // a 32 x 32 output block as four 16 x 16 tiles (4 registers each), K = 32 per instruction: the flops of TWO
        // 32x32x16 slabs per pass -> the caller halves `slabs`
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 c = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha2, hb1, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha1, hb2, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha1, hb1, c, 0, 0, 0);
          acc[t][4 * q] = c[0]; acc[t][4 * q + 1] = c[1]; acc[t][4 * q + 2] = c[2]; acc[t][4 * q + 3] = c[3];
        }
      }
    }
  }
  float s = 0.0f;
  for (int t = 0; t < NACC; ++t)
    for (int i = 0; i < 16; ++i) s += acc[t][i];
  out[blockIdx.x * 256 * WPS + threadIdx.x] = s;
  (void)l;
}

// (3) slab loop of the planned kernel.  8 waves, wave w owns n-tiles 2w, 2w+1 x two 32-row halves (64 accumulator
// registers).  Weights: [n-tile 16][slab 32][piece 2][lane 64] x 16 B (1 MB, L2-resident, every workgroup streams the same
// image like the real kernel's member-major order).  Activations: LDS [piece 2][row 64][520] f16.
// VAR 0: MFMAs only; 1: + weight loads; 2: + LDS reads; 3: both.   SHAPE 0: 32x32x16, 1: 16x16x32 (same wave tile).
constexpr int HSTR = 520;
template <int VAR, int SHAPE>
__global__ __launch_bounds__(512, 2) void slab_kernel(const f16x8 *w, const float *seed, float *out, int passes) {
  extern __shared__ f32x4 smem4[];
  _Float16 *hb = reinterpret_cast<_Float16 *>(smem4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  for (int i = tid; i < 2 * 64 * HSTR; i += 512) hb[i] = (_Float16)(seed[i & 4095] * 64.0f);
  __syncthreads();
  f32x16 acc[2][2];
  for (int t = 0; t < 2; ++t)
    for (int b = 0; b < 2; ++b)
      for (int i = 0; i < 16; ++i) acc[t][b][i] = 0.0f;
  const f16x8 *wa = w + (size_t)(2 * wave) * 32 * 2 * 64 + lane;
  const _Float16 *b0 = hb + (size_t)r * HSTR + 8 * h;
  f16x8 A[2][2][2], B[2][2][2];   // [ping-pong][tile | row half][piece]
  auto load_a = [&](f16x8 (&a)[2][2], int s) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 2; ++p) a[t][p] = wa[((size_t)(t * 32 + s) * 2 + p) * 64];
  };
  auto read_b = [&](f16x8 (&b)[2][2], int s) {
#pragma unroll
    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        b[bt][p] = *reinterpret_cast<const f16x8 *>(b0 + (size_t)p * 64 * HSTR + (size_t)bt * 32 * HSTR + 16 * s);
  };
  load_a(A[0], 0); read_b(B[0], 0);
  load_a(A[1], 1); read_b(B[1], 1);
  auto mm = [&](f32x16 &c, const f16x8 &x, const f16x8 &y) {
    if (SHAPE == 0) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0);
    } else {   // two 16x16x32 on halves of the accumulator: same pipe time as one 32x32x16, different operand meaning
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 d = {c[8 * q], c[8 * q + 1], c[8 * q + 2], c[8 * q + 3]};
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, d, 0, 0, 0);
        c[8 * q] = d[0]; c[8 * q + 1] = d[1]; c[8 * q + 2] = d[2]; c[8 * q + 3] = d[3];
      }
    }
  };
  auto step = [&](int s, int cur) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int bt = 0; bt < 2; ++bt) {
        mm(acc[t][bt], A[cur][t][1], B[cur][bt][0]);
        mm(acc[t][bt], A[cur][t][0], B[cur][bt][1]);
        mm(acc[t][bt], A[cur][t][0], B[cur][bt][0]);
      }
    __builtin_amdgcn_sched_barrier(0);
    if (VAR & 1) load_a(A[cur], (s + 2) & 31);
    if (VAR & 2) read_b(B[cur], (s + 2) & 31);
  };
  for (int it = 0; it < passes; ++it)
    for (int s = 0; s < 32; s += 2) { step(s, 0); step(s + 1, 1); }
  float v = 0.0f;
  for (int t = 0; t < 2; ++t)
    for (int b = 0; b < 2; ++b)
      for (int i = 0; i < 16; ++i) v += acc[t][b][i];
  out[blockIdx.x * 512 + tid] = v;
}

template <typename F>
static float best_ms(F launch, int reps = 3) {
  float best = 1e30f;
  for (int rep = 0; rep < reps; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float t; hipEventElapsedTime(&t, e0, e1);
    if (t < best) best = t;
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  return best;
}

static float pow2_scale(double maxabs, int top) {   // 2^k with maxabs * 2^k in [2^(top-1), 2^top)
  if (!(maxabs > 0.0)) return 1.0f;
  int e; frexp(maxabs, &e);                          // maxabs = m 2^e, m in [0.5, 1)
  return (float)ldexp(1.0, top - e);
}

int main() {
  const int K = 512;
  float *hA = (float *)malloc(32 * K * 4), *hB = (float *)malloc(K * 32 * 4);
  srand(7);
  auto rnd = []() {
    double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
    return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
  };
  float *dA, *dB, *dC, *dS;
  hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, K * 32 * 4); hipMalloc(&dC, 4 * 1024 * 4); hipMalloc(&dS, 32 * 4);
  printf("{\"K\": %d, \"max_err_over_sum_abs\": {", K);
  // case 0: the operands of round 1's probe (A wide-range, B ~ 0.05 N(0,1)); case 1: both wide-range over 12 decades;
  // case 2: B columns whose bound is 2^10 looser than their content (a loose norm bound costs nothing)
  for (int cs = 0; cs < 3; ++cs) {
    for (int i = 0; i < 32 * K; ++i) hA[i] = (float)(rnd() * exp((cs == 1 ? 4.0 : 2.0) * rnd()));
    for (int i = 0; i < K * 32; ++i) hB[i] = (float)(rnd() * (cs == 1 ? exp(4.0 * rnd()) : 0.05));
    double amax = 0.0, bmax[32];
    for (int i = 0; i < 32 * K; ++i) amax = fmax(amax, fabs((double)hA[i]));
    float hS[32];
    for (int j = 0; j < 32; ++j) {
      bmax[j] = 0.0;
      for (int k = 0; k < K; ++k) bmax[j] = fmax(bmax[j], fabs((double)hB[k * 32 + j]));
      hS[j] = pow2_scale(bmax[j] * (cs == 2 ? 1024.0 : 1.0), 14);
    }
    const float sa = pow2_scale(amax, 14);
    double ref[1024], mag[1024];
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double s = 0.0, m = 0.0;
        for (int k = 0; k < K; ++k) { s += (double)hA[i * K + k] * hB[k * 32 + j]; m += fabs((double)hA[i * K + k] * hB[k * 32 + j]); }
        ref[i * 32 + j] = s; mag[i * 32 + j] = m;
      }
    hipMemcpy(dA, hA, 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, K * 32 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dS, hS, 32 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(accuracy_kernel, dim3(1), dim3(64), 0, 0, dA, dB, K, sa, dS, dC, dC + 1024, dC + 2048, dC + 3072);
    float hC[4096];
    hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
    double e[4] = {0, 0, 0, 0};
    for (int v = 0; v < 4; ++v)
      for (int i = 0; i < 1024; ++i) e[v] = fmax(e[v], fabs((double)hC[v * 1024 + i] - ref[i]) / mag[i]);
    printf("%s\"case%d\": {\"fp32_mfma\": %.3e, \"split_bf16_6\": %.3e, \"split_f16_3\": %.3e, \"f16_2term\": %.3e}",
           cs ? ", " : "", cs, e[0], e[1], e[2], e[3]);
  }
  printf("}");

  // ---- rate ------------------------------------------------------------------------------------------------------
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount;
  float hseed[4096];
  for (int i = 0; i < 4096; ++i) hseed[i] = (float)rnd();
  float *seed, *out;
  hipMalloc(&seed, sizeof(hseed)); hipMemcpy(seed, hseed, sizeof(hseed), hipMemcpyHostToDevice);
  hipMalloc(&out, (size_t)blocks * 1024 * 4);
  const int slabs = 40000;   // ~10-20 ms per launch: long enough for the clock to settle
  // useful float32 flops: each (slab, accumulator) is one 32 x 32 x 16 product = 32768 flop; 4 accumulators per wave
  auto tf = [&](int waves, int sl, float ms) { return (double)blocks * waves * 4.0 * sl * 32768.0 / ms * 1e-9; };
  const float t60 = best_ms([&] { hipLaunchKernelGGL((rate_kernel<0, 1>), dim3(blocks), dim3(256), 0, 0, seed, out, slabs); });
  const float t31 = best_ms([&] { hipLaunchKernelGGL((rate_kernel<1, 1>), dim3(blocks), dim3(256), 0, 0, seed, out, slabs); });
  const float t32 = best_ms([&] { hipLaunchKernelGGL((rate_kernel<1, 2>), dim3(blocks), dim3(512), 0, 0, seed, out, slabs / 2); });
  const float t16 = best_ms([&] { hipLaunchKernelGGL((rate_kernel<2, 1>), dim3(blocks), dim3(256), 0, 0, seed, out, slabs / 2); });
  const float t162 = best_ms([&] { hipLaunchKernelGGL((rate_kernel<2, 2>), dim3(blocks), dim3(512), 0, 0, seed, out, slabs / 4); });
  printf(", \"cus\": %d, \"fp32_equivalent_tflops\": {\"bf16_6_32x32x16\": %.1f, \"f16_3_32x32x16\": %.1f, "
         "\"f16_3_32x32x16_2wps\": %.1f, \"f16_3_16x16x32\": %.1f, \"f16_3_16x16x32_2wps\": %.1f}}\n",
         blocks, tf(4, slabs, t60), tf(4, slabs, t31), tf(8, slabs / 2, t32), tf(4, slabs, t16), tf(8, slabs / 2, t162));

  // ---- slab loop ---------------------------------------------------------------------------------------------------
  {
    const size_t wbytes = (size_t)16 * 32 * 2 * 64 * 16;
    f16x8 *w; hipMalloc(&w, wbytes);
    _Float16 *hw = (_Float16 *)malloc(wbytes);
    for (size_t i = 0; i < wbytes / 2; ++i) hw[i] = (_Float16)(float)(rnd() * 100.0);
    hipMemcpy(w, hw, wbytes, hipMemcpyHostToDevice);
    const size_t lds = (size_t)2 * 64 * HSTR * 2;
    const int passes = 400;   // x 32 slabs
    float t[2][4];
#define RUN(V, S)                                                                                                     \
  hipFuncSetAttribute(reinterpret_cast<const void *>(slab_kernel<V, S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
  t[S][V] = best_ms([&] { hipLaunchKernelGGL((slab_kernel<V, S>), dim3(blocks), dim3(512), lds, 0, w, seed, out, passes); });
    RUN(0, 0) RUN(1, 0) RUN(2, 0) RUN(3, 0) RUN(0, 1) RUN(1, 1) RUN(2, 1) RUN(3, 1)
#undef RUN
    // per slab and workgroup: 8 waves x 12 MFMAs = 96 MFMAs = a 512 x 64 x 16 float32 product
    auto tfs = [&](float ms) { return (double)blocks * passes * 32.0 * 2.0 * 512 * 64 * 16 / ms * 1e-9; };
    printf("{\"slab_loop\": {\"ns_per_slab\": {\"mfma_only\": %.1f, \"plus_weight_loads\": %.1f, \"plus_lds_reads\": %.1f, \"both\": %.1f}, "
           "\"fp32_equivalent_tflops\": {\"mfma_only\": %.1f, \"both\": %.1f}, "
           "\"shape_16x16x32\": {\"ns_per_slab_mfma_only\": %.1f, \"ns_per_slab_both\": %.1f, \"tflops_both\": %.1f}}}\n",
           t[0][0] * 1e6 / (passes * 32), t[0][1] * 1e6 / (passes * 32), t[0][2] * 1e6 / (passes * 32), t[0][3] * 1e6 / (passes * 32),
           tfs(t[0][0]), tfs(t[0][3]), t[1][0] * 1e6 / (passes * 32), t[1][3] * 1e6 / (passes * 32), tfs(t[1][3]));
  }
  return 0;
}
