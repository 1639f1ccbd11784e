// Issue rate of the VALU instructions the f16 split can be built from (gfx950): cycles per instruction of one wave that
// issues long runs of independent instructions (s_memtime around 64 x 16 of them).  Dev tool:
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void rate_kernel(unsigned long long *out, float seed) {
  float a[16], b[16];
  unsigned q[16];
  for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; b[i] = seed * 0.5f + i; q[i] = i; }
  const float t = seed * 3.0f;
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 64; ++it) {
    if constexpr (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(t));
      REP16(X)
#undef X
    } else if constexpr (OP == 1) {
#define X(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(q[i]) : "v"(a[i]), "v"(t));
      REP16(X)
#undef X
    } else if constexpr (OP == 2) {
#define X(i) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(q[i]) : "v"(a[i]), "v"(t));
      REP16(X)
#undef X
    } else if constexpr (OP == 3) {
#define X(i) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(q[i]) : "v"(a[i]));
      REP16(X)
#undef X
    } else if constexpr (OP == 4) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(q[i]) : "v"(a[i]), "v"(b[i]));
      REP16(X)
#undef X
    } else if constexpr (OP == 5) {
#define X(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(q[i]));
      REP16(X)
#undef X
    } else if constexpr (OP == 6) {
#define X(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(a[i]) : "v"(q[i]), "v"(t), "v"(b[i]));
      REP16(X)
#undef X
    } else if constexpr (OP == 7) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(*(double *)&a[i & ~1]) : "v"(*(double *)&b[i & ~1]), "v"(*(double *)&b[(i + 2) & 14]));
      REP16(X)
#undef X
    } else if constexpr (OP == 8) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(t));
      REP16(X)
#undef X
    } else if constexpr (OP == 9) {
#define X(i) asm volatile("v_exp_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
      REP16(X)
#undef X
    } else if constexpr (OP == 10) {
#define X(i) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(q[i]) : "v"(q[(i + 1) & 15]), "v"(q[(i + 2) & 15]));
      REP16(X)
#undef X
    } else if constexpr (OP == 11) {
#define X(i) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(q[i]) : "v"(q[(i + 1) & 15]), "v"(q[(i + 2) & 15]), "v"(q[(i + 3) & 15]));
      REP16(X)
#undef X
    } else if constexpr (OP == 12) {
#define X(i) asm volatile("v_and_b32 %0, %1, %2" : "=v"(q[i]) : "v"(q[(i + 1) & 15]), "v"(q[(i + 2) & 15]));
      REP16(X)
#undef X
    } else if constexpr (OP == 13) {
#define X(i) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(a[(i + 1) & 15]));
      REP16(X)
#undef X
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  unsigned u = 0;
  for (int i = 0; i < 16; ++i) { s += a[i]; u ^= q[i]; }
  if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
  if (s == 12345.678f && u == 77) out[1] = 1;
}

template <int OP>
void run(const char *name, int waves_per_simd) {
  unsigned long long *d, h[2];
  hipMalloc(&d, 4096);
  hipMemset(d, 0, 4096);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(1), dim3(256 * waves_per_simd), 0, 0, d, 1.25f);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(1), dim3(256 * waves_per_simd), 0, 0, d, 1.25f);
  hipDeviceSynchronize();
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("%-28s waves/SIMD %d: %6.2f cycles per instruction and wave (s_memtime ticks x clock ratio not applied)\n", name,
         waves_per_simd, (double)h[0] / (64.0 * 16.0));
  hipFree(d);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>("v_fma_f32", w);
    run<8>("v_mul_f32", w);
    run<13>("v_sub_f32", w);
    run<12>("v_and_b32", w);
    run<1>("v_fma_mixlo_f16", w);
    run<2>("v_fma_mixhi_f16 (f16 src2)", w);
    run<6>("v_fma_mix_f32 (f16 src0)", w);
    run<3>("v_cvt_f16_f32", w);
    run<4>("v_cvt_pk_f16_f32", w);
    run<5>("v_cvt_f32_f16", w);
    run<7>("v_pk_mul_f32", w);
    run<10>("v_pk_fma_f16", w);
    run<11>("v_perm_b32", w);
    run<9>("v_exp_f32", w);
  }
  return 0;
}
