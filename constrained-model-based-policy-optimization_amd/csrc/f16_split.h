// Device helpers of the three-term f16 matrix path (ens_h3.hip, critic_f16.hip): a float32 operand, lifted by a power of
// two into the top of the f16 range, as two f16 pieces; a float32 product as three MFMAs; the swish / lift / split epilogue.
#pragma once
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NSTAT = 12;             // floats per member in the stats block: {scale, max column 1-norm, max |b|, max |W|} x 3 layers

__device__ __forceinline__ float swishf(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// 2^k with v 2^k in [2^13, 2^14); 1 for zero, subnormal and non-finite v
__host__ __device__ __forceinline__ float pow2_lift(float v) {
  union { float f; unsigned u; } c;
  c.f = v;
  const int ex = (int)((c.u >> 23) & 255u);
  if (ex == 0 || ex == 255) return 1.0f;
  int k = 13 - (ex - 127);
  k = k < -100 ? -100 : (k > 100 ? 100 : k);
  c.u = (unsigned)(k + 127) << 23;
  return c.f;
}

#ifndef POW2_RCP
#define POW2_RCP 1     // 0: the IEEE division (diagnostic)
#endif
// 1 / x for x a power of two in the normal range (the lifts, the weights' scales and their products): the exponent negated, one
// integer subtraction instead of the dozen instructions of an IEEE division
__host__ __device__ __forceinline__ float pow2_rcp(float x) {
#if POW2_RCP
  union { float f; unsigned u; } c;
  c.f = x;
  c.u = 0x7F000000u - c.u;
  return c.f;
#else
  return 1.0f / x;
#endif
}

__device__ __forceinline__ void split_h(float a, _Float16 &p1, _Float16 &p2) {
  p1 = (_Float16)a;
  p2 = (_Float16)(a - (float)p1);   // exact difference
}

// three-term product, smallest terms first
__device__ __forceinline__ void mm3(f32x16 &acc, const f16x8 &a1, const f16x8 &a2, const f16x8 &b1, const f16x8 &b2) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
}

// ---- two floats x t -> their two f16 pieces, packed ------------------------------------------------------------------
// p1 = f16(x t), p2 = f16(x t - p1): one v_fma_mix each (see the epilogue below).  HAZ: the pieces feed an MFMA straight from
// the registers -- hipcc pads no hazard behind an asm statement, so the wait states stand inside the strings of the high halves.
template <bool HAZ>
__device__ __forceinline__ void split2(float x0, float x1, float t, unsigned &q1, unsigned &q2) {
  unsigned a, b;   // (the low halves are written first: "=&v", the registers need no initial value)
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=&v"(a) : "v"(x0), "v"(t));
  if constexpr (HAZ) asm("v_fma_mixhi_f16 %0, %1, %2, 0\n\ts_nop 1" : "+v"(a) : "v"(x1), "v"(t));
  else asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(a) : "v"(x1), "v"(t));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=&v"(b) : "v"(x0), "v"(t), "v"(a));
  if constexpr (HAZ) asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\ts_nop 1" : "+v"(b) : "v"(x1), "v"(t), "v"(a));
  else asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(b) : "v"(x1), "v"(t), "v"(a));
  q1 = a;
  q2 = b;
}

// ---- the swish / lift / split epilogue of four accumulator values, cut into twelve pieces of at most ~20 issue cycles
// so that one piece can stand behind each MFMA of a 12-MFMA group (a wave issues in order: what stands between two MFMAs
// runs in the shadow of the first).  The split uses the mixed-precision FMAs: p1 = f16(x t) and p2 = f16(x t - p1) are
// one v_fma_mix{lo,hi}_f16 each -- the product is exact inside the FMA, so p1 + p2 = x t (1 + d), |d| <= 2^-24, with a
// single rounding per piece -- and lo / hi write the two halves of a dword, so the pieces come out packed.
// PIN: an empty volatile asm after each piece keeps the compiler from sinking it into a later piece.
struct Epi4 {
  float z[4], e[4];
  unsigned q1[2], q2[2];   // p1 / p2 of the four values, packed f16x2
};
// PRE: the caller passes inv and the bias multiplied by log2(e) and the lift tn by ln 2 -- stage 0 then yields y = z log2(e),
// the exponential reads -y through its source modifier (no separate multiply: 7 instead of 8 instructions per value), and
// the split's exact product y sigma(z) (tn ln 2) is swish(z) tn up to the float rounding of log2(e) ln 2 (1 - 2e-8).
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
// FUSE: the caller passes 1 / tn instead of tn and the lift rides in the reciprocal -- sigma(z) tn = 1 / ((1 + e) / tn), the
// division by tn as the multiplier of ONE fma that also adds the 1 (stage 4) -- so the mixed-precision FMAs of the split form
// z (sigma tn) themselves (exact inside the FMA) and the separate product z sigma (stage 7) is gone: 6 instead of 7 vector
// instructions per activation (5 + the scaling in PRE's case).  e = inf (z -> -inf) gives sigma tn = 0, e = 0 gives tn.
template <int K, bool PIN, bool PRE = false, bool FUSE = false>
__device__ __forceinline__ void epi_stage(Epi4 &s, const f32x16 &d, int q, float inv, const f32x4 &bv, float tn) {
  if constexpr (PRE && K == 1) {
    // (nothing: the scaling is in inv / bv)
  } else if constexpr (PRE && (K == 2 || K == 3)) {
    constexpr int o = 2 * (K - 2);
    s.e[o] = __builtin_amdgcn_exp2f(-s.z[o]); s.e[o + 1] = __builtin_amdgcn_exp2f(-s.z[o + 1]);
    if (PIN) asm volatile("" : "+v"(s.e[o]), "+v"(s.e[o + 1]));
  } else if constexpr (K == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) s.z[i] = __builtin_fmaf(d[4 * q + i], inv, bv[i]);
    if (PIN) asm volatile("" : "+v"(s.z[0]), "+v"(s.z[1]), "+v"(s.z[2]), "+v"(s.z[3]));
  } else if constexpr (K == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) s.e[i] = s.z[i] * -1.4426950408889634f;
    if (PIN) asm volatile("" : "+v"(s.e[0]), "+v"(s.e[1]), "+v"(s.e[2]), "+v"(s.e[3]));
  } else if constexpr (K == 2 || K == 3) {
    constexpr int o = 2 * (K - 2);
    s.e[o] = __builtin_amdgcn_exp2f(s.e[o]); s.e[o + 1] = __builtin_amdgcn_exp2f(s.e[o + 1]);
    if (PIN) asm volatile("" : "+v"(s.e[o]), "+v"(s.e[o + 1]));
  } else if constexpr (K == 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) s.e[i] = FUSE ? __builtin_fmaf(s.e[i], tn, tn) : 1.0f + s.e[i];
    if (PIN) asm volatile("" : "+v"(s.e[0]), "+v"(s.e[1]), "+v"(s.e[2]), "+v"(s.e[3]));
  } else if constexpr (K == 5 || K == 6) {
    constexpr int o = 2 * (K - 5);
#ifdef EPI_NEWTON_RCP
    if constexpr (PIN && FUSE) {
      // diagnostic: the reciprocal on the plain vector pipe (integer seed + three Newton steps, error ~5e-8) -- does a
      // transcendental instruction cost the partner wave's MFMAs more than seven plain ones?
#pragma unroll
      for (int i = o; i < o + 2; ++i) {
        const float d = fminf(s.e[i], 1.2676506e30f);
        float rr = __uint_as_float(0x7EF311C7u - __float_as_uint(d));
        rr = rr * __builtin_fmaf(-d, rr, 2.0f);
        rr = rr * __builtin_fmaf(-d, rr, 2.0f);
        rr = rr * __builtin_fmaf(-d, rr, 2.0f);
        s.e[i] = rr;
      }
      asm volatile("" : "+v"(s.e[o]), "+v"(s.e[o + 1]));
    } else
#endif
    {
      s.e[o] = __builtin_amdgcn_rcpf(s.e[o]); s.e[o + 1] = __builtin_amdgcn_rcpf(s.e[o + 1]);
      if (PIN) asm volatile("" : "+v"(s.e[o]), "+v"(s.e[o + 1]));
    }
  } else if constexpr (K == 7) {
    if constexpr (!FUSE) {
#pragma unroll
      for (int i = 0; i < 4; ++i) s.z[i] = s.z[i] * s.e[i];
      if (PIN) asm volatile("" : "+v"(s.z[0]), "+v"(s.z[1]), "+v"(s.z[2]), "+v"(s.z[3]));
    }
  } else if constexpr (K == 8) {
    // the second factor of the split's products: the lift, or sigma tn of the value itself
    const float m0 = FUSE ? s.e[0] : tn, m1 = FUSE ? s.e[1] : tn, m2 = FUSE ? s.e[2] : tn, m3 = FUSE ? s.e[3] : tn;
    // (hipcc pads no hazard behind an asm statement: where the pieces feed an MFMA straight from the registers -- the
    // tail, PIN == false -- the wait states between a VALU write and an MFMA's operand read stand inside the string)
    // (the low halves are written first, "=&v": the registers need no initial value -- a v_mov each otherwise)
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=&v"(s.q1[0]) : "v"(s.z[0]), "v"(m0));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=&v"(s.q1[1]) : "v"(s.z[2]), "v"(m2));
    if constexpr (PIN) {
      asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(s.q1[0]) : "v"(s.z[1]), "v"(m1));
      asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(s.q1[1]) : "v"(s.z[3]), "v"(m3));
    } else {
      asm("v_fma_mixhi_f16 %0, %1, %2, 0\n\ts_nop 1" : "+v"(s.q1[0]) : "v"(s.z[1]), "v"(m1));
      asm("v_fma_mixhi_f16 %0, %1, %2, 0\n\ts_nop 1" : "+v"(s.q1[1]) : "v"(s.z[3]), "v"(m3));
    }
    if (PIN) asm volatile("" : "+v"(s.q1[0]), "+v"(s.q1[1]));
  } else if constexpr (K == 9) {
    const float m0 = FUSE ? s.e[0] : tn, m1 = FUSE ? s.e[1] : tn, m2 = FUSE ? s.e[2] : tn, m3 = FUSE ? s.e[3] : tn;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=&v"(s.q2[0]) : "v"(s.z[0]), "v"(m0), "v"(s.q1[0]));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=&v"(s.q2[1]) : "v"(s.z[2]), "v"(m2), "v"(s.q1[1]));
    if constexpr (PIN) {
      asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(s.q2[0]) : "v"(s.z[1]), "v"(m1), "v"(s.q1[0]));
      asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(s.q2[1]) : "v"(s.z[3]), "v"(m3), "v"(s.q1[1]));
    } else {
      asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\ts_nop 1" : "+v"(s.q2[0]) : "v"(s.z[1]), "v"(m1), "v"(s.q1[0]));
      asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\ts_nop 1" : "+v"(s.q2[1]) : "v"(s.z[3]), "v"(m3), "v"(s.q1[1]));
    }
    if (PIN) asm volatile("" : "+v"(s.q2[0]), "+v"(s.q2[1]));
  }
}
template <bool PIN, bool PRE = false, bool FUSE = false>
__device__ __forceinline__ void epi_all(Epi4 &s, const f32x16 &d, int q, float inv, const f32x4 &bv, float tn) {
  epi_stage<0, PIN, PRE, FUSE>(s, d, q, inv, bv, tn); epi_stage<1, PIN, PRE, FUSE>(s, d, q, inv, bv, tn);
  epi_stage<2, PIN, PRE, FUSE>(s, d, q, inv, bv, tn); epi_stage<3, PIN, PRE, FUSE>(s, d, q, inv, bv, tn);
  epi_stage<4, PIN, PRE, FUSE>(s, d, q, inv, bv, tn); epi_stage<5, PIN, PRE, FUSE>(s, d, q, inv, bv, tn);
  epi_stage<6, PIN, PRE, FUSE>(s, d, q, inv, bv, tn); epi_stage<7, PIN, PRE, FUSE>(s, d, q, inv, bv, tn);
  epi_stage<8, PIN, PRE, FUSE>(s, d, q, inv, bv, tn); epi_stage<9, PIN, PRE, FUSE>(s, d, q, inv, bv, tn);
}


struct cmbpo_mlp;
// per-member statistics {power-of-two weight scale, largest column 1-norm, max |b|, max |W|} x 3 layers -> stats[E][NSTAT]
void cmbpo_internal_f16_stats(const cmbpo_mlp *m, float *stats, hipStream_t s);
// layer `layer` of the handle's fp32 packs -> two f16 images [n-tile][k-slab 16][piece 2][lane 64][8 halves], scaled by the
// member's power of two; perm: k order of an accumulator tile used as the B operand (see h3_pack_kernel)
void cmbpo_internal_f16_pack(const cmbpo_mlp *m, int layer, void *dst, size_t dst_stride, int n_tiles, int slabs, int perm,
                             const float *stats, hipStream_t s);
