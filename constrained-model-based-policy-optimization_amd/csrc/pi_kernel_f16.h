// The policy kernels with every matrix product as three f16 MFMAs on two-piece operands (f16_split.h); included by
// policy_update.hip inside its anonymous namespace, after PiArgs / PiPack / the row-tile helpers.
//
// Same tiling as pi_kernel (one persistent workgroup walks tiles of 32 samples, wave w owns hidden units 32 w .. 32 w + 31,
// weight gradients stay in registers across tiles), same results to fp32 rounding, ~3 x fewer MFMA cycles:
//   * matrices: split once per pack into images [n-tile][k-slab 16][piece 2][lane 64] of 8 halves, each matrix lifted by
//     its own power of two (pack_f16_part); they reach the MFMAs through a 4-slab register ring that runs ahead of the products
//     along the tile's fixed sequence of matrices (stream_tab), so no product starts with an L2 round trip;
//   * activations that feed a 128-deep product (h1, dh1, delta2) live in LDS as two-piece images, split once by the wave
//     that produces them (same footprint as the fp32 rows they replace); the narrow operands (x, the 32-column slices of
//     h2 / dh2 each wave contracts, the cotangent) stay fp32 in LDS and are split as they are read;
//   * lifts are powers of two known before an image is produced: 2^14 for tanh outputs, the exact tile maximum (one LDS
//     atomic max, visible behind the next barrier) where a barrier separates producer and split, and the lift of an upper
//     bound (max |x| D max |dW0| + max |db0|;  max |cot| A max |W2|) for dh1 and delta2, which are split by their producer.
//     A lift below the optimum costs nothing until the second piece leaves the f16 range, ~2^10 further down.
#pragma once
#include <utility>

#ifndef PI_DEFER_DMA
#define PI_DEFER_DMA 1      // 0: round 3's first form (the barrier that ends a tile waits for the next tile's saved activations)
#endif
constexpr float T_TANH = 16384.0f;      // |tanh| <= 1
constexpr float T_TANH_INV = 1.0f / 16384.0f;
constexpr int RSH = 2 * RS;             // row stride of a two-piece image, in halves
constexpr int P2H = HID;                // offset of the second piece inside a row, in halves
constexpr int RW = 4;                   // slabs in the matrix ring
// saved activations of a tile on this path: the LDS block [h1 two-piece image | h2 fp32 rows] as it stands, row padding
// included (2 x 32 x RS floats): cmbpo_pi_loss_grad copies it out, a product on the saved activations copies it back
// with LDS-DMA -- no registers, no split, no LDS stores
constexpr int ACT_TILE4 = 2 * BB * RS / 4;                 // float4s per tile
constexpr int ACT_PIECES = (ACT_TILE4 * 16 + 1023) / 1024; // 1-KiB wave-instructions per tile (the last one partly past the block)

// A barrier that orders the workgroup's LDS traffic only.  __syncthreads() is a workgroup-scope fence + s_barrier: hipcc waits
// vmcnt(0) in front of it, i.e. for every global load still in flight -- the per-sample batch columns requested at the top
// of a tile for its element phase, the next tile's observations, the matrix ring: each became an exposed L2 / HBM round
// trip at the next barrier (the first barrier of a tile alone waited ~3 k cycles).  Here only the LDS counter is drained;
// the compiler still tracks the loads and waits for each one where its value is used.  NOT for a barrier that publishes
// LDS-DMA data (counted in vmcnt): the one that ends a tile stays a __syncthreads().
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float pow2_inv(float t) {   // exact inverse of a power of two in [2^-126, 2^126]
  return __uint_as_float(0x7F000000u - __float_as_uint(t));
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// (split2<HAZ>: f16_split.h)
__device__ __forceinline__ void split8(const f32x4 &lo, const f32x4 &hi, float t, f16x8 &p1, f16x8 &p2) {
  unsigned a[4], b[4];
  split2<true>(lo[0], lo[1], t, a[0], b[0]);
  split2<true>(lo[2], lo[3], t, a[1], b[1]);
  split2<true>(hi[0], hi[1], t, a[2], b[2]);
  split2<true>(hi[2], hi[3], t, a[3], b[3]);
  const u32x4 q1 = {a[0], a[1], a[2], a[3]}, q2 = {b[0], b[1], b[2], b[3]};
  p1 = __builtin_bit_cast(f16x8, q1);
  p2 = __builtin_bit_cast(f16x8, q2);
}

// accumulator tile (rows n_base.., cols b) x t -> the two-piece image
__device__ __forceinline__ void store_tile_H(const f32x16 &v, int n_base, _Float16 *img, float t, int lane) {
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned q1[2], q2[2];
    split2<false>(v[4 * q], v[4 * q + 1], t, q1[0], q2[0]);
    split2<false>(v[4 * q + 2], v[4 * q + 3], t, q1[1], q2[1]);
    _Float16 *dst = img + j * RSH + n_base + 8 * q + 4 * h;
    *reinterpret_cast<uint2 *>(dst) = make_uint2(q1[0], q1[1]);
    *reinterpret_cast<uint2 *>(dst + P2H) = make_uint2(q2[0], q2[1]);
  }
}

// the tile rows n_base.. of this wave's columns, as p1 + p2 (= value x lift, exact in fp32)
__device__ __forceinline__ f32x16 load_tile_H(const _Float16 *img, int n_base, int lane) {
  const int j = lane & 31, h = lane >> 5;
  f32x16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const _Float16 *src = img + j * RSH + n_base + 8 * q + 4 * h;
    const uint2 a = *reinterpret_cast<const uint2 *>(src), b = *reinterpret_cast<const uint2 *>(src + P2H);
    // p1 + p2 as one mixed-precision FMA per value (p1 x 1.0 + p2, both f16 sources, fp32 result)
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(v[4 * q]) : "v"(a.x), "v"(b.x));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(v[4 * q + 1]) : "v"(a.x), "v"(b.x));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(v[4 * q + 2]) : "v"(a.y), "v"(b.y));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(v[4 * q + 3]) : "v"(a.y), "v"(b.y));
  }
  return v;
}

// largest |value| -> one LDS word per tile (non-negative floats order like their bits)
__device__ __forceinline__ void thread_max_put(float m, unsigned *slot, int lane) {
  m = fmaxf(m, __shfl_xor(m, 1, 64));
  m = fmaxf(m, __shfl_xor(m, 2, 64));
  if ((lane & 3) == 0) atomicMax(slot, __float_as_uint(m));
}
__device__ __forceinline__ void tile_max_put(const f32x16 &o, unsigned *slot, int lane) {
  float m = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(o[r]));
  thread_max_put(m, slot, lane);
}

// ---- the tile's sequence of matrix slabs ------------------------------------------------------------------------------
enum { I_WF0 = 0, I_WF1, I_WF2, I_WB2, I_WB1, I_VF0, I_VF1, I_VF2, I_COUNT };
struct StreamTab {
  int L;
  unsigned char img[64], slab[64];
};
constexpr void stream_add(StreamTab &t, int img, int ns) {
  for (int s = 0; s < ns; ++s) {
    t.img[t.L] = (unsigned char)img;
    t.slab[t.L] = (unsigned char)s;
    ++t.L;
  }
}
constexpr StreamTab stream_tab(int mode, int n_it, bool cached) {
  StreamTab t{};
  const int s0 = 2 * n_it;
  if (mode != MODE_FVP || !cached) {     // forward
    stream_add(t, I_WF0, s0);
    stream_add(t, I_WF1, 8);
  }
  if (mode == MODE_FVP) {                // JVP
    stream_add(t, I_VF0, s0);
    stream_add(t, I_VF1, 8);
    stream_add(t, I_WF1, 8);
    stream_add(t, I_VF2, 2);
  }
  stream_add(t, I_WF2, 2);
  if (mode != MODE_EVAL) {               // backward
    stream_add(t, I_WB2, 2);
    stream_add(t, I_WB1, 8);
  }
  return t;
}
template <int MODE, int N_IT, bool CACHED>
struct Stream {
  static constexpr StreamTab tab = stream_tab(MODE, N_IT, CACHED);
};

typedef u32x4 Ring[RW][2];

template <class ST>
__device__ __forceinline__ void ring_fill(Ring &R, const u32x4 *(&imgs)[I_COUNT], int lane_t) {
  static_for<RW>([&](auto ic) {
    constexpr int q = decltype(ic)::value;
    constexpr int im = ST::tab.img[q], sl = ST::tab.slab[q];
    R[q][0] = (imgs[im] + (2 * sl) * 64)[lane_t];        // (scalar base + constant first: saddr addressing, one VGPR offset)
    R[q][1] = (imgs[im] + (2 * sl + 1) * 64)[lane_t];
  });
}

// ---- the activation operand of a product ----------------------------------------------------------------------------
struct BImg {        // a two-piece image: row = this lane's sample row + 8 (lane >> 5)
  const _Float16 *row;
  struct Raw { u32x4 a, b; };
  __device__ __forceinline__ void fetch(int s, Raw &r) const {
    r.a = *reinterpret_cast<const u32x4 *>(row + 16 * s);
    r.b = *reinterpret_cast<const u32x4 *>(row + 16 * s + P2H);
  }
  __device__ __forceinline__ void conv(const Raw &r, f16x8 &b1, f16x8 &b2) const {
    b1 = __builtin_bit_cast(f16x8, r.a);
    b2 = __builtin_bit_cast(f16x8, r.b);
  }
};
struct BSplit {      // an fp32 row image, split as it is read: row = this lane's sample row + first column + 8 (lane >> 5)
  const float *row;
  float t;
  struct Raw { f32x4 lo, hi; };
  __device__ __forceinline__ void fetch(int s, Raw &r) const {
    r.lo = *reinterpret_cast<const f32x4 *>(row + 16 * s);
    r.hi = *reinterpret_cast<const f32x4 *>(row + 16 * s + 4);
  }
  __device__ __forceinline__ void conv(const Raw &r, f16x8 &b1, f16x8 &b2) const { split8(r.lo, r.hi, t, b1, b2); }
};

// acc[n][b] += sum_k A[n][k] X[b][k] (lifted): NS slabs of the matrix at stream position POS (in the ring), X from prov.
// Every ring slot is refilled, as soon as its MFMAs are issued, with the slab RW positions further down the stream; TAIL:
// the tile's last product leaves the refills that belong to the next tile to ring_fill (they would be live across the
// weight gradients).
template <class ST, int POS, int NS, bool TAIL, class PROV>
__device__ __forceinline__ void gemm_r(f32x16 &acc, Ring &R, const u32x4 *(&imgs)[I_COUNT], const PROV &prov, int lane_t) {
  typename PROV::Raw raw[2];
  prov.fetch(0, raw[0]);
  static_for<NS>([&](auto ic) {
    constexpr int s = decltype(ic)::value;
    if constexpr (s + 1 < NS) prov.fetch(s + 1, raw[(s + 1) & 1]);
    f16x8 b1, b2;
    prov.conv(raw[s & 1], b1, b2);
    constexpr int slot = (POS + s) % RW;
    mm3(acc, __builtin_bit_cast(f16x8, R[slot][0]), __builtin_bit_cast(f16x8, R[slot][1]), b1, b2);
    if constexpr (!(TAIL && POS + s + RW >= ST::tab.L)) {
      constexpr int q = (POS + s + RW) % ST::tab.L;
      constexpr int im = ST::tab.img[q], sl = ST::tab.slab[q];
      R[slot][0] = (imgs[im] + (2 * sl) * 64)[lane_t];
      R[slot][1] = (imgs[im] + (2 * sl + 1) * 64)[lane_t];
    }
    __builtin_amdgcn_sched_barrier(0);   // left alone, the scheduler lifts every slab's reads and splits to the top
  });
}

// ---- weight gradients: K = the tile's 32 samples ----------------------------------------------------------------------
// the two-piece fragment of eight samples b = b0 + e (b0 = 8 (lane >> 5) + ks) of column c0 + (lane & 31), from a two-piece
// [sample][unit] image: the hardware's transposed read (ds_read_b64_tr_b16: a 16-lane group reads a block of 4 rows x 16
// columns of halves, lane 4 q + p supplies the address of row q, columns 4 p .. 4 p + 3, and lane i receives column i of the
// 4 rows) delivers four samples of one unit per read -- 4 reads per fragment pair instead of 16 two-byte gathers and the
// ~12 permutes that packed them.  EXEC must be all ones (the gather crosses lanes): every call site is workgroup-uniform.
typedef __fp16 h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ void frag_T(const _Float16 *img, int ks, int c0, f16x8 &p1, f16x8 &p2, int lane) {
#ifdef PI_FRAG_GATHER      // diagnostic: round 2's two-byte gathers
  const _Float16 *q = img + (8 * (lane >> 5) + ks) * RSH + c0 + (lane & 31);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    p1[e] = q[e * RSH];
    p2[e] = q[e * RSH + P2H];
  }
  return;
#endif
  typedef __attribute__((address_space(3))) h16x4 *lds_h4;
  const int t = lane & 15, g = lane >> 4;
  const _Float16 *a = img + (8 * (g >> 1) + ks + (t >> 2)) * RSH + c0 + 16 * (g & 1) + 4 * (t & 3);
  const h16x4 lo1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)a);
  const h16x4 hi1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(a + 4 * RSH));
  const h16x4 lo2 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(a + P2H));
  const h16x4 hi2 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(a + 4 * RSH + P2H));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 l1 = __builtin_bit_cast(u32x2, lo1), h1 = __builtin_bit_cast(u32x2, hi1);
  const u32x2 l2 = __builtin_bit_cast(u32x2, lo2), h2 = __builtin_bit_cast(u32x2, hi2);
  const u32x4 q1 = {l1[0], l1[1], h1[0], h1[1]}, q2 = {l2[0], l2[1], h2[0], h2[1]};
  p1 = __builtin_bit_cast(f16x8, q1);
  p2 = __builtin_bit_cast(f16x8, q2);
}
// ... and from an fp32 row image, split as it is read
__device__ __forceinline__ void frag_Ts(const float *img, int stride, int b0, int c, float t, f16x8 &p1, f16x8 &p2) {
  const float *q = img + b0 * stride + c;
  f32x4 lo, hi;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    lo[e] = q[e * stride];
    hi[e] = q[(e + 4) * stride];
  }
  split8(lo, hi, t, p1, p2);
}
// sum of the 16 lifted values of a fragment pair (fp32 accumulation of exact products by one)
__device__ __forceinline__ float frag_sum(const f16x8 (&b1)[2], const f16x8 (&b2)[2]) {
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  const f16x2 one = {(_Float16)1.0f, (_Float16)1.0f};
  float cs = 0.0f;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f16x2 u = {b2[s][2 * e], b2[s][2 * e + 1]}, w = {b1[s][2 * e], b1[s][2 * e + 1]};
      cs = __builtin_amdgcn_fdot2(u, one, cs, false);
      cs = __builtin_amdgcn_fdot2(w, one, cs, false);
    }
  return cs;
}

// g[J][i][j] += sum_b X[b][i0 + i] Y[b][32 J + j], J = 0 .. 3; X, Y two-piece images;
// colsum += sum_b Y[b][32 jsum + j] (the two lane halves hold the two halves of the samples)
__device__ __forceinline__ void wgrad_h(f32x16 (&g)[4], const _Float16 *X, int i0, const _Float16 *Y, float unscale, float y_unscale,
                                        int jsum, float &colsum, int lane) {
  f16x8 a1[2], a2[2];
  frag_T(X, 0, i0, a1[0], a2[0], lane);
  frag_T(X, 16, i0, a1[1], a2[1], lane);
#pragma unroll
  for (int J = 0; J < 4; ++J) {
    f16x8 b1[2], b2[2];
    frag_T(Y, 0, 32 * J, b1[0], b2[0], lane);
    frag_T(Y, 16, 32 * J, b1[1], b2[1], lane);
    f32x16 tmp;
    zero(tmp);
    mm3(tmp, a1[0], a2[0], b1[0], b2[0]);
    mm3(tmp, a1[1], a2[1], b1[1], b2[1]);
    if (J == jsum) colsum += frag_sum(b1, b2) * y_unscale;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[J][r] = __builtin_fmaf(tmp[r], unscale, g[J][r]);
  }
}

// g[i][j] += sum_b X[b][i0 + i] Y[b][j0 + j]; X, Y fp32 row images with lifts tx, ty; SUM: colsum += sum_b Y[b][j0 + j]
template <bool SUM>
__device__ __forceinline__ void wgrad_s(f32x16 &g, const float *X, int sx, int i0, float tx, const float *Y, int sy, int j0,
                                        float ty, float unscale, float y_unscale, float &colsum, int lane) {
  const int i = lane & 31, b0 = 8 * (lane >> 5);
  f16x8 a1[2], a2[2], b1[2], b2[2];
  frag_Ts(X, sx, b0, i0 + i, tx, a1[0], a2[0]);
  frag_Ts(Y, sy, b0, j0 + i, ty, b1[0], b2[0]);
  frag_Ts(X, sx, b0 + 16, i0 + i, tx, a1[1], a2[1]);
  frag_Ts(Y, sy, b0 + 16, j0 + i, ty, b1[1], b2[1]);
  f32x16 tmp;
  zero(tmp);
  mm3(tmp, a1[0], a2[0], b1[0], b2[0]);
  mm3(tmp, a1[1], a2[1], b1[1], b2[1]);
  if constexpr (SUM) colsum += frag_sum(b1, b2) * y_unscale;
#pragma unroll
  for (int r = 0; r < 16; ++r) g[r] = __builtin_fmaf(tmp[r], unscale, g[r]);
}

// ---- the kernel --------------------------------------------------------------------------------------------------------
template <int MODE, int N_IT, bool CACHED>
__global__ __launch_bounds__(kThreads, 2) void pi_kernel_h(const PiArgs p) {
  using ST = Stream<MODE, N_IT, CACHED>;
  constexpr int S0 = 2 * N_IT;          // slabs of the input layer (K = 32 N_IT >= D, zero padded)
  constexpr int XS = 32 * N_IT + 4;     // row stride of the x image
  // stream positions of the products (stream_tab)
  constexpr bool FWD = (MODE != MODE_FVP) || !CACHED;
  constexpr int P_WF0 = 0, P_WF1F = S0, P_J = FWD ? S0 + 8 : 0;
  constexpr int P_VF0 = P_J, P_VF1 = P_J + S0, P_WF1J = P_J + S0 + 8, P_VF2 = P_J + S0 + 16;
  constexpr int P_WF2 = (MODE == MODE_FVP) ? P_J + S0 + 18 : S0 + 8, P_WB2 = P_WF2 + 2, P_WB1 = P_WF2 + 4;
  static_assert(ST::tab.L == (MODE == MODE_EVAL ? P_WF2 + 2 : P_WB1 + 8), "stream positions");

  extern __shared__ f32x4 smem4[];
  const PiDims d = p.d;
  float *sm = reinterpret_cast<float *>(smem4);
  float *xR = sm;                       // [BB][XS] fp32
  float *h1R = xR + BB * XS;            // h1: two-piece image
  float *h2R = h1R + BB * RS;           // h2: fp32 rows
  float *u1R = h2R + BB * RS;           // dh1 (two-piece), then the split-K reduction image (4*32*33 floats), then delta1 (fp32)
  float *u2R = u1R + BB * RS;           // dh2 (fp32), then delta2 (two-piece)
  float *wR = u2R + BB * RS;            // [BB][36] cotangent on mu; GRAD / EVAL scratch before that
  float *red = u1R;
  float *d1R = u1R;
  _Float16 *h1H = reinterpret_cast<_Float16 *>(h1R), *u1H = reinterpret_cast<_Float16 *>(u1R),
           *u2H = reinterpret_cast<_Float16 *>(u2R);

  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);   // in an SGPR: every per-wave matrix pointer is scalar
  const int n_tiles = (p.n + BB - 1) / BB;

  // this wave's part of every matrix image
  const u32x4 *imgs[I_COUNT];
  imgs[I_WF0] = p.w.F0h + (size_t)wave * (S0 * 128);
  imgs[I_WF1] = p.w.F1h + (size_t)wave * (8 * 128);
  imgs[I_WF2] = p.w.F2h + (size_t)wave * (2 * 128);      // (K split over the waves: slabs 2 w, 2 w + 1)
  imgs[I_WB2] = p.w.B2h + (size_t)wave * (2 * 128);
  imgs[I_WB1] = p.w.B1h + (size_t)wave * (8 * 128);
  imgs[I_VF0] = p.v.F0h + (size_t)wave * (S0 * 128);
  imgs[I_VF1] = p.v.F1h + (size_t)wave * (8 * 128);
  imgs[I_VF2] = p.v.F2h + (size_t)wave * (2 * 128);

  // persistent accumulators
  f32x16 gW1[4], gW0[N_IT], gW2;
#pragma unroll
  for (int t = 0; t < 4; ++t) zero(gW1[t]);
#pragma unroll
  for (int t = 0; t < N_IT; ++t) zero(gW0[t]);
  zero(gW2);
  float gbias1 = 0.0f, gbias0 = 0.0f;   // d/d b1, d/d b0 of unit 32 wave + (lane & 31); the samples split over the two lane halves
  float gb2p[4] = {0, 0, 0, 0}, glsp[4] = {0, 0, 0, 0};   // per (a = tid/32 + 8*it) partials over this thread's b
  double s_n = 0, s_ra = 0, s_rc = 0, s_kl = 0, s_cost = 0;
  Ring R;
  ring_fill<ST>(R, imgs, tid0 & 63);
  __shared__ unsigned s_mx[2][4];    // per tile parity: largest |x|, |cotangent|, |dh2|, |delta1| (float bits)
  if (tid0 < 8) s_mx[tid0 >> 2][tid0 & 3] = 0u;
  float bnd_x = 0.0f, bnd_0 = 0.0f;
  if constexpr (MODE == MODE_FVP) {
    bnd_x = (float)d.D * p.v.mx[0];      // |dh1| <= max |x| D max |dW0| + max |db0|
    bnd_0 = p.v.mx[1];
  }
  const float bnd_c = (float)d.A * p.w.mx[2];   // |delta2| <= max |cot| A max |W2|

  // zero the padded cotangent columns (a in [A, 36)) and the padded input columns (k in [D, XS)) once: nothing else
  // ever writes them
  for (int i = tid0; i < BB * 36; i += kThreads) wR[i] = 0.0f;
  for (int i = tid0; i < BB * XS; i += kThreads) xR[i] = 0.0f;
  __syncthreads();

  // the tile's observations: fetched one tile ahead into registers (N_IT == 1), see pi_kernel
  constexpr bool PRE = (N_IT == 1);
  float xpre[4 * N_IT];
  auto fetch_x = [&](int t, int tid) {
    if constexpr (!PRE) return;
    const size_t base = (size_t)t * BB * d.D;
    const int lim = min(BB, p.n - t * BB) * d.D;      // rows past the batch end read as zeros
#pragma unroll
    for (int q = 0; q < 4 * N_IT; ++q) {
      const int i = tid + q * kThreads;
      xpre[q] = (i < lim) ? p.obs[base + i] : 0.0f;
    }
  };
  if ((int)blockIdx.x < n_tiles) fetch_x(blockIdx.x, tid0);
  __shared__ float c_b2[32], c_e2[32];
  if constexpr (MODE == MODE_FVP) {
    if (tid0 < 32) {
      c_b2[tid0] = (tid0 < d.A) ? p.v.b2[tid0] : 0.0f;
      c_e2[tid0] = (tid0 < d.A) ? 2.0f * expf(2.0f * p.w.ls[tid0]) : 0.0f;
    }
    __syncthreads();
  }

  // saved activations: a tile's block [h1 image | h2 rows] travels global -> LDS by LDS-DMA (one wave-instruction = 64 lanes x
  // 16 B = 1 KiB, destination = wave-uniform base + 16 lane), requested as soon as the previous tile has no reader of the
  // two images left (behind its dW2: only dW0 follows, on x and delta1) and drained by the barrier that ends the tile
  constexpr bool ACT_DMA = !FWD;
  auto fetch_act = [&](int t, int lane_) {
    const char *src = reinterpret_cast<const char *>(p.cache_r + (size_t)t * ACT_TILE4);
    char *dst = reinterpret_cast<char *>(h1R);
#pragma unroll
    for (int k = 0; k < (ACT_PIECES + 3) / 4; ++k) {
      const int pc = wave + 4 * k;
      if (pc < ACT_PIECES)      // (wave-uniform)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)pc * 1024 + lane_ * 16),
                                         (__attribute__((address_space(3))) void *)(dst + pc * 1024), 16, 0, 0);
    }
  };
  if constexpr (ACT_DMA) {
    if ((int)blockIdx.x < n_tiles) fetch_act(blockIdx.x, tid0 & 63);
  }
  // with the saved activations the only batch column a product reads is log_std_old: fetched one tile ahead like x, so that
  // the barrier that waits for the activations (vmcnt(0)) finds no younger load of this tile in front of it
  constexpr bool LSO_PRE = ACT_DMA && PI_DEFER_DMA;
  float lso_pre = 0.0f;
  auto fetch_lso = [&](int t, int tid) {
    const int er0 = t * BB + (tid & 31), a0 = tid >> 5;
    lso_pre = (er0 < p.n && a0 < d.A) ? p.ls_old[(size_t)er0 * d.A + a0] : 0.0f;
  };
  if constexpr (LSO_PRE) {
    if ((int)blockIdx.x < n_tiles) fetch_lso(blockIdx.x, tid0);
  }
#ifdef CMBPO_STAMPS
  unsigned long long t_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
  const unsigned long long t_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  int parity = 0;
  float warm = 0.0f;       // (the warm-up touch of the next tile's saved activations: consumed where vmcnt is drained anyway)
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, parity ^= 1) {
    const int row0 = tile * BB;
    // the thread index is opaque per tile, or every per-lane address of the loop (matrix slabs, bias / batch pointers, ...)
    // is hoisted out of it as a 64-bit VGPR pointer, and spilled
    int tid_o = threadIdx.x;
    asm volatile("" : "+v"(tid_o));
    const int tid = tid_o, lane = tid & 63, j = lane & 31, h = lane >> 5;
    unsigned *mx = s_mx[parity];
    // ---- stage x in the row layout ---------------------------------------------------------------------
    float xmax = 0.0f;
    if constexpr (PRE) {
#pragma unroll
      for (int q = 0; q < 4 * N_IT; ++q) {
        const int i = tid + q * kThreads;
        if (i < BB * d.D) {
          const int b = i / d.D, k = i - b * d.D;
          xR[b * XS + k] = xpre[q];
          xmax = fmaxf(xmax, fabsf(xpre[q]));
        }
      }
    } else {
      for (int i = tid; i < BB * d.D; i += kThreads) {
        const int b = i / d.D, k = i - b * d.D;
        const float xv = (row0 + b < p.n) ? p.obs[(size_t)row0 * d.D + i] : 0.0f;
        xR[b * XS + k] = xv;
        xmax = fmaxf(xmax, fabsf(xv));
      }
    }
    thread_max_put(xmax, &mx[0], lane);   // (visible behind the next barrier)
    if constexpr (MODE == MODE_EVAL) {   // (the other modes fetch ahead later, next to their weight gradients)
      if (tile + (int)gridDim.x < n_tiles) fetch_x(tile + gridDim.x, tid);
    }
    // per-sample batch columns: requested now, used in the element phase
    float lso0 = 0.0f;
    float e_act0 = 0.0f, e_mu0 = 0.0f, e_logp = 0.0f, e_adv = 0.0f, e_cadv = 0.0f;
    {
      const int er0 = row0 + (tid & 31), a0 = tid >> 5;
      if (er0 < p.n) {
        if constexpr (MODE != MODE_FVP) {
          e_logp = p.logp_old[er0];
          e_adv = p.adv[er0];
          e_cadv = p.cadv[er0];
        }
        if (a0 < d.A) {
          if constexpr (MODE != MODE_GRAD && !LSO_PRE) lso0 = p.ls_old[(size_t)er0 * d.A + a0];
          if constexpr (MODE != MODE_FVP) e_act0 = p.act[(size_t)er0 * d.A + a0];
          if constexpr (MODE == MODE_EVAL) e_mu0 = p.mu_old[(size_t)er0 * d.A + a0];
        }
      }
    }
    f32x16 acc;
    float t_x = 1.0f, it_x = 1.0f;
    if constexpr (!FWD) {
      // ---- h1, h2 as cmbpo_pi_loss_grad left them (same parameters, same batch): the LDS block itself, so the products
      // see identical bits; it was requested behind the previous tile's dW2 (before the loop for the first tile) and the
      // barrier that ended that tile waited for it
      if constexpr (PI_DEFER_DMA) {
        // ... and THIS barrier publishes the block: its LDS-DMA (counted in vmcnt) was requested behind the previous tile's
        // dW2 and has had the rest of that tile and this tile's staging to arrive (it used to be waited for by the barrier
        // that ends a tile, a few hundred cycles after the request: ~14 % of a tile at that barrier)
        lso0 = lso_pre;
        __syncthreads();
        asm volatile("" ::"v"(warm));   // the previous tile's warm-up load must be issued, its value is not used
      } else {
        lds_barrier();
      }
      t_x = pow2_lift(__uint_as_float(mx[0]));
      it_x = pow2_inv(t_x);
    } else {
      lds_barrier();
      t_x = pow2_lift(__uint_as_float(mx[0]));
      it_x = pow2_inv(t_x);
      // ---- forward ----------------------------------------------------------------------------------
      acc = load_bias(p.w.b0, wave * 32, lane);     // (the bias in lifted units: it is there when the MFMAs are)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] *= p.w.lift[L_F0] * t_x;
      gemm_r<ST, P_WF0, S0, false>(acc, R, imgs, BSplit{xR + j * XS + 8 * h, t_x}, lane);
      {
        const float un = p.w.lift[8 + L_F0] * it_x;
        f32x16 hv;
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = cmbpo_fast_tanh(acc[r] * un);
        store_tile_H(hv, wave * 32, h1H, T_TANH, lane);
      }
      lds_barrier();
      acc = load_bias(p.w.b1, wave * 32, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] *= p.w.lift[L_F1] * T_TANH;
      gemm_r<ST, P_WF1F, 8, false>(acc, R, imgs, BImg{h1H + j * RSH + 8 * h}, lane);
      {
        const float un = p.w.lift[8 + L_F1] * T_TANH_INV;
        f32x16 hv;
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = cmbpo_fast_tanh(acc[r] * un);
        store_tile_R(hv, wave * 32, h2R, RS, lane);
      }
      lds_barrier();
      if constexpr (MODE == MODE_GRAD) {
        if (p.cache_w != nullptr) {      // both images are complete behind the barrier: the block as it stands
          f32x4 *dst = p.cache_w + (size_t)tile * ACT_TILE4;
          const f32x4 *blk = reinterpret_cast<const f32x4 *>(h1R);
          for (int i = tid; i < ACT_TILE4; i += kThreads) dst[i] = blk[i];
        }
      }
    }

    PI_STAMP(0);
    if constexpr (MODE == MODE_FVP) {
      // ---- JVP chain: dh1 = (1-h1^2)(x dW0 + db0) ; dh2 = (1-h2^2)(dh1 W1 + h1 dW1 + db1) -----------
      const float t1 = pow2_lift(__builtin_fmaf(__uint_as_float(mx[0]), bnd_x, bnd_0)), it1 = pow2_inv(t1);
      acc = load_bias(p.v.b0, wave * 32, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] *= p.v.lift[L_F0] * t_x;
      gemm_r<ST, P_VF0, S0, false>(acc, R, imgs, BSplit{xR + j * XS + 8 * h, t_x}, lane);
      {
        const float un = p.v.lift[8 + L_F0] * it_x;
        const f32x16 hh = load_tile_H(h1H, wave * 32, lane);
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = __builtin_fmaf(-hh[r] * hh[r], T_TANH_INV * T_TANH_INV, 1.0f) * (acc[r] * un);
        store_tile_H(o, wave * 32, u1H, t1, lane);
      }
      PI_STAMP(1);
      // the h1 dW1 half does not need dh1: it runs ahead of the barrier and absorbs the waves' skew
      acc = load_bias(p.v.b1, wave * 32, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] *= p.v.lift[L_F1] * T_TANH;
      gemm_r<ST, P_VF1, 8, false>(acc, R, imgs, BImg{h1H + j * RSH + 8 * h}, lane);
      {
        // h1 dW1 + db1, moved into the units of the second product (both factors are powers of two)
        const float mv = (p.v.lift[8 + L_F1] * T_TANH_INV) * (p.w.lift[L_F1] * t1);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= mv;
      }
      PI_STAMP(2);
      lds_barrier();
      PI_STAMP(3);
      gemm_r<ST, P_WF1J, 8, false>(acc, R, imgs, BImg{u1H + j * RSH + 8 * h}, lane);
      {
        const float un = p.w.lift[8 + L_F1] * it1;
        const f32x16 hh = load_tile_R(h2R, RS, wave * 32, lane);
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = (1.0f - hh[r] * hh[r]) * (acc[r] * un);
        store_tile_R(o, wave * 32, u2R, RS, lane);
        tile_max_put(o, &mx[2], lane);
      }
      PI_STAMP(4);
      // dmu = dh2 W2 + h2 dW2 (+ db2): K split over the 4 waves; the h2 dW2 half ahead of the barrier
      zero(acc);
      gemm_r<ST, P_VF2, 2, false>(acc, R, imgs, BSplit{h2R + j * RS + 32 * wave + 8 * h, T_TANH}, lane);
      PI_STAMP(5);
      lds_barrier();   // dh2 complete; every wave is done reading dh1 (u1R becomes the reduction image)
      PI_STAMP(6);
      {
        const float t_d = pow2_lift(__uint_as_float(mx[2])), it_d = pow2_inv(t_d);
        const float mv = (p.v.lift[8 + L_F2] * T_TANH_INV) * (p.w.lift[L_F2] * t_d), un = p.w.lift[8 + L_F2] * it_d;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= mv;
        gemm_r<ST, P_WF2, 2, false>(acc, R, imgs, BSplit{u2R + j * RS + 32 * wave + 8 * h, t_d}, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= un;
      }
    } else {
      // mu = h2 W2 (+ b2): K split over the 4 waves
      zero(acc);
      gemm_r<ST, P_WF2, 2, MODE == MODE_EVAL>(acc, R, imgs, BSplit{h2R + j * RS + 32 * wave + 8 * h, T_TANH}, lane);
      const float un = p.w.lift[8 + L_F2] * T_TANH_INV;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] *= un;
      if constexpr (MODE == MODE_EVAL) ring_fill<ST>(R, imgs, lane);   // the next tile's first slabs
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int a = (r & 3) + 8 * (r >> 2) + 4 * h;
      red[(wave * 32 + a) * RED_LD + j] = acc[r];
    }
    lds_barrier();

    PI_STAMP(7);
    // ---- element phase over (a, b): this thread owns b = tid & 31, a = tid/32 + 8*it ------------------
    const int eb = tid & 31, er = row0 + eb;
    const bool valid = er < p.n;
    float z_[4], mu_[4];
    float cmax = 0.0f;      // largest |cotangent| this thread writes
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int a = (tid >> 5) + 8 * it;
      z_[it] = mu_[it] = 0.0f;
      if (a < d.A) {
        float m = red[(0 * 32 + a) * RED_LD + eb];
        m += red[(1 * 32 + a) * RED_LD + eb];
        m += red[(2 * 32 + a) * RED_LD + eb];
        m += red[(3 * 32 + a) * RED_LD + eb];
        if constexpr (MODE == MODE_FVP) {
          m += c_b2[a];
          float cot = 0.0f;
          if (valid) {
            // d2 KL / d mu^2 = 1 / (exp(2 ls_old) + eps)   (network/ac_network.py:52-53)
            const float lso = (it == 0) ? lso0 : p.ls_old[(size_t)er * d.A + a];
            const float v1 = expf(2.0f * lso) + 1e-8f;
            cot = m / v1;
            glsp[it] += c_e2[a] / v1;                           // d2 KL / d log_std^2 = 2 exp(2 ls) / (...)
          }
          wR[eb * 36 + a] = cot;
          gb2p[it] += cot;
          cmax = fmaxf(cmax, fabsf(cot));
        } else {
          m += p.w.b2[a];
          mu_[it] = m;
          float term = 0.0f;
          if (valid) {
            const float ls = p.w.ls[a];
            const float sd = expf(ls) + 1e-8f;
            const float z = (((it == 0) ? e_act0 : p.act[(size_t)er * d.A + a]) - m) / sd;
            z_[it] = z;
            term = -0.5f * (z * z + 2.0f * ls + 1.8378770664093453f);   // gaussian_likelihood :46-48
          }
          wR[eb * 36 + a] = term;   // scratch: per-(b, a) log-likelihood terms
        }
      }
    }
    if constexpr (MODE != MODE_FVP) {
      lds_barrier();
      float logp = 0.0f;
      for (int a = 0; a < d.A; ++a) logp += wR[eb * 36 + a];
      const float ratio = valid ? expf(logp - e_logp) : 0.0f;                // cpo_policy.py:522
      const float adv = e_adv, cadv = e_cadv;                                // (zero past the batch end)
      if (tid < 32 && valid) {
        s_n += 1.0;
        s_ra += (double)(ratio * adv);
        s_rc += (double)(ratio * cadv);
        s_cost += (double)p.cost[er];
      }
      lds_barrier();   // every thread has read the scratch terms before they are overwritten
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int a = (tid >> 5) + 8 * it;
        if (a < d.A) {
          if constexpr (MODE == MODE_EVAL) {
            if (valid) {
              // gaussian_kl(mu, log_std, mu_old, log_std_old), ac_network.py:50-55
              const float ls = p.w.ls[a], lso = (it == 0) ? lso0 : p.ls_old[(size_t)er * d.A + a];
              const float dm = ((it == 0) ? e_mu0 : p.mu_old[(size_t)er * d.A + a]) - mu_[it];
              const float pre = 0.5f * ((dm * dm + expf(2.0f * ls)) / (expf(2.0f * lso) + 1e-8f) - 1.0f) + lso - ls;
              s_kl += (double)pre;
            }
          } else {
            const float wgt = (p.which == 0) ? -adv : cadv;
            const float ls = p.w.ls[a];
            const float sd = expf(ls);
            const float inv = 1.0f / (sd + 1e-8f);
            const float cot = wgt * ratio * z_[it] * inv;                    // d logp / d mu = z / (sd + eps)
            wR[eb * 36 + a] = cot;
            gb2p[it] += cot;
            cmax = fmaxf(cmax, fabsf(cot));
            glsp[it] += wgt * ratio * (z_[it] * z_[it] * sd * inv - 1.0f);  // d logp / d log_std
          }
        }
      }
    }
    if constexpr (MODE != MODE_EVAL) thread_max_put(cmax, &mx[1], lane);
    if (tid < 4) s_mx[parity ^ 1][tid] = 0u;     // the other parity's maxima were last read a tile ago
    lds_barrier();
    PI_STAMP(8);
    if constexpr (MODE == MODE_EVAL) continue;

    // ---- backward: delta2 = (W2 cot) (1-h2^2) ; delta1 = (W1 delta2) (1-h1^2) --------------------------
    const float cm = __uint_as_float(mx[1]);
    const float t_c = pow2_lift(cm), it_c = pow2_inv(t_c);
    const float t2 = pow2_lift(cm * bnd_c), it2 = pow2_inv(t2);
    zero(acc);
    gemm_r<ST, P_WB2, 2, false>(acc, R, imgs, BSplit{wR + j * 36 + 8 * h, t_c}, lane);
    {
      const float un = p.w.lift[8 + L_B2] * it_c;
      const f32x16 hh = load_tile_R(h2R, RS, wave * 32, lane);
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = (1.0f - hh[r] * hh[r]) * (acc[r] * un);
      store_tile_H(o, wave * 32, u2H, t2, lane);     // dh2 is dead (the barrier after the reduction image)
    }
    lds_barrier();
    PI_STAMP(9);
    zero(acc);
    gemm_r<ST, P_WB1, 8, true>(acc, R, imgs, BImg{u2H + j * RSH + 8 * h}, lane);
    {
      const float un = p.w.lift[8 + L_B1] * it2;
      const f32x16 hh = load_tile_H(h1H, wave * 32, lane);
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = __builtin_fmaf(-hh[r] * hh[r], T_TANH_INV * T_TANH_INV, 1.0f) * (acc[r] * un);
      store_tile_R(o, wave * 32, d1R, RS, lane);     // the reduction image is dead (element phase barrier)
      tile_max_put(o, &mx[3], lane);
    }
    PI_STAMP(10);
    // ---- weight gradients: K = the tile's samples ---------------------------------------------------
    if (tile + (int)gridDim.x < n_tiles) fetch_x(tile + gridDim.x, tid);
    if constexpr (LSO_PRE) {
      if (tile + (int)gridDim.x < n_tiles) fetch_lso(tile + gridDim.x, tid);
    }
    if constexpr (MODE == MODE_FVP && CACHED) {
      // touch one dword of each 128-B line of the NEXT tile's saved activations: they travel HBM -> L2 behind the
      // MFMAs below, and the loads at the top of the next iteration hit L2
      const int nxt = tile + gridDim.x;
      if (nxt < n_tiles) warm = reinterpret_cast<const float *>(p.cache_r + (size_t)nxt * ACT_TILE4)[tid * 32];
    }
    // dW1 and dW2 need delta2 / the cotangent only (complete since the previous barrier): they run while the slower
    // waves still write delta1
    wgrad_h(gW1, h1H, wave * 32, u2H, T_TANH_INV * it2, it2, wave, gbias1, lane);
    PI_STAMP(11);
    {
      float unused = 0.0f;
      wgrad_s<false>(gW2, h2R, RS, wave * 32, T_TANH, wR, 36, 0, t_c, T_TANH_INV * it_c, 0.0f, unused, lane);
    }
    lds_barrier();   // delta1 complete; no wave reads h1 / h2 any more
    PI_STAMP(12);
    if constexpr (ACT_DMA) {
      if (tile + (int)gridDim.x < n_tiles) fetch_act(tile + gridDim.x, lane);
    }
    {
      const float t_d1 = pow2_lift(__uint_as_float(mx[3])), it_d1 = pow2_inv(t_d1);
      float unused = 0.0f;
      wgrad_s<true>(gW0[0], xR, XS, 0, t_x, d1R, RS, wave * 32, t_d1, it_x * it_d1, it_d1, gbias0, lane);
      if constexpr (N_IT == 2) wgrad_s<false>(gW0[N_IT - 1], xR, XS, 32, t_x, d1R, RS, wave * 32, t_d1, it_x * it_d1, 0.0f, unused, lane);
    }
    if constexpr (!(ACT_DMA && PI_DEFER_DMA)) asm volatile("" ::"v"(warm));   // the warm-up load must be issued, its value is not used
    ring_fill<ST>(R, imgs, lane);   // the next tile's first slabs: they land behind its staging
    if constexpr (ACT_DMA && PI_DEFER_DMA) lds_barrier();   // x / delta1 have no reader left; the activations are waited for later
    else __syncthreads();           // (waits for the LDS-DMA of the next tile's saved activations as well)
    PI_STAMP(13);
  }
#ifdef CMBPO_STAMPS
  if (p.stamps && threadIdx.x == 0) {
    for (int k = 0; k < 14; ++k) p.stamps[(size_t)blockIdx.x * 16 + k] = t_acc[k];
    p.stamps[(size_t)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memrealtime() - t_rt0;
  }
#endif

  // ---- flush: this workgroup's partial vector (plain stores; reduce_parts_kernel adds the partials in a fixed order)
  const int tid = tid0, lane = tid & 63, j = lane & 31, h = lane >> 5;
  if constexpr (MODE != MODE_EVAL) {
    float *part = p.part + (size_t)blockIdx.x * p.part_ld;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
      for (int J = 0; J < 4; ++J) part[d.oW1 + (wave * 32 + row) * HID + J * 32 + j] = gW1[J][r];
#pragma unroll
      for (int t = 0; t < N_IT; ++t)
        if (32 * t + row < d.D) part[d.oW0 + (32 * t + row) * HID + wave * 32 + j] = gW0[t][r];
      if (j < d.A) part[d.oW2 + (wave * 32 + row) * d.A + j] = gW2[r];
    }
    {
      const float g1 = gbias1 + __shfl_xor(gbias1, 32, 64), g0 = gbias0 + __shfl_xor(gbias0, 32, 64);
      if (h == 0) {
        part[d.ob1 + wave * 32 + j] = g1;
        part[d.ob0 + wave * 32 + j] = g0;
      }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int a = (tid >> 5) + 8 * it;
      const float sb = half_sum(gb2p[it]), sl = half_sum(glsp[it]);
      if (a < d.A && (tid & 31) == 0) {
        part[d.ob2 + a] = sb;
        if constexpr (MODE == MODE_GRAD) part[d.ols + a] = sl;
        else part[d.ols + a] = sl * p.v.ls[a];
      }
    }
  }
  if constexpr (MODE != MODE_FVP) {
    // tid < 32 hold the per-sample sums; s_kl is spread over every thread
    __shared__ double sd[4];
    const double kl = wave_sum_d(s_kl);
    if (lane == 0) sd[wave] = kl;
    __syncthreads();
    if (wave == 0) {
      const double n = wave_sum_d(s_n), ra = wave_sum_d(s_ra), rc = wave_sum_d(s_rc), c = wave_sum_d(s_cost);
      if (lane == 0) {
        atomicAdd(&p.sums[0], n);
        atomicAdd(&p.sums[1], ra);
        atomicAdd(&p.sums[2], rc);
        atomicAdd(&p.sums[3], sd[0] + sd[1] + sd[2] + sd[3]);
        atomicAdd(&p.sums[4], c);
      }
    }
  }
}
