// Shared MFMA tile helpers (fp32-exact v_mfma_f32_32x32x2_f32) for the gfx950 kernels.
//
// Conventions (see ens_mlp.hip header): D[i][j] += A[i][k] * B[k][j]; lane l = (j|i = l & 31, h = l >> 5).
//   * "packed" A operands live in global memory as [n-tile][k-group][lane][4]: lane (i, h) holds
//     A[n0 + i][8g + 4h + s], s = 0..3 -- one coalesced 16-B load per lane per 8 k;
//   * "T-layout" B operands live in LDS as float4 [k/4][BB]: element (k, b) at ((k >> 2) * BB + b) * 4 + (k & 3);
//   * accumulator tile: lane holds column j = l & 31, rows (r & 3) + 8 (r >> 2) + 4 h, r = 0..15.
#pragma once
#include "common.h"

// One dense layer slice: acc[t][bt] += A(n-tile t) * B(b-tile bt) over k-groups
// [g0, g1).  wp points at this wave's first n-tile for k-group 0; consecutive
// n-tiles are `nt_stride` float4 apart.  lds_in is the [K/4][BB] float4 image.
template <int NT, int BT>
__device__ __forceinline__ void mfma_block(const f32x4 (&a)[NT], const f32x4 (&b)[BT], f32x16 (&acc)[NT][BT]) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int bt = 0; bt < BT; ++bt)
        acc[t][bt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], b[bt][s], acc[t][bt], 0, 0, 0);
}

// ROWS == false: B operand in the T-layout (float4 [k/4][BB]).  ROWS == true: B operand in a row layout
// [b][row_stride] of floats (k contiguous per sample); with row_stride / 4 odd (132, 36, 52, ...) the per-lane
// 16-B reads of a wave fall on distinct banks, so the same image also serves the weight-gradient MFMAs.
template <int NT, int BT, bool ROWS = false, int DEPTH = 1>
__device__ __forceinline__ void mfma_layer(const f32x4 *__restrict__ wp,
                                           size_t nt_stride, int g0, int g1,
                                           const f32x4 *lds_in, int lane,
                                           f32x16 (&acc)[NT][BT], int row_stride = 0) {
  constexpr int BB = 32 * BT;
  const int j = lane & 31, h = lane >> 5;
  // Software pipeline, one k-group deep, ping-pong register sets (no copies): the A fragments (global / L2)
  // and B fragments (LDS) of group g + 1 are requested before the 4*NT*BT MFMAs of group g are issued and
  // are first needed a full MFMA block (>= 1024 cycles) later.  The sched_barrier after each request pins
  // it above the block: left alone, hipcc sinks the loads next to their first use and the L2 latency is
  // exposed once per k-group; everything else (address arithmetic, loop control) may interleave freely.
  f32x4 a0[NT], a1[NT], b0[BT], b1[BT];
  const f32x4 *lds_lane = lds_in + h * BB + j;
  const float *row_lane = reinterpret_cast<const float *>(lds_in) + j * row_stride + 4 * h;
  auto request = [&](int g, f32x4 (&a)[NT], f32x4 (&b)[BT]) {
    const f32x4 *wg = wp + (size_t)g * 64;          // wave-uniform: scalar pointer arithmetic
#pragma unroll
    for (int t = 0; t < NT; ++t) a[t] = (wg + t * nt_stride)[lane];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
      if constexpr (!ROWS) b[bt] = lds_lane[2 * g * BB + bt * 32];
      else b[bt] = *reinterpret_cast<const f32x4 *>(row_lane + bt * 32 * row_stride + 8 * g);
    }
  };
  if constexpr (DEPTH == 2) {
    // two k-groups in flight (three register sets): a lone wave per SIMD issues a block in 1024 cycles while an L2
    // hit under load takes ~1150, so one group of lead is not enough when the partner workgroup is not computing
    f32x4 a2[NT], b2[BT];
    const int last = g1 - 1;
    request(g0, a0, b0);
    request(g0 + 1 < g1 ? g0 + 1 : last, a1, b1);
    int g = g0;
#pragma unroll 1
    for (; g + 2 < g1; g += 3) {
      request(g + 2, a2, b2);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block<NT, BT>(a0, b0, acc);
      request(g + 3 < g1 ? g + 3 : last, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block<NT, BT>(a1, b1, acc);
      request(g + 4 < g1 ? g + 4 : last, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block<NT, BT>(a2, b2, acc);
    }
    if (g < g1) mfma_block<NT, BT>(a0, b0, acc);
    if (g + 1 < g1) mfma_block<NT, BT>(a1, b1, acc);
    return;
  }
  request(g0, a0, b0);
  int g = g0;
#pragma unroll 1
  for (; g + 1 < g1; g += 2) {
#ifdef CMBPO_DIAG_NOLOAD   // diagnostic: same MFMA stream, operands never refreshed
    if (g == g0) request(g + 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block<NT, BT>(a0, b0, acc);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block<NT, BT>(a1, b1, acc);
#else
    request(g + 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block<NT, BT>(a0, b0, acc);
    request((g + 2 < g1) ? g + 2 : g + 1, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block<NT, BT>(a1, b1, acc);
#endif
  }
  if (g < g1) mfma_block<NT, BT>(a0, b0, acc);
}
