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
__device__ __forceinline__ void mfma_layer(const f32x4 *__restrict__ wp,
                                           size_t nt_stride, int g0, int g1,
                                           const f32x4 *lds_in, int lane,
                                           f32x16 (&acc)[NT][BT]) {
  constexpr int BB = 32 * BT;
  const int j = lane & 31, h = lane >> 5;
  f32x4 a_cur[NT], a_nxt[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) a_cur[t] = wp[t * nt_stride + (size_t)g0 * 64 + lane];
  for (int g = g0; g < g1; ++g) {
    const int gn = (g + 1 < g1) ? g + 1 : g;
#pragma unroll
    for (int t = 0; t < NT; ++t) a_nxt[t] = wp[t * nt_stride + (size_t)gn * 64 + lane];
    f32x4 b[BT];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) b[bt] = lds_in[(2 * g + h) * BB + bt * 32 + j];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
          acc[t][bt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t][s], b[bt][s],
                                                            acc[t][bt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) a_cur[t] = a_nxt[t];
  }
}

