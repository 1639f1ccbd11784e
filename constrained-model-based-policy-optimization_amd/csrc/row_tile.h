// Register-tile <-> LDS helpers for the fused 128-wide MLP kernels (policy_update.hip, ens_train.hip): 32 samples
// per tile on the MFMA columns, activation images in the row layout [sample][stride] (stride / 4 odd).
#pragma once
#include "common.h"

namespace {

constexpr int BB = 32;          // samples per tile

// ---- tile helpers -------------------------------------------------------------------------------------
__device__ __forceinline__ void zero(f32x16 &a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.0f;
}

// accumulator tile (rows n_base.., cols b) -> T-layout float4 [n/4][BB] and/or row layout [b][stride]
template <bool T, bool R>
__device__ __forceinline__ void store_tile(const f32x16 &v, int n_base, f32x4 *ldsT, float *ldsR, int strideR, int lane) {
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = n_base + 8 * q + 4 * h;
    f32x4 x;
#pragma unroll
    for (int s = 0; s < 4; ++s) x[s] = v[4 * q + s];
    if (T) ldsT[(n >> 2) * BB + j] = x;
    if (R) *reinterpret_cast<f32x4 *>(ldsR + j * strideR + n) = x;
  }
}

__device__ __forceinline__ f32x16 load_tile_T(const f32x4 *ldsT, int n_base, int lane) {
  const int j = lane & 31, h = lane >> 5;
  f32x16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 x = ldsT[((n_base + 8 * q + 4 * h) >> 2) * BB + j];
#pragma unroll
    for (int s = 0; s < 4; ++s) v[4 * q + s] = x[s];
  }
  return v;
}

__device__ __forceinline__ f32x16 load_bias(const float *b, int n_base, int lane) {
  const int h = lane >> 5;
  f32x16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 x = *reinterpret_cast<const f32x4 *>(b + n_base + 8 * q + 4 * h);
#pragma unroll
    for (int s = 0; s < 4; ++s) v[4 * q + s] = x[s];
  }
  return v;
}

// D[i][j] += sum_b X[b][i0 + i] * Y[b][j0 + j]   (K = the 32 samples of the tile)
__device__ __forceinline__ void wgrad_tile(f32x16 &acc, const float *X, int sx, int i0, const float *Y, int sy, int j0,
                                           int lane) {
  const int i = lane & 31, h = lane >> 5;
  const float *xp = X + h * sx + i0 + i, *yp = Y + h * sy + j0 + i;
#pragma unroll 4
  for (int s2 = 0; s2 < BB / 2; ++s2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xp[2 * s2 * sx], yp[2 * s2 * sy], acc, 0, 0, 0);
}

// sum over the 32 lanes that share h (the sample index) -> valid in lane j == 0 of each half
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// load the activation tile rows n_base.. of this wave's columns from a row-layout image
__device__ __forceinline__ f32x16 load_tile_R(const float *ldsR, int strideR, int n_base, int lane) {
  const int j = lane & 31, h = lane >> 5;
  f32x16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 x = *reinterpret_cast<const f32x4 *>(ldsR + j * strideR + n_base + 8 * q + 4 * h);
#pragma unroll
    for (int s = 0; s < 4; ++s) v[4 * q + s] = x[s];
  }
  return v;
}

__device__ __forceinline__ void store_tile_R(const f32x16 &v, int n_base, float *ldsR, int strideR, int lane) {
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 x;
#pragma unroll
    for (int s = 0; s < 4; ++s) x[s] = v[4 * q + s];
    *reinterpret_cast<f32x4 *>(ldsR + j * strideR + n_base + 8 * q + 4 * h) = x;
  }
}


}  // namespace
