// Segmented reward + cost GAE over contiguous paths and the advantage normalisation of get():
//   buffers/cpobuffer.py:179-207   CPOBuffer.finish_path  (single real-env path)
//   buffers/cpobuffer.py:249-268   CPOBuffer.get: adv = (adv - mean) / (std + EPS), cadv -= mean
//   utilities/utils.py:184-188     discount_cumsum (lfilter, float64 state)
//   utilities/mpi_tools.py:71-92   mpi_statistics_scalar
// Same arithmetic as finish_kernel in rollout_state.hip, for flat [N] arrays cut into segments by an
// offsets array (one thread per segment: the recurrence is strictly serial per path).
#include "common.h"

#include <math.h>

namespace {

struct SegArgs {
  int n_paths;
  const int32_t *offs;
  const float *rew, *val, *cost, *cval, *last_val, *last_cval;
  const uint8_t *f64_mask;  // bit0: reward deltas in float64 (float64 bootstrap), bit1: cost deltas
  double gamma, lam, cgamma, clam;
  float *adv, *ret, *cadv, *cret;
};

__global__ void gae_segments_kernel(const SegArgs a) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.n_paths) return;
  const int lo = a.offs[p], hi = a.offs[p + 1];
  const uint8_t m = a.f64_mask ? a.f64_mask[p] : 0;
  const float g32 = (float)a.gamma, cg32 = (float)a.cgamma;
  const double gl = a.gamma * a.lam, cgl = a.cgamma * a.clam;
  float vnext = a.last_val[p], cvnext = a.last_cval[p];
  double y = 0.0, cy = 0.0;
  for (int t = hi - 1; t >= lo; --t) {
    const float rw = a.rew[t], v = a.val[t], c = a.cost[t], cv = a.cval[t];
    double d, cd;
    if (m & 1) d = __dsub_rn(__dadd_rn((double)rw, __dmul_rn(a.gamma, (double)vnext)), (double)v);
    else d = (double)__fsub_rn(__fadd_rn(rw, __fmul_rn(g32, vnext)), v);
    if (m & 2) cd = __dsub_rn(__dadd_rn((double)c, __dmul_rn(a.cgamma, (double)cvnext)), (double)cv);
    else cd = (double)__fsub_rn(__fadd_rn(c, __fmul_rn(cg32, cvnext)), cv);
    y = __dadd_rn(d, __dmul_rn(gl, y));
    cy = __dadd_rn(cd, __dmul_rn(cgl, cy));
    const float adv = (float)y, cadv = (float)cy;
    a.adv[t] = adv;
    a.ret[t] = __fadd_rn(adv, v);
    a.cadv[t] = cadv;
    a.cret[t] = __fadd_rn(cadv, cv);
    vnext = v;
    cvnext = cv;
  }
}

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// st: [0] n [1] adv_mean [2] adv_std [3] cadv_mean | raw [8] sum adv [9] sum cadv [10] sum (adv-mean)^2
__global__ __launch_bounds__(256) void norm_moments(int n, const float *adv, const float *cadv, int pass, double *st) {
  __shared__ double sm[8];
  double s0 = 0, s1 = 0;
  const float mean = (float)st[1];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (pass == 0) { s0 += adv[i]; s1 += cadv[i]; }
    else { const float d = __fsub_rn(adv[i], mean); s0 += (double)__fmul_rn(d, d); }
  }
  s0 = wsum(s0); s1 = wsum(s1);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sm[w] = s0; sm[4 + w] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double a0 = sm[0] + sm[1] + sm[2] + sm[3], a1 = sm[4] + sm[5] + sm[6] + sm[7];
    if (pass == 0) { atomicAdd(&st[8], a0); atomicAdd(&st[9], a1); }
    else atomicAdd(&st[10], a0);
  }
}

__global__ void norm_finalize(int n, int pass, double *st) {
  if (pass == 0) { st[0] = n; st[1] = n > 0 ? st[8] / n : 0.0; st[3] = n > 0 ? st[9] / n : 0.0; }
  else st[2] = n > 0 ? sqrt(st[10] / n) : 0.0;
}

__global__ void norm_apply(int n, float *adv, float *cadv, const double *st) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float m = (float)st[1], den = (float)st[2] + 1e-8f, cm = (float)st[3];
  adv[i] = __fsub_rn(adv[i], m) / den;
  cadv[i] = __fsub_rn(cadv[i], cm);
}

}  // namespace

extern "C" int cmbpo_gae_segments(int n_paths, const int32_t *d_offsets, const float *d_rew, const float *d_val,
                                  const float *d_cost, const float *d_cval, const float *d_last_val,
                                  const float *d_last_cval, const uint8_t *d_f64_mask, double gamma, double lam,
                                  double cost_gamma, double cost_lam, float *d_adv, float *d_ret, float *d_cadv,
                                  float *d_cret, void *stream) {
  CMBPO_REQUIRE(n_paths >= 0, "cmbpo_gae_segments: n_paths %d < 0", n_paths);
  if (n_paths == 0) return CMBPO_OK;
  CMBPO_REQUIRE(d_offsets && d_rew && d_val && d_cost && d_cval && d_last_val && d_last_cval && d_adv && d_ret &&
                    d_cadv && d_cret,
                "cmbpo_gae_segments: NULL buffer");
  SegArgs a{n_paths, d_offsets, d_rew, d_val, d_cost, d_cval, d_last_val, d_last_cval, d_f64_mask,
            gamma, lam, cost_gamma, cost_lam, d_adv, d_ret, d_cadv, d_cret};
  hipLaunchKernelGGL(gae_segments_kernel, dim3(cmbpo_ceil_div(n_paths, 64)), dim3(64), 0, (hipStream_t)stream, a);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

extern "C" int cmbpo_adv_normalize(int n, float *d_adv, float *d_cadv, double *d_stats, void *stream) {
  CMBPO_REQUIRE(n >= 0 && d_stats, "cmbpo_adv_normalize: bad n / NULL stats");
  hipStream_t s = (hipStream_t)stream;
  CMBPO_HIP_CHECK(hipMemsetAsync(d_stats, 0, 16 * sizeof(double), s));
  if (n == 0) return CMBPO_OK;
  CMBPO_REQUIRE(d_adv && d_cadv, "cmbpo_adv_normalize: NULL buffer");
  const int blocks = cmbpo_ceil_div(n, 256) < 512 ? cmbpo_ceil_div(n, 256) : 512;
  hipLaunchKernelGGL(norm_moments, dim3(blocks), dim3(256), 0, s, n, d_adv, d_cadv, 0, d_stats);
  hipLaunchKernelGGL(norm_finalize, dim3(1), dim3(1), 0, s, n, 0, d_stats);
  hipLaunchKernelGGL(norm_moments, dim3(blocks), dim3(256), 0, s, n, d_adv, d_cadv, 1, d_stats);
  hipLaunchKernelGGL(norm_finalize, dim3(1), dim3(1), 0, s, n, 1, d_stats);
  hipLaunchKernelGGL(norm_apply, dim3(cmbpo_ceil_div(n, 256)), dim3(256), 0, s, n, d_adv, d_cadv, d_stats);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}
