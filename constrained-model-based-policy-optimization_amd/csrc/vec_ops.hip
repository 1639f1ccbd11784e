// Parameter-vector algebra of the CPO update kept on the device (P ~ 2e4 floats: one workgroup per op).
//   utilities/trust_region.py:32-45   cg(): x, r, p updates and the two dot products per iteration
//   policies/cpo_policy.py:210-266    q = v.Hv, r = w.Hv, s = w.Hw, b.b, step x = (v + nu w) / (lam + eps)
//   policies/cpo_policy.py:278        trial parameters old - step * x
// The reference runs these in NumPy on the host between sess.run calls; here they are single-workgroup
// kernels so a whole CG solve (10 x [pack direction, Fisher-vector product, all-reduce, cg step]) is enqueued
// without a host round trip.  Dot products accumulate in float64 and are rounded to float32 where the reference
// holds float32 values.
#include "common.h"

namespace {

constexpr int kT = 1024;

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-wide sum broadcast to every thread
__device__ __forceinline__ double block_sum_all(double v, double *sm) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  v = wsum(v);
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < kT / 64; ++i) t += sm[i];
  return t;
}

__global__ __launch_bounds__(kT) void cg_init_kernel(int P, const float *b, float *x, float *r, float *p, double *scal) {
  __shared__ double sm[kT / 64];
  double rr = 0.0;
  for (int i = threadIdx.x; i < P; i += kT) {
    const float bi = b[i];
    x[i] = 0.0f;
    r[i] = bi;
    p[i] = bi;
    rr += (double)bi * (double)bi;
  }
  rr = block_sum_all(rr, sm);
  if (threadIdx.x == 0) {
    scal[0] = (double)(float)rr;   // r_dot_old, a float32 value in the reference
    scal[1] = -1.0;                // no pending hand-over (cmbpo_pi_cg_iter)
  }
}

// z = hp_sum * inv_n + damping * p ; alpha = rr / (p.z + EPS) ; x += alpha p ; r -= alpha z ;
// rr_new = r.r ; p = r + (rr_new / rr) p          (utilities/trust_region.py:37-44)
__global__ __launch_bounds__(kT) void cg_step_kernel(int P, const float *hp_sum, float inv_n, float damping, float *x,
                                                     float *r, float *p, double *scal) {
  __shared__ double sm[kT / 64];
  double pz = 0.0;
  for (int i = threadIdx.x; i < P; i += kT) {
    const float z = hp_sum[i] * inv_n + damping * p[i];
    pz += (double)p[i] * (double)z;
  }
  pz = block_sum_all(pz, sm);
  const float rr_old = (float)scal[0];
  const float alpha = rr_old / ((float)pz + 1e-8f);
  double rr = 0.0;
  for (int i = threadIdx.x; i < P; i += kT) {
    const float pi = p[i];
    const float z = hp_sum[i] * inv_n + damping * pi;
    x[i] += alpha * pi;
    const float ri = r[i] - alpha * z;
    r[i] = ri;
    rr += (double)ri * (double)ri;
  }
  rr = block_sum_all(rr, sm);
  const float rr_new = (float)rr;
  const float beta = rr_new / rr_old;
  for (int i = threadIdx.x; i < P; i += kT) p[i] = r[i] + beta * p[i];
  if (threadIdx.x == 0) scal[0] = (double)rr_new;
}

__global__ __launch_bounds__(kT) void lincomb_kernel(int P, float a, const float *x, float b, const float *y, float *out) {
  for (int i = threadIdx.x + blockIdx.x * kT; i < P; i += gridDim.x * kT) {
    float v = a * x[i];
    if (y) v += b * y[i];
    out[i] = v;
  }
}

struct DotArgs {
  const float *x[8], *y[8];
};

__global__ __launch_bounds__(kT) void dots_kernel(int P, int n, const DotArgs a, double *out) {
  __shared__ double sm[kT / 64];
  for (int k = 0; k < n; ++k) {
    double s = 0.0;
    for (int i = threadIdx.x; i < P; i += kT) s += (double)a.x[k][i] * (double)a.y[k][i];
    s = block_sum_all(s, sm);
    if (threadIdx.x == 0) out[k] = s;
  }
}

}  // namespace

extern "C" int cmbpo_cg_init(int P, const float *d_b, float *d_x, float *d_r, float *d_p, double *d_scal, void *stream) {
  CMBPO_REQUIRE(P >= 1 && d_b && d_x && d_r && d_p && d_scal, "cmbpo_cg_init: bad argument");
  hipLaunchKernelGGL(cg_init_kernel, dim3(1), dim3(kT), 0, (hipStream_t)stream, P, d_b, d_x, d_r, d_p, d_scal);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

extern "C" int cmbpo_cg_step(int P, const float *d_hp_sum, double inv_n, float damping, float *d_x, float *d_r,
                             float *d_p, double *d_scal, void *stream) {
  CMBPO_REQUIRE(P >= 1 && d_hp_sum && d_x && d_r && d_p && d_scal, "cmbpo_cg_step: bad argument");
  hipLaunchKernelGGL(cg_step_kernel, dim3(1), dim3(kT), 0, (hipStream_t)stream, P, d_hp_sum, (float)inv_n, damping, d_x,
                     d_r, d_p, d_scal);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

extern "C" int cmbpo_vec_lincomb(int P, float a, const float *d_x, float b, const float *d_y, float *d_out, void *stream) {
  CMBPO_REQUIRE(P >= 1 && d_x && d_out, "cmbpo_vec_lincomb: bad argument");
  const int blocks = cmbpo_ceil_div(P, kT) < 64 ? cmbpo_ceil_div(P, kT) : 64;
  hipLaunchKernelGGL(lincomb_kernel, dim3(blocks), dim3(kT), 0, (hipStream_t)stream, P, a, d_x, b, d_y, d_out);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

extern "C" int cmbpo_vec_dots(int P, int n, const float *const *h_x, const float *const *h_y, double *d_out, void *stream) {
  CMBPO_REQUIRE(P >= 1 && n >= 1 && n <= 8 && h_x && h_y && d_out, "cmbpo_vec_dots: bad argument");
  DotArgs a;
  for (int k = 0; k < n; ++k) {
    CMBPO_REQUIRE(h_x[k] && h_y[k], "cmbpo_vec_dots: NULL vector %d", k);
    a.x[k] = h_x[k];
    a.y[k] = h_y[k];
  }
  hipLaunchKernelGGL(dots_kernel, dim3(1), dim3(kT), 0, (hipStream_t)stream, P, n, a, d_out);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}
