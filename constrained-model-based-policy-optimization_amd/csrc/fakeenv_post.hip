// FakeEnv.step after the ensemble forward: uncertainty measures, elite pick,
// delta add, static termination / cost rules.  HBM-bound: reads the
// (E, B, out) mean / var once, writes O(obs) floats per branch.
//
// Follows, line for line in arithmetic order:
//   models/fake_env.py:104-151   (std, ep_var, dkl path, elite gather, delta, statics, reward)
//   models/pens/utils.py:15-57   (gaussian_kl_np, average_dkl: all E*E ordered pairs, clip [0, 1e10])
//   models/statics.py:3-53       (no_done, hcs_cost_f, antsafe_term_fn, antsafe_c_fn incl. the
//                                 `a*b*c*z_rot >= -0.7` precedence quirk)
#include "common.h"

#include <math.h>

namespace {

constexpr int kThreads = 256;
constexpr int kRows = 8;     // branches per workgroup
constexpr int kEMaxRt = 8;   // ensemble members held in registers (largest ensemble)
constexpr int kEMax = kEMaxRt;

struct PostArgs {
  int task, ensemble, obs_dim, act_dim, out_dim;
  const float *mean, *var;
  int ld_rows;
  const float *obs, *act;
  const int32_t *elite, *row_idx, *n_rows_dev;
  int n_rows;
  float *next_obs, *rew;
  uint8_t *term;
  float *cost, *dkl_path, *ep_var_mean, *ep_var;
};

// numpy's float32 pairwise sum for n < 128 (the contiguous-axis np.mean/np.sum path):
// 8 strided partial sums, tree-combined, then the tail sequentially.
__device__ float np_sum_f32(const float *a, int n) {
  if (n < 8) {
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s = __fadd_rn(s, a[i]);
    return s;
  }
  float r[8];
  for (int k = 0; k < 8; ++k) r[k] = a[k];
  int i = 8;
  for (; i < n - (n % 8); i += 8)
    for (int k = 0; k < 8; ++k) r[k] = __fadd_rn(r[k], a[i + k]);
  float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                        __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
  for (; i < n; ++i) res = __fadd_rn(res, a[i]);
  return res;
}

// np.clip(x, lo, hi): comparisons are false for a NaN, which therefore passes through (fminf / fmaxf would drop it)
__device__ __forceinline__ float clip_np(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// EC: the ensemble size at compile time (0: read it at run time) -- with a run-time size the member loops are unrolled to
// kEMax and masked: 56 pair terms computed and selected for the 42 that exist, a third more instructions
template <int EC>
__global__ __launch_bounds__(kThreads) void fakeenv_post_kernel(const PostArgs p) {
  extern __shared__ float sm[];
  const int D = p.obs_dim;
  float *s_dkl = sm;                  // [kRows][D]
  float *s_var = s_dkl + kRows * D;   // [kRows][D]
  float *s_next = s_var + kRows * D;  // [kRows][D]
  int *s_row = reinterpret_cast<int *>(s_next + kRows * D);  // [kRows]

  const int tid = threadIdx.x;
  const int n_rows = p.n_rows_dev ? *p.n_rows_dev : p.n_rows;
  const int row0 = blockIdx.x * kRows;
  if (row0 >= n_rows) return;
  if (tid < kRows) {
    const int r = row0 + tid;
    s_row[tid] = (r < n_rows) ? (p.row_idx ? p.row_idx[r] : r) : -1;
  }
  __syncthreads();

  const int E = EC > 0 ? EC : p.ensemble;
  constexpr int kEMax = EC > 0 ? EC : ::kEMaxRt;
  const size_t mstride = (size_t)p.ld_rows * p.out_dim;
  for (int i = tid; i < kRows * D; i += kThreads) {
    const int b = i / D, d = i - b * D;
    const int r = s_row[b];
    if (r < 0) continue;
    float mu[kEMax], ls[kEMax], vr[kEMax];
    const size_t base = (size_t)r * p.out_dim + d;
#pragma unroll
    for (int e = 0; e < kEMax; ++e) {
      if (e < E) {
        mu[e] = p.mean[e * mstride + base];
        // fake_env.py:104 std = sqrt(var); average_dkl: log_std = clip(log(std), -100, 1e8); gaussian_kl_np:
        // var = exp(2 log_std).  Inside the clip range that is log_std = 0.5 log(var) and exp(2 log_std) = var up to
        // rounding (the KL is compared at 1e-4, not bit for bit): one log instead of sqrt + log + exp per member.
        const float v0 = p.var[e * mstride + base];
        float l = __fmul_rn(0.5f, __logf(v0));              // (v_log_f32: the KL is compared at 1e-4; libm's logf is ~25 instructions)
        const bool inside = l >= -100.0f && l <= 1e8f;       // false for NaN as well
        if (!inside) l = clip_np(logf(sqrtf(v0)), -100.0f, 1e8f);   // np.clip: a NaN stays a NaN
        ls[e] = l;
        vr[e] = inside ? v0 : expf(__fmul_rn(2.0f, l));
      }
    }
    // ensemble epistemic variance over ALL members (np.var, axis 0), fake_env.py:112
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < kEMax; ++e)
      if (e < E) s = __fadd_rn(s, mu[e]);
    const float mbar = s / (float)E;
    float sq = 0.0f;
#pragma unroll
    for (int e = 0; e < kEMax; ++e)
      if (e < E) {
        const float dlt = __fsub_rn(mu[e], mbar);
        sq = __fadd_rn(sq, __fmul_rn(dlt, dlt));
      }
    s_var[i] = sq / (float)E;
    // average KL over all ordered pairs (i outer, j inner), models/pens/utils.py:49-56.  Two exact savings: an a == c term
    // is 0.5 (v / (v + 1e-10) - 1) <= 0 for finite v and mu, which np.clip turns into +0 -- adding +0 to the
    // non-negative running sum changes nothing, so those are skipped (for a member whose variance is inf / NaN or whose
    // mean is non-finite the term is NaN, np.clip keeps it, and the reference's KL of that branch is NaN: added below); and
    // (mu_c - mu_a)^2 == (mu_a - mu_c)^2 bit for bit, so the squared difference of a pair is computed once.
    float dm2[kEMax][kEMax];
#pragma unroll
    for (int a = 0; a < kEMax; ++a)
#pragma unroll
      for (int c = a + 1; c < kEMax; ++c)
        if (c < E) {
          const float dm = __fsub_rn(mu[c], mu[a]);
          dm2[a][c] = __fmul_rn(dm, dm);
        }
    // 1 / (var_c + 1e-10) once per member instead of a division per ordered pair (E - 1 times fewer divisions; the
    // quotient differs from the reference's by an ulp at most)
    float rden[kEMax];
#pragma unroll
    for (int c = 0; c < kEMax; ++c)
      if (c < E) rden[c] = __builtin_amdgcn_rcpf(__fadd_rn(vr[c], 1e-10f));   // v_rcp_f32: 1 ulp
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < kEMax; ++a) {
      if (a < E) {
#pragma unroll
        for (int c = 0; c < kEMax; ++c) {
          if (c < E && c != a) {
            const float d2 = (a < c) ? dm2[a][c] : dm2[c][a];
            const float num = __fadd_rn(d2, vr[a]);
            const float q = __fmul_rn(num, rden[c]);
            float pre = __fmul_rn(0.5f, __fsub_rn(q, 1.0f));
            pre = __fsub_rn(__fadd_rn(pre, ls[c]), ls[a]);
            pre = clip_np(pre, 0.0f, 1e10f);
            acc = __fadd_rn(acc, pre);
          }
        }
        // the a == a term of a member with a non-finite variance or mean: (0 or NaN + v) / (v + 1e-10) is inf / inf or NaN
        if (!isfinite(vr[a]) || !isfinite(mu[a])) acc = __fadd_rn(acc, __builtin_nanf(""));
      }
    }
    s_dkl[i] = acc / ((float)(E * (E - 1)) + 1e-10f);
    // elite pick + delta add, fake_env.py:121-131
    const int me = p.elite[r];
    const float nx = __fadd_rn(p.mean[me * mstride + base], p.obs[(size_t)r * D + d]);
    s_next[i] = nx;
    p.next_obs[(size_t)r * D + d] = nx;
    if (p.ep_var) p.ep_var[(size_t)r * D + d] = s_var[i];
  }
  __syncthreads();

  if (tid < kRows) {
    const int r = s_row[tid];
    if (r < 0) return;
    const float *nx = s_next + tid * D;
    p.dkl_path[r] = np_sum_f32(s_dkl + tid * D, D) / (float)D;      // fake_env.py:113
    p.ep_var_mean[r] = np_sum_f32(s_var + tid * D, D) / (float)D;   // model_sampler.py:322
    const int me = p.elite[r];
    p.rew[r] = p.mean[me * mstride + (size_t)r * p.out_dim + D];    // fake_env.py:148-151
    uint8_t done = 0;
    float cost = 0.0f;
    if (p.task == CMBPO_TASK_ANTSAFE) {
      // statics.py:17-53
      bool fin = true;
      for (int d = 0; d < D; ++d) fin = fin && isfinite(nx[d]);
      const float z = nx[0];
      const float q1 = nx[2], q2 = nx[3];
      const float zrot = __fsub_rn(1.0f, __fmul_rn(2.0f, __fadd_rn(__fmul_rn(q1, q1), __fmul_rn(q2, q2))));
      const float gate = (fin && z >= 0.2f && z <= 1.0f) ? 1.0f : 0.0f;
      const bool notdone = __fmul_rn(gate, zrot) >= -0.7f;
      done = notdone ? 0 : 1;
      const float obj = (fabsf(nx[D - 1]) > 3.2f) ? 1.0f : 0.0f;
      cost = fminf(fmaxf((float)done + obj, 0.0f), 1.0f);
    } else if (p.task == CMBPO_TASK_HCS) {
      // statics.py:10-15
      const float xdist = __fmul_rn(nx[D - 1], 10.0f);
      cost = (fabsf(xdist) < 2.0f) ? 1.0f : 0.0f;
    }
    p.term[r] = done;
    p.cost[r] = cost;
  }
}

}  // namespace

extern "C" int cmbpo_fakeenv_post(int task, int ensemble, int obs_dim, int act_dim,
                                  const float *d_mean, const float *d_var, int ld_rows,
                                  const float *d_obs, const float *d_act, const int32_t *d_elite,
                                  const int32_t *d_row_idx, const int32_t *d_n_rows, int n_rows,
                                  float *d_next_obs, float *d_rew, uint8_t *d_term, float *d_cost,
                                  float *d_dkl_path, float *d_ep_var_mean, float *d_ep_var,
                                  void *stream) {
  CMBPO_REQUIRE(task >= CMBPO_TASK_DEFAULT && task <= CMBPO_TASK_ANTSAFE, "cmbpo_fakeenv_post: bad task %d", task);
  CMBPO_REQUIRE(ensemble >= 2 && ensemble <= kEMax, "cmbpo_fakeenv_post: ensemble %d not in [2, %d]", ensemble, kEMax);
  CMBPO_REQUIRE(obs_dim >= 1 && obs_dim <= 512 && act_dim >= 0, "cmbpo_fakeenv_post: bad dims");
  if (task == CMBPO_TASK_ANTSAFE)
    CMBPO_REQUIRE(obs_dim >= 5, "cmbpo_fakeenv_post: AntSafe rules need obs_dim >= 5");
  CMBPO_REQUIRE(d_mean && d_var && d_obs && d_elite && d_next_obs && d_rew && d_term && d_cost &&
                    d_dkl_path && d_ep_var_mean,
                "cmbpo_fakeenv_post: NULL buffer");
  CMBPO_REQUIRE(n_rows >= 0 && ld_rows >= n_rows, "cmbpo_fakeenv_post: n_rows %d / ld_rows %d", n_rows, ld_rows);
  if (n_rows == 0) return CMBPO_OK;
  PostArgs a{};
  a.task = task; a.ensemble = ensemble; a.obs_dim = obs_dim; a.act_dim = act_dim;
  a.out_dim = obs_dim + 1;  // delta-obs + reward (algorithms/cmbpo.py:121-123, m_learn_cost=False)
  a.mean = d_mean; a.var = d_var; a.ld_rows = ld_rows; a.obs = d_obs; a.act = d_act;
  a.elite = d_elite; a.row_idx = d_row_idx; a.n_rows_dev = d_n_rows; a.n_rows = n_rows;
  a.next_obs = d_next_obs; a.rew = d_rew; a.term = d_term; a.cost = d_cost;
  a.dkl_path = d_dkl_path; a.ep_var_mean = d_ep_var_mean; a.ep_var = d_ep_var;
  const size_t lds = (size_t)3 * kRows * obs_dim * sizeof(float) + kRows * sizeof(int);
  const dim3 grid(cmbpo_ceil_div(n_rows, kRows));
  if (ensemble == 7) hipLaunchKernelGGL(fakeenv_post_kernel<7>, grid, dim3(kThreads), lds, (hipStream_t)stream, a);   // the shipped configs
  else if (ensemble == 5) hipLaunchKernelGGL(fakeenv_post_kernel<5>, grid, dim3(kThreads), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(fakeenv_post_kernel<0>, grid, dim3(kThreads), lds, (hipStream_t)stream, a);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}
