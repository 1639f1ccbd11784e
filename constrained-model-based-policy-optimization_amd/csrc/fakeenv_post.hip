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

typedef float f32x2 __attribute__((ext_vector_type(2)));

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
#ifndef POST_WAVES
#define POST_WAVES 5      // waves per SIMD the register allocation aims at (swept 4 / 5 / 6 / 8: 57.7 / 51.1 / 51.3 / 77.4 us at 100 k rows)
#endif
template <int EC>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(POST_WAVES, POST_WAVES))) void fakeenv_post_kernel(const PostArgs p) {
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
    bool fast = EC > 0;     // every member's (mean, var) of this (row, dim) finite and far from the float range's ends
    const size_t base = (size_t)r * p.out_dim + d;
    // every load of the thread first -- 2 E + 2 requests in flight: written member by member next to their use, the loads of
    // a member waited for its own memory round trip (the out-of-range branch below keeps hipcc from hoisting them), seven
    // round trips in a row per workgroup
    float v0s[kEMax];
#pragma unroll
    for (int e = 0; e < kEMax; ++e) {
      mu[e] = p.mean[(e < E ? e : 0) * mstride + base];
      v0s[e] = p.var[(e < E ? e : 0) * mstride + base];
    }
    const int me = p.elite[r];
    const float mean_me = p.mean[me * mstride + base], obs_rd = p.obs[(size_t)r * D + d];
#pragma unroll
    for (int e = 0; e < kEMax; ++e) {
      if (e < E) {
        // fake_env.py:104 std = sqrt(var); average_dkl: log_std = clip(log(std), -100, 1e8); gaussian_kl_np:
        // var = exp(2 log_std).  Inside the clip range that is log_std = 0.5 log(var) and exp(2 log_std) = var up to
        // rounding (the KL is compared at 1e-4, not bit for bit): one log instead of sqrt + log + exp per member.
        const float v0 = v0s[e];
        float l = __fmul_rn(0.5f, __logf(v0));              // (v_log_f32: the KL is compared at 1e-4; libm's logf is ~25 instructions)
        const bool inside = l >= -100.0f && l <= 1e8f;       // false for NaN as well
        if (!inside) l = clip_np(logf(sqrtf(v0)), -100.0f, 1e8f);   // np.clip: a NaN stays a NaN
        ls[e] = l;
        vr[e] = inside ? v0 : expf(__fmul_rn(2.0f, l));
        fast = fast && inside && fabsf(mu[e]) < 1e18f && v0 < 1e30f;      // (false for a NaN mean as well)
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
    float acc;
    if (fast) {
      // The common case, on packed float32 arithmetic: two ordered pairs (a, c), (a, c + 1) per instruction, the pair term as
      //   clip(fma(fma(dm, dm, var_a), 0.5 / (var_c + 1e-10), log_std_c - 0.5) - log_std_a, 0, 1e10)
      // (the reference's sum, regrouped: the KL is compared at 1e-4, and its own pair term carries a cancellation error of a few
      // 1e-7 in `0.5 (q - 1) + log_std_c - log_std_a`), the clip as one v_med3 -- with every operand finite no NaN can arise, so
      // nothing has to pass through it -- and the a == c term left out: it is <= 0 before the clip (see below).  3.5 vector
      // instructions per ordered pair instead of 11.
      constexpr int EH = (kEMax + 1) / 2;
      f32x2 mu2[EH], hr2[EH], k2[EH];
#pragma unroll
      for (int h = 0; h < EH; ++h)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int c = 2 * h + u;
          const bool real = c < kEMax;
          mu2[h][u] = real ? mu[real ? c : 0] : 0.0f;
          hr2[h][u] = real ? 0.5f * __builtin_amdgcn_rcpf(vr[real ? c : 0] + 1e-10f) : 0.0f;
          k2[h][u] = real ? ls[real ? c : 0] - 0.5f : -1.0f;
        }
      f32x2 acc2 = {0.0f, 0.0f};
#pragma unroll
      for (int a = 0; a < kEMax; ++a) {
        const f32x2 nmu = {-mu[a], -mu[a]}, va = {vr[a], vr[a]}, nls = {-ls[a], -ls[a]};
#pragma unroll
        for (int h = 0; h < EH; ++h) {
          const f32x2 dm = mu2[h] + nmu;
          const f32x2 num = __builtin_elementwise_fma(dm, dm, va);
          const f32x2 pre = __builtin_elementwise_fma(num, hr2[h], k2[h]) + nls;
          f32x2 cl;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int c = 2 * h + u;
            cl[u] = (c < kEMax && c != a) ? __builtin_amdgcn_fmed3f(pre[u], 0.0f, 1e10f) : 0.0f;
          }
          acc2 += cl;
        }
      }
      acc = acc2[0] + acc2[1];
    } else {
      asm volatile("" ::: "memory");      // a real branch: the code below has no side effects, hipcc would run BOTH sides and select
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
      acc = 0.0f;
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
    }
    s_dkl[i] = acc / ((float)(E * (E - 1)) + 1e-10f);
    // elite pick + delta add, fake_env.py:121-131
    const float nx = __fadd_rn(mean_me, obs_rd);
    s_next[i] = nx;
    p.next_obs[(size_t)r * D + d] = nx;
    if (p.ep_var) p.ep_var[(size_t)r * D + d] = s_var[i];
  }
  __syncthreads();

#ifndef POST_TAIL_PAR
#define POST_TAIL_PAR 1      // diagnostic: 0 = one thread per row walks the row's sums alone (round 2's tail)
#endif
#if POST_TAIL_PAR
  // Eight lanes per row: numpy's pairwise sum IS eight strided partial sums -- lane k carries r[k] -- combined in a fixed tree
  // (np_sum_f32 above, operation for operation), so the row means keep their bits while the dependent chain of D additions
  // that one thread per row walked (twice, with 248 threads of the workgroup waiting) becomes D / 8 + 3.
  if (tid < 8 * kRows) {
    const int rw = tid >> 3, k = tid & 7;
    const int r = s_row[rw];
    const float *nx = s_next + rw * D;
    auto row_sum = [&](const float *a) {
      float res;
      if (D < 8) {
        res = 0.0f;
        for (int i = 0; i < D; ++i) res = __fadd_rn(res, a[i]);
        return res;
      }
      float rk = a[k];
      const int full = D - (D % 8);
      for (int i = 8; i < full; i += 8) rk = __fadd_rn(rk, a[i + k]);
      rk = __fadd_rn(rk, __shfl_down(rk, 1, 64));      // lanes 0, 2, 4, 6: r0 + r1, r2 + r3, r4 + r5, r6 + r7
      rk = __fadd_rn(rk, __shfl_down(rk, 2, 64));      // lanes 0, 4
      rk = __fadd_rn(rk, __shfl_down(rk, 4, 64));      // lane 0
      for (int i = full; i < D; ++i) rk = __fadd_rn(rk, a[i]);
      return rk;
    };
    const float sd = row_sum(s_dkl + rw * D), sv = row_sum(s_var + rw * D);
    bool fin = true;
    for (int d = k; d < D; d += 8) fin = fin && isfinite(nx[d]);
    const unsigned long long okm = __ballot(fin);
    fin = ((okm >> (8 * rw)) & 0xffull) == 0xffull;
    if (r < 0 || k != 0) return;
    p.dkl_path[r] = sd / (float)D;      // fake_env.py:113
    p.ep_var_mean[r] = sv / (float)D;   // model_sampler.py:322
#else
  if (tid < kRows) {
    const int r = s_row[tid];
    if (r < 0) return;
    const float *nx = s_next + tid * D;
    p.dkl_path[r] = np_sum_f32(s_dkl + tid * D, D) / (float)D;      // fake_env.py:113
    p.ep_var_mean[r] = np_sum_f32(s_var + tid * D, D) / (float)D;   // model_sampler.py:322
    bool fin = true;
    for (int d = 0; d < D; ++d) fin = fin && isfinite(nx[d]);
#endif
    const int me = p.elite[r];
    p.rew[r] = p.mean[me * mstride + (size_t)r * p.out_dim + D];    // fake_env.py:148-151
    uint8_t done = 0;
    float cost = 0.0f;
    if (p.task == CMBPO_TASK_ANTSAFE) {
      // statics.py:17-53
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
