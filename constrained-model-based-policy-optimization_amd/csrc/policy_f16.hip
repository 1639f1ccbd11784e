// The actor's forward pass of a rollout step (CPOPolicy.get_actions -> mlp_gaussian_policy, network/ac_network.py:99-123:
// two 128-wide tanh layers, mu = linear, pi = mu + exp(log_std) eps, logp_pi = gaussian_likelihood :46-48) with every
// float32 product as three f16 MFMAs (f16_split.h), on the pattern of critic_f16.hip: ONE wave carries a 32-row tile from
// the observation to the action without leaving its registers -- the input fragment is split in registers, an accumulator
// tile IS the next layer's B operand (the W1 / W2 images are packed in the matching k order), tanh outputs carry the fixed
// lift 2^14.  Four independent waves per workgroup; no barrier after the staging of the rows.
#include "common.h"
#include "ens_mlp_internal.h"
#include "f16_split.h"

#include "policy_f16_tile.h"

namespace {

template <int S0>
__global__ __launch_bounds__(256, 2) void policy_f16_kernel(const PfArgs a) {
  constexpr int KP = 16 * S0;
  extern __shared__ float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __shared__ int rows_s[4][32];
  const int n_rows = a.n_rows_dev ? *a.n_rows_dev : a.n_rows;
  const int row0 = (blockIdx.x * 4 + wave) * 32;
  if (row0 >= n_rows) return;                   // (waves are independent: no workgroup barrier)
  policy_tile<S0, false>(a, row0, n_rows, sm + wave * (32 * (KP + 1)), rows_s[wave], lane);
}

bool policy_eligible(const cmbpo_mlp *m) {
  return m->head == CMBPO_HEAD_GAUSS_PI && m->hidden == HP && m->act == CMBPO_ACT_TANH && m->ensemble == 1 && m->in_pad <= 64 &&
         m->out_dim <= 32 && m->o_tiles == 1 && !m->has_in_scaler && m->loaded;
}

// (re)builds the three f16 images and the statistics when the packs changed
int ensure_pf16(cmbpo_mlp *m, hipStream_t s) {
  const int S0 = m->h3_s0;
  if (m->d_h3 == nullptr) {
    m->h3_stride[0] = (size_t)NTP * S0 * 2 * 64;
    m->h3_stride[1] = (size_t)NTP * SP1 * 2 * 64;
    m->h3_stride[2] = (size_t)SP1 * 2 * 64;
    m->h3_off[0] = 0;
    m->h3_off[1] = m->h3_stride[0];
    m->h3_off[2] = m->h3_off[1] + m->h3_stride[1];
    m->h3_stats_off = m->h3_off[2] + m->h3_stride[2];
    const size_t bytes = m->h3_stats_off * 16 + (size_t)NSTAT * sizeof(float);
    if (hipMalloc(&m->d_h3, bytes) != hipSuccess) {
      (void)hipGetLastError();
      m->d_h3 = nullptr;
      cmbpo_set_error("policy_f16: hipMalloc of the f16 weight images failed");
      return CMBPO_ENOMEM;
    }
    m->h3_version = ~0ul;
  }
  if (m->h3_version == m->pack_version) return CMBPO_OK;
  float *stats = reinterpret_cast<float *>(reinterpret_cast<char *>(m->d_h3) + m->h3_stats_off * 16);
  cmbpo_internal_f16_stats(m, stats, s);
  f16x8 *base = reinterpret_cast<f16x8 *>(m->d_h3);
  cmbpo_internal_f16_pack(m, 0, base + m->h3_off[0], m->h3_stride[0], NTP, S0, 0, stats, s);
  cmbpo_internal_f16_pack(m, 1, base + m->h3_off[1], m->h3_stride[1], NTP, SP1, 1, stats, s);
  cmbpo_internal_f16_pack(m, 2, base + m->h3_off[2], m->h3_stride[2], 1, SP1, 1, stats, s);
  CMBPO_HIP_CHECK(hipGetLastError());
  m->h3_version = m->pack_version;
  return CMBPO_OK;
}

}  // namespace

bool cmbpo_internal_policy_f16_eligible(const cmbpo_mlp *m) { return policy_eligible(m); }

// the kernel arguments of the actor's images (built / refreshed on the way); *s0_out = k-slabs of its input layer
int cmbpo_internal_policy_f16_args(cmbpo_mlp *m, void *pf_args, int *s0_out, hipStream_t s) {
  const int s0 = (m->in_pad + 15) / 16;
  if (m->h3_s0 == 0) m->h3_s0 = s0 < 2 ? 2 : s0;
  if (int rc = ensure_pf16(m, s)) return rc;
  const float *blob = m->d_blob;
  const f16x8 *base = reinterpret_cast<const f16x8 *>(m->d_h3);
  PfArgs &a = *static_cast<PfArgs *>(pf_args);
  a.w0 = base + m->h3_off[0]; a.w1 = base + m->h3_off[1]; a.w2 = base + m->h3_off[2];
  a.b0 = blob + m->off_b0; a.b1 = blob + m->off_b1; a.b2 = blob + m->off_b2; a.log_std = blob + m->off_log_std;
  a.stats = reinterpret_cast<const float *>(reinterpret_cast<const char *>(m->d_h3) + m->h3_stats_off * 16);
  a.obs_dim = m->in_dim; a.act_dim = m->out_dim;
  *s0_out = m->h3_s0;
  return CMBPO_OK;
}

int cmbpo_internal_launch_policy_f16(cmbpo_mlp *m, const MlpKernelArgs &k, hipStream_t s) {
  PfArgs a{};
  int S0 = 0;
  if (int rc = cmbpo_internal_policy_f16_args(m, &a, &S0, s)) return rc;
  a.obs = k.obs; a.eps = k.eps; a.obs_dim = k.obs_dim;
  a.row_idx = k.row_idx; a.n_rows_dev = k.n_rows_dev; a.n_rows = k.n_rows;
  a.pi = k.out0; a.logp = k.out1; a.mu = k.out2; a.ls = k.out3;
  const size_t lds = (size_t)4 * 32 * (16 * S0 + 1) * sizeof(float);
  const int grid = cmbpo_ceil_div(k.n_rows, 128);
  if (S0 == 2) hipLaunchKernelGGL(policy_f16_kernel<2>, dim3(grid), dim3(256), lds, s, a);
  else if (S0 == 3) hipLaunchKernelGGL(policy_f16_kernel<3>, dim3(grid), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(policy_f16_kernel<4>, dim3(grid), dim3(256), lds, s, a);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}
