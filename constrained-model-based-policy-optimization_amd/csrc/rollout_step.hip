// One imagined step of every alive branch as ONE call (samplers/model_sampler.py:239-375): policy forward, ensemble
// forward, FakeEnv post-processing, finish decisions, pre-store finishes, store, both critics at the next observations,
// post-store finishes -- the launch sequence the host mirror otherwise makes through nine separate calls.  At the
// rollout batch sizes of the shipped configs (1e3 - 1e4 branches) a step is bound by host latency, not by the kernels.
// Every buffer is a field of the rollout struct; nothing here adds arithmetic.
#include "common.h"

extern "C" int cmbpo_rollout_step(const cmbpo_rollout_t *r, int n_alive, cmbpo_mlp_t *policy, cmbpo_mlp_t *model,
                                  cmbpo_mlp_t *v, cmbpo_mlp_t *vc, int task, int ensemble, const float *d_eps,
                                  const int32_t *d_elite, float *d_mean, float *d_var, void *stream) {
  CMBPO_REQUIRE(r && policy && model && v && vc && d_eps && d_elite && d_mean && d_var, "cmbpo_rollout_step: NULL argument");
  CMBPO_REQUIRE(n_alive >= 1 && n_alive <= r->B, "cmbpo_rollout_step: n_alive %d not in [1, B=%d]", n_alive, r->B);
  // the per-step arrays are inputs of the bookkeeping kernels (const in the struct) and outputs of the forward passes
  auto w = [](const float *p) { return const_cast<float *>(p); };
  int rc;
  if ((rc = cmbpo_policy_forward(policy, r->cur_obs, r->obs_dim, d_eps, r->alive_idx, nullptr, n_alive, w(r->act_t), w(r->logp_t),
                                 w(r->mu_t), w(r->ls_t), stream)))
    return rc;
  if ((rc = cmbpo_ens_forward(model, r->cur_obs, r->obs_dim, r->act_t, r->act_dim, r->alive_idx, nullptr, n_alive, r->B,
                              d_mean, d_var, stream)))
    return rc;
  if ((rc = cmbpo_fakeenv_post(task, ensemble, r->obs_dim, r->act_dim, d_mean, d_var, r->B, r->cur_obs, r->act_t, d_elite,
                               r->alive_idx, nullptr, n_alive, w(r->next_obs), w(r->rew_t), const_cast<uint8_t *>(r->term_t), w(r->cost_t),
                               w(r->dkl_t), w(r->epv_t), nullptr, stream)))
    return rc;
  if (n_alive <= cmbpo_rollout_book_pre_max_rows() && !r->use_host_budget) {
    if ((rc = cmbpo_rollout_book_pre(r, n_alive, stream))) return rc;     // decide + finish(PRE) + store in one launch
  } else {
    if ((rc = cmbpo_rollout_decide(r, stream))) return rc;
    if ((rc = cmbpo_rollout_finish(r, 0, stream))) return rc;
    if ((rc = cmbpo_rollout_store(r, stream))) return rc;
  }
  if (cmbpo_critic_pair_supported(v, vc)) {      // both critics in one launch (csrc/critic_f16.hip)
    if ((rc = cmbpo_critic_pair_predict(v, vc, r->next_obs, r->obs_dim, r->alive_idx, nullptr, n_alive, w(r->v_n), w(r->vc_n), stream)))
      return rc;
  } else {
    if ((rc = cmbpo_ens_predict_mean(v, r->next_obs, r->obs_dim, r->alive_idx, nullptr, n_alive, w(r->v_n), stream))) return rc;
    if ((rc = cmbpo_ens_predict_mean(vc, r->next_obs, r->obs_dim, r->alive_idx, nullptr, n_alive, w(r->vc_n), stream))) return rc;
  }
  if (n_alive <= cmbpo_rollout_book_pre_max_rows() && !r->use_host_budget) {
    if ((rc = cmbpo_rollout_book_post(r, n_alive, stream))) return rc;    // finish(POST) + compaction in one launch
    return 1;                                                              // the alive list has been rebuilt: swap it
  }
  return cmbpo_rollout_finish(r, 1, stream);
}
