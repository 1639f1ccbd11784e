// One imagined step of every alive branch as ONE call (samplers/model_sampler.py:239-375): policy forward, ensemble
// forward, FakeEnv post-processing, finish decisions, pre-store finishes, store, both critics at the next observations,
// post-store finishes -- the launch sequence the host mirror otherwise makes through nine separate calls.  At the
// rollout batch sizes of the shipped configs (1e3 - 1e4 branches) a step is bound by host latency, not by the kernels.
// Every buffer is a field of the rollout struct; nothing here adds arithmetic.
#include "common.h"
#include "ens_mlp_internal.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <atomic>
#include <mutex>

int cmbpo_internal_book_post_mirror(const cmbpo_rollout_t *r, int n_alive, uint32_t *d_host_out, uint32_t seq, int spec, int min_alive,
                                    double stop_total, void *stream);
int cmbpo_internal_scalars_mirror(const cmbpo_rollout_t *r, uint32_t *d_host_out, uint32_t seq, void *stream);
int cmbpo_internal_book_pre(const cmbpo_rollout_t *r, int n_alive, int spec, int with_vec, void *stream);
int cmbpo_internal_store_nostats(const cmbpo_rollout_t *r, void *stream);
int cmbpo_internal_finish_post_fold(const cmbpo_rollout_t *r, void *stream);
int cmbpo_internal_spec_words(const cmbpo_rollout_t *r, int begin, void *stream);

// A step enqueued AHEAD of the previous step's counters (cmbpo_rollout_run, small batches): n_alive is then an upper bound (the
// alive count only falls), every kernel reads the row count on the device, and the step is void if the previous one met a stop
// test (min_alive / stop_total, decided by its last kernel: book_post_kernel).
struct Ahead {
  bool on = false;
  int min_alive = 0;
  double stop_total = 0.0;
};

// policy_ready: this step's actions are already in act_t / logp_t / mu_t / ls_t (the previous step's critic launch carried
// the actor along).  d_eps_next != NULL: let this step's critic launch carry the actor for the NEXT step (its noise), if the
// three networks allow it; *next_ready says whether it did.
static int step_impl(const cmbpo_rollout_t *r, int n_alive, cmbpo_mlp_t *policy, cmbpo_mlp_t *model, cmbpo_mlp_t *v,
                     cmbpo_mlp_t *vc, int task, int ensemble, const float *d_eps, const int32_t *d_elite, float *d_mean,
                     float *d_var, bool policy_ready, const float *d_eps_next, bool *next_ready, uint32_t *d_mirror, uint32_t seq,
                     const Ahead &ahead, void *stream) {
  CMBPO_REQUIRE(r && policy && model && v && vc && d_eps && d_elite && d_mean && d_var, "cmbpo_rollout_step: NULL argument");
  CMBPO_REQUIRE(n_alive >= 1 && n_alive <= r->B, "cmbpo_rollout_step: n_alive %d not in [1, B=%d]", n_alive, r->B);
  // the per-step arrays are inputs of the bookkeeping kernels (const in the struct) and outputs of the forward passes
  auto w = [](const float *p) { return const_cast<float *>(p); };
  int rc;
  *next_ready = false;
  const int32_t *d_n = ahead.on ? r->iscal + CMBPO_I_N_EFF : nullptr;     // the forward kernels' row count (device) / n_alive (host)
  if (ahead.on)
    CMBPO_REQUIRE(n_alive <= cmbpo_rollout_book_pre_max_rows() && !r->use_host_budget && d_mirror && cmbpo_critic_pair_supported(v, vc),
                  "cmbpo_rollout_run: look-ahead needs the small-batch step");
  if (!policy_ready &&
      (rc = cmbpo_policy_forward(policy, r->cur_obs, r->obs_dim, d_eps, r->alive_idx, d_n, n_alive, w(r->act_t), w(r->logp_t),
                                 w(r->mu_t), w(r->ls_t), stream)))
    return rc;
  if ((rc = cmbpo_ens_forward(model, r->cur_obs, r->obs_dim, r->act_t, r->act_dim, r->alive_idx, d_n, n_alive, r->B,
                              d_mean, d_var, stream)))
    return rc;
  if ((rc = cmbpo_fakeenv_post(task, ensemble, r->obs_dim, r->act_dim, d_mean, d_var, r->B, r->cur_obs, r->act_t, d_elite,
                               r->alive_idx, d_n, n_alive, w(r->next_obs), w(r->rew_t), const_cast<uint8_t *>(r->term_t), w(r->cost_t),
                               w(r->dkl_t), w(r->epv_t), nullptr, stream)))
    return rc;
  const bool small = ahead.on || (n_alive <= cmbpo_rollout_book_pre_max_rows() && !r->use_host_budget);
  // small batches: decide + finish(PRE) + the store's scalar half in one launch; its vector half (obs, act, mu, log_std) rides
  // in the critics' launch below where that is the one-wave-per-member kernel (8 us as a launch of its own at 1000 rows)
  const bool vec_rides = small && cmbpo_critic_pair_supported(v, vc) && n_alive < cmbpo_internal_critic_big_min();
  if (small) {
    if ((rc = cmbpo_internal_book_pre(r, n_alive, ahead.on ? 1 : 0, vec_rides ? 0 : 1, stream))) return rc;
  } else {
    // large batches.  No branch can finish before the store unless the uncertainty test or a sample budget is on; the step's
    // sums are folded by the first workgroup of finish(POST) below instead of a launch of their own (nothing reads the
    // accumulators in between).  The budget exchange across shards reads them on the host: it keeps the separate launches.
    const bool fold_late = !r->use_host_budget;
    if ((rc = cmbpo_rollout_decide(r, stream))) return rc;
    if (r->uncertainty_mode || r->max_samples != 0 || r->use_host_budget)
      if ((rc = cmbpo_rollout_finish(r, 0, stream))) return rc;
    if ((rc = fold_late ? cmbpo_internal_store_nostats(r, stream) : cmbpo_rollout_store(r, stream))) return rc;
  }
  if (cmbpo_critic_pair_supported(v, vc)) {      // both critics in one launch (csrc/critic_f16.hip)
    // ... and, where it fits, the actor for the next step as one more wave per tile: both read next_obs, the store above has
    // consumed this step's actions, and nothing below reads them
    static const int ride_max = getenv("CMBPO_RIDE_MAX_ROWS") ? atoi(getenv("CMBPO_RIDE_MAX_ROWS")) : (1 << 30);
    const bool ride = d_eps_next != nullptr && n_alive <= ride_max && n_alive < cmbpo_internal_critic_big_min() &&
                      cmbpo_internal_critic_pair_can_ride(v, vc, policy);     // (large batches: the member-after-member kernel)
    CmbpoStoreVec sv{};
    if (vec_rides) {
      const float *src[4] = {r->cur_obs, r->act_t, r->mu_t, r->ls_t};
      float *dst[4] = {r->obs_buf, r->act_buf, r->mu_buf, r->ls_buf};
      const int dim[4] = {r->obs_dim, r->act_dim, r->act_dim, r->act_dim};
      for (int f = 0; f < 4; ++f) { sv.src[f] = src[f]; sv.dst[f] = dst[f]; sv.dim[f] = dim[f]; }
      sv.fin_code = r->fin_code;
      sv.col_off = (size_t)r->ptr * (size_t)r->B;
    }
    if ((rc = cmbpo_internal_critic_pair_ride(v, vc, r->next_obs, r->obs_dim, r->alive_idx, d_n, n_alive, w(r->v_n), w(r->vc_n),
                                              ride ? policy : nullptr, d_eps_next, w(r->act_t), w(r->logp_t), w(r->mu_t), w(r->ls_t),
                                              stream, vec_rides ? &sv : nullptr)))
      return rc;
    *next_ready = ride;
  } else {
    if ((rc = cmbpo_ens_predict_mean(v, r->next_obs, r->obs_dim, r->alive_idx, nullptr, n_alive, w(r->v_n), stream))) return rc;
    if ((rc = cmbpo_ens_predict_mean(vc, r->next_obs, r->obs_dim, r->alive_idx, nullptr, n_alive, w(r->vc_n), stream))) return rc;
  }
  if (small) {
    // finish(POST) + compaction in one launch (with the step's counters mirrored to the host, if asked)
    if ((rc = d_mirror ? cmbpo_internal_book_post_mirror(r, n_alive, d_mirror, seq, ahead.on ? 1 : 0, ahead.min_alive, ahead.stop_total,
                                                         stream)
                       : cmbpo_rollout_book_post(r, n_alive, stream)))
      return rc;
    return 1;                                                              // the alive list has been rebuilt: swap it
  }
  return r->use_host_budget ? cmbpo_rollout_finish(r, 1, stream) : cmbpo_internal_finish_post_fold(r, stream);
}

extern "C" int cmbpo_rollout_step(const cmbpo_rollout_t *r, int n_alive, cmbpo_mlp_t *policy, cmbpo_mlp_t *model,
                                  cmbpo_mlp_t *v, cmbpo_mlp_t *vc, int task, int ensemble, const float *d_eps,
                                  const int32_t *d_elite, float *d_mean, float *d_var, void *stream) {
  bool unused = false;
  return step_impl(r, n_alive, policy, model, v, vc, task, ensemble, d_eps, d_elite, d_mean, d_var, false, nullptr, &unused, nullptr, 0u,
                   Ahead{}, stream);
}

// Several steps in one call: what ModelSampler.sample()'s caller does between two steps (read the step's counters, swap the
// alive lists and the cur / next arrays, advance the column) without leaving native code -- at 1e3 branches a step is
// ~85 us of kernels and the interpreter between two steps was a quarter of it.  *r is updated in place exactly as the
// per-step caller would leave it.  Stops after max_steps, when no branch is alive, when at most min_alive are (the
// reference's `alive_ratio <= 0.1`, algorithms/cmbpo.py:358-359), when total_samples reaches stop_total (NaN: no such test; :356-357)
// or when the buffer is full.  d_eps / d_elite: the draws of the first step, one step further every eps_stride /
// elite_stride elements.  h_scalars: [max_steps][384 B] (pinned), the counters of every step taken.
// Host-mapped mirror of a step's counters (one 512-byte slot per step of a call): the single-workgroup bookkeeping kernel that
// ends a small-batch step writes them there and raises a sequence word; the host polls it.  NULL if the mapping cannot be had
// (the copy + stream synchronisation of cmbpo_rollout_read_scalars is used instead).
namespace {
constexpr int kMirrorSlots = 64, kMirrorDwords = 128, kMaxDevices = 64;
// one mirror per device (the device pointer of a mapping belongs to the device that was current when it was taken);
// the sequence word of a slot only has to differ from the slot's previous one: one process-wide atomic counter
struct Mirror {
  uint32_t *h = nullptr, *d = nullptr;
  bool tried = false;
};
Mirror g_mirror[kMaxDevices];
std::mutex g_mirror_mu;
std::atomic<uint32_t> g_seq{0};

uint32_t next_seq() {
  uint32_t s = g_seq.fetch_add(1, std::memory_order_relaxed) + 1;
  return s ? s : g_seq.fetch_add(1, std::memory_order_relaxed) + 1;      // never 0
}

Mirror &mirror() {
  static Mirror none;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return none;
  std::lock_guard<std::mutex> lock(g_mirror_mu);
  Mirror &m = g_mirror[dev];
  if (!m.tried) {
    m.tried = true;
    const char *e = getenv("CMBPO_STEP_MIRROR");
    if (e && e[0] == '0') return m;
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, (size_t)kMirrorSlots * kMirrorDwords * 4, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) == hipSuccess &&
        hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
      memset(h, 0, (size_t)kMirrorSlots * kMirrorDwords * 4);
      m.h = static_cast<uint32_t *>(h);
      m.d = static_cast<uint32_t *>(d);
    } else {
      (void)hipGetLastError();
      if (h) (void)hipHostFree(h);
    }
  }
  return m;
}

// wait until the kernel raised `seq` in the slot; after 2 s of wall clock fall back to a stream synchronisation
int wait_mirror(const uint32_t *slot, uint32_t seq, hipStream_t s) {
  const volatile uint32_t *flag = slot + 96;
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (;;) {
    for (int spin = 0; spin < 4096; ++spin) {
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return CMBPO_OK;
      __builtin_ia32_pause();
    }
    timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > 2.0) break;
  }
  CMBPO_HIP_CHECK(hipStreamSynchronize(s));
  if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return CMBPO_OK;
  cmbpo_set_error("cmbpo_rollout_run: the step's counters never arrived in the host mirror");
  return CMBPO_EHIP;
}
}  // namespace

extern "C" int cmbpo_rollout_run(cmbpo_rollout_t *r, int n_alive, cmbpo_mlp_t *policy, cmbpo_mlp_t *model, cmbpo_mlp_t *v,
                                 cmbpo_mlp_t *vc, int task, int ensemble, const float *d_eps, const int32_t *d_elite,
                                 long eps_stride, long elite_stride, float *d_mean, float *d_var, int max_steps,
                                 double stop_total, int min_alive, void *h_scalars, int *steps_done, int *n_alive_out,
                                 int *list_swaps, void *stream) {
  CMBPO_REQUIRE(r && h_scalars && steps_done && n_alive_out && list_swaps, "cmbpo_rollout_run: NULL argument");
  CMBPO_REQUIRE(max_steps >= 1, "cmbpo_rollout_run: max_steps %d", max_steps);
  CMBPO_REQUIRE(!r->use_host_budget, "cmbpo_rollout_run: the cross-shard budget exchange needs the per-step path");
  int done = 0, swaps = 0;
  bool ready = false;     // the actor for the step about to run was evaluated by the previous step's critic launch
  auto advance = [&](bool swap_lists) {     // what the per-step caller does between two steps
    if (swap_lists) {
      int32_t *t = r->alive_idx; r->alive_idx = r->alive_idx_out; r->alive_idx_out = t;
      ++swaps;
    }
    { const float *t = r->cur_obs; r->cur_obs = r->next_obs; r->next_obs = t; }
    { const float *t = r->v_t; r->v_t = r->v_n; r->v_n = t; }
    { const float *t = r->vc_t; r->vc_t = r->vc_n; r->vc_n = t; }
    r->ptr += 1;
  };
  static const bool ahead_env = !(getenv("CMBPO_STEP_AHEAD") && getenv("CMBPO_STEP_AHEAD")[0] == '0');
  while (done < max_steps && n_alive > 0 && r->ptr < r->T) {
    Mirror &mir = mirror();
    if (ahead_env && mir.d && n_alive <= cmbpo_rollout_book_pre_max_rows() && cmbpo_critic_pair_supported(v, vc)) {
      // ---- small batches, to the end of the call: step k + 1 is enqueued BEFORE the host waits for step k's counters, so the
      // GPU never idles across the host's round trip (13 of 93 us per step at 1000 branches).  Nothing the host does between
      // two steps needs the counters -- the lists are swapped after every small-batch step, the column advances by one -- except
      // the stop tests: book_post_kernel takes them on the device and raises the halt word that voids the step behind it; the
      // host reads the same word from the mirrored block, so both sides always agree on the last step.
      Ahead ah;
      ah.on = true; ah.min_alive = min_alive; ah.stop_total = stop_total;
      if (int rc = cmbpo_internal_spec_words(r, 1, stream)) return rc;
      struct Pend { int slot; uint32_t seq; };
      auto enqueue = [&](int k, Pend &p) -> int {       // step k of this call, from the struct as step k - 1 left it
        const float *eps_next = k + 1 < max_steps ? d_eps + (size_t)(k + 1) * eps_stride : nullptr;
        bool next_ready = false;
        p.slot = k % kMirrorSlots;
        p.seq = next_seq();
        const int rc = step_impl(r, n_alive, policy, model, v, vc, task, ensemble, d_eps + (size_t)k * eps_stride,
                                 d_elite + (size_t)k * elite_stride, d_mean, d_var, ready, eps_next, &next_ready,
                                 mir.d + (size_t)p.slot * kMirrorDwords, p.seq, ah, stream);
        ready = next_ready;
        if (rc == 0) cmbpo_set_error("cmbpo_rollout_run: look-ahead step took the large-batch path");
        return rc == 1 ? CMBPO_OK : (rc == 0 ? CMBPO_EINVAL : rc);
      };
      Pend cur{}, nxt{};
      if (int rc = enqueue(done, cur)) return rc;
      bool halted = false;
      for (;;) {
        advance(true);
        const bool more = done + 1 < max_steps && r->ptr < r->T;     // (the host's own limits; n_alive stays an upper bound)
        if (more)
          if (int rc = enqueue(done + 1, nxt)) return rc;
        const uint32_t *h_slot = mir.h + (size_t)cur.slot * kMirrorDwords;
        if (int rc2 = wait_mirror(h_slot, cur.seq, (hipStream_t)stream)) return rc2;
        char *h = static_cast<char *>(h_scalars) + (size_t)done * 384;
        memcpy(h, h_slot, 384);
        const int32_t *isc = reinterpret_cast<const int32_t *>(h);
        n_alive = isc[CMBPO_I_N_ALIVE];
        halted = isc[CMBPO_I_HALT] != 0;
        ++done;
        if (halted || !more) break;
        cur = nxt;
      }
      if (halted)       // (the void step behind the last one has run by then: stream order)
        if (int rc = cmbpo_internal_spec_words(r, 0, stream)) return rc;
      break;
    }
    const float *eps_next = done + 1 < max_steps ? d_eps + (size_t)(done + 1) * eps_stride : nullptr;
    bool next_ready = false;
    const int slot = done % kMirrorSlots;
    const uint32_t seq = next_seq();
    uint32_t *d_slot = mir.d ? mir.d + (size_t)slot * kMirrorDwords : nullptr;
    int rc = step_impl(r, n_alive, policy, model, v, vc, task, ensemble, d_eps + (size_t)done * eps_stride,
                       d_elite + (size_t)done * elite_stride, d_mean, d_var, ready, eps_next, &next_ready, d_slot, seq, Ahead{}, stream);
    ready = next_ready;
    if (rc != 0 && rc != 1) return rc;
    char *h = static_cast<char *>(h_scalars) + (size_t)done * 384;
    if (rc == 1 && d_slot != nullptr) {
      // small batch: the step's last kernel wrote the counters into the mirror -- wait for its sequence word
      const uint32_t *h_slot = mir.h + (size_t)slot * kMirrorDwords;
      if (int rc2 = wait_mirror(h_slot, seq, (hipStream_t)stream)) return rc2;
      memcpy(h, h_slot, 384);
    } else if (d_slot != nullptr) {
      const uint32_t *h_slot = mir.h + (size_t)slot * kMirrorDwords;
      if (int rc2 = cmbpo_internal_scalars_mirror(r, d_slot, seq, stream)) return rc2;
      // large batches (no rider in the critics' launch): the actor for the next step goes out BEHIND the counters and ahead of
      // the host's wait for them -- it needs neither (it evaluates every row this step stepped, outputs slot indexed, exactly
      // what the rider does), and the host's round trip hides behind it
      static const bool early_actor = !(getenv("CMBPO_EARLY_ACTOR") && getenv("CMBPO_EARLY_ACTOR")[0] == '0');
      if (early_actor && eps_next != nullptr && !ready && r->ptr + 1 < r->T) {
        auto w = [](const float *q) { return const_cast<float *>(q); };
        if (int rc2 = cmbpo_policy_forward(policy, r->next_obs, r->obs_dim, eps_next, r->alive_idx, nullptr, n_alive, w(r->act_t),
                                           w(r->logp_t), w(r->mu_t), w(r->ls_t), stream))
          return rc2;
        ready = true;
      }
      if (int rc2 = wait_mirror(h_slot, seq, (hipStream_t)stream)) return rc2;
      memcpy(h, h_slot, 384);
    } else if (int rc2 = cmbpo_rollout_read_scalars(r, h, stream)) {         // the host sync of the step
      return rc2;
    }
    const int32_t *isc = reinterpret_cast<const int32_t *>(h);
    const double *dsc = reinterpret_cast<const double *>(h + 128);
    const int fin = isc[CMBPO_I_N_FIN_PRE] + isc[CMBPO_I_N_FIN_POST];
    bool swap_lists = rc == 1;
    if (rc == 0 && fin > 0) {
      if (int rc2 = cmbpo_rollout_compact(r, stream)) return rc2;           // the alive list only changes when a branch finished
      swap_lists = true;
    }
    n_alive -= fin;
    advance(swap_lists);
    ++done;
    if (n_alive <= min_alive) break;
    if (stop_total == stop_total && dsc[CMBPO_D_TOTAL_SAMPLES] >= stop_total) break;     // NaN: no such test
  }
  *steps_done = done;
  *n_alive_out = n_alive;
  *list_swaps = swaps;
  return CMBPO_OK;
}
