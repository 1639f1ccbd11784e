/*
 * cmbpo_hip.h -- C-ABI of the MI355X (gfx950) hot path for CMBPO.
 *
 * The reference (anyboby/Constrained-Model-Based-Policy-Optimization) is pure
 * Python on TF 1.14 + NumPy: it has no FFI.  The "binding" a maintainer adds is
 * therefore a ctypes stub; every entry point below names
 * the reference Python interface (file:line, relative to the reference tree)
 * whose arithmetic it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, a negative CMBPO_E* code otherwise;
 *     cmbpo_last_error() returns a static, human-readable string for the last
 *     failure on the calling thread.
 *   - all pointers named d_* are DEVICE pointers owned by the caller (the
 *     Python side allocates them as torch tensors); h_* are HOST pointers.
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on
 *     that stream, nothing in here synchronises or allocates per call.
 *   - handles own only packed weight copies; one host thread per handle.
 *   - float is IEEE fp32; masks are uint8_t (0/1); indices are int32_t.
 */
#ifndef CMBPO_HIP_H
#define CMBPO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMBPO_OK 0
#define CMBPO_EINVAL (-1)   /* bad argument / unsupported shape            */
#define CMBPO_EHIP (-2)     /* a HIP runtime call failed                   */
#define CMBPO_ENOMEM (-3)   /* allocation failed                           */
#define CMBPO_ESTATE (-4)   /* handle used before weights were loaded      */

/* activations (models/pens/fc.py:13-20, network/ac_network.py:26-33) */
#define CMBPO_ACT_SWISH 0
#define CMBPO_ACT_TANH 1

/* output heads */
#define CMBPO_HEAD_PROB 0     /* mean|logvar split, models/pens/pe.py:789-838 */
#define CMBPO_HEAD_DETMEAN 1  /* single head, mean over members, pe.py:338-343,648-669 */
#define CMBPO_HEAD_GAUSS_PI 2 /* tanh-MLP Gaussian policy, network/ac_network.py:99-123 */

/* static termination / cost rules (models/statics.py:56-69) */
#define CMBPO_TASK_DEFAULT 0  /* no_done, zero (bool) cost: Hopper/Humanoid */
#define CMBPO_TASK_HCS 1      /* HalfCheetahSafe-v2: no_done + hcs_cost_f    */
#define CMBPO_TASK_ANTSAFE 2  /* AntSafe-v2: antsafe_term_fn + antsafe_c_fn  */

const char *cmbpo_last_error(void);
int cmbpo_version(void);

/* Tuning knob (no reference counterpart): branches per workgroup of the 512-wide
 * ensemble kernel, 32 (two workgroups per CU) or 64 (one, half the L2 weight traffic). */
int cmbpo_set_block_rows(int rows);
/* Tuning knob: start-up stagger of the second dispatch batch of the 512-wide ensemble kernel, in units
 * of s_sleep(127) (~8k cycles); 0 disables it. */
int cmbpo_set_stagger(int sleeps);
/* Tuning knob: how the ensemble kernels walk their (member, row-tile) items: 0 = one workgroup per item
 * (hardware dispatch), 1 = persistent workgroups with static striding, 2 = persistent with a device work
 * counter. */
int cmbpo_set_dispatch_mode(int mode);
/* Matrix path of the 512-wide probabilistic ensemble forward (cmbpo_ens_forward / cmbpo_rollout_step): same
 * function, same float32 inputs / outputs, two ways to form the float32 products of its three GEMMs.
 * CMBPO_ENS_FP32: v_mfma_f32_32x32x2f32.  CMBPO_ENS_SPLIT_BF16 (default): every operand is split exactly into
 * three bf16 pieces (3 x 8 = 24 mantissa bits) and a.b = sum_{i+j<=4} a_i b_j runs as six
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- every partial product exact, dropped terms <= 2^-24 |ab|,
 * measured error 6.2e-7 of sum|a_k b_k| at K = 512 against 7.6e-7 for the fp32 MFMA chain
 * (tools/split_bf16_probe.hip) -- at about twice the matrix rate.  Both pass the same parity tests.  The
 * environment variable CMBPO_ENS_SPLIT=0/1/2 sets the initial value.  The switch also covers the critics
 * (cmbpo_ens_predict_mean at 128 hidden units, one output; paths 1 and 2 both mean the bf16 split there); every other
 * shape / head uses fp32 MFMAs.
 * CMBPO_ENS_SPLIT_F16 (default since round 2, csrc/ens_h3.hip): every operand is lifted by a power of two into the top
 * of the f16 range and split exactly into two f16 pieces (11 + 1 + 11 significant bits = a float32 rounding);
 * a.b = a2 b1 + a1 b2 + a1 b1 runs as three v_mfma_f32_32x32x16_f16 with fp32 accumulation -- measured error 3.2e-7 of
 * sum|a_k b_k| at K = 512 (tools/split_f16_probe.hip) -- with scales derived from norm bounds so that no finite input
 * can overflow a piece.  Rows of a call below cmbpo_set_ens_f16_min_rows take the bf16 path (smaller items). */
#define CMBPO_ENS_FP32 0
#define CMBPO_ENS_SPLIT_BF16 1
#define CMBPO_ENS_SPLIT_F16 2
int cmbpo_set_ens_matrix_path(int path);
int cmbpo_get_ens_matrix_path(void);
/* Tuning knob: calls of the 512-wide probabilistic forward with fewer rows than this use CMBPO_ENS_SPLIT_BF16 even when
 * CMBPO_ENS_SPLIT_F16 is selected (its 128-row items leave CUs idle at small rollout batches); 0 = never. */
int cmbpo_set_ens_f16_min_rows(int rows);
/* Tuning knob: 32-row tiles per item of the CMBPO_ENS_SPLIT_F16 kernel: 4 (128-row items, the throughput shape), 2 or 1
 * (the same kernel for rollout batches too small to give every CU a 128-row item); 0 (default) chooses by row count. */
int cmbpo_set_ens_f16_row_tiles(int rt);

/* ------------------------------------------------------------------------ *
 * Ensemble MLP handle: a 3-layer (in -> H -> H -> O) ensemble of E members.
 * Replaces the TF variables of models/pens/pe.py:150-213 (PE.finalize: layers
 * FC0..FC2 + TensorStandardScaler in/out) and, for HEAD_GAUSS_PI, the `pi`
 * scope of network/ac_network.py:99-123 (dense x3 + log_std).
 * ------------------------------------------------------------------------ */
typedef struct cmbpo_mlp cmbpo_mlp_t;

/* hidden must be 128, 256 or 512 (256: the general fp32-MFMA kernels only).  out_width is the network's last-layer width
 * (2*out_dim for HEAD_PROB, out_dim otherwise), <= 128. */
int cmbpo_mlp_create(cmbpo_mlp_t **out, int ensemble, int in_dim, int hidden,
                     int out_width, int activation, int head);
void cmbpo_mlp_destroy(cmbpo_mlp_t *m);

/* Weights in the reference's layout (models/pens/fc.py:135-166): W[E,in,out]
 * row-major, b[E,out] (the reference's [E,1,out]).  Scaler vectors are the
 * TensorStandardScaler mu/var ([1,dim], models/pens/utils.py:100-115); pass
 * NULL for "no scaler".  h_log_std (HEAD_GAUSS_PI only) is the [out] vector of
 * network/ac_network.py:104.  Packs on the host, copies on `stream`. */
int cmbpo_mlp_load(cmbpo_mlp_t *m, const float *h_w0, const float *h_b0,
                   const float *h_w1, const float *h_b1, const float *h_w2,
                   const float *h_b2, const float *h_in_mu,
                   const float *h_in_var, const float *h_out_mu,
                   const float *h_out_var, const float *h_log_std,
                   void *stream);

/* A Gaussian policy's parameters as ONE flat device vector in the order of
 * get_vars('pi') (network/ac_network.py:35-36, the order trust_region.py:21-25
 * assigns them in): W0[in,H] | b0 | W1[H,H] | b1 | W2[H,out] | b2 | log_std.
 * The trust-region update leaves the accepted parameters on the device
 * (policies/cpo_policy.py:278-300); this packs them into a handle that was
 * loaded once through cmbpo_mlp_load (one network, GAUSS_PI head, no scalers)
 * without a trip through the host. */
int cmbpo_mlp_load_policy_flat(cmbpo_mlp_t *m, const float *d_flat, void *stream);

/* PE.predict_ensemble, 2-D input path (models/pens/pe.py:688-697 ->
 * _compile_outputs(scale_output=True) :789-838 -> FC.compute_output_tensor
 * models/pens/fc.py:74-95 -> TensorStandardScaler models/pens/utils.py:156-187).
 * x = [obs | act] per row (the concat of models/fake_env.py:81); d_act may be
 * NULL with act_dim 0.  Row i of the call is branch d_row_idx[i] (or i when
 * d_row_idx is NULL); d_n_rows (device int, may be NULL) overrides n_rows so a
 * captured launch follows a device-side alive count.  Outputs are indexed
 * [member][branch slot][out_dim] with leading dimension ld_rows. */
int cmbpo_ens_forward(cmbpo_mlp_t *m, const float *d_obs, int obs_dim,
                      const float *d_act, int act_dim, const int32_t *d_row_idx,
                      const int32_t *d_n_rows, int n_rows, int ld_rows,
                      float *d_mean, float *d_var, void *stream);

/* PE.predict for the critics (models/pens/pe.py:648-669; mean over ALL members
 * :338-343), as used by CPOPolicy.get_v/get_vc (policies/cpo_policy.py:825-835).
 * d_out is [branch slot][out_dim]. */
int cmbpo_ens_predict_mean(cmbpo_mlp_t *m, const float *d_obs, int obs_dim,
                           const int32_t *d_row_idx, const int32_t *d_n_rows,
                           int n_rows, float *d_out, void *stream);

/* Both critics of a rollout step in one launch: CPOPolicy.get_v and get_vc (policies/cpo_policy.py:825-835) on the same
 * observation rows -- two loaded HEAD_DETMEAN ensembles of 128 hidden units, swish, one output, equal input width and
 * member count (<= 4).  Same function as two cmbpo_ens_predict_mean calls; float32 products run as three f16 MFMAs
 * (csrc/critic_f16.hip), one wave per (critic, member).  d_v / d_vc are [branch slot]. */
int cmbpo_critic_pair_supported(const cmbpo_mlp_t *v, const cmbpo_mlp_t *vc);
int cmbpo_critic_pair_predict(cmbpo_mlp_t *v, cmbpo_mlp_t *vc, const float *d_obs, int obs_dim,
                              const int32_t *d_row_idx, const int32_t *d_n_rows, int n_rows,
                              float *d_v, float *d_vc, void *stream);

/* mlp_gaussian_policy forward (network/ac_network.py:99-123) behind
 * CPOPolicy.get_action_outs (policies/cpo_policy.py:801-823): mu = MLP(obs),
 * pi = mu + eps*exp(log_std), logp_pi = gaussian_likelihood(pi, mu, log_std)
 * (:46-48).  eps is an input (the reference draws tf.random_normal, :109);
 * outputs indexed by branch slot: d_pi,d_mu,d_logstd [.,act_dim], d_logp [.]. */
int cmbpo_policy_forward(cmbpo_mlp_t *m, const float *d_obs, int obs_dim,
                         const float *d_eps, const int32_t *d_row_idx,
                         const int32_t *d_n_rows, int n_rows, float *d_pi,
                         float *d_logp, float *d_mu, float *d_logstd,
                         void *stream);

/* FakeEnv.step after the ensemble forward (models/fake_env.py:104-151):
 * std = sqrt(var); ens_ep_var = var_E(mean[..., :obs]); dkl_path = mean_d
 * average_dkl(mean, std) (models/pens/utils.py:15-57, all E members);
 * next_obs = mean[elite[b], b, :obs] + obs (delta model, deterministic=True);
 * r = mean[elite[b], b, obs]; term / cost from models/statics.py evaluated on
 * (obs, act, next_obs).  d_elite is the per-row member index (the draw of
 * models/fake_env.py:174-178, injected by the caller).  Row addressing as in
 * cmbpo_ens_forward.  d_cost is float (bool cast for the default task),
 * d_term is uint8.  d_ep_var_mean = mean over obs dims of ens_ep_var (what
 * samplers/model_sampler.py:322,343 consume); d_ep_var [.,obs_dim] optional. */
int cmbpo_fakeenv_post(int task, int ensemble, int obs_dim, int act_dim,
                       const float *d_mean, const float *d_var, int ld_rows,
                       const float *d_obs, const float *d_act,
                       const int32_t *d_elite, const int32_t *d_row_idx,
                       const int32_t *d_n_rows, int n_rows, float *d_next_obs,
                       float *d_rew, uint8_t *d_term, float *d_cost,
                       float *d_dkl_path, float *d_ep_var_mean,
                       float *d_ep_var, void *stream);

/* ------------------------------------------------------------------------ *
 * Device-resident rollout state: the fused counterpart of ModelSampler
 * (samplers/model_sampler.py:203-444) + ModelBuffer (buffers/modelbuffer.py).
 * Every array is owned by the caller; branch slots never move (mask-in-place
 * + an ordered alive list instead of the reference's per-step compaction,
 * samplers/model_sampler.py:300-311).  Buffers are TIME-MAJOR [T][B][...]
 * so that per-step stores and the per-branch GAE scan are both coalesced; the
 * reference's (B, T) order is restored by cmbpo_buffer_flatten (get()).
 * ------------------------------------------------------------------------ */
typedef struct cmbpo_rollout {
  int32_t B, T, obs_dim, act_dim;
  int32_t max_path_length;  /* sampler horizon (model_sampler.py:352)          */
  int32_t ptr;              /* column of the next store (modelbuffer.py:135)   */
  int32_t uncertainty_mode; /* rollout_mode == 'uncertainty' (:275-279)        */
  int32_t rank, world;      /* shard id / count (budget rule offsets)          */
  int64_t max_samples;      /* budget of sample(max_samples); 0: none (the reference tests `if max_samples:`, so a
                             * negative budget early-terminates every surviving branch, model_sampler.py:282-287) */
  double dkl_lim;           /* set_rollout_dkl (:169-170)                      */
  double gamma, lam, cost_gamma, cost_lam; /* modelbuffer.py:41-51             */
  /* ordered alive list (ascending slot ids) and scalars */
  int32_t *alive_idx;       /* [B] current list                                */
  int32_t *alive_idx_out;   /* [B] list written by cmbpo_rollout_compact       */
  int32_t *iscal;           /* [32] see CMBPO_I_*                              */
  double *dscal;            /* [32] see CMBPO_D_*                              */
  int32_t use_host_budget;  /* 1: take the two fields below instead of local counts */
  int32_t host_rank_off;    /* surviving rows on lower ranks (global index order)   */
  int64_t host_excess;      /* global rows to early-terminate this step              */
  uint8_t *alive;           /* [B] !terminated_paths_mask                      */
  uint8_t *fin_code;        /* [B] 0 keep, 1 finish w/ bootstrap, 2 terminal   */
  int32_t *len;             /* [B] populated entries of the branch             */
  /* per-step values, slot indexed */
  const float *cur_obs;     /* [B,obs]  observation the step starts from       */
  const float *next_obs;    /* [B,obs]                                         */
  const float *act_t, *logp_t, *mu_t, *ls_t; /* [B,act] / [B]                  */
  const float *v_t, *vc_t;  /* [B] critics at cur_obs                          */
  const float *v_n, *vc_n;  /* [B] critics at next_obs                         */
  const float *rew_t, *cost_t, *dkl_t, *epv_t; /* [B]                          */
  const uint8_t *term_t;    /* [B]                                             */
  double *dkl_acc, *path_ret, *path_cost, *path_dyn_var; /* [B] (float64 in the reference) */
  double *store_part;   /* [ceil(B / 64)][8] per-workgroup sampler sums of cmbpo_rollout_store (scratch) */
  /* time-major buffers */
  float *obs_buf, *act_buf, *mu_buf, *ls_buf;             /* [T,B,dim]         */
  float *rew_buf, *val_buf, *cost_buf, *cval_buf, *logp_buf; /* [T,B]          */
  float *adv_buf, *ret_buf, *cadv_buf, *cret_buf;         /* [T,B]             */
} cmbpo_rollout_t;

/* iscal slots */
#define CMBPO_I_N_ALIVE 0   /* length of alive_idx                              */
#define CMBPO_I_N_UNC 1     /* too-uncertain rows of this step (local)          */
#define CMBPO_I_N_FIN_PRE 2 /* rows finished before the store                   */
#define CMBPO_I_N_STORED 3  /* rows stored this step                            */
#define CMBPO_I_N_ALIVE_OUT 4 /* length of alive_idx_out after compact          */
#define CMBPO_I_SIZE 5      /* populated entries in the buffer (pool.size)      */
#define CMBPO_I_N_FIN_POST 6 /* rows finished after the store (horizon / terminal) */
/* (8 .. 11: the row other shards gather -- {n_alive, n_unc, total_samples, 0}) */
#define CMBPO_I_HALT 12     /* cmbpo_rollout_run's look-ahead: the step just taken met a stop test, the step enqueued behind it is void */
#define CMBPO_I_N_EFF 13    /* ... and the row count that step's forward kernels read (0 once halted)   */
/* dscal slots (sampler accumulators, model_sampler.py:314-333) */
#define CMBPO_D_TOTAL_SAMPLES 0
#define CMBPO_D_TOTAL_COST 1
#define CMBPO_D_TOTAL_REW 2
#define CMBPO_D_TOTAL_VS 3
#define CMBPO_D_TOTAL_CVS 4
#define CMBPO_D_TOTAL_DKL 5
#define CMBPO_D_TOTAL_DYN_EP_VAR 6
#define CMBPO_D_MAX_DKL 7
#define CMBPO_D_MAX_PATH_RETURN 8
#define CMBPO_D_DKL_SUM_T 9     /* sum of dkl_t over the rows stepped this call */
#define CMBPO_D_STEP_MAX_DKL 10
#define CMBPO_D_SUM_PATH_RET 11
#define CMBPO_D_SUM_PATH_COST 12

/* ModelSampler.reset + ModelBuffer.reset (model_sampler.py:203-237,
 * modelbuffer.py:53-98): all B branches alive, lists / accumulators zeroed.
 * Does not touch the big buffers (no realloc, no memset: entries are only
 * read below `len`). */
int cmbpo_rollout_reset(const cmbpo_rollout_t *r, void *stream);

/* Uncertainty test on accumulated + new DKL BEFORE storing and the budget
 * early-termination of the first n surviving rows by index
 * (model_sampler.py:275-287) -> fin_code, counters. */
int cmbpo_rollout_decide(const cmbpo_rollout_t *r, void *stream);
/* Counting half of the above only: writes iscal[8..11] = {n_alive, n_unc,
 * total_samples, 0}, the row a sharded run all-gathers; the host turns the
 * gathered rows into (host_excess, host_rank_off) for cmbpo_rollout_decide
 * (budget rule across shards; dist.budget_plan). */
int cmbpo_rollout_count(const cmbpo_rollout_t *r, void *stream);

/* ModelBuffer.finish_path_multiple (modelbuffer.py:138-182) = reward + cost GAE
 * (utilities/utils.py:184-188 discount_cumsum, float64 recurrence) for the rows
 * selected by `mode`, then marks them terminated:
 *   0 PRE : rows with fin_code != 0, bootstrap V/VC of the pre-step obs (:290,401-407)
 *   1 POST: stored rows; horizon (path_length >= max_path_length-1, :350-353) finishes
 *           all with V/VC(next_obs); else env-terminal rows with last_val = 0 but
 *           last_cval = VC(next_obs) (:357-367)
 *   2 ALL : finish_all_paths (:418-444), bootstrap V/VC of the current obs. */
int cmbpo_rollout_finish(const cmbpo_rollout_t *r, int mode, void *stream);

/* ModelBuffer.store_multiple for the surviving rows at column ptr + the sampler
 * accumulators (modelbuffer.py:114-135, model_sampler.py:314-346). */
int cmbpo_rollout_store(const cmbpo_rollout_t *r, void *stream);

/* Rebuild the ordered alive list from `alive` (replaces the boolean-mask
 * compaction of model_sampler.py:300-311,361-372). */
int cmbpo_rollout_compact(const cmbpo_rollout_t *r, void *stream);
/* The whole step in one call (single-GPU jobs without a cross-shard budget exchange): cmbpo_policy_forward ->
 * cmbpo_ens_forward -> cmbpo_fakeenv_post -> decide -> finish(PRE) -> store -> cmbpo_ens_predict_mean x 2 at next_obs ->
 * finish(POST), every buffer taken from *r (slot-indexed d_eps [B, act], d_elite [B]; scratch d_mean / d_var
 * [E, B, obs + 1]).  n_alive = the host's copy of iscal[CMBPO_I_N_ALIVE]. */
/* Small rollout batches: decide + finish(PRE) + store (+ its statistics) as one single-workgroup launch for up to
 * cmbpo_rollout_book_pre_max_rows() alive rows (same decisions and per-branch arithmetic as the separate calls;
 * single-rank path).  cmbpo_rollout_step uses it by itself. */
int cmbpo_rollout_book_pre_max_rows(void);
int cmbpo_rollout_book_pre(const cmbpo_rollout_t *r, int n_alive, void *stream);
/* ... and finish(POST) + the compaction likewise: the ordered alive list of the survivors is ALWAYS written to
 * alive_idx_out (iscal[CMBPO_I_N_ALIVE] updated), so the host swaps the two lists after every such step.
 * cmbpo_rollout_step returns 1 (instead of 0) when it took this path. */
int cmbpo_rollout_book_post(const cmbpo_rollout_t *r, int n_alive, void *stream);
/* The step's counters and accumulators (iscal[32] | dscal[32], one 384-byte block) into host memory + a stream
 * synchronisation: the one host sync of a rollout step (the reference's per-step `alive_ratio`,
 * samplers/model_sampler.py:371-375). */
int cmbpo_rollout_read_scalars(const cmbpo_rollout_t *r, void *h_out384, void *stream);
int cmbpo_rollout_step(const cmbpo_rollout_t *r, int n_alive, cmbpo_mlp_t *policy, cmbpo_mlp_t *model,
                       cmbpo_mlp_t *v, cmbpo_mlp_t *vc, int task, int ensemble, const float *d_eps,
                       const int32_t *d_elite, float *d_mean, float *d_var, void *stream);

/* Several steps in one call (single-rank path; replaces the body of the rollout loop of algorithms/cmbpo.py:352-360
 * around ModelSampler.sample): step -> counters -> swap of the alive lists and of the cur / next arrays, repeated
 * until max_steps, no branch alive, at most min_alive alive (`alive_ratio <= 0.1`), total_samples >= stop_total
 * (stop_total = NaN: no such test; a threshold <= 0 is reached by the first step, as in the reference's
 * `_total_samples + samples_added >= .99 * approx_model_batch`) or a full buffer.  *r is left as the caller of cmbpo_rollout_step would leave it.
 * d_eps / d_elite: the draws of the first step; step k reads them k * eps_stride / k * elite_stride elements on.
 * h_scalars: [max_steps][384 bytes] of host memory (pinned), the counters of every step taken (iscal | dscal);
 * *list_swaps = how often alive_idx / alive_idx_out changed places. */
int cmbpo_rollout_run(cmbpo_rollout_t *r, int n_alive, cmbpo_mlp_t *policy, cmbpo_mlp_t *model, cmbpo_mlp_t *v,
                      cmbpo_mlp_t *vc, int task, int ensemble, const float *d_eps, const int32_t *d_elite,
                      long eps_stride, long elite_stride, float *d_mean, float *d_var, int max_steps,
                      double stop_total, int min_alive, void *h_scalars, int *steps_done, int *n_alive_out,
                      int *list_swaps, void *stream);

/* ModelBuffer.get (modelbuffer.py:184-226): d_offsets[B+1] = exclusive scan of
 * len; d_stats[8] = {n, adv_mean, adv_std, cadv_mean, ret_mean, cret_mean}
 * (two-pass mean / std of utilities/mpi_tools.py:71-92).  With d_gstats
 * (global statistics after an all-reduce) NULL the local ones are used. */
int cmbpo_buffer_offsets(const cmbpo_rollout_t *r, int32_t *d_offsets, void *stream);
int cmbpo_buffer_moments(const cmbpo_rollout_t *r, int pass, double *d_stats, void *stream);
/* The same offsets and statistics on ONE GPU as two launches (four beyond 32 768 branches, where the fold of the
 * workgroups' partial sums is a launch of its own; r->world <= 1; replaces cmbpo_buffer_offsets + the four
 * cmbpo_buffer_moments passes of ModelBuffer.get, buffers/modelbuffer.py:184-204 / utilities/mpi_tools.py:71-92): the
 * scan of the path lengths and the first moments in one pass over the buffers, the offsets and the centred second
 * moment in another; the last workgroup of each launch adds the workgroups' partial sums in a fixed order. */
int cmbpo_buffer_prepare(const cmbpo_rollout_t *r, int32_t *d_offsets, double *d_stats, void *stream);
/* Flatten in branch-major, time-minor order into the 12-array list
 * [obs, act, adv, cadv, ret, cret, logp, val, cval, cost, log_std, mu]
 * (modelbuffer.py:212-218), normalising adv by (mean, std + 1e-8) and centring
 * cadv (:198-204).  d_out[12] are device pointers sized for offsets[B] rows. */
int cmbpo_buffer_flatten(const cmbpo_rollout_t *r, const int32_t *d_offsets,
                         const double *d_stats, float *const *h_out12, void *stream);

/* ------------------------------------------------------------------------ *
 * CPO trust-region update: the policy-graph fetches of CPOAgent.update_pi
 * (policies/cpo_policy.py:153-300) as fused HIP kernels.  All results are RAW
 * SUMS over the samples of this call (the caller divides by the global sample
 * count after an all-reduce, instead of mpi_avg's equal-shard assumption,
 * utilities/mpi_tools.py:67-69).  Parameter vectors are flat float32[P] in the
 * order of get_vars('pi') (network/ac_network.py:35-36): W0[obs,H], b0,
 * W1[H,H], b1, W2[H,act], b2, log_std.  Hidden width H: 128 (every shipped
 * experiment; both matrix paths) or 256 (configs/baseconfig/base.py:7, the
 * default: fp32 MFMAs whatever cmbpo_set_pi_matrix_path says); obs <= 64,
 * act <= 32.
 * ------------------------------------------------------------------------ */
typedef struct cmbpo_pi cmbpo_pi_t;

typedef struct cmbpo_pi_batch {   /* the actor feed (actor_phs, cpo_policy.py:479-487) */
  int32_t n, obs_dim, act_dim;
  const float *obs;        /* [n,obs]                                       */
  const float *act;        /* [n,act]                                       */
  const float *adv;        /* [n]                                           */
  const float *cadv;       /* [n]                                           */
  const float *logp_old;   /* [n]                                           */
  const float *cost;       /* [n]   cur_cost_ph                             */
  const float *mu_old;     /* [n,act] pi_info 'mu'                          */
  const float *logstd_old; /* [n,act] pi_info 'log_std'                     */
} cmbpo_pi_batch_t;

int cmbpo_pi_create(cmbpo_pi_t **out, int obs_dim, int hidden, int act_dim);
void cmbpo_pi_destroy(cmbpo_pi_t *h);
int cmbpo_pi_num_params(const cmbpo_pi_t *h);

/* set_pi_params (utilities/trust_region.py:21-25, cpo_policy.py:559): d_flat is
 * a DEVICE vector; packing into MFMA fragment order happens on the device. */
int cmbpo_pi_set_params(cmbpo_pi_t *h, const float *d_flat, void *stream);

/* flat_g / flat_b with their losses (cpo_policy.py:169-171, 522-555):
 * which = 0: d_vec = sum_n grad(-ratio*adv)   (divide by N, subtract ent_reg on
 *            the log_std block, to get flat_g); which = 1: sum_n grad(ratio*cadv).
 * d_sums[8] = {n, sum ratio*adv, sum ratio*cadv, -, sum cost}. */
int cmbpo_pi_loss_grad(cmbpo_pi_t *h, const cmbpo_pi_batch_t *b, int which,
                       float *d_vec, double *d_sums, void *stream);

/* hessian_vector_product(d_kl, pi_params) (utilities/trust_region.py:15-19) in
 * its Gauss-Newton / Fisher form: d_vec = sum_n J^T diag(1/(exp(2 ls_old)+1e-8))
 * J v on the MLP block and sum_n 2 exp(2 ls)/(exp(2 ls_old)+1e-8) * v on log_std.
 * The caller divides by N and adds damping_coeff * v (cpo_policy.py:550-552). */
int cmbpo_pi_fvp(cmbpo_pi_t *h, const cmbpo_pi_batch_t *b, const float *d_v,
                 float *d_vec, void *stream);

/* The reference re-feeds the batch through the whole policy graph for every one
 * of the 10-20 Hx evaluations of an update (cpo_policy.py:168, 550-552), although
 * parameters and batch are fixed from flat_g to the end of the CG solves.  With
 * enable != 0, cmbpo_pi_loss_grad saves the two hidden activation images of the
 * batch (1 KB per sample, owned by the handle, grow-only) and every later
 * Fisher-vector product on the same (obs pointer, n) reads them instead of
 * recomputing the forward chain -- bit-identical results.  The saved images are
 * dropped by cmbpo_pi_set_params and by any call of this function; the caller
 * must not change the batch in place between loss_grad and the products.  Off by
 * default; if the memory cannot be had the products silently recompute. */
int cmbpo_pi_keep_activations(cmbpo_pi_t *h, int enable);
/* Fisher-vector product launches that read saved activations so far (a launch
 * captured into the CG graph counts once); diagnostics / tests. */
long cmbpo_pi_saved_activation_uses(const cmbpo_pi_t *h);

/* Arithmetic of the policy kernels' matrix products (forward, JVP, backward and
 * weight-gradient): 1 (default) = three f16 MFMAs on two-piece operands (11 + 1 +
 * 11 mantissa bits, fp32 accumulation: the error of a product stays at fp32's,
 * see DESIGN.md 3a), 0 = fp32 MFMAs throughout (round 1).  Environment
 * CMBPO_PI_F16=0 selects 0 at load. */
void cmbpo_set_pi_matrix_path(int path);
int cmbpo_get_pi_matrix_path(void);

/* [d_kl, pi_loss, surr_cost] at the current parameters (set_and_eval,
 * cpo_policy.py:278-280): d_sums[8] = {n, sum ratio*adv, sum ratio*cadv,
 * sum_n sum_a kl, sum cost}. */
int cmbpo_pi_eval(cmbpo_pi_t *h, const cmbpo_pi_batch_t *b, double *d_sums,
                  void *stream);

/* CPOBuffer.finish_path (buffers/cpobuffer.py:179-207) / any set of contiguous
 * paths cut out of flat [N] arrays by d_offsets[n_paths+1]: reward + cost GAE
 * (utilities/utils.py:184-188) with per-path bootstraps.  d_f64_mask (may be
 * NULL) bit0/bit1: compute the reward/cost deltas in float64, which is what
 * np.append does to the float32 buffers when the bootstrap is not float32
 * (the np.zeros((1,)) of samplers/cpo_sampler.py:205-212). */
int cmbpo_gae_segments(int n_paths, const int32_t *d_offsets, const float *d_rew,
                       const float *d_val, const float *d_cost, const float *d_cval,
                       const float *d_last_val, const float *d_last_cval,
                       const uint8_t *d_f64_mask, double gamma, double lam,
                       double cost_gamma, double cost_lam, float *d_adv,
                       float *d_ret, float *d_cadv, float *d_cret, void *stream);

/* CPOBuffer.get normalisation (buffers/cpobuffer.py:261-268): in place
 * adv = (adv - mean) / (std + 1e-8), cadv -= mean(cadv), two-pass statistics of
 * utilities/mpi_tools.py:71-87.  d_stats[16]: [0] n [1] adv_mean [2] adv_std
 * [3] cadv_mean. */
int cmbpo_adv_normalize(int n, float *d_adv, float *d_cadv, double *d_stats, void *stream);

/* Parameter-vector algebra of the update kept on the device (csrc/vec_ops.hip).
 * cg_init / cg_step are utilities/trust_region.py:32-45 (x = 0, r = p = b; then per
 * iteration z = hp_sum * inv_n + damping * p, alpha = rr / (p.z + 1e-8), x += alpha p,
 * r -= alpha z, p = r + (rr_new / rr) p); d_scal[0] carries r.r.  d_hp_sum is the raw
 * cmbpo_pi_fvp output for direction d_p. */
int cmbpo_cg_init(int P, const float *d_b, float *d_x, float *d_r, float *d_p, double *d_scal, void *stream);
int cmbpo_cg_step(int P, const float *d_hp_sum, double inv_n, float damping, float *d_x, float *d_r,
                  float *d_p, double *d_scal, void *stream);
/* The whole solve x = cg(Hx, b) in one call (single-GPU jobs): cg_init, then `iters` times [cmbpo_pi_fvp on direction
 * d_p into d_vec, cg_step].  With use_graph != 0 the iterations after the first are captured into a hipGraph that is
 * cached per handle (re-captured when a pointer / size changes); cmbpo_pi_cg_release drops it. */
int cmbpo_pi_cg_solve(cmbpo_pi_t *h, const cmbpo_pi_batch_t *b, const float *d_b, double inv_n, float damping,
                      int iters, float *d_x, float *d_r, float *d_p, float *d_vec, double *d_scal,
                      int use_graph, void *stream);
void cmbpo_pi_cg_release(cmbpo_pi_t *h);
/* The iteration cmbpo_pi_cg_solve repeats: Fisher-vector product of direction d_p (kept as per-workgroup partial
 * vectors), z = Hp / N + damping p, alpha, x, r, beta, p updated by three multi-workgroup kernels with ordered float64
 * dot products.  d_scal[0] = r.r; d_scal[1] < 0 on entry to the first iteration; cmbpo_pi_cg_commit after the last. */
int cmbpo_pi_cg_iter(cmbpo_pi_t *h, const cmbpo_pi_batch_t *b, double inv_n, float damping, float *d_x,
                     float *d_r, float *d_p, double *d_scal, void *stream);
int cmbpo_pi_cg_commit(double *d_scal, void *stream);
long cmbpo_pi_cg_graph_launches(void);   /* graph replays so far; -1: stream capture unavailable, eager loop in use */
long cmbpo_pi_cg_graph_captures(void);   /* graphs captured so far: one per (handle, solution vector) while the arguments repeat */
/* d_out = a * d_x + b * d_y (d_y may be NULL): Hx = hvp / N + damping v, the step x = (v + nu w) / (lam + eps)
 * (policies/cpo_policy.py:266) and the trial parameters old - step * x (:278). */
int cmbpo_vec_lincomb(int P, float a, const float *d_x, float b, const float *d_y, float *d_out, void *stream);
/* d_out[k] = sum_i x_k[i] * y_k[i], k < n <= 8 (q, r, s, b.b of policies/cpo_policy.py:212-226); h_x / h_y are
 * host arrays of device pointers. */
int cmbpo_vec_dots(int P, int n, const float *const *h_x, const float *const *h_y, double *d_out, void *stream);

/* ---- ensemble training (csrc/ens_train.hip) -------------------------
 * The TensorFlow train_op of the reference ensembles (models/pens/pe.py:252-274,312-318:
 * train_loss = sum_e loss_e + sum_l decay_l * l2_loss(W_l), tf.train.AdamOptimizer(lr)) for the
 * two losses the shipped configs use: 'MSPE' on HEAD_PROB handles (pe.py:921-973, dynamics)
 * and 'MSE' on HEAD_DETMEAN handles (pe.py:840-919 with inc_var_loss=False, critics).  The
 * trainer owns the row-major master weights and Adam moments and keeps the handle's packed
 * weights (what cmbpo_ens_forward / cmbpo_ens_predict_mean read) in step with them. */
typedef struct cmbpo_trainer cmbpo_trainer_t;

/* decays[3] = weight_decay of the three FC layers (pe_factory.py:50-55: decay/4, decay/2,
 * decay); max_batch bounds the rows of one step (activations are kept for the backward pass). */
int cmbpo_trainer_create(cmbpo_trainer_t **out, cmbpo_mlp_t *m, int max_batch, float lr,
                         const double *decays);
void cmbpo_trainer_destroy(cmbpo_trainer_t *t);
/* Train loss (PE.finalize, models/pens/pe.py:240-300).  DEFAULT: 'MSPE' (_mspe_loss, :921-973) for probabilistic
 * heads, 'MSE' (_nll_loss(inc_var_loss=False), :840-919) for deterministic ones -- what the shipped configs use.
 * NLL: _nll_loss(inc_var_loss=True) = mean 0.5 exp(-log_var)(mean - t)^2 + mean 0.5 log_var, probabilistic heads
 * only (the class default of PE, pe.py:65).  `self.loss` (cmbpo_trainer_losses) is the same for all of them. */
#define CMBPO_LOSS_DEFAULT 0
#define CMBPO_LOSS_NLL 1
int cmbpo_trainer_set_loss(cmbpo_trainer_t *t, int loss);
/* Host weights in the reference variable layout (W[E][in][out], b[E][out]) -> masters and the
 * handle's packed images; the Adam state is untouched (the reference keeps it across train()
 * calls, pe.py:318).  get_weights copies the masters back (checkpointing, models/pens/pe.py:736-764). */
int cmbpo_trainer_set_weights(cmbpo_trainer_t *t, const float *h_w0, const float *h_b0,
                              const float *h_w1, const float *h_b1, const float *h_w2,
                              const float *h_b2, void *stream);
int cmbpo_trainer_get_weights(cmbpo_trainer_t *t, float *h_w0, float *h_b0, float *h_w1,
                              float *h_b1, float *h_w2, float *h_b2, void *stream);
int cmbpo_trainer_reset_optimizer(cmbpo_trainer_t *t, void *stream);
/* Adam moments (which: 0 = m, 1 = v) in the layout of the weights, and the step count t of
 * lr_t = lr sqrt(1 - b2^t) / (1 - b1^t): optimizer checkpoint / resume (the reference saves
 * optimizer.variables() with the model, pe.py:318,736-764). */
int cmbpo_trainer_get_moments(cmbpo_trainer_t *t, int which, float *h_w0, float *h_b0, float *h_w1,
                              float *h_b1, float *h_w2, float *h_b2, void *stream);
int cmbpo_trainer_set_moments(cmbpo_trainer_t *t, int which, const float *h_w0, const float *h_b0,
                              const float *h_w1, const float *h_b1, const float *h_w2,
                              const float *h_b2, long steps_done, void *stream);
/* TensorStandardScaler.fit result (models/pens/utils.py:119-138) -> the handle, without touching
 * the weights; NULL pairs are left as they are. */
int cmbpo_mlp_set_scalers(cmbpo_mlp_t *m, const float *h_in_mu, const float *h_in_var,
                          const float *h_out_mu, const float *h_out_var, void *stream);
/* One sess.run(train_op) (pe.py:541-563): member e trains on rows d_idx[e * idx_stride + b],
 * b < batch, of d_inputs[N][in_dim] / d_targets[N][target_dim] (inputs[batch_idxs], pe.py:543-547);
 * d_idx == NULL takes rows 0..batch-1 for every member. */
int cmbpo_trainer_step(cmbpo_trainer_t *t, const float *d_inputs, int in_dim,
                       const float *d_targets, int target_dim, const int32_t *d_idx,
                       int idx_stride, int batch, void *stream);
/* The minibatch loop of one epoch (pe.py:541-563): ceil(n_rows / batch) train steps on columns
 * [k * batch, (k + 1) * batch) of the index lists, enqueued without returning to the caller. */
int cmbpo_trainer_epoch(cmbpo_trainer_t *t, const float *d_inputs, int in_dim,
                        const float *d_targets, int target_dim, const int32_t *d_idx,
                        int idx_stride, int n_rows, int batch, void *stream);
/* sess.run(self.loss) (pe.py:264,582-603,629-635): d_losses[e] = 0.5 * mean over rows and
 * target dims of (mean head - scaled target)^2; idx_stride 0 evaluates every member on the same
 * rows (the tiled holdout set). */
int cmbpo_trainer_losses(cmbpo_trainer_t *t, const float *d_inputs, int in_dim,
                         const float *d_targets, int target_dim, const int32_t *d_idx,
                         int idx_stride, int n_rows, float *d_losses, void *stream);
long cmbpo_trainer_steps_done(const cmbpo_trainer_t *t);

#ifdef __cplusplus
}
#endif
#endif /* CMBPO_HIP_H */
