// Internal (not part of the C-ABI): shared between ens_mlp.hip and ens_train.hip.
#pragma once
#include "common.h"

#include <vector>

#define CMBPO_HEAD_TRAIN 3   // raw outputs + exported activations, per-member row gather (training forward)

struct MlpKernelArgs {
  // packed weights, float4 units
  const f32x4 *wp0, *wp1, *wp2;
  size_t wp0_stride, wp1_stride, wp2_stride;  // per member, in float4
  const float *b0, *b1, *b2;                  // [E][HID],[E][HID],[E][o_pad]
  const float *in_mu, *in_sig;                // [in_dim] or nullptr; sig = max(sqrt(var), 1e-2)
  const float *out_mu, *out_sig, *out_lsig2;  // [out_dim] or nullptr; lsig2 = 2 log(sig)
  const float *log_std;                       // [out_dim] (policy head)
  int ensemble, e_chunk;
  int in_dim, in_pad;  // in_pad multiple of 8
  int o_width, o_tiles, out_dim;
  // inputs
  const float *obs;
  int obs_dim;
  const float *act;
  int act_dim;
  const float *eps;
  const int32_t *row_idx;
  const int32_t *n_rows_dev;
  int n_rows;
  int ld_rows;
  // outputs
  float *out0;  // mean | predict-mean | pi
  float *out1;  // var  |              | logp
  float *out2;  //                     | mu
  float *out3;  //                     | log_std broadcast
  unsigned long long *stamps;  // diagnostic builds only
  int stagger_sleeps;          // s_sleep(127) repetitions for the second dispatch batch (0 = off)
  int tiles, n_items, n_cu;    // persistent grid: items = member chunks x row tiles, member-major
  int *work_counter;           // device counter for dynamic item claiming (nullptr: static striding)
  // training forward (HEAD_TRAIN): rows are gathered through a per-member index list and every intermediate
  // the backward pass needs is exported, indexed [member][batch row][...]
  int row_idx_stride;          // elements between the index lists of consecutive members
  float *tr_x;                 // [E][n_rows][in_pad]  scaled inputs
  float *tr_h1, *tr_g1;        // [E][n_rows][HID]     swish(z1), swish'(z1)
  float *tr_h2, *tr_g2;        // [E][n_rows][HID]
  const float *tr_targets;     // [N][tr_tdim] regression targets (rows gathered like the inputs); with tr_loss_part
  int tr_tdim;
  double *tr_loss_part;        // [items][3] per-tile loss statistics (nullptr: not wanted)
  float *tr_opmax;             // [items][8] largest |x|, |h1|, |h2| of the tile in slots 0..2 (nullptr: not wanted): the
                               // f16 weight-gradient kernel lifts its operands by the member's maxima (ens_train.hip)
};

struct cmbpo_mlp {
  int ensemble, in_dim, in_pad, hidden, o_width, o_tiles, out_dim, act, head;
  bool loaded, has_in_scaler, has_out_scaler;
  float *d_blob;  // one allocation holding everything below
  size_t blob_floats;
  // offsets (in floats) into the blob
  size_t off_wp0, off_wp1, off_wp2, off_b0, off_b1, off_b2;
  size_t off_in_mu, off_in_var, off_out_mu, off_out_var, off_out_lsig2, off_log_std;   // *_var hold sigma
  std::vector<float> h_blob;
  // split-bf16 forward (ens_split.hip): three bf16 images of the packed weights, rebuilt when the packs change
  void *d_split = nullptr;
  unsigned long pack_version = 0, split_version = ~0ul;
  size_t sp_off[3] = {0, 0, 0}, sp_stride[3] = {0, 0, 0};   // 16-B units
  // three-term f16 forward (ens_h3.hip): two f16 images of the packed weights + per-member statistics
  void *d_h3 = nullptr;
  unsigned long h3_version = ~0ul;
  size_t h3_off[3] = {0, 0, 0}, h3_stride[3] = {0, 0, 0}, h3_stats_off = 0;   // 16-B units
  int h3_s0 = 0, h3_otp = 0;   // k-slabs of the input layer, padded output tiles
};

// fills the weight / scaler pointers of `a` from the handle and launches the kernel matching (hidden, act, head);
// head_override >= 0 replaces the handle's head (HEAD_TRAIN on a PROB / DETMEAN handle).
int cmbpo_internal_launch_mlp(cmbpo_mlp *m, MlpKernelArgs &a, hipStream_t s, int head_override);
// HEAD_PROB, 512-wide, swish: the same forward on the bf16 matrix cores (fp32 products as six bf16 MFMAs); `a` filled
// as for cmbpo_internal_launch_mlp
int cmbpo_internal_launch_split(cmbpo_mlp *m, MlpKernelArgs &a, hipStream_t s);
// HEAD_PROB, 512-wide, swish: the same forward with three f16 MFMAs per float32 product, 128-row items (ens_h3.hip)
bool cmbpo_internal_h3_eligible(const cmbpo_mlp *m);
int cmbpo_internal_launch_h3(cmbpo_mlp *m, MlpKernelArgs &a, hipStream_t s);
// the actor (128-wide tanh + gaussian head, one member) on the three-term f16 path (policy_f16.hip)
bool cmbpo_internal_policy_f16_eligible(const cmbpo_mlp *m);
int cmbpo_internal_launch_policy_f16(cmbpo_mlp *m, const MlpKernelArgs &a, hipStream_t s);
int cmbpo_internal_policy_f16_args(cmbpo_mlp *m, void *pf_args, int *s0_out, hipStream_t s);
// both critics and, riding along as one more wave per tile, the actor at the same rows (critic_f16.hip)
int cmbpo_internal_critic_big_min();
bool cmbpo_internal_critic_pair_can_ride(const cmbpo_mlp *v, const cmbpo_mlp *vc, const cmbpo_mlp *policy);
// optional passenger of the critics' launch at small batches: the vector half of the rollout step's store (obs, act, mu, log_std
// of every row that was not finished before the store -> column `col_off / B` of the buffers), copied by each tile's workgroup for
// its own 32 rows before the rider may overwrite the per-step arrays (store_vec_kernel as a launch of its own otherwise)
struct CmbpoStoreVec {
  const float *src[4];
  float *dst[4];
  int dim[4];
  const uint8_t *fin_code;
  size_t col_off;        // ptr * B: first row of the column
};
int cmbpo_internal_critic_pair_ride(cmbpo_mlp *v, cmbpo_mlp *vc, const float *d_obs, int obs_dim, const int32_t *d_row_idx,
                                    const int32_t *d_n_rows, int n_rows, float *d_v, float *d_vc, cmbpo_mlp *policy,
                                    const float *d_eps, float *d_pi, float *d_logp, float *d_mu, float *d_ls, void *stream,
                                    const CmbpoStoreVec *store_vec = nullptr);
// HEAD_DETMEAN, 128-wide, swish, one output (the critics) on the same matrix path
int cmbpo_internal_launch_critic_split(cmbpo_mlp *m, MlpKernelArgs &a, hipStream_t s);
