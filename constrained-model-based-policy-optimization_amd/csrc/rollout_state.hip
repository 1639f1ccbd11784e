// Device-resident rollout bookkeeping: the fused counterpart of
//   samplers/model_sampler.py:239-444  (ModelSampler.sample / _finish_paths / finish_all_paths)
//   buffers/modelbuffer.py:53-226      (ModelBuffer.reset / store_multiple / finish_path_multiple / get)
//   utilities/utils.py:184-188         (discount_cumsum, un-weighted branch: lfilter([1],[1,-g*l]) in float64)
//   utilities/mpi_tools.py:71-92       (mpi_statistics_scalar: two-pass mean / std)
//
// All kernels here are HBM / latency bound integer-and-scan work: branch slots never move, an ordered
// alive list replaces the reference's per-step boolean-mask compaction, and the (T, B) time-major
// buffers make both the per-step store and the per-branch backward GAE recurrence coalesced
// (adjacent lanes = adjacent branches).
#include "common.h"
#include <type_traits>

#include <math.h>
#include <stdlib.h>

namespace {

constexpr int kScanThreads = 1024;

// ---- block-level helpers (wave = 64 lanes) -----------------------------------------------------
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// exclusive scan over the block (blockDim.x <= 1024); *total = block sum.  `sm` needs 17 ints.
__device__ __forceinline__ int block_excl_scan(int v, int *sm, int *total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  const int inc = wave_incl_scan(v, lane);
  __syncthreads();  // sm reuse across calls
  if (lane == 63) sm[w] = inc;
  __syncthreads();
  if (w == 0) {
    int t = (lane < nw) ? sm[lane] : 0;
    t = wave_incl_scan(t, lane);
    if (lane < nw) sm[lane] = t;
    if (lane == nw - 1) sm[16] = t;
  }
  __syncthreads();
  *total = sm[16];
  return inc - v + (w > 0 ? sm[w - 1] : 0);
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}

// block sum; result valid in thread 0.  `sm` needs 16 doubles.
__device__ __forceinline__ double block_sum(double v, double *sm) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  double t = 0.0;
  if (w == 0) {
    t = (lane < nw) ? sm[lane] : 0.0;
    t = wave_sum(t);
  }
  return t;
}

__device__ __forceinline__ double block_max(double v, double *sm) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  double t = 0.0;
  if (w == 0) {
    t = (lane < nw) ? sm[lane] : -1e300;
    t = wave_max(t);
  }
  return t;
}

// NS block sums followed by NM block maxima with the barriers of ONE reduction: per value exactly the operations of block_sum /
// block_max above (same shuffle trees, same order: same bits), results valid in thread 0.  `sm` needs (NS + NM) * 16 doubles.
template <int NS, int NM>
__device__ __forceinline__ void block_reduce_many(double (&v)[NS + NM], double *sm) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < NS + NM; ++k) v[k] = k < NS ? wave_sum(v[k]) : wave_max(v[k]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NS + NM; ++k) sm[k * 16 + w] = v[k];
  }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int k = 0; k < NS + NM; ++k) {
      double t = (lane < nw) ? sm[k * 16 + lane] : (k < NS ? 0.0 : -1e300);
      v[k] = k < NS ? wave_sum(t) : wave_max(t);
    }
  } else {
#pragma unroll
    for (int k = 0; k < NS + NM; ++k) v[k] = 0.0;
  }
}

// ---- reset -------------------------------------------------------------------------------------
__global__ void reset_kernel(const cmbpo_rollout_t r) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < r.B) {
    r.alive[b] = 1;
    r.fin_code[b] = 0;
    r.len[b] = 0;
    r.alive_idx[b] = b;
    r.dkl_acc[b] = 0.0;
    r.path_ret[b] = 0.0;
    r.path_cost[b] = 0.0;
    r.path_dyn_var[b] = 0.0;
  }
  if (b < 32) {
    r.iscal[b] = (b == CMBPO_I_N_ALIVE) ? r.B : 0;
    r.dscal[b] = 0.0;
  }
}

// ---- decide: uncertainty test + budget early termination (model_sampler.py:275-287) ------------
__device__ __forceinline__ bool too_uncertain(const cmbpo_rollout_t &r, int b) {
  // next_dkl = _dyn_dkl_path[alive] (float64) + dyn_dkl_path (float32) >= dkl_lim
  return r.uncertainty_mode && (r.dkl_acc[b] + (double)r.dkl_t[b] >= r.dkl_lim);
}

// pass 1 (grid): uncertainty flags -> fin_code; per-workgroup {count, sum of dkl_t} into the scratch array (no counter
// to clear beforehand, no atomics: pass 2 adds the partials in workgroup order)
__global__ __launch_bounds__(256) void decide_flags_kernel(const cmbpo_rollout_t r) {
  __shared__ double sm_d[16];
  const int n = r.iscal[CMBPO_I_N_ALIVE];
  int cnt = 0;
  double dsum = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int b = r.alive_idx[i];
    const bool u = too_uncertain(r, b);
    r.fin_code[b] = u ? 1 : 0;
    cnt += u ? 1 : 0;
    dsum += (double)r.dkl_t[b];
  }
  const double c = block_sum((double)cnt, sm_d);
  const double d = block_sum(dsum, sm_d);
  if (threadIdx.x == 0) {
    r.store_part[2 * blockIdx.x] = c;
    r.store_part[2 * blockIdx.x + 1] = d;
  }
}

// pass 2 (one workgroup): this step's counters, then the budget rule -- the first `excess` surviving rows in index
// order are finished too.  Exits at once when the budget is not exceeded (every step but the last one or two).
__global__ __launch_bounds__(kScanThreads) void decide_budget_kernel(const cmbpo_rollout_t r, int count_only, int n_parts) {
  __shared__ int sm_i[17];
  __shared__ double sm_d[16];
  __shared__ int s_unc;
  const int tid = threadIdx.x;
  const int n = r.iscal[CMBPO_I_N_ALIVE];
  {
    double c = 0.0, d = 0.0;
    for (int w = tid; w < n_parts; w += kScanThreads) {
      c += r.store_part[2 * w];
      d += r.store_part[2 * w + 1];
    }
    c = block_sum(c, sm_d);
    d = block_sum(d, sm_d);
    if (tid == 0) {
      s_unc = (int)c;
      r.iscal[CMBPO_I_N_UNC] = (int)c;
      r.dscal[CMBPO_D_DKL_SUM_T] = d;
      r.iscal[CMBPO_I_N_STORED] = 0;
      r.iscal[CMBPO_I_N_FIN_POST] = 0;
    }
    __syncthreads();
  }
  const int n_unc = s_unc;
  if (tid == 0) {
    // the row other shards gather: {n_alive, n_unc, total_samples, 0}
    r.iscal[8] = n;
    r.iscal[9] = n_unc;
    r.iscal[10] = (int)r.dscal[CMBPO_D_TOTAL_SAMPLES];
    r.iscal[11] = 0;
    r.iscal[CMBPO_I_N_FIN_PRE] = n_unc;
  }
  if (count_only || r.max_samples == 0) return;    // `if max_samples:` -- a negative budget IS a budget (model_sampler.py:282)
  long long excess;
  int rank_off = 0;
  if (r.use_host_budget) {
    excess = r.host_excess;
    rank_off = r.host_rank_off;
  } else {
    // n = total + alive - too_uncertain; n = max(n - max_samples, 0)   (model_sampler.py:283-284)
    excess = (long long)r.dscal[CMBPO_D_TOTAL_SAMPLES] + n - n_unc - (long long)r.max_samples;
  }
  if (excess <= 0) return;
  int carry = 0, nfin = 0;
  for (int base = 0; base < n; base += kScanThreads) {
    const int i = base + tid;
    int b = -1, flag = 0;
    if (i < n) {
      b = r.alive_idx[i];
      flag = r.fin_code[b] ? 0 : 1;
    }
    int total;
    const int excl = block_excl_scan(flag, sm_i, &total);
    // early_term[:n] = True over the surviving rows in index order
    if (flag && ((long long)rank_off + carry + excl < excess)) {
      r.fin_code[b] = 1;
      nfin += 1;
    }
    carry += total;
    if ((long long)rank_off + carry >= excess) break;   // uniform: carry is a block-wide value
  }
  const double nf = block_sum((double)nfin, sm_d);
  if (tid == 0) r.iscal[CMBPO_I_N_FIN_PRE] = n_unc + (int)nf;
}

// reward + cost GAE of one finished path and its termination mark (modelbuffer.py:138-182; discount_cumsum =
// lfilter([1], [1, -g*l]) on the reversed row with a float64 state, utilities/utils.py:184-188)
__device__ __forceinline__ void gae_finish_path(const cmbpo_rollout_t &r, int b, float lv, float lcv, bool zero_boot) {
  const int L = r.len[b];
  const size_t B = (size_t)r.B;
  const float g32 = (float)r.gamma, cg32 = (float)r.cost_gamma;
  const double gl = r.gamma * r.lam, cgl = r.cost_gamma * r.cost_lam;
  double y = 0.0, cy = 0.0;
  float vnext = lv, cvnext = lcv;
  for (int t = L - 1; t >= 0; --t) {
    const size_t o = (size_t)t * B + b;
    const float rw = r.rew_buf[o], v = r.val_buf[o], c = r.cost_buf[o], cv = r.cval_buf[o];
    double delta;
    if (zero_boot) {
      // float64 arithmetic: rews/vals were promoted by the float64 zeros bootstrap
      delta = __dsub_rn(__dadd_rn((double)rw, __dmul_rn(r.gamma, (double)vnext)), (double)v);
    } else {
      delta = (double)__fsub_rn(__fadd_rn(rw, __fmul_rn(g32, vnext)), v);
    }
    const float cdelta = __fsub_rn(__fadd_rn(c, __fmul_rn(cg32, cvnext)), cv);
    // lfilter([1], [1, -g*l]) on the reversed row, float64 state: y = x + (g*l)*y_prev
    y = __dadd_rn(delta, __dmul_rn(gl, y));
    cy = __dadd_rn((double)cdelta, __dmul_rn(cgl, cy));
    const float adv = (float)y, cadv = (float)cy;
    r.adv_buf[o] = adv;
    r.ret_buf[o] = __fadd_rn(adv, v);
    r.cadv_buf[o] = cadv;
    r.cret_buf[o] = __fadd_rn(cadv, cv);
    vnext = v;
    cvnext = cv;
  }
  r.alive[b] = 0;
}

// ---- finish: reward + cost GAE, then mark terminated (modelbuffer.py:138-182) -------------------
__device__ __forceinline__ void store_stats_body(const cmbpo_rollout_t &r);
// fold_stats (the rollout step at large batches, mode 1): workgroup 0 first folds the sums store_kernel's tiles left -- the
// store is a launch further back, so no atomics are needed, and nothing before the end of this kernel reads the accumulators
// (a launch of its own until round 3: store_stats_kernel)
__global__ __launch_bounds__(256) void finish_kernel(const cmbpo_rollout_t r, int mode, int fold_stats) {
  if (fold_stats && blockIdx.x == 0) store_stats_body(r);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = r.iscal[CMBPO_I_N_ALIVE];
  if (i >= n) return;
  const int b = r.alive_idx[i];
  float lv, lcv;
  bool zero_boot = false;
  if (mode == 0) {
    const uint8_t code = r.fin_code[b];
    if (!code) return;
    lv = r.v_t[b];
    lcv = r.vc_t[b];
    if (code == 2) {  // finish_path_multiple called with float64 zeros as last_val
      lv = 0.0f;
      zero_boot = true;
    }
  } else if (mode == 1) {
    if (r.fin_code[b]) return;  // finished before the store
    // path_length (= ptr + 1 after the store) >= max_path_length - 1, model_sampler.py:352
    const bool horizon = (r.ptr + 1 >= r.max_path_length - 1);
    if (horizon) {
      lv = r.v_n[b];
    } else if (r.term_t[b]) {
      lv = 0.0f;
      zero_boot = true;  // np.zeros -> float64 rews/vals in the reference (model_sampler.py:404)
    } else {
      return;
    }
    lcv = r.vc_n[b];  // terminal branches still bootstrap the cost value (:364)
    atomicAdd(&r.iscal[CMBPO_I_N_FIN_POST], 1);
  } else {
    if (!r.alive[b]) return;
    lv = r.v_t[b];
    lcv = r.vc_t[b];
  }
  gae_finish_path(r, b, lv, lcv, zero_boot);
}

// ---- store: transition -> column ptr, sampler accumulators ---------------------------------------
// The copy wants many small workgroups (memory-level parallelism: 196 workgroups of 512 rows moved 54 MB at 0.9 TB/s),
// the sampler accumulators want few writers (13 contended atomics per workgroup: 1563 workgroups took 170 us).  So a
// tile is 64 rows -- wave 0 handles the scalar fields and reduces the tile's sums with shuffles -- every workgroup
// writes its 8 sums to r.store_part, and store_stats_kernel (one workgroup) folds them into iscal / dscal in a fixed
// order: no atomics, reproducible accumulators.
constexpr int kStoreRows = 64;

__global__ __launch_bounds__(256) void store_kernel(const cmbpo_rollout_t r) {
  __shared__ int s_slot[kStoreRows];   // branch slot of each row of the tile, -1: not stored
  const int n = r.iscal[CMBPO_I_N_ALIVE];
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * kStoreRows;
  if (row0 >= n) return;
  const size_t B = (size_t)r.B;
  const int D = r.obs_dim, A = r.act_dim;
  if (tid < kStoreRows) {
    double a_cnt = 0, a_cost = 0, a_rew = 0, a_v = 0, a_vc = 0, a_epv = 0, a_maxdkl = 0, a_maxret = 0;
    const int i = row0 + tid;
    int b = -1;
    if (i < n) {
      b = r.alive_idx[i];
      if (r.fin_code[b]) b = -1;
    }
    s_slot[tid] = b;
    if (b >= 0) {
      const size_t col = (size_t)r.ptr * B + b;
      const float rw = r.rew_t[b], c = r.cost_t[b], v = r.v_t[b], vc = r.vc_t[b];
      const float epv = r.epv_t[b], dk = r.dkl_t[b];
      r.rew_buf[col] = rw;
      r.val_buf[col] = v;
      r.cost_buf[col] = c;
      r.cval_buf[col] = vc;
      r.logp_buf[col] = r.logp_t[b];
      r.len[b] = r.ptr + 1;
      const double pr = r.path_ret[b] + (double)rw;
      r.path_ret[b] = pr;
      r.path_cost[b] += (double)c;
      r.path_dyn_var[b] += (double)epv;
      r.dkl_acc[b] += (double)dk;
      a_cnt = 1.0; a_cost = c; a_rew = rw; a_v = v; a_vc = vc;
      a_epv = (double)epv * D;
      a_maxdkl = (double)dk;
      a_maxret = pr;
    }
    const double v0 = wave_sum(a_cnt), v1 = wave_sum(a_cost), v2 = wave_sum(a_rew), v3 = wave_sum(a_v);
    const double v4 = wave_sum(a_vc), v5 = wave_sum(a_epv), v6 = wave_max(a_maxdkl), v7 = wave_max(a_maxret);
    if (tid == 0) {
      double *dst = r.store_part + (size_t)blockIdx.x * 8;
      dst[0] = v0; dst[1] = v1; dst[2] = v2; dst[3] = v3; dst[4] = v4; dst[5] = v5; dst[6] = v6; dst[7] = v7;
    }
  }
  __syncthreads();
  // vector fields: consecutive threads -> consecutive elements of a row (rows of a dense alive list are
  // adjacent slots, so both sides are contiguous)
  const int rows_here = min(kStoreRows, n - row0);
  const float *vsrc[4] = {r.cur_obs, r.act_t, r.mu_t, r.ls_t};
  float *vdst[4] = {r.obs_buf, r.act_buf, r.mu_buf, r.ls_buf};
  const int vdim[4] = {D, A, A, A};
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int dim = vdim[f];
    const float inv = 1.0f / (float)dim;
    for (int e = tid; e < rows_here * dim; e += 256) {
      const int k = (int)(((float)e + 0.5f) * inv);   // exact for e < 2^16
      const int d = e - k * dim;
      const int b = s_slot[k];
      if (b >= 0) vdst[f][((size_t)r.ptr * B + b) * dim + d] = vsrc[f][(size_t)b * dim + d];
    }
  }
}

// fold the per-tile sums of store_kernel into the counters / accumulators (the 256 threads of one workgroup, fixed order)
__device__ __forceinline__ void store_stats_body(const cmbpo_rollout_t &r) {
  __shared__ double sm_d[16];
  const int n = r.iscal[CMBPO_I_N_ALIVE];
  const int n_wg = (n + kStoreRows - 1) / kStoreRows;
  double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int w = threadIdx.x; w < n_wg; w += 256) {
    const double *src = r.store_part + (size_t)w * 8;
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] += src[k];
    t[6] = fmax(t[6], src[6]);
    t[7] = fmax(t[7], src[7]);
  }
  double s[8];
#pragma unroll
  for (int k = 0; k < 6; ++k) s[k] = block_sum(t[k], sm_d);
  s[6] = block_max(t[6], sm_d);
  s[7] = block_max(t[7], sm_d);
  if (threadIdx.x == 0 && s[0] > 0.0) {
    const double cnt = s[0];
    const double dkl_mean = r.dscal[CMBPO_D_DKL_SUM_T] / (double)n;  // np.mean over the rows stepped
    r.iscal[CMBPO_I_N_STORED] += (int)cnt;
    r.iscal[CMBPO_I_SIZE] += (int)cnt;
    r.dscal[CMBPO_D_TOTAL_SAMPLES] += cnt;
    r.dscal[CMBPO_D_TOTAL_COST] += s[1];
    r.dscal[CMBPO_D_TOTAL_REW] += s[2];
    r.dscal[CMBPO_D_SUM_PATH_RET] += s[2];
    r.dscal[CMBPO_D_SUM_PATH_COST] += s[1];
    r.dscal[CMBPO_D_TOTAL_VS] += s[3];
    r.dscal[CMBPO_D_TOTAL_CVS] += s[4];
    r.dscal[CMBPO_D_TOTAL_DYN_EP_VAR] += s[5];
    r.dscal[CMBPO_D_TOTAL_DKL] += dkl_mean * cnt;
    r.dscal[CMBPO_D_MAX_DKL] = fmax(r.dscal[CMBPO_D_MAX_DKL], s[6]);
    r.dscal[CMBPO_D_MAX_PATH_RETURN] = fmax(r.dscal[CMBPO_D_MAX_PATH_RETURN], s[7]);
  }
}

__global__ __launch_bounds__(256) void store_stats_kernel(const cmbpo_rollout_t r) { store_stats_body(r); }

// ---- small rollout batches: decide -> finish(PRE) -> store -> statistics as ONE workgroup --------------------------------
// At the shipped configurations' 1e3 - 1e4 branches each of the five kernels above is a few microseconds of work behind a
// launch boundary (decide 2 x 5, finish 5, store 9 + 6 us of a 140 us step at 1000 branches).  Up to kBookMax alive rows
// one 1024-thread workgroup walks them in a fixed order: same decisions, same per-branch arithmetic, the step's sums added
// in a fixed tree.  (Single-rank path: the cross-shard budget exchange keeps the separate kernels.)
// Round 3 (tools/sweep_book_max.sh): with the large-batch step's launches pipelined and its actor ahead of the host's wait, the
// separate kernels win from ~1000 rows on (4 % at 1250 and 2500 rows, 8 % at 4000; the one-workgroup path wins by 5 - 10 % at
// 400 - 700): the threshold came down from 4096.  (CMBPO_BOOK_MAX: 0 .. 65536; the kernels walk any count)
constexpr int kBookMaxDefault = 1024;
static const int kBookMax = [] {
  const char *e = getenv("CMBPO_BOOK_MAX");
  const int v = e ? atoi(e) : kBookMaxDefault;
  return v < 0 ? 0 : (v > 65536 ? 65536 : v);
}();

// spec (cmbpo_rollout_run's look-ahead): the step was enqueued before the host saw the previous step's counters -- it is void
// when that step raised the halt word (book_post_kernel)
__global__ __launch_bounds__(kScanThreads) void book_pre_kernel(const cmbpo_rollout_t r, int spec) {
  __shared__ int sm_i[17];
  __shared__ double sm_d[16];
  __shared__ double sm_m[8 * 16];
  __shared__ int s_unc;
  const int tid = threadIdx.x;
  if (spec && r.iscal[CMBPO_I_HALT]) return;
  const int n = r.iscal[CMBPO_I_N_ALIVE];
  // (1) uncertainty flags + this step's counters (decide_flags_kernel / head of decide_budget_kernel)
  {
    double c = 0.0, d = 0.0;
    for (int i = tid; i < n; i += kScanThreads) {
      const int b = r.alive_idx[i];
      const bool u = too_uncertain(r, b);
      r.fin_code[b] = u ? 1 : 0;
      c += u ? 1.0 : 0.0;
      d += (double)r.dkl_t[b];
    }
    double cd[2] = {c, d};
    block_reduce_many<2, 0>(cd, sm_m);     // (one pair of barriers for both sums)
    c = cd[0]; d = cd[1];
    if (tid == 0) {
      s_unc = (int)c;
      r.iscal[CMBPO_I_N_UNC] = (int)c;
      r.dscal[CMBPO_D_DKL_SUM_T] = d;
      r.iscal[CMBPO_I_N_STORED] = 0;
      r.iscal[CMBPO_I_N_FIN_POST] = 0;
      r.iscal[8] = n; r.iscal[9] = (int)c; r.iscal[10] = (int)r.dscal[CMBPO_D_TOTAL_SAMPLES]; r.iscal[11] = 0;
      r.iscal[CMBPO_I_N_FIN_PRE] = (int)c;
    }
    __syncthreads();
  }
  const int n_unc = s_unc;
  // (2) budget rule: the first `excess` surviving rows in index order are finished too (model_sampler.py:282-287)
  if (r.max_samples != 0) {
    const long long excess = (long long)r.dscal[CMBPO_D_TOTAL_SAMPLES] + n - n_unc - (long long)r.max_samples;
    if (excess > 0) {
      int carry = 0, nfin = 0;
      for (int base = 0; base < n; base += kScanThreads) {
        const int i = base + tid;
        int b = -1, flag = 0;
        if (i < n) {
          b = r.alive_idx[i];
          flag = r.fin_code[b] ? 0 : 1;
        }
        int total;
        const int excl = block_excl_scan(flag, sm_i, &total);
        if (flag && ((long long)carry + excl < excess)) {
          r.fin_code[b] = 1;
          nfin += 1;
        }
        carry += total;
        if ((long long)carry >= excess) break;   // uniform: carry is a block-wide value
      }
      const double nf = block_sum((double)nfin, sm_d);
      if (tid == 0) r.iscal[CMBPO_I_N_FIN_PRE] = n_unc + (int)nf;
    }
  }
  // (every thread set the codes of its own rows only: rows i = tid, tid + 1024, ... in all three passes)
  // (3) finish(PRE) + (4) the scalar half of the store
  const size_t B = (size_t)r.B;
  const int D = r.obs_dim;
  double a_cnt = 0, a_cost = 0, a_rew = 0, a_v = 0, a_vc = 0, a_epv = 0, a_maxdkl = 0, a_maxret = 0;
  for (int i = tid; i < n; i += kScanThreads) {
    const int b = r.alive_idx[i];
    const uint8_t code = r.fin_code[b];
    if (code) {
      gae_finish_path(r, b, code == 2 ? 0.0f : r.v_t[b], r.vc_t[b], code == 2);
      continue;
    }
    const size_t col = (size_t)r.ptr * B + b;
    const float rw = r.rew_t[b], c = r.cost_t[b], v = r.v_t[b], vc = r.vc_t[b];
    const float epv = r.epv_t[b], dk = r.dkl_t[b];
    r.rew_buf[col] = rw;
    r.val_buf[col] = v;
    r.cost_buf[col] = c;
    r.cval_buf[col] = vc;
    r.logp_buf[col] = r.logp_t[b];
    r.len[b] = r.ptr + 1;
    const double pr = r.path_ret[b] + (double)rw;
    r.path_ret[b] = pr;
    r.path_cost[b] += (double)c;
    r.path_dyn_var[b] += (double)epv;
    r.dkl_acc[b] += (double)dk;
    a_cnt += 1.0; a_cost += c; a_rew += rw; a_v += v; a_vc += vc;
    a_epv += (double)epv * D;
    a_maxdkl = fmax(a_maxdkl, (double)dk);
    a_maxret = fmax(a_maxret, pr);
  }
  // (the vector fields -- obs, act, mu, log_std: 53 floats per row at AntSafe shapes -- are copied by store_vec_kernel on
  // many CUs: one workgroup moving them took 25 of this kernel's 34 us at 1000 rows)
  // (5) the step's sums into the accumulators (store_stats_kernel)
  // (the eight reductions behind one pair of barriers: sixteen of them were a third of this kernel at 1000 rows)
  double red[8] = {a_cnt, a_cost, a_rew, a_v, a_vc, a_epv, a_maxdkl, a_maxret};
  block_reduce_many<6, 2>(red, sm_m);
  const double s0 = red[0], s1 = red[1], s2 = red[2], s3 = red[3], s4 = red[4], s5 = red[5], s6 = red[6], s7 = red[7];
  if (tid == 0 && s0 > 0.0) {
    const double dkl_mean = r.dscal[CMBPO_D_DKL_SUM_T] / (double)n;  // np.mean over the rows stepped
    r.iscal[CMBPO_I_N_STORED] += (int)s0;
    r.iscal[CMBPO_I_SIZE] += (int)s0;
    r.dscal[CMBPO_D_TOTAL_SAMPLES] += s0;
    r.dscal[CMBPO_D_TOTAL_COST] += s1;
    r.dscal[CMBPO_D_TOTAL_REW] += s2;
    r.dscal[CMBPO_D_SUM_PATH_RET] += s2;
    r.dscal[CMBPO_D_SUM_PATH_COST] += s1;
    r.dscal[CMBPO_D_TOTAL_VS] += s3;
    r.dscal[CMBPO_D_TOTAL_CVS] += s4;
    r.dscal[CMBPO_D_TOTAL_DYN_EP_VAR] += s5;
    r.dscal[CMBPO_D_TOTAL_DKL] += dkl_mean * s0;
    r.dscal[CMBPO_D_MAX_DKL] = fmax(r.dscal[CMBPO_D_MAX_DKL], s6);
    r.dscal[CMBPO_D_MAX_PATH_RETURN] = fmax(r.dscal[CMBPO_D_MAX_PATH_RETURN], s7);
  }
}

// the vector half of the store for book_pre_kernel's decisions: 64-row tiles on as many workgroups
__global__ __launch_bounds__(256) void store_vec_kernel(const cmbpo_rollout_t r, int spec) {
  __shared__ int s_slot[kStoreRows];
  if (spec && r.iscal[CMBPO_I_HALT]) return;
  const int n = r.iscal[CMBPO_I_N_ALIVE];
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * kStoreRows;
  if (row0 >= n) return;
  if (tid < kStoreRows) {
    const int i = row0 + tid;
    int b = -1;
    if (i < n) {
      b = r.alive_idx[i];
      if (r.fin_code[b]) b = -1;
    }
    s_slot[tid] = b;
  }
  __syncthreads();
  const size_t B = (size_t)r.B;
  const int rows_here = min(kStoreRows, n - row0);
  const float *vsrc[4] = {r.cur_obs, r.act_t, r.mu_t, r.ls_t};
  float *vdst[4] = {r.obs_buf, r.act_buf, r.mu_buf, r.ls_buf};
  const int vdim[4] = {r.obs_dim, r.act_dim, r.act_dim, r.act_dim};
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int dim = vdim[f];
    const float inv = 1.0f / (float)dim;
    for (int e = tid; e < rows_here * dim; e += 256) {
      const int k = (int)(((float)e + 0.5f) * inv);   // exact for e < 2^16
      const int d = e - k * dim;
      const int b = s_slot[k];
      if (b >= 0) vdst[f][((size_t)r.ptr * B + b) * dim + d] = vsrc[f][(size_t)b * dim + d];
    }
  }
}

// finish(POST) + compaction as one workgroup (small rollout batches): the horizon / environment-terminal finishes of
// finish_kernel(mode 1), then the ordered alive list of the survivors into alive_idx_out.  The list is ALWAYS written (a
// copy when nothing finished), so the host swaps alive_idx <-> alive_idx_out after every step without a second look.
// host_out (optional): a host-mapped, coherent block of 128 dwords -- the step's counters (iscal | dscal, 96 dwords) are
// written straight into it, then a system-scope fence, then `seq` at dword 96: the host polls that word instead of paying a
// copy + a stream synchronisation per step (cmbpo_rollout_run)
// spec: the step may be void (see book_pre_kernel); a step that is not decides whether the NEXT one is -- the caller's stop tests
// (at most min_alive rows left, total_samples >= stop_total unless NaN; algorithms/cmbpo.py:356-359) are taken here, the halt
// word travels in the mirrored block and iscal[CMBPO_I_N_EFF] (the row count the next step's forward kernels read) drops to 0.
__global__ __launch_bounds__(kScanThreads) void book_post_kernel(const cmbpo_rollout_t r, uint32_t *host_out, uint32_t seq, int spec,
                                                                 int min_alive, double stop_total) {
  __shared__ int sm_i[17];
  __shared__ double sm_d[16];
  const int tid = threadIdx.x;
  if (spec && r.iscal[CMBPO_I_HALT]) return;
  const int n = r.iscal[CMBPO_I_N_ALIVE];
  const bool horizon = (r.ptr + 1 >= r.max_path_length - 1);     // path_length after the store, model_sampler.py:352
  int nfin = 0;
  for (int i = tid; i < n; i += kScanThreads) {
    const int b = r.alive_idx[i];
    if (r.fin_code[b]) continue;              // finished before the store
    if (horizon) {
      gae_finish_path(r, b, r.v_n[b], r.vc_n[b], false);
      ++nfin;
    } else if (r.term_t[b]) {
      gae_finish_path(r, b, 0.0f, r.vc_n[b], true);   // float64 zeros bootstrap; the cost value still bootstraps (:364)
      ++nfin;
    }
  }
  const double nf = block_sum((double)nfin, sm_d);
  if (tid == 0) r.iscal[CMBPO_I_N_FIN_POST] = (int)nf;
  __syncthreads();                              // (every thread marked its own rows only; the scan below re-reads them)
  int carry = 0;
  for (int base = 0; base < n; base += kScanThreads) {
    const int i = base + tid;
    int b = -1, flag = 0;
    if (i < n) {
      b = r.alive_idx[i];
      flag = r.alive[b] ? 1 : 0;
    }
    int total;
    const int excl = block_excl_scan(flag, sm_i, &total);
    if (flag) r.alive_idx_out[carry + excl] = b;
    carry += total;
  }
  if (tid == 0) {
    r.iscal[CMBPO_I_N_ALIVE_OUT] = carry;
    r.iscal[CMBPO_I_N_ALIVE] = carry;           // the caller swaps alive_idx <-> alive_idx_out
    if (spec) {
      const int halt = (carry <= 0 || carry <= min_alive || (stop_total == stop_total && r.dscal[CMBPO_D_TOTAL_SAMPLES] >= stop_total)) ? 1 : 0;
      r.iscal[CMBPO_I_HALT] = halt;
      r.iscal[CMBPO_I_N_EFF] = halt ? 0 : carry;
    }
    if (host_out != nullptr) {
      // (this thread wrote the last counters itself; the accumulators were left by book_pre_kernel, an earlier launch)
      const uint4 *src = reinterpret_cast<const uint4 *>(r.iscal);      // iscal[32] | dscal[32]: one 384-byte block
      uint4 *dst = reinterpret_cast<uint4 *>(host_out);
      uint4 q[24];
#pragma unroll
      for (int i = 0; i < 24; ++i) q[i] = src[i];
#pragma unroll
      for (int i = 0; i < 24; ++i) dst[i] = q[i];
      __threadfence_system();
      *reinterpret_cast<volatile uint32_t *>(host_out + 96) = seq;
    }
  }
}

// ---- compact: ordered alive list from the alive mask ---------------------------------------------
// Three short multi-workgroup kernels instead of one workgroup walking the whole list (123 us at 100 k branches, on
// every step of an 'uncertainty' rollout): per-chunk survivor counts, a one-workgroup exclusive scan of the counts,
// and the scatter.  The integer scratch is r.store_part (free again once the step's store has been folded).
__global__ __launch_bounds__(kScanThreads) void compact_count_kernel(const cmbpo_rollout_t r) {
  __shared__ int sm_i[17];
  const int n = r.iscal[CMBPO_I_N_ALIVE];
  const int i = blockIdx.x * kScanThreads + threadIdx.x;
  if ((int)blockIdx.x * kScanThreads >= n) return;
  int flag = 0;
  if (i < n) flag = r.alive[r.alive_idx[i]] ? 1 : 0;
  int total;
  (void)block_excl_scan(flag, sm_i, &total);
  if (threadIdx.x == 0) reinterpret_cast<int *>(r.store_part)[blockIdx.x] = total;
}

__global__ __launch_bounds__(kScanThreads) void compact_scan_kernel(const cmbpo_rollout_t r, int n_chunks_max) {
  __shared__ int sm_i[17];
  int *cnt = reinterpret_cast<int *>(r.store_part);
  const int n = r.iscal[CMBPO_I_N_ALIVE];
  const int n_chunks = (n + kScanThreads - 1) / kScanThreads;
  int carry = 0;
  for (int base = 0; base < n_chunks; base += kScanThreads) {
    const int c = base + threadIdx.x;
    const int v = (c < n_chunks) ? cnt[c] : 0;
    int total;
    const int excl = block_excl_scan(v, sm_i, &total);
    if (c < n_chunks) cnt[c] = carry + excl;
    carry += total;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    cnt[n_chunks_max] = n;                        // the scatter still needs the old length
    r.iscal[CMBPO_I_N_ALIVE_OUT] = carry;
    r.iscal[CMBPO_I_N_ALIVE] = carry;             // the caller swaps alive_idx <-> alive_idx_out
  }
}

__global__ __launch_bounds__(kScanThreads) void compact_scatter_kernel(const cmbpo_rollout_t r, int n_chunks_max) {
  __shared__ int sm_i[17];
  const int *cnt = reinterpret_cast<const int *>(r.store_part);
  const int n = cnt[n_chunks_max];
  if ((int)blockIdx.x * kScanThreads >= n) return;
  const int i = blockIdx.x * kScanThreads + threadIdx.x;
  int b = -1, flag = 0;
  if (i < n) {
    b = r.alive_idx[i];
    flag = r.alive[b] ? 1 : 0;
  }
  int total;
  const int excl = block_excl_scan(flag, sm_i, &total);
  if (flag) r.alive_idx_out[cnt[blockIdx.x] + excl] = b;
}

// ---- get(): offsets, moments, flatten ------------------------------------------------------------
// offsets = exclusive scan of the path lengths, as three short multi-workgroup kernels like the compaction (one
// workgroup walking 100 k lengths took 110 us): per-chunk sums, a one-workgroup scan of the sums, per-chunk scans.
// Integer scratch: r.store_part (idle between the rollout and the next reset).
__global__ __launch_bounds__(kScanThreads) void offsets_count_kernel(const cmbpo_rollout_t r) {
  __shared__ int sm_i[17];
  const int b = blockIdx.x * kScanThreads + threadIdx.x;
  int total;
  (void)block_excl_scan(b < r.B ? r.len[b] : 0, sm_i, &total);
  if (threadIdx.x == 0) reinterpret_cast<int *>(r.store_part)[blockIdx.x] = total;
}

__global__ __launch_bounds__(kScanThreads) void offsets_scan_kernel(const cmbpo_rollout_t r, int n_chunks, int32_t *offs) {
  __shared__ int sm_i[17];
  int *cnt = reinterpret_cast<int *>(r.store_part);
  int carry = 0;
  for (int base = 0; base < n_chunks; base += kScanThreads) {
    const int c = base + threadIdx.x;
    const int v = (c < n_chunks) ? cnt[c] : 0;
    int total;
    const int excl = block_excl_scan(v, sm_i, &total);
    if (c < n_chunks) cnt[c] = carry + excl;
    carry += total;
  }
  if (threadIdx.x == 0) offs[r.B] = carry;
}

__global__ __launch_bounds__(kScanThreads) void offsets_write_kernel(const cmbpo_rollout_t r, int32_t *offs) {
  __shared__ int sm_i[17];
  const int b = blockIdx.x * kScanThreads + threadIdx.x;
  int total;
  const int excl = block_excl_scan(b < r.B ? r.len[b] : 0, sm_i, &total);
  if (b < r.B) offs[b] = reinterpret_cast<const int *>(r.store_part)[blockIdx.x] + excl;
}

// stats layout: [0] n  [1] adv_mean  [2] adv_std  [3] cadv_mean  [4] ret_mean  [5] cret_mean
//               [8] n (raw)  [9] sum adv  [10] sum cadv  [11] sum ret  [12] sum cret  [13] sum (adv-mean)^2
// Every workgroup leaves its float64 partial sums in r.store_part[block][8]; moments_fold_kernel adds them in block
// order -- no atomics, the statistics (and with them the normalised advantages) are bitwise reproducible.
__global__ __launch_bounds__(256) void moments_kernel(const cmbpo_rollout_t r, int pass, const double *st) {
  __shared__ double sm_d[16];
  const size_t B = (size_t)r.B;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
  const float mean = (float)st[1];
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < r.B; b += gridDim.x * blockDim.x) {
    const int L = r.len[b];
    for (int t = 0; t < L; ++t) {
      const size_t o = (size_t)t * B + b;
      if (pass == 0) {
        s0 += 1.0; s1 += r.adv_buf[o]; s2 += r.cadv_buf[o]; s3 += r.ret_buf[o]; s4 += r.cret_buf[o];
      } else {
        const float d = __fsub_rn(r.adv_buf[o], mean);  // (x - mean)**2 in float32 (mpi_tools.py:86)
        s0 += (double)__fmul_rn(d, d);
      }
    }
  }
  double *part = r.store_part + (size_t)blockIdx.x * 8;
  const double a0 = block_sum(s0, sm_d);
  if (pass == 0) {
    const double a1 = block_sum(s1, sm_d), a2 = block_sum(s2, sm_d);
    const double a3 = block_sum(s3, sm_d), a4 = block_sum(s4, sm_d);
    if (threadIdx.x == 0) { part[0] = a0; part[1] = a1; part[2] = a2; part[3] = a3; part[4] = a4; }
  } else if (threadIdx.x == 0) {
    part[0] = a0;
  }
}

__global__ __launch_bounds__(64) void moments_fold_kernel(const cmbpo_rollout_t r, int pass, int n_parts, double *st) {
  // lane l adds parts l, l + 64, ... in that order, then the 64 lane sums are added in a fixed tree: same result every run
  const int lane = threadIdx.x;
  const int nk = pass == 0 ? 5 : 1;
  for (int k = 0; k < nk; ++k) {
    double acc = 0.0;
    for (int i = lane; i < n_parts; i += 64) acc += r.store_part[(size_t)i * 8 + k];
    acc = wave_sum(acc);
    if (lane == 0) st[pass == 0 ? 8 + k : 13] = acc;
  }
}

__global__ void moments_finalize(int pass, double *st) {
  if (pass == 1) {
    const double n = st[8];
    st[0] = n;
    st[1] = n > 0 ? st[9] / n : 0.0;
    st[3] = n > 0 ? st[10] / n : 0.0;
    st[4] = n > 0 ? st[11] / n : 0.0;
    st[5] = n > 0 ? st[12] / n : 0.0;
  } else {
    const double n = st[8];
    st[2] = n > 0 ? sqrt(st[13] / n) : 0.0;
  }
}

// ---- get() on one GPU: the scan and both moment passes as TWO launches ----------------------------------------------------
// (the separate kernels above remain for the sharded path, where an all-reduce stands between the passes).  A workgroup =
// 256 consecutive branches: its path-length sum and its float64 partial moments go to r.store_part[block][0..5]; the LAST
// workgroup to arrive (device-scope counter) scans the block sums, adds the partial moments in block order -- the statistics
// stay bitwise reproducible -- and finishes the means.  The second launch writes every branch's offset and the partial sums
// of (adv - mean)^2 (float32 differences, utilities/mpi_tools.py:86), and its last workgroup finishes the standard deviation.
// store_part row: [0] n  [1] sum adv  [2] sum cadv  [3] sum ret  [4] sum cret  [5] samples of the block  [6] its offset  [7] sum (adv - mean)^2
__device__ __forceinline__ bool last_block_arrives(int *counter) {
  __shared__ int s_last;
  __threadfence();                      // this workgroup's partial row is visible device-wide before the count
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(counter, 1) == (int)gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (s_last) __threadfence();          // acquire: the other workgroups' rows
  return s_last != 0;
}

// what the last workgroup of pass 1 does (256 threads): scan of the block sums, moments in block order, the means
__device__ __forceinline__ void get_fold1(const cmbpo_rollout_t &r, int nblk, int32_t *offs, double *st) {
  __shared__ int sm_i[17];
  const int tid = threadIdx.x;
  int carry = 0;
  for (int base = 0; base < nblk; base += 256) {
    const int c = base + tid;
    const int v = c < nblk ? (int)__builtin_nontemporal_load(&r.store_part[(size_t)c * 8 + 5]) : 0;
    int tot;
    const int excl = block_excl_scan(v, sm_i, &tot);
    if (c < nblk) r.store_part[(size_t)c * 8 + 6] = (double)(carry + excl);
    carry += tot;
  }
  if (tid < 64) {
    double acc[5];
    for (int k = 0; k < 5; ++k) {
      double a = 0.0;
      for (int i = tid; i < nblk; i += 64) a += __builtin_nontemporal_load(&r.store_part[(size_t)i * 8 + k]);
      acc[k] = wave_sum(a);
    }
    if (tid == 0) {
      const double n = acc[0];
      offs[r.B] = carry;
      for (int k = 0; k < 5; ++k) st[8 + k] = acc[k];
      st[13] = 0.0; st[2] = 0.0;
      st[0] = n;
      st[1] = n > 0 ? acc[1] / n : 0.0;
      st[3] = n > 0 ? acc[2] / n : 0.0;
      st[4] = n > 0 ? acc[3] / n : 0.0;
      st[5] = n > 0 ? acc[4] / n : 0.0;
    }
  }
}
__device__ __forceinline__ void get_fold2(const cmbpo_rollout_t &r, int nblk, double *st) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    double a = 0.0;
    for (int i = tid; i < nblk; i += 64) a += __builtin_nontemporal_load(&r.store_part[(size_t)i * 8 + 7]);
    a = wave_sum(a);
    if (tid == 0) {
      const double n = st[8];
      st[13] = a;
      st[2] = n > 0 ? sqrt(a / n) : 0.0;
    }
  }
}
__global__ __launch_bounds__(256) void get_fold_kernel(const cmbpo_rollout_t r, int pass, int nblk, int32_t *offs, double *st) {
  if (pass == 1) get_fold1(r, nblk, offs, st);
  else get_fold2(r, nblk, st);
}

// fold_here: the last workgroup to arrive folds (few workgroups: the returning atomics of hundreds of workgroups on one
// counter serialise at the memory side, ~50 ns each -- then a one-workgroup launch of get_fold_kernel is cheaper)
__global__ __launch_bounds__(256) void get_pass1_kernel(const cmbpo_rollout_t r, int32_t *offs, double *st, int fold_here) {
  __shared__ double sm_d[16];
  __shared__ int sm_i[17];
  const size_t B = (size_t)r.B;
  const int tid = threadIdx.x, b = blockIdx.x * 256 + tid;
  const int L = b < r.B ? r.len[b] : 0;
  double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
  for (int t = 0; t < L; ++t) {
    const size_t o = (size_t)t * B + b;
    s1 += r.adv_buf[o]; s2 += r.cadv_buf[o]; s3 += r.ret_buf[o]; s4 += r.cret_buf[o];
  }
  int total;
  (void)block_excl_scan(L, sm_i, &total);
  const double a1 = block_sum(s1, sm_d), a2 = block_sum(s2, sm_d), a3 = block_sum(s3, sm_d), a4 = block_sum(s4, sm_d);
  double *part = r.store_part + (size_t)blockIdx.x * 8;
  if (tid == 0) { part[0] = (double)total; part[1] = a1; part[2] = a2; part[3] = a3; part[4] = a4; part[5] = (double)total; }
  if (!fold_here || !last_block_arrives(&r.iscal[30])) return;
  get_fold1(r, gridDim.x, offs, st);
  if (tid == 0) r.iscal[30] = 0;        // ready for the next get()
}

__global__ __launch_bounds__(256) void get_pass2_kernel(const cmbpo_rollout_t r, int32_t *offs, double *st, int fold_here) {
  __shared__ double sm_d[16];
  __shared__ int sm_i[17];
  const size_t B = (size_t)r.B;
  const int tid = threadIdx.x, b = blockIdx.x * 256 + tid;
  const int L = b < r.B ? r.len[b] : 0;
  const float mean = (float)st[1];
  double s0 = 0;
  for (int t = 0; t < L; ++t) {
    const float d = __fsub_rn(r.adv_buf[(size_t)t * B + b], mean);  // (x - mean)**2 in float32 (mpi_tools.py:86)
    s0 += (double)__fmul_rn(d, d);
  }
  int total;
  const int excl = block_excl_scan(L, sm_i, &total);
  double *part = r.store_part + (size_t)blockIdx.x * 8;
  if (b < r.B) offs[b] = (int)part[6] + excl;
  const double a0 = block_sum(s0, sm_d);
  if (tid == 0) part[7] = a0;
  if (!fold_here || !last_block_arrives(&r.iscal[31])) return;
  get_fold2(r, gridDim.x, st);
  if (tid == 0) r.iscal[31] = 0;
}

// flatten: time-major [t][b][.] buffers -> the reference's branch-major / time-minor list (modelbuffer.py:212-218).
// Both sides of the copy want long contiguous runs: in the buffers the rows of consecutive BRANCHES at one step are
// adjacent, in the output the rows of consecutive STEPS of one branch are.  So a workgroup takes a tile of branches, reads
// step by step (runs of tile x dim floats, one float per lane, consecutive lanes on consecutive addresses), transposes
// through LDS ([branch][step][dim]) and writes each branch's samples as one run -- the runs of a tile are adjacent, the
// tile's output is a single contiguous block.  (Reading the output order straight from the buffers touched a 116-byte run
// per sample: 2.9 TB/s.)  Vector fields: 16-branch tiles (63 KB of LDS for obs at T = 34); scalar fields: 64-branch tiles.
#ifndef FLAT_WAVES
#define FLAT_WAVES 3
#endif
#ifndef FLAT_WIDE_TILES
#define FLAT_WIDE_TILES 1      // 0: 16-branch tiles whatever the rollout's length (diagnostic)
#endif
constexpr int kVecTile = FLAT_WIDE_TILES ? 64 : 16;      // (16 at 34 steps of AntSafe shapes by the 64 KB rule below; 64 after a rollout of <= 8 steps)
constexpr int kFlatRows = 64;

struct FlatArgs {
  float *out[12];
};

// one vector field of the tile: buffer -> LDS ([branch][step][dim]).  Steps go round-robin over the four waves, NU steps
// per wave and pass, and every load of a pass is requested before the first LDS write: a workgroup pays the HBM latency
// once per pass, not once per step (at T = 34 one pass covers the tile).
template <int NU, int KV>
__device__ __forceinline__ void flat_vec_in(float *tile, const float *src, int dim, int nb, int T, int lmax, size_t B, int lane,
                                            int wave, int tid) {
  // KV 16-byte loads per lane and step: runs of up to 256 KV floats (2: 16-branch tiles of long rollouts; 8: the 64-branch
  // tiles of short ones)
  const int run = nb * dim;                 // floats of the tile at one step: contiguous in the buffer
  const int c = (T - 1) * dim;
  // step t, element j = bl * dim + d  ->  tile[(bl * T + t) * dim + d] = tile[j + bl * c + t * dim]
  if ((run & 3) == 0 && run <= 256 * KV && ((B * dim) & 3) == 0) {
    int j0[KV], d0[KV], i0[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      j0[k] = 4 * (lane + 64 * k);
      const int jj = j0[k] < run ? j0[k] : 0;
      const int bl = jj / dim;
      d0[k] = jj - bl * dim;
      i0[k] = jj + bl * c;
    }
    const int kmax = (run + 255) >> 8;      // loads per lane that reach into the run (wave-uniform)
    for (int tb = wave; tb < lmax; tb += 4 * NU) {
      f32x4 v[NU][KV];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int t = min(tb + 4 * u, T - 1);
        const float *st = src + (size_t)t * B * dim;
        if constexpr (KV == 2) {     // (the long-rollout form exactly as it was: one flag, hoisted out of the loop)
          v[u][0] = *reinterpret_cast<const f32x4 *>(st + (j0[0] < run ? j0[0] : 0));
          if (kmax > 1) v[u][1] = *reinterpret_cast<const f32x4 *>(st + (j0[1] < run ? j0[1] : 0));
        } else {
#pragma unroll
          for (int k = 0; k < KV; ++k)
            if (k == 0 || k < kmax) v[u][k] = *reinterpret_cast<const f32x4 *>(st + (j0[k] < run ? j0[k] : 0));
        }
      }
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int k = 0; k < KV; ++k)
          if (tb + 4 * u < lmax && j0[k] < run && (k == 0 || k < kmax)) {
            float *q = tile + i0[k] + (tb + 4 * u) * dim;
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i + (d0[k] + i >= dim ? c : 0)] = v[u][k][i];
          }
    }
  } else {
    for (int e = tid; e < lmax * run; e += 256) {
      const int t = e / run, j = e - t * run;
      tile[j + ((j / dim) * (T - 1) + t) * dim] = src[(size_t)t * B * dim + j];
    }
  }
}

// LDS -> output: the tile's samples form ONE contiguous block of the output (branch-major: the runs of consecutive
// branches are adjacent); sample p of the block is step t of branch bl, smap[p] = bl * T + t.  Every path full: the tile
// already is the block.  Ragged paths: the block is written element by element in output order (consecutive lanes on
// consecutive addresses whatever the path lengths are -- one short run per branch and wave left an 'uncertainty' rollout's
// get() at 0.1 of the HBM rate).
__device__ __forceinline__ void flat_vec_out(const float *tile, float *dst, int dim, int nb, int T, int cnt, int o0,
                                             const unsigned short *smap, int tid) {
  const int n = cnt * dim;
  if (cnt == nb * T) {
    if ((((size_t)o0 * dim) & 3) == 0 && (n & 3) == 0) {
      for (int e = 4 * tid; e < n; e += 1024) *reinterpret_cast<f32x4 *>(dst + e) = *reinterpret_cast<const f32x4 *>(tile + e);
    } else {
      for (int e = tid; e < n; e += 256) dst[e] = tile[e];
    }
  } else {
    const float inv = 1.0f / (float)dim;
    for (int e = tid; e < n; e += 256) {
      int p = (int)(((float)e + 0.5f) * inv);       // e / dim (exact: e < 2^18)
      p -= (p * dim > e) ? 1 : 0;
      p += ((p + 1) * dim <= e) ? 1 : 0;
      dst[e] = tile[(int)smap[p] * dim + (e - p * dim)];
    }
  }
}

// SHORT: the rollout ended within 8 steps -- tiles of up to 64 branches through the 2-step x 8-load pass; else tiles of up to
// 16 branches through the 9-step x 2-load pass (one instantiation of the kernel each: with both passes in one body the
// long-rollout shapes lost 7 % of their flatten)
template <bool SHORT>
__device__ __forceinline__ void flatten_vec_body(const cmbpo_rollout_t &r, const int32_t *offs, const FlatArgs &fa, int vt, int Tt,
                                                 int blk) {
  constexpr int VT_MAX = SHORT ? kVecTile : 16;
  // Tt = the steps the rollout took (<= r.T): no path is longer, so the tiles are laid out -- and the LDS sized -- for Tt
  // steps (an 'uncertainty' rollout that ended after 5 of 34 steps: 9 KB per workgroup instead of 63, eight workgroups
  // per CU instead of two)
  extern __shared__ float tile[];          // [vt][Tt][obs_dim], then [3][vt][Tt][act_dim]; vt <= kVecTile branches
  __shared__ int lens[VT_MAX], loffs[VT_MAX + 1];
  // (behind the tiles: output position inside the tile's block -> bl * Tt + t, vt * Tt entries)
  unsigned short *smap = reinterpret_cast<unsigned short *>(tile + (size_t)vt * Tt * max(r.obs_dim, 3 * r.act_dim));
  const int b0 = blk * vt;
  const int nb = min(r.B - b0, vt);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < VT_MAX) lens[tid] = tid < nb ? r.len[b0 + tid] : 0;
  if (tid <= VT_MAX) loffs[tid] = offs[min(b0 + tid, r.B)];
  __syncthreads();
  const int o0 = loffs[0], cnt = loffs[nb] - o0;
  if (cnt == 0) return;
  int lmax = 0;
  for (int i = 0; i < nb; ++i) lmax = max(lmax, lens[i]);      // (broadcast reads, pipelined: cheaper than a shuffle tree)
  const size_t B = (size_t)r.B;
  const int T = Tt, D = r.obs_dim, A = r.act_dim;
  if (cnt != nb * T) {
    for (int e = tid; e < nb * T; e += 256) {
      const int bl = e / T, t = e - bl * T;
      if (t < lens[bl]) smap[loffs[bl] - o0 + t] = (unsigned short)e;
    }
  }
  // 4 waves x NU steps per pass, every load of a pass in flight before the first LDS write: 9 covers T <= 36 in one pass; a
  // rollout that ended after a few steps ('uncertainty' mode) takes the 2-step form -- steps past the last are clamped
  // copies of it, and seven of nine were
  auto body = [&](auto NUC, auto KVC) {
    constexpr int NU = decltype(NUC)::value, KV = decltype(KVC)::value;
    // obs (output 0)
    flat_vec_in<NU, KV>(tile, r.obs_buf + (size_t)b0 * D, D, nb, T, lmax, B, lane, wave, tid);
    __syncthreads();
    flat_vec_out(tile, fa.out[0] + (size_t)o0 * D, D, nb, T, cnt, o0, smap, tid);
    __syncthreads();
    // act (1), log_std (10), mu (11): three tiles side by side
    const size_t ts = (size_t)vt * T * A;
    flat_vec_in<NU, KV>(tile, r.act_buf + (size_t)b0 * A, A, nb, T, lmax, B, lane, wave, tid);
    flat_vec_in<NU, KV>(tile + ts, r.ls_buf + (size_t)b0 * A, A, nb, T, lmax, B, lane, wave, tid);
    flat_vec_in<NU, KV>(tile + 2 * ts, r.mu_buf + (size_t)b0 * A, A, nb, T, lmax, B, lane, wave, tid);
    __syncthreads();
    flat_vec_out(tile, fa.out[1] + (size_t)o0 * A, A, nb, T, cnt, o0, smap, tid);
    flat_vec_out(tile + ts, fa.out[10] + (size_t)o0 * A, A, nb, T, cnt, o0, smap, tid);
    flat_vec_out(tile + 2 * ts, fa.out[11] + (size_t)o0 * A, A, nb, T, cnt, o0, smap, tid);
  };
  if constexpr (SHORT) body(std::integral_constant<int, 2>{}, std::integral_constant<int, FLAT_WIDE_TILES ? 8 : 2>{});
  else body(std::integral_constant<int, 9>{}, std::integral_constant<int, 2>{});
}

// ROWS branches per workgroup: 64 (16 lanes x 16 bytes cover a step's row of the tile), or 16 for buffers too small to give
// every CU a 64-branch tile (10 000 branches are 157 tiles of 64 on 256 CUs)
template <int ROWS>
__device__ __forceinline__ void flatten_scalar_body(const cmbpo_rollout_t &r, const int32_t *offs, const double *st,
                                                    const FlatArgs &fa, int Tt, int blk) {
  extern __shared__ float tile[];          // [8][ROWS][Tt + 1] | position map [ROWS * Tt] (ushort)
  __shared__ int loffs[ROWS + 1];
  const int b0 = blk * ROWS;
  const int nb = min(r.B - b0, ROWS);
  const int tid = threadIdx.x;
  if (tid <= ROWS) loffs[tid] = offs[min(b0 + tid, r.B)];
  __syncthreads();
  const int o0 = loffs[0], cnt = loffs[nb] - o0;
  if (cnt == 0) return;
  const size_t B = (size_t)r.B;
  const int T = Tt, TS = T + 1;
  const size_t fs = (size_t)ROWS * TS;               // floats of one field's tile
  unsigned short *map = reinterpret_cast<unsigned short *>(tile + 8 * fs);   // output position -> bl * TS + t
  int lmax = 0;
  for (int e = tid; e < nb * T; e += 256) {
    const int bl = e / T, t = e - bl * T;
    if (t < loffs[bl + 1] - loffs[bl]) map[loffs[bl] - o0 + t] = (unsigned short)(bl * TS + t);
  }
  for (int i = 0; i < nb; ++i) lmax = max(lmax, loffs[i + 1] - loffs[i]);
  const float adv_mean = (float)st[1], adv_den = (float)st[2] + 1e-8f, cadv_mean = (float)st[3];
  // adv 2, cadv 3, ret 4, cret 5, logp 6, val 7, cval 8, cost 9: all eight fields in one pass
  const float *ssrc[8] = {r.adv_buf, r.cadv_buf, r.ret_buf, r.cret_buf, r.logp_buf, r.val_buf, r.cval_buf, r.cost_buf};
  if (nb == ROWS && (B & 3) == 0) {
    // LPS lanes x 16 bytes = the tile's branches of one step; steps ts, ts + G, ...; every load of a pass is requested
    // before the first LDS write
    constexpr int LPS = ROWS / 4, G = 256 / LPS;
    const int l16 = tid % LPS, ts = tid / LPS;
    auto pass = [&](auto NUC) {
      constexpr int NUS = decltype(NUC)::value;
      for (int tb = ts; tb < lmax; tb += G * NUS) {
        f32x4 v[8][NUS];
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
          for (int u = 0; u < NUS; ++u) {
            const int t = min(tb + G * u, T - 1);
            v[g][u] = *reinterpret_cast<const f32x4 *>(ssrc[g] + (size_t)t * B + b0 + 4 * l16);
          }
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
          for (int u = 0; u < NUS; ++u)
            if (tb + G * u < lmax) {
              float *tl = tile + g * fs + (4 * l16) * TS + tb + G * u;
#pragma unroll
              for (int i = 0; i < 4; ++i) tl[i * TS] = v[g][u][i];
            }
      }
    };
    // (64-branch tiles: 16 step groups x 3 = T <= 48 in one pass; T <= 16 needs one step per group, not two clamped copies more)
    if (ROWS == 64 && T > G) pass(std::integral_constant<int, 3>{});
    else pass(std::integral_constant<int, 1>{});
  } else {
    const int bl_in = tid % ROWS, tq = tid / ROWS;    // lanes over branches, the rest over steps
    for (int g = 0; g < 8; ++g) {
      const float *src = ssrc[g] + b0 + min(bl_in, nb - 1);
      float *tl = tile + g * fs + bl_in * TS;
      for (int t = tq; t < lmax; t += 256 / ROWS) {
        const float x = src[(size_t)t * B];
        if (bl_in < nb) tl[t] = x;
      }
    }
  }
  __syncthreads();
  for (int pq = tid; pq < cnt; pq += 256) {
    const int code = map[pq];
    const size_t o = (size_t)o0 + pq;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      float v = tile[g * fs + code];
      if (g == 0) v = __fsub_rn(v, adv_mean) / adv_den;     // modelbuffer.py:199
      if (g == 1) v = __fsub_rn(v, cadv_mean);              // modelbuffer.py:204
      fa.out[2 + g][o] = v;
    }
  }
}

int check_rollout(const cmbpo_rollout_t *r, const char *who) {
  CMBPO_REQUIRE(r != nullptr, "%s: rollout struct is NULL", who);
  CMBPO_REQUIRE(r->B >= 1 && r->T >= 1 && r->T <= 255 && r->obs_dim >= 1 && r->act_dim >= 1,
                "%s: bad sizes B=%d T=%d obs=%d act=%d", who, r->B, r->T, r->obs_dim, r->act_dim);
  CMBPO_REQUIRE(r->alive_idx && r->alive_idx_out && r->iscal && r->dscal && r->alive && r->fin_code && r->len,
                "%s: NULL state array", who);
  CMBPO_REQUIRE(r->ptr >= 0 && r->ptr <= r->T, "%s: ptr %d outside [0, T=%d]", who, r->ptr, r->T);
  return CMBPO_OK;
}

}  // namespace

extern "C" int cmbpo_rollout_reset(const cmbpo_rollout_t *r, void *stream) {
  if (int rc = check_rollout(r, "cmbpo_rollout_reset")) return rc;
  CMBPO_REQUIRE(r->dkl_acc && r->path_ret && r->path_cost && r->path_dyn_var, "cmbpo_rollout_reset: NULL accumulators");
  const int n = r->B > 32 ? r->B : 32;
  hipLaunchKernelGGL(reset_kernel, dim3(cmbpo_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, *r);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

static int launch_decide(const cmbpo_rollout_t *r, int count_only, hipStream_t s) {
  const int blocks = cmbpo_ceil_div(r->B, 256) < 512 ? cmbpo_ceil_div(r->B, 256) : 512;
  hipLaunchKernelGGL(decide_flags_kernel, dim3(blocks), dim3(256), 0, s, *r);
  hipLaunchKernelGGL(decide_budget_kernel, dim3(1), dim3(kScanThreads), 0, s, *r, count_only, blocks);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

extern "C" int cmbpo_rollout_decide(const cmbpo_rollout_t *r, void *stream) {
  if (int rc = check_rollout(r, "cmbpo_rollout_decide")) return rc;
  CMBPO_REQUIRE(r->dkl_t && r->dkl_acc && r->store_part, "cmbpo_rollout_decide: NULL dkl arrays / scratch");
  CMBPO_REQUIRE(r->world >= 1 && r->rank >= 0 && r->rank < r->world, "cmbpo_rollout_decide: bad rank/world");
  return launch_decide(r, 0, (hipStream_t)stream);
}

extern "C" int cmbpo_rollout_count(const cmbpo_rollout_t *r, void *stream) {
  if (int rc = check_rollout(r, "cmbpo_rollout_count")) return rc;
  CMBPO_REQUIRE(r->dkl_t && r->dkl_acc && r->store_part, "cmbpo_rollout_count: NULL dkl arrays / scratch");
  return launch_decide(r, 1, (hipStream_t)stream);
}

extern "C" int cmbpo_rollout_finish(const cmbpo_rollout_t *r, int mode, void *stream) {
  if (int rc = check_rollout(r, "cmbpo_rollout_finish")) return rc;
  CMBPO_REQUIRE(mode >= 0 && mode <= 2, "cmbpo_rollout_finish: bad mode %d", mode);
  CMBPO_REQUIRE(r->rew_buf && r->val_buf && r->cost_buf && r->cval_buf && r->adv_buf && r->ret_buf &&
                    r->cadv_buf && r->cret_buf,
                "cmbpo_rollout_finish: NULL buffer");
  if (mode == 1) CMBPO_REQUIRE(r->v_n && r->vc_n && r->term_t, "cmbpo_rollout_finish: POST needs v_n, vc_n, term_t");
  else CMBPO_REQUIRE(r->v_t && r->vc_t, "cmbpo_rollout_finish: needs v_t, vc_t");
  hipLaunchKernelGGL(finish_kernel, dim3(cmbpo_ceil_div(r->B, 256)), dim3(256), 0, (hipStream_t)stream, *r, mode, 0);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

extern "C" int cmbpo_rollout_store(const cmbpo_rollout_t *r, void *stream) {
  if (int rc = check_rollout(r, "cmbpo_rollout_store")) return rc;
  CMBPO_REQUIRE(r->ptr < r->T, "cmbpo_rollout_store: buffer full (ptr %d == T)", r->ptr);  // modelbuffer.py:115
  CMBPO_REQUIRE(r->cur_obs && r->act_t && r->logp_t && r->mu_t && r->ls_t && r->v_t && r->vc_t && r->rew_t &&
                    r->cost_t && r->dkl_t && r->epv_t,
                "cmbpo_rollout_store: NULL step array");
  CMBPO_REQUIRE(r->obs_buf && r->act_buf && r->mu_buf && r->ls_buf && r->rew_buf && r->val_buf && r->cost_buf &&
                    r->cval_buf && r->logp_buf,
                "cmbpo_rollout_store: NULL buffer");
  CMBPO_REQUIRE(r->store_part != nullptr, "cmbpo_rollout_store: NULL store_part scratch");
  hipLaunchKernelGGL(store_kernel, dim3(cmbpo_ceil_div(r->B, kStoreRows)), dim3(256), 0, (hipStream_t)stream, *r);
  hipLaunchKernelGGL(store_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *r);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

// the rollout step at large batches: the store without its statistics launch, and the finish(POST) that folds them
int cmbpo_internal_store_nostats(const cmbpo_rollout_t *r, void *stream) {
  if (int rc = check_rollout(r, "cmbpo_rollout_store")) return rc;
  CMBPO_REQUIRE(r->ptr < r->T, "cmbpo_rollout_store: buffer full (ptr %d == T)", r->ptr);
  CMBPO_REQUIRE(r->cur_obs && r->act_t && r->logp_t && r->mu_t && r->ls_t && r->v_t && r->vc_t && r->rew_t && r->cost_t && r->dkl_t &&
                    r->epv_t && r->obs_buf && r->act_buf && r->mu_buf && r->ls_buf && r->rew_buf && r->val_buf && r->cost_buf &&
                    r->cval_buf && r->logp_buf && r->store_part,
                "cmbpo_rollout_store: NULL array");
  hipLaunchKernelGGL(store_kernel, dim3(cmbpo_ceil_div(r->B, kStoreRows)), dim3(256), 0, (hipStream_t)stream, *r);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}
int cmbpo_internal_finish_post_fold(const cmbpo_rollout_t *r, void *stream) {
  if (int rc = check_rollout(r, "cmbpo_rollout_finish")) return rc;
  CMBPO_REQUIRE(r->v_n && r->vc_n && r->term_t && r->rew_buf && r->val_buf && r->cost_buf && r->cval_buf && r->adv_buf && r->ret_buf &&
                    r->cadv_buf && r->cret_buf && r->store_part,
                "cmbpo_rollout_finish (POST): NULL array");
  hipLaunchKernelGGL(finish_kernel, dim3(cmbpo_ceil_div(r->B, 256)), dim3(256), 0, (hipStream_t)stream, *r, 1, 1);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

extern "C" int cmbpo_rollout_book_pre_max_rows(void) { return kBookMax; }

extern "C" int cmbpo_rollout_book_pre(const cmbpo_rollout_t *r, int n_alive, void *stream) {
  if (int rc = check_rollout(r, "cmbpo_rollout_book_pre")) return rc;
  CMBPO_REQUIRE(n_alive >= 0 && n_alive <= kBookMax, "cmbpo_rollout_book_pre: %d alive rows, at most %d", n_alive, kBookMax);
  CMBPO_REQUIRE(!r->use_host_budget, "cmbpo_rollout_book_pre: the cross-shard budget exchange uses the separate kernels");
  CMBPO_REQUIRE(r->ptr < r->T, "cmbpo_rollout_book_pre: buffer full (ptr %d == T)", r->ptr);
  CMBPO_REQUIRE(r->dkl_t && r->dkl_acc && r->cur_obs && r->act_t && r->logp_t && r->mu_t && r->ls_t && r->v_t && r->vc_t &&
                    r->rew_t && r->cost_t && r->epv_t,
                "cmbpo_rollout_book_pre: NULL step array");
  CMBPO_REQUIRE(r->obs_buf && r->act_buf && r->mu_buf && r->ls_buf && r->rew_buf && r->val_buf && r->cost_buf && r->cval_buf &&
                    r->logp_buf && r->adv_buf && r->ret_buf && r->cadv_buf && r->cret_buf,
                "cmbpo_rollout_book_pre: NULL buffer");
  if (n_alive == 0) return CMBPO_OK;
  hipLaunchKernelGGL(book_pre_kernel, dim3(1), dim3(kScanThreads), 0, (hipStream_t)stream, *r, 0);
  hipLaunchKernelGGL(store_vec_kernel, dim3(cmbpo_ceil_div(n_alive, kStoreRows)), dim3(256), 0, (hipStream_t)stream, *r, 0);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

// the same inside cmbpo_rollout_step / _run.  spec: the step was enqueued ahead of the previous step's counters (n_alive is an
// upper bound then).  with_vec == 0: the vector half of the store rides in the critics' launch (CmbpoStoreVec) instead of
// store_vec_kernel.
int cmbpo_internal_book_pre(const cmbpo_rollout_t *r, int n_alive, int spec, int with_vec, void *stream) {
  if (int rc = check_rollout(r, "cmbpo_rollout_book_pre")) return rc;
  CMBPO_REQUIRE(n_alive >= 1 && n_alive <= kBookMax && !r->use_host_budget && r->ptr < r->T, "book_pre (step): bad state");
  CMBPO_REQUIRE(r->dkl_t && r->dkl_acc && r->cur_obs && r->act_t && r->logp_t && r->mu_t && r->ls_t && r->v_t && r->vc_t && r->rew_t &&
                    r->cost_t && r->epv_t && r->obs_buf && r->act_buf && r->mu_buf && r->ls_buf && r->rew_buf && r->val_buf &&
                    r->cost_buf && r->cval_buf && r->logp_buf && r->adv_buf && r->ret_buf && r->cadv_buf && r->cret_buf,
                "cmbpo_rollout_book_pre: NULL array");
  hipLaunchKernelGGL(book_pre_kernel, dim3(1), dim3(kScanThreads), 0, (hipStream_t)stream, *r, spec);
  if (with_vec)
    hipLaunchKernelGGL(store_vec_kernel, dim3(cmbpo_ceil_div(n_alive, kStoreRows)), dim3(256), 0, (hipStream_t)stream, *r, spec);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

// look-ahead bookkeeping of cmbpo_rollout_run: begin = no halt, the forward kernels' row count = the alive count; end = no halt
__global__ void spec_words_kernel(const cmbpo_rollout_t r, int begin) {
  r.iscal[CMBPO_I_HALT] = 0;
  r.iscal[CMBPO_I_N_EFF] = begin ? r.iscal[CMBPO_I_N_ALIVE] : 0;
}
int cmbpo_internal_spec_words(const cmbpo_rollout_t *r, int begin, void *stream) {
  CMBPO_REQUIRE(r && r->iscal, "look-ahead words: NULL argument");
  hipLaunchKernelGGL(spec_words_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, *r, begin);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

extern "C" int cmbpo_rollout_book_post(const cmbpo_rollout_t *r, int n_alive, void *stream) {
  if (int rc = check_rollout(r, "cmbpo_rollout_book_post")) return rc;
  CMBPO_REQUIRE(n_alive >= 0 && n_alive <= kBookMax, "cmbpo_rollout_book_post: %d alive rows, at most %d", n_alive, kBookMax);
  CMBPO_REQUIRE(r->v_n && r->vc_n && r->term_t && r->rew_buf && r->val_buf && r->cost_buf && r->cval_buf && r->adv_buf &&
                    r->ret_buf && r->cadv_buf && r->cret_buf,
                "cmbpo_rollout_book_post: NULL array");
  hipLaunchKernelGGL(book_post_kernel, dim3(1), dim3(kScanThreads), 0, (hipStream_t)stream, *r, (uint32_t *)nullptr, 0u, 0, 0, 0.0);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

// the counters into the host mirror behind whatever ran before on the stream (large batches: the step ends with multi-workgroup
// kernels, so a one-thread launch does what book_post_kernel's tail does for small ones)
__global__ void scalars_mirror_kernel(const cmbpo_rollout_t r, uint32_t *host_out, uint32_t seq) {
  const uint4 *src = reinterpret_cast<const uint4 *>(r.iscal);
  uint4 *dst = reinterpret_cast<uint4 *>(host_out);
  uint4 q[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) q[i] = src[i];
#pragma unroll
  for (int i = 0; i < 24; ++i) dst[i] = q[i];
  __threadfence_system();
  *reinterpret_cast<volatile uint32_t *>(host_out + 96) = seq;
}
int cmbpo_internal_scalars_mirror(const cmbpo_rollout_t *r, uint32_t *d_host_out, uint32_t seq, void *stream) {
  CMBPO_REQUIRE(r && r->iscal && r->dscal && d_host_out, "scalars mirror: NULL argument");
  CMBPO_REQUIRE(reinterpret_cast<const char *>(r->dscal) == reinterpret_cast<const char *>(r->iscal) + 128,
                "scalars mirror: iscal[32] and dscal[32] must be one 384-byte block");
  hipLaunchKernelGGL(scalars_mirror_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, *r, d_host_out, seq);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

// the same with the counters mirrored into host-mapped memory (d_host_out: device address of 128 dwords) and `seq` behind them
// spec: see book_post_kernel
int cmbpo_internal_book_post_mirror(const cmbpo_rollout_t *r, int n_alive, uint32_t *d_host_out, uint32_t seq, int spec, int min_alive,
                                    double stop_total, void *stream) {
  if (int rc = check_rollout(r, "cmbpo_rollout_book_post")) return rc;
  CMBPO_REQUIRE(n_alive >= 0 && n_alive <= kBookMax, "cmbpo_rollout_book_post: %d alive rows, at most %d", n_alive, kBookMax);
  CMBPO_REQUIRE(r->v_n && r->vc_n && r->term_t && r->rew_buf && r->val_buf && r->cost_buf && r->cval_buf && r->adv_buf &&
                    r->ret_buf && r->cadv_buf && r->cret_buf,
                "cmbpo_rollout_book_post: NULL array");
  CMBPO_REQUIRE(reinterpret_cast<const char *>(r->dscal) == reinterpret_cast<const char *>(r->iscal) + 128,
                "cmbpo_rollout_book_post: iscal[32] and dscal[32] must be one 384-byte block");
  hipLaunchKernelGGL(book_post_kernel, dim3(1), dim3(kScanThreads), 0, (hipStream_t)stream, *r, d_host_out, seq, spec, min_alive,
                     stop_total);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

// the step's counters and accumulators (32 int32 + 32 float64, contiguous) into host memory, then wait for the stream:
// the one host synchronisation of a rollout step, as a single call (h_out384: 384 bytes, pinned for a true async copy)
extern "C" int cmbpo_rollout_read_scalars(const cmbpo_rollout_t *r, void *h_out384, void *stream) {
  CMBPO_REQUIRE(r && r->iscal && r->dscal && h_out384, "cmbpo_rollout_read_scalars: NULL argument");
  CMBPO_REQUIRE(reinterpret_cast<const char *>(r->dscal) == reinterpret_cast<const char *>(r->iscal) + 128,
                "cmbpo_rollout_read_scalars: iscal[32] and dscal[32] must be one 384-byte block");
  CMBPO_HIP_CHECK(hipMemcpyAsync(h_out384, r->iscal, 384, hipMemcpyDeviceToHost, (hipStream_t)stream));
  CMBPO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return CMBPO_OK;
}

extern "C" int cmbpo_rollout_compact(const cmbpo_rollout_t *r, void *stream) {
  if (int rc = check_rollout(r, "cmbpo_rollout_compact")) return rc;
  CMBPO_REQUIRE(r->store_part != nullptr, "cmbpo_rollout_compact: NULL store_part scratch");
  const int chunks = cmbpo_ceil_div(r->B, kScanThreads);
  hipLaunchKernelGGL(compact_count_kernel, dim3(chunks), dim3(kScanThreads), 0, (hipStream_t)stream, *r);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(kScanThreads), 0, (hipStream_t)stream, *r, chunks);
  hipLaunchKernelGGL(compact_scatter_kernel, dim3(chunks), dim3(kScanThreads), 0, (hipStream_t)stream, *r, chunks);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

extern "C" int cmbpo_buffer_offsets(const cmbpo_rollout_t *r, int32_t *d_offsets, void *stream) {
  if (int rc = check_rollout(r, "cmbpo_buffer_offsets")) return rc;
  CMBPO_REQUIRE(d_offsets != nullptr, "cmbpo_buffer_offsets: NULL offsets");
  hipStream_t s = (hipStream_t)stream;
  const int chunks = cmbpo_ceil_div(r->B, kScanThreads);     // <= ceil(B / 64) * 2 ints of r->store_part
  hipLaunchKernelGGL(offsets_count_kernel, dim3(chunks), dim3(kScanThreads), 0, s, *r);
  hipLaunchKernelGGL(offsets_scan_kernel, dim3(1), dim3(kScanThreads), 0, s, *r, chunks, d_offsets);
  hipLaunchKernelGGL(offsets_write_kernel, dim3(chunks), dim3(kScanThreads), 0, s, *r, d_offsets);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

extern "C" int cmbpo_buffer_moments(const cmbpo_rollout_t *r, int pass, double *d_stats, void *stream) {
  if (int rc = check_rollout(r, "cmbpo_buffer_moments")) return rc;
  CMBPO_REQUIRE(pass >= 0 && pass <= 3 && d_stats, "cmbpo_buffer_moments: bad pass %d / NULL stats", pass);
  hipStream_t s = (hipStream_t)stream;
  if (pass == 0) CMBPO_HIP_CHECK(hipMemsetAsync(d_stats, 0, 16 * sizeof(double), s));
  if (pass == 0 || pass == 2) {
    // one partial row of r->store_part per workgroup: at most ceil(B / 64) rows exist
    const int cap = cmbpo_ceil_div(r->B, 64);
    int blocks = cmbpo_ceil_div(r->B, 256) < 1024 ? cmbpo_ceil_div(r->B, 256) : 1024;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(moments_kernel, dim3(blocks), dim3(256), 0, s, *r, pass == 0 ? 0 : 1, d_stats);
    hipLaunchKernelGGL(moments_fold_kernel, dim3(1), dim3(64), 0, s, *r, pass == 0 ? 0 : 1, blocks, d_stats);
  } else {
    hipLaunchKernelGGL(moments_finalize, dim3(1), dim3(1), 0, s, pass, d_stats);
  }
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

extern "C" int cmbpo_buffer_prepare(const cmbpo_rollout_t *r, int32_t *d_offsets, double *d_stats, void *stream) {
  if (int rc = check_rollout(r, "cmbpo_buffer_prepare")) return rc;
  CMBPO_REQUIRE(d_offsets && d_stats, "cmbpo_buffer_prepare: NULL argument");
  CMBPO_REQUIRE(r->world <= 1, "cmbpo_buffer_prepare: a sharded buffer needs the all-reduce between the passes "
                               "(cmbpo_buffer_offsets / cmbpo_buffer_moments)");
  hipStream_t s = (hipStream_t)stream;
  const int blocks = cmbpo_ceil_div(r->B, 256);      // one partial row each: <= ceil(B / 64) rows exist
  const int fold_here = blocks <= 128 ? 1 : 0;
  hipLaunchKernelGGL(get_pass1_kernel, dim3(blocks), dim3(256), 0, s, *r, d_offsets, d_stats, fold_here);
  if (!fold_here) hipLaunchKernelGGL(get_fold_kernel, dim3(1), dim3(256), 0, s, *r, 1, blocks, d_offsets, d_stats);
  hipLaunchKernelGGL(get_pass2_kernel, dim3(blocks), dim3(256), 0, s, *r, d_offsets, d_stats, fold_here);
  if (!fold_here) hipLaunchKernelGGL(get_fold_kernel, dim3(1), dim3(256), 0, s, *r, 2, blocks, d_offsets, d_stats);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

// Both halves of the flatten in ONE launch: the first n_vec workgroups carry the vector fields' tiles, the rest the eight scalar
// fields' (two launches of ~10 us each around 20-40 us of copying at the shipped configurations' 1e4 branches; the scalar tiles
// fill the CUs the vector tiles leave).  Dynamic LDS: the larger of the two layouts.
namespace {
// (three waves per SIMD: the two load passes' registers sit at 168-169 VGPRs, the edge between three waves and two -- two cost
// every shape but the short ragged one 14-35 % of its flatten)
template <int ROWS, bool SHORT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FLAT_WAVES))) void flatten_kernel(const cmbpo_rollout_t r, const int32_t *offs, const double *st, const FlatArgs fa,
                                                      int vt, int Tt, int n_vec) {
  if ((int)blockIdx.x < n_vec) flatten_vec_body<SHORT>(r, offs, fa, vt, Tt, (int)blockIdx.x);
  else flatten_scalar_body<ROWS>(r, offs, st, fa, Tt, (int)blockIdx.x - n_vec);
}
}  // namespace

extern "C" int cmbpo_buffer_flatten(const cmbpo_rollout_t *r, const int32_t *d_offsets, const double *d_stats,
                                    float *const *h_out12, void *stream) {
  if (int rc = check_rollout(r, "cmbpo_buffer_flatten")) return rc;
  CMBPO_REQUIRE(d_offsets && d_stats && h_out12, "cmbpo_buffer_flatten: NULL argument");
  FlatArgs fa;
  for (int k = 0; k < 12; ++k) {
    CMBPO_REQUIRE(h_out12[k] != nullptr, "cmbpo_buffer_flatten: output %d is NULL", k);
    fa.out[k] = h_out12[k];
  }
  hipStream_t s = (hipStream_t)stream;
  const int dmax = r->obs_dim > r->act_dim ? r->obs_dim : r->act_dim;
  static int n_cu = 0;
  if (n_cu == 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  // no path is longer than the steps the rollout took: tiles (and their LDS) are sized for those
  const int Tt = r->ptr > 0 ? r->ptr : 1;
  // branches per workgroup of the vector fields: as many as keep the [branch][step][dim] tile within 64 KB (two
  // workgroups per CU; 16 at AntSafe shapes and 34 steps, 64 after a rollout that ended within 8 steps), fewer while the
  // buffer has less than four tiles per CU
  static const int vt_max = getenv("CMBPO_FLAT_VT") ? atoi(getenv("CMBPO_FLAT_VT")) : kVecTile;
  const int vt_cap = Tt <= 8 ? kVecTile : 16;     // (64-branch tiles only after short rollouts: the 2 x 8-load pass)
  int vt = vt_max < 1 ? 1 : (vt_max > vt_cap ? vt_cap : vt_max);
  const int dtile = dmax > 3 * r->act_dim ? dmax : 3 * r->act_dim;     // obs alone, then act | log_std | mu side by side
  while (vt > 1 && (size_t)vt * Tt * dtile * sizeof(float) > 64 * 1024) vt >>= 1;
  // ... and as keep a step's run of the tile (vt x dim floats) inside the 8 x 16 bytes per lane of the short rollouts' load
  // pass (flat_vec_in; a longer run falls back to its element-wise loop)
  if (FLAT_WIDE_TILES && Tt <= 8)
    while (vt > 4 && vt * dmax > 2048) vt >>= 1;
  while (vt > 4 && cmbpo_ceil_div(r->B, vt) < 4 * n_cu) vt >>= 1;
  const int rows = cmbpo_ceil_div(r->B, kFlatRows) < 2 * n_cu ? 16 : kFlatRows;
  const size_t lds_v = (size_t)vt * Tt * dtile * sizeof(float) + (((size_t)vt * Tt * sizeof(unsigned short) + 15) & ~(size_t)15);
  const size_t lds_s = (size_t)8 * rows * (Tt + 1) * sizeof(float) + (size_t)rows * Tt * sizeof(unsigned short);
  CMBPO_REQUIRE(lds_v <= 150 * 1024 && lds_s <= 150 * 1024, "cmbpo_buffer_flatten: T = %d, dim = %d exceed the LDS tiles", Tt, dmax);
  const size_t lds = lds_v > lds_s ? lds_v : lds_s;
  const bool is_short = Tt <= 8;
  const int n_vec = cmbpo_ceil_div(r->B, vt);
  static size_t attr[2][2] = {{64 * 1024, 64 * 1024}, {64 * 1024, 64 * 1024}};
  auto launch = [&](auto kern, size_t &granted, int n_scalar) -> int {
    if (lds > granted) {
      CMBPO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      granted = lds;
    }
    hipLaunchKernelGGL(kern, dim3(n_vec + n_scalar), dim3(256), lds, s, *r, d_offsets, d_stats, fa, vt, Tt, n_vec);
    return CMBPO_OK;
  };
  int rc;
  if (rows == 16) {
    rc = is_short ? launch(flatten_kernel<16, true>, attr[0][1], cmbpo_ceil_div(r->B, 16))
                  : launch(flatten_kernel<16, false>, attr[0][0], cmbpo_ceil_div(r->B, 16));
  } else {
    rc = is_short ? launch(flatten_kernel<kFlatRows, true>, attr[1][1], cmbpo_ceil_div(r->B, kFlatRows))
                  : launch(flatten_kernel<kFlatRows, false>, attr[1][0], cmbpo_ceil_div(r->B, kFlatRows));
  }
  if (rc) return rc;
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}
