// Ensemble training step for gfx950: the TensorFlow train_op of the reference's
// probabilistic ensemble, as hand-written kernels over device-resident data.
//
// Replaces
//   models/pens/pe.py:252-274,312-315   train_loss = sum_e loss_e + sum_l decay_l * l2_loss(W_l); Adam apply
//   models/pens/pe.py:921-973           _mspe_loss (dynamics ensemble, loss 'MSPE')
//   models/pens/pe.py:840-919           _nll_loss(inc_var_loss=False) (critics, loss 'MSE'; also `self.loss`,
//                                       the per-member holdout loss that ranks the elites)
//   models/pens/fc.py:74-95,167-168     batched per-member matmul + swish, weight decay
//   models/pens/pe.py:543-551           inputs[batch_idxs]: the per-member bootstrap gather (done in the kernel)
//
// One step = training forward (ens_mlp_kernel<HEAD_TRAIN>: raw outputs + exported activations and swish
// derivatives + per-tile loss statistics) -> fused backward chain (d(loss)/d(output) computed while its tile is
// staged, then two transposed-weight GEMMs; the delta of the middle layer never leaves the CU un-multiplied) -> three weight-gradient GEMMs with the batch as
// the K dimension (fp32 MFMA, split-K partials, bias gradients as column sums of the same operands; no atomics:
// the step is bitwise reproducible; one launch for the 512-wide ensembles) -> Adam with the TensorFlow update
// rule, which also re-packs the new weights into the MFMA layouts the forward / backward kernels read (the master
// copy stays row-major for checkpoints).  The 128-wide MSE ensembles (critics) run the whole gradient in one
// fused kernel instead (fused_mse_step_kernel).
#include "common.h"
#include "ens_mlp_internal.h"
#include "f16_split.h"
#include "mfma_tile.h"
#include "row_tile.h"

#include <math.h>
#include <new>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------------------
// loss: per-member sums, then d(train_loss)/d(raw output)
// ------------------------------------------------------------------------------------------------------------
struct LossArgs {
  const float *o;          // [E][B][O] raw network output
  const float *targets;    // [N][D]
  const int32_t *idx;      // [E][idx_stride] rows of `targets` (nullptr: row b)
  int idx_stride;
  const float *out_mu, *out_sig;   // output scaler (nullptr: identity)
  int E, B, O, D, OPk, prob;
  double *sums;            // [3][E]: sum (mean - t)^2 | sum (var - mse)^2 | sum log_var^2
  float *d3;               // [E][B][OPk]
};

__device__ __forceinline__ float scaled_target(const LossArgs &p, int e, int b, int d) {
  const int row = p.idx ? p.idx[(size_t)e * p.idx_stride + b] : b;
  float t = p.targets[(size_t)row * p.D + d];
  if (p.out_mu) t = (t - p.out_mu[d]) / p.out_sig[d];   // TensorStandardScaler.transform, models/pens/utils.py:156
  return t;
}

__global__ __launch_bounds__(kThreads) void loss_sums_kernel(const LossArgs p) {
  const int e = blockIdx.y;
  double s_mse = 0.0, s_vl = 0.0, s_lv = 0.0;
  const int n = p.B * p.D;
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
    const int b = i / p.D, d = i - b * p.D;
    const float *orow = p.o + ((size_t)e * p.B + b) * p.O;
    const float diff = orow[d] - scaled_target(p, e, b, d);
    const float mse = diff * diff;
    s_mse += (double)mse;
    if (p.prob) {
      const float lv = orow[p.D + d];
      const float dv = expf(lv) - mse;
      s_vl += (double)(dv * dv);
      s_lv += (double)(lv * lv);
    }
  }
  __shared__ double sm[3][kThreads / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  s_mse = wave_sum(s_mse); s_vl = wave_sum(s_vl); s_lv = wave_sum(s_lv);
  if (lane == 0) { sm[0][w] = s_mse; sm[1][w] = s_vl; sm[2][w] = s_lv; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = 0.0;
    for (int i = 0; i < kThreads / 64; ++i) t += sm[threadIdx.x][i];
    atomicAdd(p.sums + (size_t)threadIdx.x * p.E + e, t);
  }
}

// `self.loss` of the reference for n rows seen so far: per-member 0.5 * mean (mean - t)^2
__global__ void loss_finalize_kernel(const double *sums, int E, double inv_count, float *out) {
  const int e = threadIdx.x;
  if (e < E) out[e] = (float)(0.5 * sums[e] * inv_count);
}

// ------------------------------------------------------------------------------------------------------------
// backward chain:  d2 = (d3 W2^T) * g2 ;  d1 = (d2 W1^T) * g1      (g = swish'(z) exported by the forward)
// Same transposed-GEMM formulation as the forward (see ens_mlp.hip): rows of the batch on the MFMA columns, the
// transposed weights packed [n-tile][k-group][lane][4], the d2 tile handed from the first GEMM to the second
// through LDS.  One workgroup = 32 batch rows of one member.
// ------------------------------------------------------------------------------------------------------------
struct BwdArgs {
  const f32x4 *wpb2, *wpb1;
  size_t wpb2_stride, wpb1_stride;   // per member, float4 units
  const float *d3, *g2, *g1;
  float *d2, *d1;
  int B, OPk;
  // fused output delta (d3 == nullptr on entry is not used: `fuse` selects): d(train_loss)/d(raw output) computed while
  // the tile is staged, from the raw outputs, the gathered targets and the forward's per-tile loss statistics
  int fuse, prob, O, D, n_items;
  const float *o, *targets, *out_mu, *out_sig;
  const int32_t *idx;
  int idx_stride;
  const double *loss_part;   // [n_items][3]
  float *d3_out;             // [E][B][OPk], written for the weight-gradient kernel
  float *opmax;              // [E * tiles][8]: largest |d1|, |d2|, |d3| of the tile into slots 3..5 (nullptr: not wanted)
};

template <int HID>
__global__ __launch_bounds__(kThreads, 2) void bwd_chain_kernel(const BwdArgs p) {
  constexpr int BB = 32, NT = HID / 128, KG_H = HID / 8;
  extern __shared__ f32x4 smem[];
  f32x4 *hbuf = smem;                    // d2 tile, T-layout [HID/4][BB]
  f32x4 *xbuf = smem + HID / 4 * BB;     // d3 tile, T-layout [OPk/4][BB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int e = blockIdx.y, row0 = blockIdx.x * BB;
  const int kgo = p.OPk / 8;
  __shared__ float s_dmax[3][kThreads / 64];   // per-wave maxima of |d3|, |d2|, |d1| of this tile
  float dmax = 0.0f;
  if (!p.fuse) {
    float *xf = reinterpret_cast<float *>(xbuf);
    for (int i = tid; i < BB * p.OPk; i += kThreads) {
      const int b = i / p.OPk, k = i - b * p.OPk;
      float v = 0.0f;
      if (row0 + b < p.B) v = p.d3[((size_t)e * p.B + row0 + b) * p.OPk + k];
      xf[((k >> 2) * BB + b) * 4 + (k & 3)] = v;
      dmax = fmaxf(dmax, fabsf(v));
    }
  } else {
    // Output deltas d(train_loss)/d(raw output).
    // MSPE (pe.py:921-973): total_e = mean (m - t)^2 + ratio * mean (var - sg(mse))^2 + 0.05 * mean_all lv^2, with
    // ratio = 0.05 * mean_all(mse) / mean_all((var - mse)^2) a constant of the step; train_loss = sum_e total_e, so the
    // regulariser (a scalar broadcast onto every member) counts E times.
    // MSE (pe.py:911-919): total_e = mean 0.5 (o - t)^2.
    // The two sums behind `ratio` come from the forward's per-tile statistics, added here in tile order by every
    // workgroup (same order everywhere: reproducible).
    // NLL (pe.py:840-919, inc_var_loss = True; prob == 2): total_e = mean 0.5 exp(-lv) (m - t)^2 + mean 0.5 lv; no
    // coupling between members.  (Its max_logvar / min_logvar variables only carry a constant regulariser gradient
    // and never enter the network, pe.py:198-209,263,789-838: they are host-side bookkeeping, pens.PE.)
    __shared__ double s_tot[2][kThreads / 64];
    float ratio = 0.0f;
    if (p.prob == 1) {
      double tm = 0.0, tv = 0.0;
      for (int w = tid; w < p.n_items; w += kThreads) {
        tm += p.loss_part[(size_t)w * 3];
        tv += p.loss_part[(size_t)w * 3 + 1];
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        tm += __shfl_down(tm, o, 64);
        tv += __shfl_down(tv, o, 64);
      }
      if (lane == 0) { s_tot[0][wave] = tm; s_tot[1][wave] = tv; }
      __syncthreads();
      tm = (s_tot[0][0] + s_tot[0][1]) + (s_tot[0][2] + s_tot[0][3]);
      tv = (s_tot[1][0] + s_tot[1][1]) + (s_tot[1][2] + s_tot[1][3]);
      ratio = (float)(0.05 * tm / tv);
    }
    const float inv_bd = 1.0f / ((float)p.B * (float)p.D);
    float *xf = reinterpret_cast<float *>(xbuf);
    for (int i = tid; i < BB * p.OPk; i += kThreads) {
      const int b = i / p.OPk, k = i - b * p.OPk;
      float v = 0.0f;
      const int row = row0 + b;
      if (row < p.B) {
        if (k < p.O) {
          const float *orow = p.o + ((size_t)e * p.B + row) * p.O;
          const int src = p.idx ? p.idx[(size_t)e * p.idx_stride + row] : row;
          const int dd = (k < p.D) ? k : k - p.D;
          float t = p.targets[(size_t)src * p.D + dd];
          if (p.out_mu) t = (t - p.out_mu[dd]) / p.out_sig[dd];
          const float diff = orow[dd] - t;
          if (!p.prob) {
            v = diff * inv_bd;
          } else if (p.prob == 2) {
            const float iv = expf(-orow[p.D + dd]);
            v = (k < p.D) ? iv * diff * inv_bd : (0.5f - 0.5f * iv * diff * diff) * inv_bd;
          } else if (k < p.D) {
            v = 2.0f * diff * inv_bd;
          } else {
            const float lv = orow[k], var = expf(lv);
            v = (2.0f * ratio * (var - diff * diff) * var + 0.1f * lv) * inv_bd;
          }
        }
        p.d3_out[((size_t)e * p.B + row) * p.OPk + k] = v;
      }
      xf[((k >> 2) * BB + b) * 4 + (k & 3)] = v;
      dmax = fmaxf(dmax, fabsf(v));
    }
  }
  auto wave_max_to = [&](float m, float *slot) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) slot[wave] = m;
  };
  wave_max_to(dmax, s_dmax[0]);
  dmax = 0.0f;
  const int j = lane & 31, h = lane >> 5;
  const bool valid = row0 + j < p.B;
  const size_t grow = ((size_t)e * p.B + row0 + j) * HID;
  const int n_base = wave * NT * 32;
  f32x4 gq[NT][4];
  auto load_g = [&](const float *g) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n_base + t * 32 + 8 * q + 4 * h;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (valid) v = *reinterpret_cast<const f32x4 *>(g + grow + n);
        gq[t][q] = v;
      }
  };
  f32x16 acc[NT][1];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][0][r] = 0.0f;
  };
  load_g(p.g2);      // in flight under the first GEMM
  zero_acc();
  __syncthreads();
  mfma_layer<NT, 1>(p.wpb2 + e * p.wpb2_stride + (size_t)(wave * NT) * kgo * 64, (size_t)kgo * 64, 0, kgo, xbuf, lane, acc);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n_base + t * 32 + 8 * q + 4 * h;
      f32x4 v;
#pragma unroll
      for (int s = 0; s < 4; ++s) v[s] = acc[t][0][4 * q + s] * gq[t][q][s];
      hbuf[(n >> 2) * BB + j] = v;
      if (valid) {
        *reinterpret_cast<f32x4 *>(p.d2 + grow + n) = v;
        dmax = fmaxf(fmaxf(dmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      }
    }
  wave_max_to(dmax, s_dmax[1]);
  dmax = 0.0f;
  load_g(p.g1);
  zero_acc();
  __syncthreads();
  mfma_layer<NT, 1>(p.wpb1 + e * p.wpb1_stride + (size_t)(wave * NT) * KG_H * 64, (size_t)KG_H * 64, 0, KG_H, hbuf, lane, acc);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n_base + t * 32 + 8 * q + 4 * h;
      f32x4 v;
#pragma unroll
      for (int s = 0; s < 4; ++s) v[s] = acc[t][0][4 * q + s] * gq[t][q][s];
      if (valid) {
        *reinterpret_cast<f32x4 *>(p.d1 + grow + n) = v;
        dmax = fmaxf(fmaxf(dmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      }
    }
  if (p.opmax) {
    wave_max_to(dmax, s_dmax[2]);
    __syncthreads();
    if (tid < 3) {
      const float *sm = s_dmax[2 - tid];      // slots 3, 4, 5 = d1, d2, d3
      p.opmax[((size_t)e * gridDim.x + blockIdx.x) * 8 + 3 + tid] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// The backward chain of a 512-wide ensemble on the f16 matrix path (three v_mfma_f32_32x32x16_f16 per float32 product on
// two-piece operands, f16_split.h; arithmetic and lifts as in ens_h3.hip).  The fp32 kernel above walks 32-row tiles, two
// workgroups per CU: every tile streams the member's 1 MB of W1^T from L2 -- 470 MB per step at batch 2048, which is what
// its 111 us are.  Here an item is 64 rows of one member on 8 waves (one workgroup per CU, one round of items at batch
// 2048): half the weight stream per row, the products at the f16 pipe's rate.
//   d3 (computed while the tile is staged, as above) -> fp32 tile in LDS -> the row's lift from its largest magnitude ->
//   two-piece f16 image [row][k] -> d2 = (d3 W2^T) * g2 (A = the W2^T image from L2, wave w owns n-tiles 2 w, 2 w + 1 x both
//   row tiles) -> stored, and split into the [64][512] two-piece image under a lift known before the product (|d2| <=
//   1.1 max |W2| OPk max |d3[row]|) -> d1 = (d2 W1^T) * g1 (A fragments a slab ahead through a three-slot ring) -> stored.
// ------------------------------------------------------------------------------------------------------------
struct BwdHArgs {
  BwdArgs b;
  const f16x8 *w2t, *w1t;          // images [member][n-tile 16][k-slab][piece 2][lane 64] of 8 halves
  size_t w2t_stride, w1t_stride;    // per member, 16-byte units
  const float *stats;               // [E][NSTAT]: the weights' lift of layer l at [4 l], max |W| at [4 l + 3]
  int s3;                           // k-slabs of the first product (OPk <= 16 s3)
  int tiles32;                      // 32-row tiles of the batch (the per-tile maxima are indexed by those)
};

constexpr int kBhThreads = 512, kBhRows = 64, kBhHid = 512;
constexpr int kBhD2Str = kBhHid + 8;      // halves per row of the d2 image: 16-byte slots per row odd -> conflict-free reads
__host__ __device__ constexpr int bh_d3str(int s3) { return 16 * s3 + 8; }
__host__ __device__ constexpr size_t bh_lds_bytes(int s3) {
  return (size_t)2 * kBhRows * kBhD2Str * 2 + (size_t)2 * kBhRows * bh_d3str(s3) * 2 + (size_t)4 * kBhRows * 4;
}

__global__ __launch_bounds__(kBhThreads) void bwd_chain_h_kernel(const BwdHArgs a) {
  const BwdArgs &p = a.b;
  extern __shared__ f32x4 smem[];
  _Float16 *d2img = reinterpret_cast<_Float16 *>(smem);                         // [2 pieces][64][kBhD2Str]
  float *d3f = reinterpret_cast<float *>(smem);                                 // [64][OPk] fp32 (aliases d2img: consumed first)
  _Float16 *d3img = d2img + (size_t)2 * kBhRows * kBhD2Str;                     // [2 pieces][64][D3S]
  const int D3S = bh_d3str(a.s3);
  float *r_inv3 = reinterpret_cast<float *>(d3img + (size_t)2 * kBhRows * D3S); // [64] 1 / (s2 t3)
  float *r_t3 = r_inv3 + kBhRows, *r_t2 = r_t3 + kBhRows, *r_inv2 = r_t2 + kBhRows;
  __shared__ unsigned s_rowmax[kBhRows];
  __shared__ float s_dmax[3][kBhThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int e = blockIdx.y, row0 = blockIdx.x * kBhRows;
  const float *st = a.stats + (size_t)e * NSTAT;
  if (tid < kBhRows) s_rowmax[tid] = 0u;
  float dmax = 0.0f;
  // ---- output deltas (the fp32 kernel's arithmetic, operation for operation) -> fp32 tile, row maxima ---------------------
  {
    __shared__ double s_tot[2][kBhThreads / 64];
    float ratio = 0.0f;
    if (p.prob == 1) {
      double tm = 0.0, tv = 0.0;
      for (int w = tid; w < p.n_items; w += kBhThreads) {
        tm += p.loss_part[(size_t)w * 3];
        tv += p.loss_part[(size_t)w * 3 + 1];
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        tm += __shfl_down(tm, o, 64);
        tv += __shfl_down(tv, o, 64);
      }
      if (lane == 0) { s_tot[0][wave] = tm; s_tot[1][wave] = tv; }
      __syncthreads();
      tm = 0.0; tv = 0.0;
      for (int w = 0; w < kBhThreads / 64; ++w) { tm += s_tot[0][w]; tv += s_tot[1][w]; }
      ratio = (float)(0.05 * tm / tv);
    } else {
      __syncthreads();
    }
    const float inv_bd = 1.0f / ((float)p.B * (float)p.D);
    for (int i = tid; i < kBhRows * p.OPk; i += kBhThreads) {
      const int b = i / p.OPk, k = i - b * p.OPk;
      float v = 0.0f;
      const int row = row0 + b;
      if (row < p.B) {
        if (p.fuse) {
          if (k < p.O) {
            const float *orow = p.o + ((size_t)e * p.B + row) * p.O;
            const int src = p.idx ? p.idx[(size_t)e * p.idx_stride + row] : row;
            const int dd = (k < p.D) ? k : k - p.D;
            float t = p.targets[(size_t)src * p.D + dd];
            if (p.out_mu) t = (t - p.out_mu[dd]) / p.out_sig[dd];
            const float diff = orow[dd] - t;
            if (!p.prob) {
              v = diff * inv_bd;
            } else if (p.prob == 2) {
              const float iv = expf(-orow[p.D + dd]);
              v = (k < p.D) ? iv * diff * inv_bd : (0.5f - 0.5f * iv * diff * diff) * inv_bd;
            } else if (k < p.D) {
              v = 2.0f * diff * inv_bd;
            } else {
              const float lv = orow[k], var = expf(lv);
              v = (2.0f * ratio * (var - diff * diff) * var + 0.1f * lv) * inv_bd;
            }
          }
          p.d3_out[((size_t)e * p.B + row) * p.OPk + k] = v;
        } else {
          v = p.d3[((size_t)e * p.B + row) * p.OPk + k];
        }
      }
      d3f[b * p.OPk + k] = v;
      const float av = fabsf(v);
      dmax = fmaxf(dmax, av);
      if (av > 0.0f) atomicMax(&s_rowmax[b], __float_as_uint(av));     // (non-negative floats order like their bit patterns)
    }
  }
  auto wave_max_to = [&](float m, float *slot) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) slot[wave] = m;
  };
  wave_max_to(dmax, s_dmax[0]);
  dmax = 0.0f;
  __syncthreads();
  if (tid < kBhRows) {
    const float m3 = __uint_as_float(s_rowmax[tid]);
    const float t3 = pow2_lift(m3);
    // |d2[row][n]| = |g2 sum_k d3[row][k] W2[n][k]| <= 1.1 OPk max |W2| max |d3[row]|   (swish' <= 1.0999)
    const float t2 = pow2_lift(1.1f * (float)p.OPk * st[11] * m3);
    r_t3[tid] = t3;
    r_inv3[tid] = 1.0f / (st[8] * t3);
    r_t2[tid] = t2;
    r_inv2[tid] = 1.0f / (st[4] * t2);
  }
  __syncthreads();
  {
    const int KP = 16 * a.s3;
    for (int i = tid; i < kBhRows * KP; i += kBhThreads) {
      const int b = i / KP, k = i - b * KP;
      const float v = k < p.OPk ? d3f[b * p.OPk + k] : 0.0f;
      _Float16 q1, q2;
      split_h(v * r_t3[b], q1, q2);
      d3img[(size_t)b * D3S + k] = q1;
      d3img[(size_t)kBhRows * D3S + (size_t)b * D3S + k] = q2;
    }
  }
  // this lane's rows of the two row tiles, its export offsets
  const bool valid[2] = {row0 + r < p.B, row0 + 32 + r < p.B};
  const size_t grow[2] = {((size_t)e * p.B + row0 + r) * kBhHid, ((size_t)e * p.B + row0 + 32 + r) * kBhHid};
  f32x4 gq[2][2][4];      // [n-tile of the wave][row tile][quad]
  auto load_g = [&](const float *g) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = 32 * (2 * wave + t) + 8 * q + 4 * hh;
          f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
          if (valid[bt]) v = *reinterpret_cast<const f32x4 *>(g + grow[bt] + n);
          gq[t][bt][q] = v;
        }
  };
  f32x16 acc[2][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][bt][i] = 0.0f;
  };
  auto read_b = [&](f16x8 (&x)[2], const _Float16 *img, int str, int bt, int sl) {
    const _Float16 *q = img + (size_t)(32 * bt + r) * str + 16 * sl + 8 * hh;
    x[0] = *reinterpret_cast<const f16x8 *>(q);
    x[1] = *reinterpret_cast<const f16x8 *>(q + (size_t)kBhRows * str);
  };
  // ---- d2 = (d3 W2^T) * g2 -----------------------------------------------------------------------------------------------
  load_g(p.g2);
  zero_acc();
  {
    const f16x8 *w2 = a.w2t + (size_t)e * a.w2t_stride + lane;       // + ((tile * s3 + s) * 2 + piece) * 64
    f16x8 A2[4][2][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f16x8 *q = w2 + ((size_t)((2 * wave + t) * a.s3 + (s < a.s3 ? s : 0)) * 2) * 64;
        A2[s][t][0] = q[0]; A2[s][t][1] = q[64];
      }
    __syncthreads();           // the d3 image is complete
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s < a.s3) {
#pragma unroll
        for (int bt = 0; bt < 2; ++bt) {
          f16x8 bf[2];
          read_b(bf, d3img, D3S, bt, s);
#pragma unroll
          for (int t = 0; t < 2; ++t) mm3(acc[t][bt], A2[s][t][0], A2[s][t][1], bf[0], bf[1]);
        }
      }
  }
  // (every wave has read the fp32 tile that aliases the d2 image long ago: the barrier above)
#pragma unroll
  for (int bt = 0; bt < 2; ++bt) {
    const float inv3 = r_inv3[32 * bt + r], t2 = r_t2[32 * bt + r];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = 32 * (2 * wave + t) + 8 * q + 4 * hh;
        f32x4 v;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = (acc[t][bt][4 * q + s] * inv3) * gq[t][bt][q][s];
        if (valid[bt]) {
          *reinterpret_cast<f32x4 *>(p.d2 + grow[bt] + n) = v;
          dmax = fmaxf(fmaxf(dmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
        unsigned q1[2], q2[2];
        split2<false>(v[0], v[1], t2, q1[0], q2[0]);
        split2<false>(v[2], v[3], t2, q1[1], q2[1]);
        _Float16 *c1 = d2img + (size_t)(32 * bt + r) * kBhD2Str + n;
        *reinterpret_cast<uint2 *>(c1) = make_uint2(q1[0], q1[1]);
        *reinterpret_cast<uint2 *>(c1 + (size_t)kBhRows * kBhD2Str) = make_uint2(q2[0], q2[1]);
      }
  }
  wave_max_to(dmax, s_dmax[1]);
  dmax = 0.0f;
  // ---- d1 = (d2 W1^T) * g1 -----------------------------------------------------------------------------------------------
  load_g(p.g1);
  zero_acc();
  {
    const f16x8 *w1 = a.w1t + (size_t)e * a.w1t_stride + lane;       // + ((tile * 32 + s) * 2 + piece) * 64
    constexpr int S1 = kBhHid / 16, RING = 3;
    f16x8 A1[RING][2][2];
    auto load_a = [&](f16x8 (&x)[2][2], int s) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f16x8 *q = w1 + ((size_t)((2 * wave + t) * S1 + s) * 2) * 64;
        x[t][0] = q[0]; x[t][1] = q[64];
      }
    };
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) load_a(A1[s], s);
    __syncthreads();           // the d2 image is complete
#pragma unroll
    for (int s = 0; s < S1; ++s) {
      if (s + RING - 1 < S1) load_a(A1[(s + RING - 1) % RING], s + RING - 1);
#pragma unroll
      for (int bt = 0; bt < 2; ++bt) {
        f16x8 bf[2];
        read_b(bf, d2img, kBhD2Str, bt, s);
#pragma unroll
        for (int t = 0; t < 2; ++t) mm3(acc[t][bt], A1[s % RING][t][0], A1[s % RING][t][1], bf[0], bf[1]);
      }
    }
  }
#pragma unroll
  for (int bt = 0; bt < 2; ++bt) {
    const float inv2 = r_inv2[32 * bt + r];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = 32 * (2 * wave + t) + 8 * q + 4 * hh;
        f32x4 v;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = (acc[t][bt][4 * q + s] * inv2) * gq[t][bt][q][s];
        if (valid[bt]) {
          *reinterpret_cast<f32x4 *>(p.d1 + grow[bt] + n) = v;
          dmax = fmaxf(fmaxf(dmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
      }
  }
  if (p.opmax) {
    wave_max_to(dmax, s_dmax[2]);
    __syncthreads();
    if (tid < 3) {
      const float *sm = s_dmax[2 - tid];      // slots 3, 4, 5 = d1, d2, d3
      float m = 0.0f;
      for (int w = 0; w < kBhThreads / 64; ++w) m = fmaxf(m, sm[w]);
      // (the weight-gradient kernel reads the maxima per 32-row tile: this item covers two of them)
      for (int h2 = 0; h2 < 2; ++h2) {
        const int t32 = 2 * (int)blockIdx.x + h2;
        if (t32 < a.tiles32) p.opmax[((size_t)e * a.tiles32 + t32) * 8 + 3 + tid] = m;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// The training forward of a 512-wide probabilistic ensemble on the same path (what ens_mlp_kernel<512, 1, swish,
// HEAD_TRAIN> computes: per-member bootstrap rows, raw outputs, the scaled inputs, h = swish(z) and g = swish'(z) of both
// layers as [member][row][feature] arrays, the tile's loss statistics and largest operand magnitudes).  An item is 64 rows
// of one member on 8 waves.  x, h1 and h2 pass through LDS as two-piece f16 images, every row lifted by its own largest
// magnitude: the activations of a layer stay in the accumulator registers between the pass that exports them and takes
// the row maxima (LDS atomics on the float bits) and the pass, a barrier later, that splits them under the exact lift.
// Wave w owns hidden n-tiles 2 w, 2 w + 1 x both row tiles; the 64-wide output layer runs on waves 0-3 (one (output tile,
// row tile) pair each, the whole K): 96 of an item's 3 744 MFMAs per SIMD.
// ------------------------------------------------------------------------------------------------------------
struct FwdHArgs {
  const float *inputs;              // [N][I]
  const int32_t *idx; int idx_stride;     // [E][idx_stride] rows of `inputs` / `targets` (nullptr: row b)
  int n_rows, I, IP, s0;            // batch rows, input width, padded width, k-slabs of the input layer (IP <= 16 s0)
  const float *in_mu, *in_sig;      // input scaler or nullptr
  const f16x8 *w0, *w1, *w2;        // images [member][n-tile][k-slab][piece][lane]; w2: 2 n-tiles, plain k order
  size_t w0_stride, w1_stride, w2_stride;
  const float *b0, *b1, *b2; int b2_ld;   // [E][512], [E][512], [E][b2_ld]
  const float *stats;               // [E][NSTAT]: the weights' lifts at [0], [4], [8]
  int O, D;                         // raw outputs per row (<= 64), target dims
  float *x, *h1, *g1, *h2, *g2;     // exports (nullptr: none)
  float *o;                         // [E][n_rows][O]
  const float *targets, *out_mu, *out_sig;
  double *loss_part;                // [E * tiles32][3] or nullptr
  float *opmax;                     // [E * tiles32][8] slots 0..2 or nullptr
  int tiles32;
};

constexpr int kFhXStr = 16 * 4 + 8;       // halves per row of the x image (up to four input slabs)
constexpr int kFhOStr = 65;               // floats per row of the output tile
__host__ __device__ constexpr size_t fh_lds_bytes() {
  return (size_t)2 * kBhRows * kBhD2Str * 2 + (size_t)2 * kBhRows * kFhXStr * 2 + (size_t)(2 * kBhHid + 64) * 4;
}
static_assert((size_t)kBhRows * kFhOStr * 4 <= (size_t)2 * kBhRows * kFhXStr * 2, "the output tile aliases the x image");

__global__ __launch_bounds__(kBhThreads) void fwd_train_h_kernel(const FwdHArgs a) {
  extern __shared__ f32x4 smem[];
  _Float16 *himg = reinterpret_cast<_Float16 *>(smem);                          // [2 pieces][64][kBhD2Str]: h1, then h2
  _Float16 *ximg = himg + (size_t)2 * kBhRows * kBhD2Str;                       // [2 pieces][64][kFhXStr]
  float *otile = reinterpret_cast<float *>(ximg);                               // [64][kFhOStr] (aliases the x image: dead by then)
  float *bias0 = reinterpret_cast<float *>(ximg + (size_t)2 * kBhRows * kFhXStr), *bias1 = bias0 + kBhHid, *bias2 = bias1 + kBhHid;
  __shared__ int s_rows[kBhRows];
  __shared__ float s_inv0[kBhRows];
  __shared__ unsigned s_hmax[2][kBhRows];
  __shared__ float s_opmax[3][kBhThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int e = blockIdx.y, row0 = blockIdx.x * kBhRows;
  const float *st = a.stats + (size_t)e * NSTAT;
  if (tid < kBhRows) {
    const int row = row0 + tid;
    s_rows[tid] = row < a.n_rows ? (a.idx ? a.idx[(size_t)e * a.idx_stride + row] : row) : -1;
    s_hmax[0][tid] = 0u; s_hmax[1][tid] = 0u;
  }
  bias0[tid] = a.b0[(size_t)e * kBhHid + tid];
  bias1[tid] = a.b1[(size_t)e * kBhHid + tid];
  if (tid < 64) bias2[tid] = tid < a.O ? a.b2[(size_t)e * a.b2_ld + tid] : 0.0f;
  __syncthreads();
  // ---- the scaled input rows: export, the row's lift, the two-piece image (eight threads per row) -----------------------
  float xmax_t = 0.0f;
  {
    const int b = tid >> 3, c = tid & 7;
    const int src = s_rows[b];
    float xs[8];
    float m = 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = c + 8 * u;
      float x = 0.0f;
      if (k < a.I && src >= 0) {
        x = a.inputs[(size_t)src * a.I + k];
        if (a.in_mu) x = (x - a.in_mu[k]) / a.in_sig[k];       // TensorStandardScaler.transform, models/pens/utils.py:156
      }
      xs[u] = x;
      m = fmaxf(m, fabsf(x));
      if (a.x && src >= 0 && k < a.IP) a.x[((size_t)e * a.n_rows + row0 + b) * a.IP + k] = x;
    }
    xmax_t = m;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const float t0 = pow2_lift(m);
    if (c == 0) s_inv0[b] = 1.0f / (st[0] * t0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = c + 8 * u;
      if (k < 16 * a.s0) {
        _Float16 q1, q2;
        split_h(xs[u] * t0, q1, q2);
        ximg[(size_t)b * kFhXStr + k] = q1;
        ximg[(size_t)kBhRows * kFhXStr + (size_t)b * kFhXStr + k] = q2;
      }
    }
  }
  auto wave_max_to = [&](float m, float *slot) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) slot[wave] = m;
  };
  wave_max_to(xmax_t, s_opmax[0]);
  const bool valid[2] = {row0 + r < a.n_rows, row0 + 32 + r < a.n_rows};
  const size_t grow[2] = {((size_t)e * a.n_rows + row0 + r) * kBhHid, ((size_t)e * a.n_rows + row0 + 32 + r) * kBhHid};
  f32x16 acc[2][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][bt][i] = 0.0f;
  };
  auto read_b = [&](f16x8 (&x)[2], const _Float16 *img, int str, int bt, int sl) {
    const _Float16 *q = img + (size_t)(32 * bt + r) * str + 16 * sl + 8 * hh;
    x[0] = *reinterpret_cast<const f16x8 *>(q);
    x[1] = *reinterpret_cast<const f16x8 *>(q + (size_t)kBhRows * str);
  };
  // z -> h = swish(z), g = swish'(z) (swish_with_grad of the fp32 kernel); exports; the rows' largest |h| (pass 1), then, a
  // barrier later, the two-piece image of h under the rows' exact lifts (pass 2).  Returns the lifts' inverses 1 / (s t).
  auto activate_export = [&](const float *bias, const float (&inv)[2], float *eh, float *eg, unsigned *rowmax, float *opslot) {
    float hmax_t = 0.0f;
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      float hm = 0.0f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = 32 * (2 * wave + t) + 8 * q + 4 * hh;
          const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + n);
          f32x4 hv, gv;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float z = __builtin_fmaf(acc[t][bt][4 * q + s], inv[bt], bv[s]);
            const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
            hv[s] = z * sg;
            gv[s] = sg * (1.0f + z * (1.0f - sg));
            acc[t][bt][4 * q + s] = hv[s];
            hm = fmaxf(hm, fabsf(hv[s]));
          }
          if (eh && valid[bt]) {
            *reinterpret_cast<f32x4 *>(eh + grow[bt] + n) = hv;
            *reinterpret_cast<f32x4 *>(eg + grow[bt] + n) = gv;
          }
        }
      if (hm > 0.0f) atomicMax(&rowmax[32 * bt + r], __float_as_uint(hm));
      if (valid[bt]) hmax_t = fmaxf(hmax_t, hm);
    }
    wave_max_to(hmax_t, opslot);
  };
  auto split_image = [&](const unsigned *rowmax, float s_w, float (&inv_next)[2]) {
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      const float tl = pow2_lift(__uint_as_float(rowmax[32 * bt + r]));
      inv_next[bt] = 1.0f / (s_w * tl);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = 32 * (2 * wave + t) + 8 * q + 4 * hh;
          unsigned q1[2], q2[2];
          split2<false>(acc[t][bt][4 * q], acc[t][bt][4 * q + 1], tl, q1[0], q2[0]);
          split2<false>(acc[t][bt][4 * q + 2], acc[t][bt][4 * q + 3], tl, q1[1], q2[1]);
          _Float16 *c1 = himg + (size_t)(32 * bt + r) * kBhD2Str + n;
          *reinterpret_cast<uint2 *>(c1) = make_uint2(q1[0], q1[1]);
          *reinterpret_cast<uint2 *>(c1 + (size_t)kBhRows * kBhD2Str) = make_uint2(q2[0], q2[1]);
        }
    }
  };
  // ---- layer 0 -----------------------------------------------------------------------------------------------------------
  zero_acc();
  {
    const f16x8 *w0 = a.w0 + (size_t)e * a.w0_stride + lane;       // + ((tile * s0 + s) * 2 + piece) * 64
    f16x8 A0[4][2][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f16x8 *q = w0 + ((size_t)((2 * wave + t) * a.s0 + (s < a.s0 ? s : 0)) * 2) * 64;
        A0[s][t][0] = q[0]; A0[s][t][1] = q[64];
      }
    __syncthreads();           // the x image and the rows' lifts are complete
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s < a.s0) {
#pragma unroll
        for (int bt = 0; bt < 2; ++bt) {
          f16x8 bf[2];
          read_b(bf, ximg, kFhXStr, bt, s);
#pragma unroll
          for (int t = 0; t < 2; ++t) mm3(acc[t][bt], A0[s][t][0], A0[s][t][1], bf[0], bf[1]);
        }
      }
  }
  float inv[2] = {s_inv0[r], s_inv0[32 + r]};
  activate_export(bias0, inv, a.h1, a.g1, s_hmax[0], s_opmax[1]);
  __syncthreads();             // the rows' largest |h1| are complete
  split_image(s_hmax[0], st[4], inv);
  // ---- layer 1 -----------------------------------------------------------------------------------------------------------
  zero_acc();
  {
    const f16x8 *w1 = a.w1 + (size_t)e * a.w1_stride + lane;       // + ((tile * 32 + s) * 2 + piece) * 64
    constexpr int S1 = kBhHid / 16, RING = 3;
    f16x8 A1[RING][2][2];
    auto load_a = [&](f16x8 (&x)[2][2], int s) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f16x8 *q = w1 + ((size_t)((2 * wave + t) * S1 + s) * 2) * 64;
        x[t][0] = q[0]; x[t][1] = q[64];
      }
    };
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) load_a(A1[s], s);
    __syncthreads();           // the h1 image is complete
#pragma unroll
    for (int s = 0; s < S1; ++s) {
      if (s + RING - 1 < S1) load_a(A1[(s + RING - 1) % RING], s + RING - 1);
#pragma unroll
      for (int bt = 0; bt < 2; ++bt) {
        f16x8 bf[2];
        read_b(bf, himg, kBhD2Str, bt, s);
#pragma unroll
        for (int t = 0; t < 2; ++t) mm3(acc[t][bt], A1[s % RING][t][0], A1[s % RING][t][1], bf[0], bf[1]);
      }
    }
  }
  activate_export(bias1, inv, a.h2, a.g2, s_hmax[1], s_opmax[2]);
  __syncthreads();             // the rows' largest |h2| are complete, and every wave has read the h1 image
  split_image(s_hmax[1], st[8], inv);
  __syncthreads();             // the h2 image is complete
  // ---- output layer: waves 0-3, one (output tile, row tile) pair each ---------------------------------------------------
  if (wave < 4) {
    const int tt = wave & 1, bt = wave >> 1;
    const f16x8 *w2 = a.w2 + (size_t)e * a.w2_stride + lane + (size_t)tt * 32 * 2 * 64;    // + (s * 2 + piece) * 64
    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.0f;
    constexpr int S1 = kBhHid / 16;
    f16x8 A2[2][2];
    A2[0][0] = w2[0]; A2[0][1] = w2[64];
#pragma unroll 4
    for (int s = 0; s < S1; ++s) {
      const int sn = s + 1 < S1 ? s + 1 : s;
      A2[(s + 1) & 1][0] = w2[(size_t)sn * 128]; A2[(s + 1) & 1][1] = w2[(size_t)sn * 128 + 64];
      f16x8 bf[2];
      read_b(bf, himg, kBhD2Str, bt, s);
      mm3(o, A2[s & 1][0], A2[s & 1][1], bf[0], bf[1]);
    }
    const float inv2 = inv[bt];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = 32 * tt + (i & 3) + 8 * (i >> 2) + 4 * hh;
      otile[(32 * bt + r) * kFhOStr + n] = __builtin_fmaf(o[i], inv2, bias2[n]);
    }
  }
  __syncthreads();
  // ---- raw outputs, the tile's loss statistics and operand maxima (as the fp32 kernel) -------------------------------------
  for (int i = tid; i < kBhRows * a.O; i += kBhThreads) {
    const int b = i / a.O, n = i - b * a.O;
    if (s_rows[b] >= 0) a.o[((size_t)e * a.n_rows + row0 + b) * a.O + n] = otile[b * kFhOStr + n];
  }
  const size_t item0 = (size_t)e * a.tiles32 + 2 * blockIdx.x;
  const bool second = 2 * (int)blockIdx.x + 1 < a.tiles32;
  if (a.opmax && tid < 3) {
    const float *sm = s_opmax[tid];
    float m = 0.0f;
    for (int w = 0; w < kBhThreads / 64; ++w) m = fmaxf(m, sm[w]);
    a.opmax[item0 * 8 + tid] = m;
    if (second) a.opmax[(item0 + 1) * 8 + tid] = m;
  }
  if (a.loss_part) {
    const int D = a.D;
    const bool prob = a.O == 2 * D;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int i = tid; i < kBhRows * D; i += kBhThreads) {
      const int b = i / D, d = i - b * D;
      const int src = s_rows[b];
      if (src < 0) continue;
      float t = a.targets[(size_t)src * D + d];
      if (a.out_mu) t = (t - a.out_mu[d]) / a.out_sig[d];
      const float diff = otile[b * kFhOStr + d] - t;
      const float mse = diff * diff;
      s0 += (double)mse;
      if (prob) {
        const float lv = otile[b * kFhOStr + D + d];
        const float dv = expf(lv) - mse;
        s1 += (double)(dv * dv);
        s2 += (double)(lv * lv);
      }
    }
    __shared__ double s_loss[3][kBhThreads / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s0 += __shfl_down(s0, o, 64);
      s1 += __shfl_down(s1, o, 64);
      s2 += __shfl_down(s2, o, 64);
    }
    if (lane == 0) { s_loss[0][wave] = s0; s_loss[1][wave] = s1; s_loss[2][wave] = s2; }
    __syncthreads();
    if (tid < 3) {
      double tsum = 0.0;
      for (int w = 0; w < kBhThreads / 64; ++w) tsum += s_loss[tid][w];
      a.loss_part[item0 * 3 + tid] = tsum;                 // (the backward kernel adds the entries of all 32-row tiles:
      if (second) a.loss_part[(item0 + 1) * 3 + tid] = 0.0;   //  this item's sums stand in the first of its two)
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// weight gradients:  C[e][m][n] = sum_b A[e][b][m] * Bm[e][b][n]      (K of the GEMM = the batch)
// Both operands are read straight from their [row][feature] arrays: a lane's 16-B (A) / 8-B (Bm) load of one
// batch row feeds 4 / 2 MFMA tiles whose rows / columns are interleaved (tile mi holds m0 + 4 i + mi), so every
// global load is a contiguous 512-B / 256-B run per batch row and no operand is staged or transposed.
// Wave tile 128 x 64, the 4 waves of a workgroup split the workgroup's batch range and reduce through LDS; the
// grid's K split writes partial sums that the Adam kernel adds in a fixed order.
// ------------------------------------------------------------------------------------------------------------
struct WgradArgs {
  const float *A; int lda;
  const float *Bm; int ldb;
  float *out;                  // [ks][E][out_member]
  size_t out_member, out_part;
  int ldc, transposed, n_out;  // C[m][n] -> out[m * ldc + n] (transposed: out[n * ldc + m]), n < n_out
  int B, rows_per_wg, n_tiles;
  // bias gradient of the same layer = column sums of one operand, accumulated next to the MFMAs:
  // mode 1: columns of Bm (workgroups with mt == 0), mode 2: columns of A (workgroups with nt == 0)
  float *bias_out;             // [ks][E][bias_member]
  size_t bias_member, bias_part;
  int bias_mode, bias_n;
  // f16 path: largest |value| of each operand per 32-row tile (left by the kernels that produce the arrays; a workgroup
  // takes the maximum over its member's tiles)
  const float *amax, *bmax;    // [E][max_tiles][kOpMax], already offset to this operand's slot
  int max_tiles;
};
constexpr int kOpMax = 8;      // slots per tile: 0 x, 1 h1, 2 h2, 3 d1, 4 d2, 5 d3

__global__ __launch_bounds__(kThreads, 2) void wgrad_kernel(const WgradArgs p) {
  extern __shared__ float red[];   // [2][128][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int mt = blockIdx.x / p.n_tiles, nt = blockIdx.x - mt * p.n_tiles;
  const int e = blockIdx.z;
  const int m0 = mt * 128, n0 = nt * 64;
  const int rpw = p.rows_per_wg / 4;
  const int start = blockIdx.y * p.rows_per_wg + wave * rpw;
  const int end = min(start + rpw, p.B);
  const float *ap = p.A + (size_t)e * p.B * p.lda + m0 + 4 * i;
  const float *bp = p.Bm + (size_t)e * p.B * p.ldb + n0 + 2 * i;
  const bool bmask = n0 + 2 * i < p.ldb;

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
  __shared__ float bred[4][128];
  const bool sum_b = p.bias_mode == 1 && mt == 0, sum_a = p.bias_mode == 2 && nt == 0;   // workgroup-uniform
  f32x4 asum = {0.0f, 0.0f, 0.0f, 0.0f};
  float2 bsum = {0.0f, 0.0f};

  // Software pipeline as in mfma_layer: the 8 batch rows of the next block are requested (ping-pong register
  // sets) before the 32 MFMAs of the current one.  Lanes whose columns lie beyond ldb read a valid dummy address:
  // an MFMA column only depends on its own lane's B value and those columns are never stored.
  if (start < end) {
    const float *bp_safe = bmask ? bp : p.Bm;
    auto request = [&](int r, f32x4 (&a)[4], float2 (&b)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t row = (size_t)(r + 2 * u + h);
        a[u] = *reinterpret_cast<const f32x4 *>(ap + row * p.lda);
        b[u] = *reinterpret_cast<const float2 *>(bp_safe + row * p.ldb);
      }
    };
    auto block = [&](const f32x4 (&a)[4], const float2 (&b)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][mi], b[u].x, acc[mi][0], 0, 0, 0);
          acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][mi], b[u].y, acc[mi][1], 0, 0, 0);
        }
      if (sum_b) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { bsum.x += b[u].x; bsum.y += b[u].y; }
      }
      if (sum_a) {
#pragma unroll
        for (int u = 0; u < 4; ++u) asum += a[u];
      }
    };
    f32x4 a0[4], a1[4];
    float2 b0[4], b1[4];
    const int full_end = start + ((end - start) & ~7);
    int r = start;
    if (r < full_end) {
      request(r, a0, b0);
#pragma unroll 1
      for (; r + 8 < full_end; r += 16) {
        request(r + 8, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        block(a0, b0);
        request(r + 16 < full_end ? r + 16 : r + 8, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        block(a1, b1);
      }
      if (r < full_end) block(a0, b0);
    }
    if (full_end < end) {   // ragged tail (batch not a multiple of 8 rows per wave): masked, not pipelined
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = full_end + 2 * u + h;
        a0[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        b0[u] = float2{0.0f, 0.0f};
        if (row < end) {
          a0[u] = *reinterpret_cast<const f32x4 *>(ap + (size_t)row * p.lda);
          b0[u] = *reinterpret_cast<const float2 *>(bp_safe + (size_t)row * p.ldb);
        }
      }
      block(a0, b0);
    }
  }

  auto spill = [&](int slot) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((slot * 128) + (mi * 2 + ni) * 16 + r) * 64 + lane] = acc[mi][ni][r];
  };
  auto absorb = [&](int slot) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] += red[((slot * 128) + (mi * 2 + ni) * 16 + r) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);   // one tile of LDS reads in flight at a time (register budget)
      }
  };
  if (sum_b) {   // lanes i and i + 32 hold the even / odd batch rows of the same two columns
    const float x = bsum.x + __shfl_xor(bsum.x, 32, 64), y = bsum.y + __shfl_xor(bsum.y, 32, 64);
    if (h == 0) { bred[wave][2 * i] = x; bred[wave][2 * i + 1] = y; }
  }
  if (sum_a) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float x = asum[c] + __shfl_xor(asum[c], 32, 64);
      if (h == 0) bred[wave][4 * i + c] = x;
    }
  }
  if (wave >= 2) spill(wave - 2);
  __syncthreads();
  if ((sum_b && tid < 64) || (sum_a && tid < 128)) {
    const int col = (sum_b ? n0 : m0) + tid;
    if (col < p.bias_n)
      p.bias_out[(size_t)blockIdx.y * p.bias_part + (size_t)e * p.bias_member + col] =
          (bred[0][tid] + bred[1][tid]) + (bred[2][tid] + bred[3][tid]);
  }
  if (wave < 2) absorb(wave);
  __syncthreads();
  if (wave == 1) spill(0);
  __syncthreads();
  if (wave != 0) return;
  absorb(0);

  float *out = p.out + (size_t)blockIdx.y * p.out_part + (size_t)e * p.out_member;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int irow = (r & 3) + 8 * (r >> 2) + 4 * h;
    if (!p.transposed) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + 4 * irow + mi;
        const int n = n0 + 2 * i;
        float *dst = out + (size_t)m * p.ldc + n;
        if (n + 1 < p.n_out) {
          if ((p.ldc & 1) == 0) *reinterpret_cast<float2 *>(dst) = float2{acc[mi][0][r], acc[mi][1][r]};
          else { dst[0] = acc[mi][0][r]; dst[1] = acc[mi][1][r]; }
        } else if (n < p.n_out) {
          dst[0] = acc[mi][0][r];
        }
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + 2 * i + ni;
        if (n < p.n_out)
          *reinterpret_cast<f32x4 *>(out + (size_t)n * p.ldc + m0 + 4 * irow) =
              f32x4{acc[0][ni][r], acc[1][ni][r], acc[2][ni][r], acc[3][ni][r]};
      }
    }
  }
}

// Workgroup-tiled variant for the 512-wide ensembles.  The register-direct kernel above moves (128 + 64) x 4 bytes per
// 2 x 128 x 64 x 2 flops -- 21 flop/B, i.e. ~7 TB/s of L2 traffic at the MFMA peak, and it measured 52 TFLOP/s on dW1.
// Here the 4 waves form a WM x WN grid of 128 x 64 wave tiles over the SAME batch rows and share the operands through
// LDS: per 8-row block the workgroup stages one [8][128 WM] slab of A and one [8][64 WN] slab of B (global -> registers
// one block ahead -> LDS, double buffered, one barrier per block), every wave reads its fragments back as one
// ds_read_b128 / ds_read_b64 per row pair.  (2 x 2): 256 x 128 tile, 43 flop/B; (4 x 1): 512 x 64 for the narrow
// operands of the first / last layer.  No cross-wave reduction: each wave owns its output tile.
template <int WM, int WN>
__device__ __forceinline__ void wgrad_lds_body(const WgradArgs &p, f32x4 *slab_mem, int bx, int by, int e) {
  constexpr int TM = 128 * WM, TN = 64 * WN;
  constexpr int A4 = 8 * TM / 4, B4 = 8 * TN / 4;          // float4 per slab
  constexpr int A_PER = A4 / kThreads, B_PER = (B4 + kThreads - 1) / kThreads;
  f32x4 *slab[2] = {slab_mem, slab_mem + A4 + B4};
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int mt = bx / p.n_tiles, nt = bx - mt * p.n_tiles;
  const int m0 = mt * TM, n0 = nt * TN;
  const int start = by * p.rows_per_wg;
  const int end = min(start + p.rows_per_wg, p.B);
  const float *Abase = p.A + (size_t)e * p.B * p.lda + m0;
  const float *Bbase = p.Bm + (size_t)e * p.B * p.ldb + n0;

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
  const bool sum_b = p.bias_mode == 1 && mt == 0 && wm == 0, sum_a = p.bias_mode == 2 && nt == 0 && wn == 0;
  f32x4 asum = {0.0f, 0.0f, 0.0f, 0.0f};
  float2 bsum = {0.0f, 0.0f};

  // two register staging sets: a slab is requested TWO blocks (~2 x 2048 MFMA cycles) before it is written to LDS --
  // one block of lead does not cover an L2 miss (the operands stream from HBM / MALL, they are read once per XCD)
  f32x4 ra0[A_PER], rb0[B_PER], ra1[A_PER], rb1[B_PER];
  auto fetch = [&](int r0, f32x4 (&ra)[A_PER], f32x4 (&rb)[B_PER]) {
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
      const int q = tid + u * kThreads;
      const int row = q / (TM / 4), c4 = q - row * (TM / 4);
      ra[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (r0 + row < end) ra[u] = *reinterpret_cast<const f32x4 *>(Abase + (size_t)(r0 + row) * p.lda + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < B_PER; ++u) {
      const int q = tid + u * kThreads;
      const int row = q / (TN / 4), c4 = q - row * (TN / 4);
      rb[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (q < B4 && r0 + row < end && n0 + 4 * c4 < p.ldb)
        rb[u] = *reinterpret_cast<const f32x4 *>(Bbase + (size_t)(r0 + row) * p.ldb + 4 * c4);
    }
  };
  auto stash = [&](int buf, const f32x4 (&ra)[A_PER], const f32x4 (&rb)[B_PER]) {
#pragma unroll
    for (int u = 0; u < A_PER; ++u) slab[buf][tid + u * kThreads] = ra[u];
#pragma unroll
    for (int u = 0; u < B_PER; ++u)
      if (tid + u * kThreads < B4) slab[buf][A4 + tid + u * kThreads] = rb[u];
  };
  auto compute = [&](int buf) {
    const float *As = reinterpret_cast<const float *>(slab[buf]) + wm * 128 + 4 * i;
    const float *Bs = reinterpret_cast<const float *>(slab[buf] + A4) + wn * 64 + 2 * i;
    f32x4 a[4];
    float2 b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = *reinterpret_cast<const f32x4 *>(As + (2 * u + h) * TM);
      b[u] = *reinterpret_cast<const float2 *>(Bs + (2 * u + h) * TN);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][mi], b[u].x, acc[mi][0], 0, 0, 0);
        acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][mi], b[u].y, acc[mi][1], 0, 0, 0);
      }
    if (sum_b) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { bsum.x += b[u].x; bsum.y += b[u].y; }
    }
    if (sum_a) {
#pragma unroll
      for (int u = 0; u < 4; ++u) asum += a[u];
    }
  };

  if (start < end) {
    fetch(start, ra0, rb0);
    if (start + 8 < end) fetch(start + 8, ra1, rb1);
    stash(0, ra0, rb0);
    __syncthreads();
    // even blocks live in slab[0], odd blocks in slab[1]
#pragma unroll 1
    for (int r0 = start; r0 < end; r0 += 16) {
      if (r0 + 16 < end) fetch(r0 + 16, ra0, rb0);
      compute(0);
      if (r0 + 8 < end) stash(1, ra1, rb1);
      __syncthreads();
      if (r0 + 8 >= end) break;
      if (r0 + 24 < end) fetch(r0 + 24, ra1, rb1);
      compute(1);
      if (r0 + 16 < end) stash(0, ra0, rb0);
      __syncthreads();
    }
  }

  const int m0w = m0 + wm * 128, n0w = n0 + wn * 64;
  if (sum_b) {
    const float x = bsum.x + __shfl_xor(bsum.x, 32, 64), y = bsum.y + __shfl_xor(bsum.y, 32, 64);
    float *dst = p.bias_out + (size_t)by * p.bias_part + (size_t)e * p.bias_member;
    if (h == 0) {
      if (n0w + 2 * i < p.bias_n) dst[n0w + 2 * i] = x;
      if (n0w + 2 * i + 1 < p.bias_n) dst[n0w + 2 * i + 1] = y;
    }
  }
  if (sum_a) {
    float *dst = p.bias_out + (size_t)by * p.bias_part + (size_t)e * p.bias_member;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float x = asum[c] + __shfl_xor(asum[c], 32, 64);
      if (h == 0 && m0w + 4 * i + c < p.bias_n) dst[m0w + 4 * i + c] = x;
    }
  }
  float *out = p.out + (size_t)by * p.out_part + (size_t)e * p.out_member;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int irow = (r & 3) + 8 * (r >> 2) + 4 * h;
    if (!p.transposed) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int m = m0w + 4 * irow + mi;
        const int n = n0w + 2 * i;
        float *dst = out + (size_t)m * p.ldc + n;
        if (n + 1 < p.n_out) {
          if ((p.ldc & 1) == 0) *reinterpret_cast<float2 *>(dst) = float2{acc[mi][0][r], acc[mi][1][r]};
          else { dst[0] = acc[mi][0][r]; dst[1] = acc[mi][1][r]; }
        } else if (n < p.n_out) {
          dst[0] = acc[mi][0][r];
        }
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = n0w + 2 * i + ni;
        if (n < p.n_out)
          *reinterpret_cast<f32x4 *>(out + (size_t)n * p.ldc + m0w + 4 * irow) =
              f32x4{acc[0][ni][r], acc[1][ni][r], acc[2][ni][r], acc[3][ni][r]};
      }
    }
  }
}

// The same tiling on the f16 pipe: each float32 product as three v_mfma_f32_32x32x16_f16 on two-piece operands (f16_split.h).
// K of an MFMA = 16 batch rows, so a block is 16 rows: global -> registers two blocks ahead -> split (one power-of-two lift
// per (member, operand), from the largest magnitude the producing kernel left in `amax / bmax`: the batch is the K dimension,
// so a lift may not vary from row to row) -> LDS as two-piece [row][feature] images -> the hardware's transposed read
// (ds_read_b64_tr_b16: four batch rows of one feature per read, cdna_hip_programming.md T10) delivers MFMA fragments with
// the batch on the k index.  24 MFMAs of 32 cycles per block and wave against the 64 of 64 cycles the fp32 form needs for
// the same 16 rows.  Row strides of the images are 16 dwords past a multiple of 64 (8 for the 512-wide A slab, which has to
// fit two workgroups per CU): the four rows of a transposed read fall on different banks.
template <int WM, int WN>
__device__ __forceinline__ void wgrad_f16_body(const WgradArgs &p, char *smem, int bx, int by, int e) {
  constexpr int TM = 128 * WM, TN = 64 * WN;
  constexpr int SA = TM + (WM == 4 ? 16 : 32), SB = TN + 32;     // row strides, halves
  constexpr int A_BYTES = 2 * 16 * SA * 2, B_BYTES = 2 * 16 * SB * 2, BUF = A_BYTES + B_BYTES;
  constexpr int A4 = 16 * TM / 4, B4 = 16 * TN / 4;              // float4 per block
  constexpr int A_PER = A4 / kThreads, B_PER = (B4 + kThreads - 1) / kThreads;
  static_assert(A4 % kThreads == 0 && (kThreads % (TM / 4)) == 0 && (kThreads % (TN / 4)) == 0, "a thread keeps its columns");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int mt = bx / p.n_tiles, nt = bx - mt * p.n_tiles;
  const int m0 = mt * TM, n0 = nt * TN;
  const int start = by * p.rows_per_wg;
  const int end = min(start + p.rows_per_wg, p.B);
  const float *Abase = p.A + (size_t)e * p.B * p.lda + m0;
  const float *Bbase = p.Bm + (size_t)e * p.B * p.ldb + n0;
  // the member's largest operand magnitudes -> one power-of-two lift per operand
  float la, lb;
  {
    float ma = 0.0f, mb = 0.0f;
    for (int k = tid; k < p.max_tiles; k += kThreads) {
      ma = fmaxf(ma, p.amax[((size_t)e * p.max_tiles + k) * kOpMax]);
      mb = fmaxf(mb, p.bmax[((size_t)e * p.max_tiles + k) * kOpMax]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ma = fmaxf(ma, __shfl_xor(ma, o, 64)); mb = fmaxf(mb, __shfl_xor(mb, o, 64)); }
    float *red = reinterpret_cast<float *>(smem);
    if (lane == 0) { red[wave] = ma; red[4 + wave] = mb; }
    __syncthreads();
    la = pow2_lift(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    lb = pow2_lift(fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
    __syncthreads();
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
  const bool sum_b = p.bias_mode == 1 && mt == 0, sum_a = p.bias_mode == 2 && nt == 0;
  f32x4 asum = {0.0f, 0.0f, 0.0f, 0.0f}, bsum = {0.0f, 0.0f, 0.0f, 0.0f};   // this thread's columns, over the blocks' rows

  // register staging: ONE set -- a block is requested one block (24 MFMAs per wave, interleaved with the CU's other
  // workgroup) before it is split into LDS.  A second set (two blocks of lead, as the fp32 form has) does not fit beside
  // the 128 accumulators, the fragments and the split's temporaries: 588 B of scratch per lane for the 256 x 128 tiles,
  // 860 B for the 512 x 64 ones (TWO_SETS keeps the loop for a build that finds the registers).
  constexpr bool TWO_SETS = false;
  f32x4 ra0[A_PER], rb0[B_PER], ra1[TWO_SETS ? A_PER : 1], rb1[TWO_SETS ? B_PER : 1];
  auto fetch = [&](int r0, f32x4 (&ra)[A_PER], f32x4 (&rb)[B_PER]) {
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
      const int q = tid + u * kThreads;
      const int row = q / (TM / 4), c4 = q - row * (TM / 4);
      ra[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (r0 + row < end) ra[u] = *reinterpret_cast<const f32x4 *>(Abase + (size_t)(r0 + row) * p.lda + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < B_PER; ++u) {
      const int q = tid + u * kThreads;
      const int row = q / (TN / 4), c4 = q - row * (TN / 4);
      rb[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (q < B4 && r0 + row < end && n0 + 4 * c4 < p.ldb)
        rb[u] = *reinterpret_cast<const f32x4 *>(Bbase + (size_t)(r0 + row) * p.ldb + 4 * c4);
    }
  };
  auto stash = [&](int buf, const f32x4 (&ra)[A_PER], const f32x4 (&rb)[B_PER]) {
    _Float16 *Ai = reinterpret_cast<_Float16 *>(smem + buf * BUF), *Bi = reinterpret_cast<_Float16 *>(smem + buf * BUF + A_BYTES);
#pragma unroll
    for (int u = 0; u < A_PER; ++u) {
      const int q = tid + u * kThreads;
      const int row = q / (TM / 4), c4 = q - row * (TM / 4);
      unsigned q1[2], q2[2];
      split2<false>(ra[u][0], ra[u][1], la, q1[0], q2[0]);
      split2<false>(ra[u][2], ra[u][3], la, q1[1], q2[1]);
      *reinterpret_cast<uint2 *>(Ai + row * SA + 4 * c4) = make_uint2(q1[0], q1[1]);
      *reinterpret_cast<uint2 *>(Ai + (16 + row) * SA + 4 * c4) = make_uint2(q2[0], q2[1]);
      if (sum_a) asum += ra[u];
    }
#pragma unroll
    for (int u = 0; u < B_PER; ++u) {
      const int q = tid + u * kThreads;
      if (q < B4) {
        const int row = q / (TN / 4), c4 = q - row * (TN / 4);
        unsigned q1[2], q2[2];
        split2<false>(rb[u][0], rb[u][1], lb, q1[0], q2[0]);
        split2<false>(rb[u][2], rb[u][3], lb, q1[1], q2[1]);
        *reinterpret_cast<uint2 *>(Bi + row * SB + 4 * c4) = make_uint2(q1[0], q1[1]);
        *reinterpret_cast<uint2 *>(Bi + (16 + row) * SB + 4 * c4) = make_uint2(q2[0], q2[1]);
        if (sum_b) bsum += rb[u];
      }
    }
  };
  // fragment of the block's 16 rows (k = 8 (lane >> 5) + j) of columns c0 + (lane & 31): two transposed reads per piece
  typedef __fp16 h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
  typedef __attribute__((address_space(3))) h16x4 *lds_h4;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  auto frag = [&](const _Float16 *img, int stride, int c0, f16x8 &p1, f16x8 &p2) {
    const int t = lane & 15, g = lane >> 4;
    const _Float16 *a = img + (8 * (g >> 1) + (t >> 2)) * stride + c0 + 16 * (g & 1) + 4 * (t & 3);
    const u32x2 l1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)a));
    const u32x2 h1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(a + 4 * stride)));
    const u32x2 l2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(a + 16 * stride)));
    const u32x2 h2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(a + 20 * stride)));
    const u32x4 q1 = {l1[0], l1[1], h1[0], h1[1]}, q2 = {l2[0], l2[1], h2[0], h2[1]};
    p1 = __builtin_bit_cast(f16x8, q1);
    p2 = __builtin_bit_cast(f16x8, q2);
  };
  auto compute = [&](int buf) {
    const _Float16 *Ai = reinterpret_cast<const _Float16 *>(smem + buf * BUF), *Bi = reinterpret_cast<const _Float16 *>(smem + buf * BUF + A_BYTES);
    f16x8 b1[2], b2[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) frag(Bi, SB, wn * 64 + 32 * ni, b1[ni], b2[ni]);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      f16x8 a1, a2;
      frag(Ai, SA, wm * 128 + 32 * mi, a1, a2);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) mm3(acc[mi][ni], a1, a2, b1[ni], b2[ni]);
      __builtin_amdgcn_sched_barrier(0);      // one A fragment pair live at a time (hoisted, the four of them spill)
    }
  };

  if (start < end) {
    if constexpr (TWO_SETS) {
      fetch(start, ra0, rb0);
      if (start + 16 < end) fetch(start + 16, ra1, rb1);
      stash(0, ra0, rb0);
      __syncthreads();
      // even blocks live in buffer 0, odd blocks in buffer 1
#pragma unroll 1
      for (int r0 = start; r0 < end; r0 += 32) {
        if (r0 + 32 < end) fetch(r0 + 32, ra0, rb0);
        compute(0);
        if (r0 + 16 < end) stash(1, ra1, rb1);
        __syncthreads();
        if (r0 + 16 >= end) break;
        if (r0 + 48 < end) fetch(r0 + 48, ra1, rb1);
        compute(1);
        if (r0 + 32 < end) stash(0, ra0, rb0);
        __syncthreads();
      }
    } else {
      fetch(start, ra0, rb0);
      stash(0, ra0, rb0);
      __syncthreads();
      int buf = 0;
#pragma unroll 1
      for (int r0 = start; r0 < end; r0 += 16, buf ^= 1) {
        const bool more = r0 + 16 < end;
        if (more) fetch(r0 + 16, ra0, rb0);
        compute(buf);
        if (more) stash(buf ^ 1, ra0, rb0);
        __syncthreads();
      }
    }
  }

  // ---- bias gradients: the threads that staged the same columns (tid, tid + TM / 4, ...) add their partial sums in order
  if (sum_a || sum_b) {
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem);
    float *dst = p.bias_out + (size_t)by * p.bias_part + (size_t)e * p.bias_member;
    if (sum_a) {
      constexpr int NP = kThreads / (TM / 4);
      *reinterpret_cast<f32x4 *>(red + 4 * tid) = asum;          // [part][TM]: tid = part * (TM / 4) + c4
      __syncthreads();
      for (int c = tid; c < TM; c += kThreads) {
        float v = 0.0f;
#pragma unroll
        for (int q = 0; q < NP; ++q) v += red[q * TM + c];
        if (m0 + c < p.bias_n) dst[m0 + c] = v;
      }
    } else {
      constexpr int NP = kThreads / (TN / 4);
      *reinterpret_cast<f32x4 *>(red + 4 * tid) = bsum;
      __syncthreads();
      for (int c = tid; c < TN; c += kThreads) {
        float v = 0.0f;
#pragma unroll
        for (int q = 0; q < NP; ++q) v += red[q * TN + c];
        if (n0 + c < p.bias_n) dst[n0 + c] = v;
      }
    }
  }

  const float un = 1.0f / (la * lb);          // both powers of two
  const int m0w = m0 + wm * 128, n0w = n0 + wn * 64;
  float *out = p.out + (size_t)by * p.out_part + (size_t)e * p.out_member;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0w + 32 * ni + i;
      if (!p.transposed) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0w + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (n < p.n_out) out[(size_t)m * p.ldc + n] = acc[mi][ni][r] * un;
        }
      } else if (n < p.n_out) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = m0w + 32 * mi + 8 * q + 4 * h;
          *reinterpret_cast<f32x4 *>(out + (size_t)n * p.ldc + m) =
              f32x4{acc[mi][ni][4 * q] * un, acc[mi][ni][4 * q + 1] * un, acc[mi][ni][4 * q + 2] * un, acc[mi][ni][4 * q + 3] * un};
        }
      }
    }
}

// All three weight-gradient GEMMs of a 512-wide ensemble in one launch: the square layer's 256 x 128 tiles first, the
// two narrow layers' 512 x 64 tiles behind them, so the small grids fill the CUs the big one frees instead of each
// running alone on half of the chip.  Workgroup index -> (layer segment, tile, K split, member).
struct WgradAllArgs {
  WgradArgs g[3];          // order of execution: layer 1, layer 0, layer 2
  int first[4];            // first workgroup of each segment
  int tiles[3], ks[3];     // tiles per (member, split), K splits
  int n_chunks0;           // (member, split) chunks of the first segment; its index space is padded to a multiple of 64
};

constexpr int kWgradF16Lds = 2 * (2 * 16 * (512 + 16) * 2 + 2 * 16 * (64 + 32) * 2);   // the (4 x 1) tiles' two buffers: 79 872 B

template <bool F16>
__global__ __launch_bounds__(kThreads, 2) void wgrad_all_kernel(const WgradAllArgs p) {
  extern __shared__ f32x4 slab_mem[];     // fp32: 2 x (8 x 512 / 4 + 8 x 64 / 4) float4; f16: kWgradF16Lds bytes
  const int blk = blockIdx.x;
  const int seg = blk < p.first[1] ? 0 : (blk < p.first[2] ? 1 : 2);
  int q = blk - p.first[seg];
  if (seg == 0) {
    // XCD-aware order for the square layer: the 8 tiles of one (member, K split) chunk read the same rows of A and B,
    // and workgroups go to the 8 XCDs round-robin by index -- so chunk c is given the indices congruent to c mod 8
    // (its tiles 8 apart, i.e. consecutive in that XCD's queue): the chunk's ~1 MB of operands is fetched from HBM once into
    // ONE L2 instead of once per XCD (PMC: 230 MB fetched per launch against 124 MB algorithmic before).
    const int T = p.tiles[0];                       // 8
    const int xcd = q & 7, slot = q >> 3;            // slot = position in this XCD's share
    const int bx = slot % T;
    const int c = (slot / T) * 8 + xcd;
    if (c >= p.n_chunks0) return;
    if constexpr (F16) wgrad_f16_body<2, 2>(p.g[0], reinterpret_cast<char *>(slab_mem), bx, c % p.ks[0], c / p.ks[0]);
    else wgrad_lds_body<2, 2>(p.g[0], slab_mem, bx, c % p.ks[0], c / p.ks[0]);
    return;
  }
  const int bx = q % p.tiles[seg];
  q /= p.tiles[seg];
  const int by = q % p.ks[seg], e = q / p.ks[seg];
  if constexpr (F16) wgrad_f16_body<4, 1>(p.g[seg], reinterpret_cast<char *>(slab_mem), bx, by, e);
  else wgrad_lds_body<4, 1>(p.g[seg], slab_mem, bx, by, e);
}

// ------------------------------------------------------------------------------------------------------------
// Fused step for the small deterministic ensembles (critics: 128-wide, MSE, output width <= 32).  At these sizes
// the separate kernels above are latency chains of a few microseconds each, so the whole gradient -- gather,
// scaler, forward, output delta, backward chain, weight and bias gradients -- runs in ONE kernel built like the
// policy-update kernel (policy_update.hip): tiles of 32 rows on the MFMA columns, every activation image once in
// LDS in the row layout, swish' kept in registers between the forward and the backward epilogue, weight gradients
// as MFMAs over the tile's rows accumulated in registers across the workgroup's tiles.  Each workgroup writes its
// partial gradient with plain stores (Adam adds the partials in a fixed order: still no atomics).
// ------------------------------------------------------------------------------------------------------------
struct FusedArgs {
  const f32x4 *F0, *F1, *F2, *B1, *B2;
  size_t sF0, sF1, sF2, sB1, sB2;          // per member, float4 units
  const float *b0, *b1, *b2;
  int b2_ld;
  const float *in_mu, *in_sig, *out_mu, *out_sig;
  const float *inputs, *targets;
  const int32_t *idx;
  int idx_stride;
  int E, I, IP, kg0, O, kga, B;
  float *pW[3], *pB[3];                    // partial buffers [workgroup][E][...]
  size_t sW[3], sB[3];                     // floats per partial
};

// One workgroup per CU (the grid is at most 32 x E workgroups) buys a 512-register budget, and that changes the
// shape of the kernel: with one 32-column n-tile per wave a k-group is only 4 MFMAs (256 cycles), far too short to
// hide an L2 round trip behind a one-group prefetch, so the member's weights are fetched ONCE at kernel start -- the
// forward fragments (~110 registers per lane) stay in registers for every tile, W1^T (64 KB) and the biases sit in LDS
// next to the 76 KB of activation images -- and the rows and targets of the next tile are gathered into registers
// while the current tile computes.
template <int KG>
__device__ __forceinline__ void load_frags(f32x4 (&a)[KG], const f32x4 *wp, int kg, int lane) {
#pragma unroll
  for (int g = 0; g < KG; ++g) {
    a[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (g < kg) a[g] = wp[(size_t)g * 64 + lane];
  }
}

// acc += A(frags) * B, B = rows [sample][stride] in LDS, k-groups [0, kg) starting at column k0
template <int KG>
__device__ __forceinline__ void mfma_frags(const f32x4 (&a)[KG], int kg, const float *rows, int stride, int k0, int lane,
                                           f32x16 &acc) {
  const float *rl = rows + (lane & 31) * stride + 4 * (lane >> 5) + k0;
#pragma unroll
  for (int g = 0; g < KG; ++g) {
    if (g < kg) {
      const f32x4 b = *reinterpret_cast<const f32x4 *>(rl + 8 * g);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][s], b[s], acc, 0, 0, 0);
    }
  }
}

template <int N_IT>
__global__ __launch_bounds__(kThreads, 1) void fused_mse_step_kernel(const FusedArgs p) {
  constexpr int HID = 128, KGH = HID / 8, RS = HID + 4, RED_LD = BB + 1;
  constexpr int KG0 = 4 * N_IT;             // k-groups of the (padded) input width
  extern __shared__ f32x4 smem4[];
  float *sm = reinterpret_cast<float *>(smem4);
  const int XS = p.IP + 4;
  float *xR = sm;                       // [BB][XS]
  float *h1R = xR + BB * XS;            // [BB][RS]
  float *h2R = h1R + BB * RS;
  float *u1R = h2R + BB * RS;           // split-K reduction image of the output layer, then delta1
  float *u2R = u1R + BB * RS;           // delta2
  float *wR = u2R + BB * RS;            // [BB][36] output delta
  float *biasL = wR + BB * 36;          // [b0 | b1] of the member
  f32x4 *bk1L = reinterpret_cast<f32x4 *>(biasL + 2 * HID);   // W1^T fragments [n-tile][k-group][lane] (64 KB)
  float *red = u1R, *d1R = u1R, *d2R = u2R;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int e = blockIdx.y;
  const int n_tiles = (p.B + BB - 1) / BB;
  const float inv_bd = 1.0f / ((float)p.B * (float)p.O);

  // ---- the member's weights, once: forward fragments in registers, W1^T and the biases in LDS -----------------
  f32x4 f0[KG0], f1[KGH], f2[4], bk2[4];
  load_frags<KG0>(f0, p.F0 + e * p.sF0 + (size_t)wave * p.kg0 * 64, p.kg0, lane);
  load_frags<KGH>(f1, p.F1 + e * p.sF1 + (size_t)wave * KGH * 64, KGH, lane);
  load_frags<4>(f2, p.F2 + e * p.sF2 + (size_t)wave * 4 * 64, 4, lane);        // K split: k-groups 4w .. 4w + 3
  load_frags<4>(bk2, p.B2 + e * p.sB2 + (size_t)wave * p.kga * 64, p.kga, lane);
  {
    const f32x4 *src = p.B1 + e * p.sB1;
    f32x4 tmp[KGH];
#pragma unroll
    for (int u = 0; u < KGH; ++u) tmp[u] = src[tid + u * kThreads];     // 4 n-tiles x 16 k-groups x 64 lanes
#pragma unroll
    for (int u = 0; u < KGH; ++u) bk1L[tid + u * kThreads] = tmp[u];
    biasL[tid] = (tid < HID) ? p.b0[(size_t)e * HID + tid] : p.b1[(size_t)e * HID + tid - HID];
  }
  const f32x4 *bk1W = bk1L + (size_t)wave * KGH * 64 + lane;
  const float *b2 = p.b2 + (size_t)e * p.b2_ld;

  f32x16 gW1[4], gW0[N_IT], gW2;
#pragma unroll
  for (int t = 0; t < 4; ++t) zero(gW1[t]);
#pragma unroll
  for (int t = 0; t < N_IT; ++t) zero(gW0[t]);
  zero(gW2);
  float gbias = 0.0f;                       // thread n < 128: db1[n]; thread 128 + n: db0[n]
  float gb2p[4] = {0.0f, 0.0f, 0.0f, 0.0f};

  for (int i = tid; i < BB * 36; i += kThreads) wR[i] = 0.0f;   // padded delta columns stay zero

  // gather (and scale) this thread's share of a tile's rows / targets into registers
  float xn[KG0], tn[4];
  auto gather = [&](int tile) {
    const int row0 = tile * BB;
#pragma unroll
    for (int u = 0; u < KG0; ++u) {
      const int i = tid + u * kThreads;
      xn[u] = 0.0f;
      if (i < p.IP * BB) {
        const int b = i / p.IP, k = i - b * p.IP;
        const int r = row0 + b;
        if (r < p.B && k < p.I) {
          const int src = p.idx ? p.idx[(size_t)e * p.idx_stride + r] : r;
          float v = p.inputs[(size_t)src * p.I + k];
          if (p.in_mu) v = (v - p.in_mu[k]) / p.in_sig[k];
          xn[u] = v;
        }
      }
    }
    const int er = row0 + (tid & 31);
    const int src = (er < p.B) ? (p.idx ? p.idx[(size_t)e * p.idx_stride + er] : er) : -1;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int a = (tid >> 5) + 8 * it;
      tn[it] = 0.0f;
      if (a < p.O && src >= 0) {
        float t = p.targets[(size_t)src * p.O + a];
        if (p.out_mu) t = (t - p.out_mu[a]) / p.out_sig[a];
        tn[it] = t;
      }
    }
  };
  if ((int)blockIdx.x < n_tiles) gather(blockIdx.x);
  __syncthreads();

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int row0 = tile * BB;
#pragma unroll
    for (int u = 0; u < KG0; ++u) {
      const int i = tid + u * kThreads;
      if (i < p.IP * BB) {
        const int b = i / p.IP, k = i - b * p.IP;
        xR[b * XS + k] = xn[u];
      }
    }
    float tc[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) tc[it] = tn[it];
    __syncthreads();
    if (tile + (int)gridDim.x < n_tiles) gather(tile + gridDim.x);   // in flight under this tile's chain
    // ---- forward -------------------------------------------------------------------------------------------
    f32x16 acc = load_bias(biasL, wave * 32, lane), g1, g2;
    mfma_frags<KG0>(f0, p.kg0, xR, XS, 0, lane, acc);
    {
      f32x16 hv;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float z = acc[r], sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
        hv[r] = z * sg;
        g1[r] = sg * (1.0f + z * (1.0f - sg));
      }
      store_tile_R(hv, wave * 32, h1R, RS, lane);
    }
    __syncthreads();
    acc = load_bias(biasL + HID, wave * 32, lane);
    mfma_frags<KGH>(f1, KGH, h1R, RS, 0, lane, acc);
    {
      f32x16 hv;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float z = acc[r], sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
        hv[r] = z * sg;
        g2[r] = sg * (1.0f + z * (1.0f - sg));
      }
      store_tile_R(hv, wave * 32, h2R, RS, lane);
    }
    __syncthreads();
    zero(acc);                            // output layer: K split over the 4 waves
    mfma_frags<4>(f2, 4, h2R, RS, wave * 32, lane, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int a = (r & 3) + 8 * (r >> 2) + 4 * h;
      red[(wave * 32 + a) * RED_LD + j] = acc[r];
    }
    __syncthreads();
    // ---- output delta: d(sum_e mean 0.5 (o - t)^2) / d o = (o - t) / (B O), pe.py:911-919 ------------------
    {
      const int eb = tid & 31;
      const bool valid = row0 + eb < p.B;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int a = (tid >> 5) + 8 * it;
        if (a < p.O) {
          float m = red[(0 * 32 + a) * RED_LD + eb];
          m += red[(1 * 32 + a) * RED_LD + eb];
          m += red[(2 * 32 + a) * RED_LD + eb];
          m += red[(3 * 32 + a) * RED_LD + eb];
          m += b2[a];
          const float cot = valid ? (m - tc[it]) * inv_bd : 0.0f;
          wR[eb * 36 + a] = cot;
          gb2p[it] += cot;
        }
      }
    }
    __syncthreads();
    // ---- backward: delta2 = (W2 d3) g2 ; delta1 = (W1 delta2) g1 -----------------------------------------
    zero(acc);
    mfma_frags<4>(bk2, p.kga, wR, 36, 0, lane, acc);
    {
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = g2[r] * acc[r];
      store_tile_R(o, wave * 32, d2R, RS, lane);
    }
    __syncthreads();
    zero(acc);
    {
      const float *rl = d2R + j * RS + 4 * h;
#pragma unroll
      for (int g = 0; g < KGH; ++g) {
        const f32x4 a = bk1W[g * 64], b = *reinterpret_cast<const f32x4 *>(rl + 8 * g);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], b[s2], acc, 0, 0, 0);
      }
    }
    {
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = g1[r] * acc[r];
      store_tile_R(o, wave * 32, d1R, RS, lane);     // the reduction image is dead (barrier after the delta phase)
    }
    __syncthreads();
    // ---- weight gradients: K = the tile's rows -----------------------------------------------------------
#pragma unroll
    for (int J = 0; J < 4; ++J) wgrad_tile(gW1[J], h1R, RS, wave * 32, d2R, RS, J * 32, lane);
#pragma unroll
    for (int t = 0; t < N_IT; ++t) wgrad_tile(gW0[t], xR, XS, 32 * t, d1R, RS, wave * 32, lane);
    wgrad_tile(gW2, h2R, RS, wave * 32, wR, 36, 0, lane);
    {
      const float *img = (tid < HID) ? d2R : d1R;
      const int n = tid & (HID - 1);
      float sb = 0.0f;
#pragma unroll 8
      for (int b = 0; b < BB; ++b) sb += img[b * RS + n];
      gbias += sb;
    }
    __syncthreads();
  }

  // ---- this workgroup's partial gradient ----------------------------------------------------------------------
  const size_t part = blockIdx.x;
  float *oW0 = p.pW[0] + part * p.sW[0] + (size_t)e * p.I * HID;
  float *oW1 = p.pW[1] + part * p.sW[1] + (size_t)e * HID * HID;
  float *oW2 = p.pW[2] + part * p.sW[2] + (size_t)e * HID * p.O;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
    for (int J = 0; J < 4; ++J) oW1[(wave * 32 + row) * HID + J * 32 + j] = gW1[J][r];
#pragma unroll
    for (int t = 0; t < N_IT; ++t)
      if (32 * t + row < p.I) oW0[(32 * t + row) * HID + wave * 32 + j] = gW0[t][r];
    if (j < p.O) oW2[(wave * 32 + row) * p.O + j] = gW2[r];
  }
  {
    const int l = (tid < HID) ? 1 : 0;
    p.pB[l][part * p.sB[l] + (size_t)e * HID + (tid & (HID - 1))] = gbias;
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int a = (tid >> 5) + 8 * it;
    const float sb = half_sum(gb2p[it]);
    if (a < p.O && (tid & 31) == 0) p.pB[2][part * p.sB[2] + (size_t)e * p.O + a] = sb;
  }
}

// ------------------------------------------------------------------------------------------------------------
// Adam (tf.train.AdamOptimizer: m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2); w -= lr_t m / (sqrt(v) + eps)) and
// re-packing of the updated weight into the forward ([n-tile][k-group][lane][4] of W) and backward (same layout
// of W^T) MFMA images.
// ------------------------------------------------------------------------------------------------------------
struct AdamWArgs {
  float *W, *m, *v;
  const float *parts; int n_parts; size_t part_stride;
  float decay;
  int E, K, N;
  float *fwd; int f_kg; size_t f_stride;   // forward pack: k-groups per n-tile, floats per member
  float *bwd; int b_kg; size_t b_stride;   // backward pack (nullptr for the first layer)
  float lr_t, b1, b2, eps;
  int apply;                               // 0: pack only (weights loaded from the host)
  float *wmax_part;                        // [blocks] largest |w| of each block's 256 consecutive elements (nullptr: not wanted);
                                           // a member's elements are a whole number of blocks: the f16 backward chain lifts W^T by it
};

__device__ __forceinline__ size_t pack_index(int k, int n, int kg) {
  return ((((size_t)(n >> 5) * kg + (k >> 3)) * 64 + ((k >> 2) & 1) * 32 + (n & 31)) << 2) + (k & 3);
}

__device__ __forceinline__ void adam_w(const AdamWArgs &p, unsigned block) {
  const size_t per = (size_t)p.K * p.N;
  const size_t i = (size_t)block * kThreads + threadIdx.x;
  float w = 0.0f;
  if (i < per * p.E) {
    const int e = (int)(i / per);
    const size_t r = i - (size_t)e * per;
    const int k = (int)(r / p.N), n = (int)(r - (size_t)k * p.N);
    w = p.W[i];
    if (p.apply) {
      float g = 0.0f;
#pragma unroll 8
      for (int s = 0; s < p.n_parts; ++s) g += p.parts[(size_t)s * p.part_stride + i];   // fixed order: reproducible
      g += p.decay * w;                      // d/dw of decay * l2_loss(w), models/pens/fc.py:167-168
      float m = p.m[i], v = p.v[i];
      m += (g - m) * (1.0f - p.b1);
      v += (g * g - v) * (1.0f - p.b2);
      w -= p.lr_t * m / (sqrtf(v) + p.eps);
      p.m[i] = m; p.v[i] = v; p.W[i] = w;
    }
    p.fwd[(size_t)e * p.f_stride + pack_index(k, n, p.f_kg)] = w;
    if (p.bwd) p.bwd[(size_t)e * p.b_stride + pack_index(n, k, p.b_kg)] = w;
  }
  if (p.wmax_part) {      // (uniform per tensor)
    __shared__ float s_wm[kThreads / 64];
    float m = fabsf(w);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) s_wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) p.wmax_part[block] = fmaxf(fmaxf(s_wm[0], s_wm[1]), fmaxf(s_wm[2], s_wm[3]));
  }
}

// Every f16 image of a training step in ONE launch.  Image i (of n_img): fp32 pack [n-tile][k-group][lane][4] of layer
// layer[i] -> two-piece image [n-tile][k-slab 16][piece 2][lane 64] of 8 halves (the layout of h3_pack_kernel, plain k order),
// lifted by the member's power of two, which every workgroup takes itself from the block maxima Adam left for the layer (a
// member's fragments of an image are a whole number of 256-thread workgroups); the first workgroup of a member in an image that
// `writes_stats` also leaves {lift, max |W|} in stats[e][4 l], [4 l + 3] for the kernels that unscale the products.
struct PackAllArgs {
  const float *src[5]; size_t src_stride[5];     // fp32 packs, floats per member
  f16x8 *dst[5]; size_t dst_stride[5];           // images, 16-byte units per member
  int kg[5], src_tiles[5], n_tiles[5], slabs[5], layer[5], writes_stats[5];
  unsigned first[6];                             // workgroups [first[i], first[i + 1]) belong to image i
  const float *wmax[3]; int blocks[3];           // per layer: [E][blocks] block maxima
  float *stats;
  int n_img, E;
};

__global__ __launch_bounds__(kThreads) void train_pack_all_kernel(const PackAllArgs p) {
  int img = 0;
#pragma unroll
  for (int i = 1; i < 5; ++i)
    if (i < p.n_img && blockIdx.x >= p.first[i]) img = i;
  const int tid = threadIdx.x;
  const long per = (long)p.n_tiles[img] * p.slabs[img] * 64;       // lanes of one member's image
  const long idx = (long)(blockIdx.x - p.first[img]) * kThreads + tid;
  const int e = (int)(idx / per);
  const int l = p.layer[img];
  // the member's largest |W| of this layer
  __shared__ float s_m[kThreads / 64];
  float m = 0.0f;
  {
    const float *part = p.wmax[l] + (size_t)e * p.blocks[l];
    for (int i = tid; i < p.blocks[l]; i += kThreads) m = fmaxf(m, part[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((tid & 63) == 0) s_m[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
  }
  const float scale = pow2_lift(m);
  const int rem = (int)(idx - (long)e * per);
  if (p.writes_stats[img] && rem == 0) {
    p.stats[(size_t)e * NSTAT + 4 * l] = scale;
    p.stats[(size_t)e * NSTAT + 4 * l + 3] = m;
  }
  const int lane = rem & 63, sl = (rem >> 6) % p.slabs[img], tile = (rem >> 6) / p.slabs[img];
  const int r = lane & 31, h = lane >> 5;
  const float *sp = p.src[img] + (size_t)e * p.src_stride[img];
  const int kg = p.kg[img];
  f16x8 p1, p2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 16 * sl + 8 * h + j;
    float v = 0.0f;
    if (tile < p.src_tiles[img] && (k >> 3) < kg) v = sp[(((size_t)tile * kg + (k >> 3)) * 64 + ((k >> 2) & 1) * 32 + r) * 4 + (k & 3)];
    _Float16 q1, q2;
    split_h(v * scale, q1, q2);
    p1[j] = q1; p2[j] = q2;
  }
  f16x8 *d = p.dst[img] + (size_t)e * p.dst_stride[img] + ((size_t)tile * p.slabs[img] + sl) * 2 * 64 + lane;
  d[0] = p1; d[64] = p2;
}

struct AdamBArgs {
  float *B, *m, *v;
  const float *g; int n_parts; size_t part_stride;   // [n_parts][E][N]
  float *blob; int blob_ld;
  int E, N;
  float lr_t, b1, b2, eps;
  int apply;
};

__device__ __forceinline__ void adam_b(const AdamBArgs &p, unsigned block) {
  const int i = block * kThreads + threadIdx.x;
  if (i >= p.E * p.N) return;
  const int e = i / p.N, n = i - e * p.N;
  float w = p.B[i];
  if (p.apply) {
    float g = 0.0f;
#pragma unroll 8
    for (int s = 0; s < p.n_parts; ++s) g += p.g[(size_t)s * p.part_stride + i];
    float m = p.m[i], v = p.v[i];
    m += (g - m) * (1.0f - p.b1);
    v += (g * g - v) * (1.0f - p.b2);
    w -= p.lr_t * m / (sqrtf(v) + p.eps);
    p.m[i] = m; p.v[i] = v; p.B[i] = w;
  }
  p.blob[(size_t)e * p.blob_ld + n] = w;
}

// all six parameter tensors in one launch: blocks [first[k], first[k + 1]) belong to tensor k (W0 W1 W2 b0 b1 b2)
struct AdamAllArgs {
  AdamWArgs w[3];
  AdamBArgs b[3];
  unsigned first[7];
};

__global__ __launch_bounds__(kThreads) void adam_all_kernel(const AdamAllArgs p) {
  const unsigned blk = blockIdx.x;
  if (blk < p.first[1]) adam_w(p.w[0], blk);
  else if (blk < p.first[2]) adam_w(p.w[1], blk - p.first[1]);
  else if (blk < p.first[3]) adam_w(p.w[2], blk - p.first[2]);
  else if (blk < p.first[4]) adam_b(p.b[0], blk - p.first[3]);
  else if (blk < p.first[5]) adam_b(p.b[1], blk - p.first[4]);
  else adam_b(p.b[2], blk - p.first[5]);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
struct cmbpo_trainer {
  cmbpo_mlp *m;
  int E, I, IP, H, O, OPk, D, prob, max_batch;
  int nll;                       // probabilistic heads: 0 = 'MSPE' train loss, 1 = 'NLL' (cmbpo_trainer_set_loss)
  float lr, b1, b2, eps;
  float decay[3];
  long step;
  int ks[3];                     // partial gradients per tensor: the grid K split of the weight-gradient GEMMs
                                 // (fixed at max_batch), or the workgroups per member of the fused step
  int fused;                     // 1: fused_mse_step_kernel computes the whole gradient
  float *pool;                   // one allocation
  float *W[3], *Bv[3], *mW[3], *vW[3], *mB[3], *vB[3];
  float *wpb1, *wpb2;
  float *x, *h1, *g1, *h2, *g2, *o, *d3, *d2, *d1;
  float *parts[3], *dB[3];
  double *sums;
  double *loss_part;             // [E * ceil(max_batch / 32)][3] per-tile loss statistics of the training forward
  size_t wsize[3], bsize[3];
  float *opmax;                  // [E][ceil(max_batch / 32)][kOpMax] per-tile largest |x|, |h1|, |h2|, |d1|, |d2|, |d3| of the step
  // the backward chain on the f16 path (bwd_chain_h_kernel; 512-wide, at most 64 padded outputs): two-piece images of W1^T and
  // W2^T and the members' weight statistics, rebuilt from the packs at every step (the packs change with every Adam step)
  void *b16 = nullptr;           // one allocation: W1^T | W2^T | W0 | W1 | W2 images | stats | block maxima of |W0|, |W1|, |W2|
  size_t b16_w2t_off = 0, b16_f_off[3] = {0, 0, 0}, b16_stats_off = 0;   // 16-byte units
  int b16_s3 = 0, b16_s0 = 0;    // k-slabs of the backward chain's first product / of the input layer
  int b16_blocks[3] = {0, 0, 0}; // 256-element blocks per member of W0, W1, W2 (Adam leaves each block's largest |w|)
  bool b16_fwd = false;          // the training forward runs on the f16 path too (probabilistic head, <= 64 inputs / outputs)
  unsigned long b16_version = ~0ul;   // pack version the images were built from
};

namespace {

// 0: the training GEMMs as fp32 MFMAs (rounds 1-2); 1: three f16 MFMAs per product on split operands (512-wide ensembles)
int g_train_f16 = -1;
bool train_f16() {
  if (g_train_f16 < 0) {
    const char *e = getenv("CMBPO_TRAIN_F16");
    g_train_f16 = (e && e[0] == '0') ? 0 : 1;
  }
  return g_train_f16 != 0;
}

int wgrad_rows_per_wg(int batch, int ks) {
  int rows = cmbpo_ceil_div(batch, ks);
  return (rows + 15) / 16 * 16;   // whole 16-row blocks (the f16 form's K per MFMA; the fp32 form needs a multiple of 8)
}

void fill_wgrad_args(cmbpo_trainer *t, int layer, int batch, WgradArgs &a, int &n_cols);

int launch_wgrad(cmbpo_trainer *t, int layer, int batch, hipStream_t s) {
  WgradArgs a{};
  const int H = t->H;
  int n_cols;
  fill_wgrad_args(t, layer, batch, a, n_cols);
  a.n_tiles = cmbpo_ceil_div(n_cols, 64);
  const size_t lds = 2 * 128 * 64 * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    CMBPO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad_kernel, dim3((H / 128) * a.n_tiles, t->ks[layer], t->E), dim3(kThreads), lds, s, a);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

// the three GEMMs of a 512-wide ensemble in one launch (wgrad_all_kernel)
int launch_wgrad_all(cmbpo_trainer *t, int batch, hipStream_t s) {
  WgradAllArgs all{};
  const int order[3] = {1, 0, 2};
  int blocks = 0;
  for (int k = 0; k < 3; ++k) {
    const int layer = order[k];
    int n_cols;
    fill_wgrad_args(t, layer, batch, all.g[k], n_cols);
    all.g[k].n_tiles = cmbpo_ceil_div(n_cols, layer == 1 ? 128 : 64);
    all.tiles[k] = (layer == 1 ? t->H / 256 : t->H / 512) * all.g[k].n_tiles;
    all.ks[k] = t->ks[layer];
    all.first[k] = blocks;
    if (k == 0) {   // XCD-aware index space: chunks in groups of 8 (one per XCD), `tiles` slots each
      all.n_chunks0 = all.ks[0] * t->E;
      blocks += cmbpo_ceil_div(all.n_chunks0, 8) * 8 * all.tiles[0];
    } else {
      blocks += all.tiles[k] * all.ks[k] * t->E;
    }
  }
  all.first[3] = blocks;
  if (train_f16()) {
    static bool attr_set = false;
    if (!attr_set) {
      CMBPO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_all_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kWgradF16Lds));
      attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_all_kernel<true>, dim3(blocks), dim3(kThreads), kWgradF16Lds, s, all);
  } else {
    hipLaunchKernelGGL(wgrad_all_kernel<false>, dim3(blocks), dim3(kThreads), 2 * (8 * 512 / 4 + 8 * 64 / 4) * sizeof(f32x4), s, all);
  }
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

void fill_wgrad_args(cmbpo_trainer *t, int layer, int batch, WgradArgs &a, int &n_cols) {
  const int H = t->H;
  if (layer == 0) {         // dW0[k][n] = sum_b x[b][k] d1[b][n]: computed as (d1^T x), written transposed
    a.A = t->d1; a.lda = H; a.Bm = t->x; a.ldb = t->IP; n_cols = t->IP;
    a.ldc = H; a.transposed = 1; a.n_out = t->I;
    a.bias_mode = 2; a.bias_n = H;          // db0 = column sums of d1
    a.amax = t->opmax + 3; a.bmax = t->opmax + 0;
  } else if (layer == 1) {  // dW1 = h1^T d2
    a.A = t->h1; a.lda = H; a.Bm = t->d2; a.ldb = H; n_cols = H;
    a.ldc = H; a.transposed = 0; a.n_out = H;
    a.bias_mode = 1; a.bias_n = H;          // db1 = column sums of d2
    a.amax = t->opmax + 1; a.bmax = t->opmax + 4;
  } else {                  // dW2 = h2^T d3
    a.A = t->h2; a.lda = H; a.Bm = t->d3; a.ldb = t->OPk; n_cols = t->OPk;
    a.ldc = t->O; a.transposed = 0; a.n_out = t->O;
    a.bias_mode = 1; a.bias_n = t->O;       // db2 = column sums of d3
    a.amax = t->opmax + 2; a.bmax = t->opmax + 5;
  }
  a.bias_out = t->dB[layer];
  a.bias_member = t->bsize[layer] / t->E;
  a.bias_part = t->bsize[layer];
  a.out = t->parts[layer];
  a.out_member = t->wsize[layer] / t->E;
  a.out_part = t->wsize[layer];
  a.B = batch;
  a.rows_per_wg = wgrad_rows_per_wg(batch, t->ks[layer]);
  a.max_tiles = cmbpo_ceil_div(batch, 32);
}

template <int HID>
int launch_bwd(const BwdArgs &a, int tiles, int E, size_t lds, hipStream_t s) {
  static size_t attr_bytes = 0;
  if (lds > attr_bytes) {
    CMBPO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(bwd_chain_kernel<HID>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_bytes = lds;
  }
  hipLaunchKernelGGL(bwd_chain_kernel<HID>, dim3(tiles, E), dim3(kThreads), lds, s, a);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

// 0 / 1: the backward chain as fp32 MFMAs / on the f16 path where the shape allows (CMBPO_TRAIN_BWD_F16=0 turns it off)
bool bwd_f16() {
  static const bool on = !(getenv("CMBPO_TRAIN_BWD_F16") && getenv("CMBPO_TRAIN_BWD_F16")[0] == '0');
  return on && train_f16();
}

// the images of the f16 training kernels from the current packs, in one launch (train_pack_all_kernel): the members' lifts come
// from the block maxima the last Adam step (or weight load) left, not from a pass over the packs (h3_stats_kernel: 24 us)
int prepare_f16(cmbpo_trainer *t, hipStream_t s) {
  cmbpo_mlp *m = t->m;
  if (t->b16_version == m->pack_version) return CMBPO_OK;
  const int E = t->E, H = t->H;
  f16x8 *base = reinterpret_cast<f16x8 *>(t->b16);
  float *stats = reinterpret_cast<float *>(base + t->b16_stats_off);
  float *wm0 = stats + (size_t)E * NSTAT, *wm1 = wm0 + (size_t)E * t->b16_blocks[0], *wm2 = wm1 + (size_t)E * t->b16_blocks[1];
  const size_t w1t_stride = (size_t)(H / 32) * (H / 16) * 2 * 64, w2t_stride = (size_t)(H / 32) * t->b16_s3 * 2 * 64;
  PackAllArgs pa{};
  pa.wmax[0] = wm0; pa.wmax[1] = wm1; pa.wmax[2] = wm2;
  for (int l = 0; l < 3; ++l) pa.blocks[l] = t->b16_blocks[l];
  pa.stats = stats; pa.E = E;
  int n = 0;
  unsigned blocks = 0;
  auto add = [&](const float *src, size_t src_stride, int kg, int src_tiles, f16x8 *dst, size_t dst_stride, int n_tiles, int slabs,
                 int layer, int writes) {
    pa.src[n] = src; pa.src_stride[n] = src_stride; pa.kg[n] = kg; pa.src_tiles[n] = src_tiles;
    pa.dst[n] = dst; pa.dst_stride[n] = dst_stride; pa.n_tiles[n] = n_tiles; pa.slabs[n] = slabs; pa.layer[n] = layer;
    pa.writes_stats[n] = writes;
    pa.first[n] = blocks;
    blocks += (unsigned)((size_t)n_tiles * slabs * 64 * E / kThreads);
    ++n;
  };
  add(t->wpb1, (size_t)(H / 32) * (H / 8) * 256, H / 8, H / 32, base, w1t_stride, H / 32, H / 16, 1, 1);
  add(t->wpb2, (size_t)(H / 32) * (t->OPk / 8) * 256, t->OPk / 8, H / 32, base + t->b16_w2t_off, w2t_stride, H / 32, t->b16_s3, 2, 1);
  if (t->b16_fwd) {
    float *blob = m->d_blob;
    add(blob + m->off_wp0, (size_t)(H / 32) * (t->IP / 8) * 256, t->IP / 8, H / 32, base + t->b16_f_off[0],
        (size_t)(H / 32) * t->b16_s0 * 2 * 64, H / 32, t->b16_s0, 0, 1);
    add(blob + m->off_wp1, (size_t)(H / 32) * (H / 8) * 256, H / 8, H / 32, base + t->b16_f_off[1], w1t_stride, H / 32, H / 16, 1, 0);
    add(blob + m->off_wp2, (size_t)m->o_tiles * (H / 8) * 256, H / 8, m->o_tiles, base + t->b16_f_off[2], (size_t)2 * (H / 16) * 2 * 64, 2,
        H / 16, 2, 0);
  }
  pa.first[n] = blocks;
  pa.n_img = n;
  hipLaunchKernelGGL(train_pack_all_kernel, dim3(blocks), dim3(kThreads), 0, s, pa);
  CMBPO_HIP_CHECK(hipGetLastError());
  t->b16_version = m->pack_version;
  return CMBPO_OK;
}

int launch_bwd_h(cmbpo_trainer *t, const BwdArgs &b, int batch, hipStream_t s) {
  const int E = t->E, H = t->H;
  if (int rc = prepare_f16(t, s)) return rc;
  f16x8 *base = reinterpret_cast<f16x8 *>(t->b16);
  BwdHArgs a{};
  a.b = b;
  a.w1t = base; a.w2t = base + t->b16_w2t_off;
  a.w1t_stride = (size_t)(H / 32) * (H / 16) * 2 * 64; a.w2t_stride = (size_t)(H / 32) * t->b16_s3 * 2 * 64;
  a.stats = reinterpret_cast<const float *>(base + t->b16_stats_off);
  a.s3 = t->b16_s3;
  a.tiles32 = cmbpo_ceil_div(batch, 32);
  const size_t lds = bh_lds_bytes(t->b16_s3);
  static size_t attr_bytes = 0;
  if (lds > attr_bytes) {
    CMBPO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(bwd_chain_h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
    attr_bytes = lds;
  }
  hipLaunchKernelGGL(bwd_chain_h_kernel, dim3(cmbpo_ceil_div(batch, kBhRows), E), dim3(kBhThreads), lds, s, a);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

// the training forward on the f16 path (fwd_train_h_kernel); arguments as run_forward
int launch_fwd_h(cmbpo_trainer *t, const float *d_inputs, const int32_t *d_idx, int idx_stride, int rows, bool exports, hipStream_t s,
                 const float *d_targets) {
  cmbpo_mlp *m = t->m;
  const int H = t->H;
  if (int rc = prepare_f16(t, s)) return rc;
  f16x8 *base = reinterpret_cast<f16x8 *>(t->b16);
  FwdHArgs a{};
  a.inputs = d_inputs; a.idx = d_idx; a.idx_stride = idx_stride;
  a.n_rows = rows; a.I = t->I; a.IP = t->IP; a.s0 = t->b16_s0;
  a.in_mu = m->has_in_scaler ? m->d_blob + m->off_in_mu : nullptr;
  a.in_sig = m->has_in_scaler ? m->d_blob + m->off_in_var : nullptr;
  a.w0 = base + t->b16_f_off[0]; a.w1 = base + t->b16_f_off[1]; a.w2 = base + t->b16_f_off[2];
  a.w0_stride = (size_t)(H / 32) * t->b16_s0 * 2 * 64; a.w1_stride = (size_t)(H / 32) * (H / 16) * 2 * 64;
  a.w2_stride = (size_t)2 * (H / 16) * 2 * 64;
  a.b0 = m->d_blob + m->off_b0; a.b1 = m->d_blob + m->off_b1; a.b2 = m->d_blob + m->off_b2; a.b2_ld = m->o_tiles * 32;
  a.stats = reinterpret_cast<const float *>(base + t->b16_stats_off);
  a.O = t->O; a.D = t->D;
  if (exports) { a.x = t->x; a.h1 = t->h1; a.g1 = t->g1; a.h2 = t->h2; a.g2 = t->g2; a.opmax = t->opmax; }
  a.o = t->o;
  if (d_targets) {
    a.targets = d_targets; a.loss_part = t->loss_part;
    a.out_mu = m->has_out_scaler ? m->d_blob + m->off_out_mu : nullptr;
    a.out_sig = m->has_out_scaler ? m->d_blob + m->off_out_var : nullptr;
  }
  a.tiles32 = cmbpo_ceil_div(rows, 32);
  static bool attr = false;
  if (!attr) {
    CMBPO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(fwd_train_h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)fh_lds_bytes()));
    attr = true;
  }
  hipLaunchKernelGGL(fwd_train_h_kernel, dim3(cmbpo_ceil_div(rows, kBhRows), t->E), dim3(kBhThreads), fh_lds_bytes(), s, a);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

// masters -> packs (apply == 0) or one Adam step (apply == 1)
int launch_update(cmbpo_trainer *t, int apply, float lr_t, hipStream_t s) {
  cmbpo_mlp *m = t->m;
  ++m->pack_version;   // the packed images change: derived images (ens_split.hip) are stale
  const int E = t->E, H = t->H;
  const int Ks[3] = {t->I, H, H}, Ns[3] = {H, H, t->O};
  float *fwd[3] = {m->d_blob + m->off_wp0, m->d_blob + m->off_wp1, m->d_blob + m->off_wp2};
  const int f_kg[3] = {t->IP / 8, H / 8, H / 8};
  const size_t f_stride[3] = {(size_t)(H / 32) * (t->IP / 8) * 256, (size_t)(H / 32) * (H / 8) * 256,
                              (size_t)m->o_tiles * (H / 8) * 256};
  float *bwd[3] = {nullptr, t->wpb1, t->wpb2};
  const int b_kg[3] = {0, H / 8, t->OPk / 8};
  const size_t b_stride[3] = {0, (size_t)(H / 32) * (H / 8) * 256, (size_t)(H / 32) * (t->OPk / 8) * 256};
  float *blob_b[3] = {m->d_blob + m->off_b0, m->d_blob + m->off_b1, m->d_blob + m->off_b2};
  const int blob_ld[3] = {H, H, m->o_tiles * 32};
  AdamAllArgs all{};
  unsigned blocks = 0;
  for (int l = 0; l < 3; ++l) {
    AdamWArgs &a = all.w[l];
    a.W = t->W[l]; a.m = t->mW[l]; a.v = t->vW[l];
    a.parts = t->parts[l]; a.n_parts = t->ks[l]; a.part_stride = t->wsize[l];
    a.decay = t->decay[l];
    a.E = E; a.K = Ks[l]; a.N = Ns[l];
    a.fwd = fwd[l]; a.f_kg = f_kg[l]; a.f_stride = f_stride[l];
    a.bwd = bwd[l]; a.b_kg = b_kg[l]; a.b_stride = b_stride[l];
    a.lr_t = lr_t; a.b1 = t->b1; a.b2 = t->b2; a.eps = t->eps; a.apply = apply;
    a.wmax_part = nullptr;
    if (t->b16) {
      float *wm = reinterpret_cast<float *>(reinterpret_cast<f16x8 *>(t->b16) + t->b16_stats_off) + (size_t)E * NSTAT;
      a.wmax_part = wm + (size_t)E * ((l >= 1 ? t->b16_blocks[0] : 0) + (l >= 2 ? t->b16_blocks[1] : 0));
    }
    all.first[l] = blocks;
    blocks += (unsigned)((t->wsize[l] + kThreads - 1) / kThreads);
  }
  for (int l = 0; l < 3; ++l) {
    AdamBArgs &b = all.b[l];
    b.B = t->Bv[l]; b.m = t->mB[l]; b.v = t->vB[l];
    b.g = t->dB[l]; b.n_parts = t->ks[l]; b.part_stride = t->bsize[l];
    b.blob = blob_b[l]; b.blob_ld = blob_ld[l];
    b.E = E; b.N = Ns[l];
    b.lr_t = lr_t; b.b1 = t->b1; b.b2 = t->b2; b.eps = t->eps; b.apply = apply;
    all.first[3 + l] = blocks;
    blocks += (unsigned)cmbpo_ceil_div(E * Ns[l], kThreads);
  }
  all.first[6] = blocks;
  hipLaunchKernelGGL(adam_all_kernel, dim3(blocks), dim3(kThreads), 0, s, all);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

int launch_fused(cmbpo_trainer *t, const float *d_inputs, const float *d_targets, const int32_t *d_idx, int idx_stride,
                 int batch, hipStream_t s) {
  cmbpo_mlp *m = t->m;
  const int H = t->H;
  FusedArgs a{};
  a.F0 = reinterpret_cast<const f32x4 *>(m->d_blob + m->off_wp0);
  a.F1 = reinterpret_cast<const f32x4 *>(m->d_blob + m->off_wp1);
  a.F2 = reinterpret_cast<const f32x4 *>(m->d_blob + m->off_wp2);
  a.B1 = reinterpret_cast<const f32x4 *>(t->wpb1);
  a.B2 = reinterpret_cast<const f32x4 *>(t->wpb2);
  a.sF0 = (size_t)(H / 32) * (t->IP / 8) * 64; a.sF1 = (size_t)(H / 32) * (H / 8) * 64;
  a.sF2 = (size_t)m->o_tiles * (H / 8) * 64;
  a.sB1 = (size_t)(H / 32) * (H / 8) * 64; a.sB2 = (size_t)(H / 32) * (t->OPk / 8) * 64;
  a.b0 = m->d_blob + m->off_b0; a.b1 = m->d_blob + m->off_b1; a.b2 = m->d_blob + m->off_b2;
  a.b2_ld = m->o_tiles * 32;
  a.in_mu = m->has_in_scaler ? m->d_blob + m->off_in_mu : nullptr;
  a.in_sig = m->has_in_scaler ? m->d_blob + m->off_in_var : nullptr;
  a.out_mu = m->has_out_scaler ? m->d_blob + m->off_out_mu : nullptr;
  a.out_sig = m->has_out_scaler ? m->d_blob + m->off_out_var : nullptr;
  a.inputs = d_inputs; a.targets = d_targets; a.idx = d_idx; a.idx_stride = idx_stride;
  a.E = t->E; a.I = t->I; a.IP = t->IP; a.kg0 = t->IP / 8; a.O = t->O; a.kga = t->OPk / 8; a.B = batch;
  for (int l = 0; l < 3; ++l) {
    a.pW[l] = t->parts[l]; a.sW[l] = t->wsize[l];
    a.pB[l] = t->dB[l]; a.sB[l] = t->bsize[l];
  }
  const size_t lds = ((size_t)32 * (t->IP + 4) + 4 * 32 * 132 + 32 * 36 + 2 * 128) * sizeof(float) + 64 * 1024;
  const int G = t->ks[0];
  // every partial slot is written: workgroups beyond the batch's tiles store zeros
  if (t->IP <= 32) {
    static bool set1 = false;
    if (!set1) {
      CMBPO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(fused_mse_step_kernel<1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
      set1 = true;
    }
    hipLaunchKernelGGL(fused_mse_step_kernel<1>, dim3(G, t->E), dim3(kThreads), lds, s, a);
  } else {
    static bool set2 = false;
    if (!set2) {
      CMBPO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(fused_mse_step_kernel<2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
      set2 = true;
    }
    hipLaunchKernelGGL(fused_mse_step_kernel<2>, dim3(G, t->E), dim3(kThreads), lds, s, a);
  }
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

bool fwd_f16() {
  static const bool on = !(getenv("CMBPO_TRAIN_FWD_F16") && getenv("CMBPO_TRAIN_FWD_F16")[0] == '0');
  return on && train_f16();
}

int run_forward(cmbpo_trainer *t, const float *d_inputs, const int32_t *d_idx, int idx_stride, int rows, bool exports,
                hipStream_t s, const float *d_targets = nullptr) {
  if (t->b16 && t->b16_fwd && fwd_f16()) return launch_fwd_h(t, d_inputs, d_idx, idx_stride, rows, exports, s, d_targets);
  MlpKernelArgs a{};
  a.obs = d_inputs; a.obs_dim = t->I; a.act = nullptr; a.act_dim = 0;
  a.row_idx = d_idx; a.row_idx_stride = idx_stride;
  a.n_rows_dev = nullptr; a.n_rows = rows; a.ld_rows = rows;
  a.out0 = t->o;
  if (exports) { a.tr_x = t->x; a.tr_h1 = t->h1; a.tr_g1 = t->g1; a.tr_h2 = t->h2; a.tr_g2 = t->g2; }
  if (d_targets) { a.tr_targets = d_targets; a.tr_tdim = t->D; a.tr_loss_part = t->loss_part; }
  if (exports) a.tr_opmax = t->opmax;          // (indexed by item = member * tiles + tile, like the loss statistics)
  return cmbpo_internal_launch_mlp(t->m, a, s, CMBPO_HEAD_TRAIN);
}

void fill_loss_args(cmbpo_trainer *t, LossArgs &l, const float *d_targets, const int32_t *d_idx, int idx_stride, int rows) {
  cmbpo_mlp *m = t->m;
  l.o = t->o; l.targets = d_targets; l.idx = d_idx; l.idx_stride = idx_stride;
  l.out_mu = m->has_out_scaler ? m->d_blob + m->off_out_mu : nullptr;
  l.out_sig = m->has_out_scaler ? m->d_blob + m->off_out_var : nullptr;
  l.E = t->E; l.B = rows; l.O = t->O; l.D = t->D; l.OPk = t->OPk; l.prob = t->prob;
  l.sums = t->sums; l.d3 = t->d3;
}

}  // namespace

extern "C" int cmbpo_trainer_create(cmbpo_trainer_t **out, cmbpo_mlp_t *m, int max_batch, float lr,
                                    const double *decays) {
  CMBPO_REQUIRE(out && m && decays, "cmbpo_trainer_create: NULL argument");
  CMBPO_REQUIRE(m->head == CMBPO_HEAD_PROB || m->head == CMBPO_HEAD_DETMEAN,
                "cmbpo_trainer_create: only ensemble handles (HEAD_PROB / HEAD_DETMEAN) train");
  CMBPO_REQUIRE(m->act == CMBPO_ACT_SWISH, "cmbpo_trainer_create: only swish ensembles train");
  CMBPO_REQUIRE(max_batch >= 1 && max_batch <= (1 << 20), "cmbpo_trainer_create: max_batch %d out of range", max_batch);
  cmbpo_trainer *t = new (std::nothrow) cmbpo_trainer();
  if (!t) { cmbpo_set_error("cmbpo_trainer_create: out of host memory"); return CMBPO_ENOMEM; }
  t->m = m;
  t->E = m->ensemble; t->I = m->in_dim; t->IP = m->in_pad; t->H = m->hidden; t->O = m->o_width;
  t->OPk = (m->o_width + 7) / 8 * 8; t->D = m->out_dim; t->prob = m->head == CMBPO_HEAD_PROB;
  t->max_batch = max_batch;
  t->nll = 0;
  t->lr = lr; t->b1 = 0.9f; t->b2 = 0.999f; t->eps = 1e-8f;
  for (int l = 0; l < 3; ++l) t->decay[l] = (float)decays[l];
  t->step = 0;
  const int E = t->E, H = t->H;
  t->wsize[0] = (size_t)E * t->I * H; t->wsize[1] = (size_t)E * H * H; t->wsize[2] = (size_t)E * H * t->O;
  t->bsize[0] = (size_t)E * H; t->bsize[1] = (size_t)E * H; t->bsize[2] = (size_t)E * t->O;
  // grid K split: enough workgroups to cover the chip about twice, at least 8 batch rows per workgroup
  const int n_tiles[3] = {cmbpo_ceil_div(t->IP, 64), H / 64, cmbpo_ceil_div(t->OPk, 64)};
  t->fused = (H == 128 && !t->prob && t->O <= 32 && t->IP <= 64) ? 1 : 0;
  for (int l = 0; l < 3; ++l) {
    if (t->fused) {
      // workgroups per member.  A CU delivers ~0.6 TFLOP/s of fp32 MFMA, i.e. ~6 us for the ~4 MFLOP of one 32-row
      // tile of a 128-wide net: the step is fastest spread one tile per CU (64 x E workgroups at batch 2048), at the
      // price of Adam adding up to 64 partial gradients (16 MB of reads, a few us)
      int g = cmbpo_ceil_div(max_batch, 32);
      t->ks[l] = g > 64 ? 64 : (g < 1 ? 1 : g);
      continue;
    }
    int wgs = (H / 128) * n_tiles[l] * E;
    int ks = cmbpo_ceil_div(512, wgs);
    if (H == 512) {
      // 512-wide: the three GEMMs are ONE launch (wgrad_all_kernel) -- together they should fill one round of two
      // workgroups per CU, not one round each.  Measured on the f16 kernels (tools/sweep_wgrad_ks.sh, E = 7: 6 for the
      // square layer and 8 for the narrow ones against round 2's 9 / 16: 284 -> 243 us per step at batch 2048, 193 -> 180 at
      // 512, 953 -> 919 at 8192): fewer, longer workgroups hide the operand loads better and leave Adam fewer partials.
      const int narrow = (cmbpo_ceil_div(t->IP, 64) + cmbpo_ceil_div(t->OPk, 64)) * E * 8;
      const int square = (H / 256) * (H / 128) * E;
      ks = (l == 1) ? (int)(0.85f * (float)(512 - narrow) / (float)square) : 8;
    }
    if (const char *env = getenv("CMBPO_WGRAD_KS")) {   // tuning aid: grid K split of the square layer
      if (l == 1 && atoi(env) > 0) ks = atoi(env);
    }
    if (const char *env = getenv("CMBPO_WGRAD_KS_NARROW")) {   // ... and of the two narrow layers
      if (l != 1 && atoi(env) > 0) ks = atoi(env);
    }
    const int max_ks = max_batch / 32 < 1 ? 1 : max_batch / 32;
    if (ks > max_ks) ks = max_ks;
    if (ks > 16) ks = 16;
    t->ks[l] = ks < 1 ? 1 : ks;
  }
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; };
  size_t oW[3], oB[3], omW[3], ovW[3], omB[3], ovB[3], oP[3], odB[3];
  for (int l = 0; l < 3; ++l) {
    oW[l] = take(t->wsize[l]); omW[l] = take(t->wsize[l]); ovW[l] = take(t->wsize[l]);
    oB[l] = take(t->bsize[l]); omB[l] = take(t->bsize[l]); ovB[l] = take(t->bsize[l]);
    oP[l] = take(t->wsize[l] * t->ks[l]); odB[l] = take(t->bsize[l] * t->ks[l]);
  }
  const size_t owpb1 = take((size_t)E * (H / 32) * (H / 8) * 256);
  const size_t owpb2 = take((size_t)E * (H / 32) * (t->OPk / 8) * 256);
  const size_t rows = (size_t)E * max_batch;
  const size_t ox = take(rows * t->IP), oh1 = take(rows * H), og1 = take(rows * H), oh2 = take(rows * H),
               og2 = take(rows * H), oo = take(rows * t->O), od3 = take(rows * t->OPk), od2 = take(rows * H),
               od1 = take(rows * H);
  const size_t osums = take(2 * 3 * (size_t)E);   // doubles
  const size_t olp = take(2 * 3 * (size_t)E * cmbpo_ceil_div(max_batch, 32));   // doubles
  const size_t oom = take((size_t)E * cmbpo_ceil_div(max_batch, 32) * kOpMax);
  hipError_t err = hipMalloc(reinterpret_cast<void **>(&t->pool), off * sizeof(float));
  if (err != hipSuccess) {
    cmbpo_set_error("cmbpo_trainer_create: hipMalloc(%zu) failed: %s", off * sizeof(float), hipGetErrorString(err));
    delete t;
    return CMBPO_ENOMEM;
  }
  err = hipMemset(t->pool, 0, off * sizeof(float));   // Adam moments, pack padding
  if (err != hipSuccess) {
    cmbpo_set_error("cmbpo_trainer_create: hipMemset failed: %s", hipGetErrorString(err));
    (void)hipFree(t->pool);
    delete t;
    return CMBPO_EHIP;
  }
  float *P = t->pool;
  for (int l = 0; l < 3; ++l) {
    t->W[l] = P + oW[l]; t->mW[l] = P + omW[l]; t->vW[l] = P + ovW[l];
    t->Bv[l] = P + oB[l]; t->mB[l] = P + omB[l]; t->vB[l] = P + ovB[l];
    t->parts[l] = P + oP[l]; t->dB[l] = P + odB[l];
  }
  t->wpb1 = P + owpb1; t->wpb2 = P + owpb2;
  t->x = P + ox; t->h1 = P + oh1; t->g1 = P + og1; t->h2 = P + oh2; t->g2 = P + og2;
  t->o = P + oo; t->d3 = P + od3; t->d2 = P + od2; t->d1 = P + od1;
  t->sums = reinterpret_cast<double *>(P + osums);
  t->loss_part = reinterpret_cast<double *>(P + olp);
  t->opmax = P + oom;
  if (H == 512 && !t->fused && t->OPk <= 64) {
    // images for the f16 training kernels (bwd_chain_h_kernel, fwd_train_h_kernel); without them the fp32 kernels run
    t->b16_s3 = cmbpo_ceil_div(t->OPk, 16);
    t->b16_s0 = cmbpo_ceil_div(t->IP, 16);
    t->b16_fwd = t->prob && t->IP <= 64 && t->O <= 64 && m->o_tiles <= 2;
    const size_t w1t = (size_t)E * (H / 32) * (H / 16) * 2 * 64, w2t = (size_t)E * (H / 32) * t->b16_s3 * 2 * 64;
    const size_t f0 = (size_t)E * (H / 32) * t->b16_s0 * 2 * 64, f2 = (size_t)E * 2 * (H / 16) * 2 * 64;
    t->b16_w2t_off = w1t;
    t->b16_f_off[0] = w1t + w2t; t->b16_f_off[1] = t->b16_f_off[0] + f0; t->b16_f_off[2] = t->b16_f_off[1] + w1t;
    t->b16_stats_off = t->b16_f_off[2] + f2;
    t->b16_blocks[0] = (int)((size_t)t->I * H / kThreads);
    t->b16_blocks[1] = (int)((size_t)H * H / kThreads);
    t->b16_blocks[2] = (int)((size_t)H * t->O / kThreads);
    const size_t tail_floats = (size_t)E * NSTAT + (size_t)E * (t->b16_blocks[0] + t->b16_blocks[1] + t->b16_blocks[2]);
    if (hipMalloc(&t->b16, t->b16_stats_off * 16 + tail_floats * sizeof(float)) != hipSuccess) {
      (void)hipGetLastError();
      t->b16 = nullptr;
    } else {
      (void)hipMemset(reinterpret_cast<char *>(t->b16) + t->b16_stats_off * 16, 0, tail_floats * sizeof(float));
    }
  }
  *out = t;
  return CMBPO_OK;
}

extern "C" void cmbpo_trainer_destroy(cmbpo_trainer_t *t) {
  if (!t) return;
  if (t->pool) (void)hipFree(t->pool);
  if (t->b16) (void)hipFree(t->b16);
  delete t;
}

extern "C" int cmbpo_trainer_set_weights(cmbpo_trainer_t *t, const float *h_w0, const float *h_b0, const float *h_w1,
                                         const float *h_b1, const float *h_w2, const float *h_b2, void *stream) {
  CMBPO_REQUIRE(t && h_w0 && h_b0 && h_w1 && h_b1 && h_w2 && h_b2, "cmbpo_trainer_set_weights: NULL argument");
  hipStream_t s = (hipStream_t)stream;
  const float *hw[3] = {h_w0, h_w1, h_w2}, *hb[3] = {h_b0, h_b1, h_b2};
  for (int l = 0; l < 3; ++l) {
    CMBPO_HIP_CHECK(hipMemcpyAsync(t->W[l], hw[l], t->wsize[l] * sizeof(float), hipMemcpyHostToDevice, s));
    CMBPO_HIP_CHECK(hipMemcpyAsync(t->Bv[l], hb[l], t->bsize[l] * sizeof(float), hipMemcpyHostToDevice, s));
  }
  // the packed images (and their zero padding) are rebuilt from the masters
  cmbpo_mlp *m = t->m;
  CMBPO_HIP_CHECK(hipMemsetAsync(m->d_blob + m->off_wp0, 0, (m->off_in_mu - m->off_wp0) * sizeof(float), s));
  const int rc = launch_update(t, 0, 0.0f, s);
  if (rc != CMBPO_OK) return rc;
  CMBPO_HIP_CHECK(hipStreamSynchronize(s));   // the host arrays may be pageable temporaries
  m->loaded = true;
  return CMBPO_OK;
}

extern "C" int cmbpo_trainer_get_weights(cmbpo_trainer_t *t, float *h_w0, float *h_b0, float *h_w1, float *h_b1,
                                         float *h_w2, float *h_b2, void *stream) {
  CMBPO_REQUIRE(t && h_w0 && h_b0 && h_w1 && h_b1 && h_w2 && h_b2, "cmbpo_trainer_get_weights: NULL argument");
  hipStream_t s = (hipStream_t)stream;
  float *hw[3] = {h_w0, h_w1, h_w2}, *hb[3] = {h_b0, h_b1, h_b2};
  for (int l = 0; l < 3; ++l) {
    CMBPO_HIP_CHECK(hipMemcpyAsync(hw[l], t->W[l], t->wsize[l] * sizeof(float), hipMemcpyDeviceToHost, s));
    CMBPO_HIP_CHECK(hipMemcpyAsync(hb[l], t->Bv[l], t->bsize[l] * sizeof(float), hipMemcpyDeviceToHost, s));
  }
  CMBPO_HIP_CHECK(hipStreamSynchronize(s));
  return CMBPO_OK;
}

// Adam state in and out (which: 0 first moment, 1 second moment): optimizer checkpoint / resume
extern "C" int cmbpo_trainer_get_moments(cmbpo_trainer_t *t, int which, float *h_w0, float *h_b0, float *h_w1,
                                         float *h_b1, float *h_w2, float *h_b2, void *stream) {
  CMBPO_REQUIRE(t && h_w0 && h_b0 && h_w1 && h_b1 && h_w2 && h_b2, "cmbpo_trainer_get_moments: NULL argument");
  CMBPO_REQUIRE(which == 0 || which == 1, "cmbpo_trainer_get_moments: which must be 0 (m) or 1 (v)");
  hipStream_t s = (hipStream_t)stream;
  float *hw[3] = {h_w0, h_w1, h_w2}, *hb[3] = {h_b0, h_b1, h_b2};
  for (int l = 0; l < 3; ++l) {
    CMBPO_HIP_CHECK(hipMemcpyAsync(hw[l], which ? t->vW[l] : t->mW[l], t->wsize[l] * sizeof(float), hipMemcpyDeviceToHost, s));
    CMBPO_HIP_CHECK(hipMemcpyAsync(hb[l], which ? t->vB[l] : t->mB[l], t->bsize[l] * sizeof(float), hipMemcpyDeviceToHost, s));
  }
  CMBPO_HIP_CHECK(hipStreamSynchronize(s));
  return CMBPO_OK;
}

extern "C" int cmbpo_trainer_set_moments(cmbpo_trainer_t *t, int which, const float *h_w0, const float *h_b0,
                                         const float *h_w1, const float *h_b1, const float *h_w2, const float *h_b2,
                                         long steps_done, void *stream) {
  CMBPO_REQUIRE(t && h_w0 && h_b0 && h_w1 && h_b1 && h_w2 && h_b2, "cmbpo_trainer_set_moments: NULL argument");
  CMBPO_REQUIRE(which == 0 || which == 1, "cmbpo_trainer_set_moments: which must be 0 (m) or 1 (v)");
  CMBPO_REQUIRE(steps_done >= 0, "cmbpo_trainer_set_moments: negative step count");
  hipStream_t s = (hipStream_t)stream;
  const float *hw[3] = {h_w0, h_w1, h_w2}, *hb[3] = {h_b0, h_b1, h_b2};
  for (int l = 0; l < 3; ++l) {
    CMBPO_HIP_CHECK(hipMemcpyAsync(which ? t->vW[l] : t->mW[l], hw[l], t->wsize[l] * sizeof(float), hipMemcpyHostToDevice, s));
    CMBPO_HIP_CHECK(hipMemcpyAsync(which ? t->vB[l] : t->mB[l], hb[l], t->bsize[l] * sizeof(float), hipMemcpyHostToDevice, s));
  }
  CMBPO_HIP_CHECK(hipStreamSynchronize(s));
  t->step = steps_done;
  return CMBPO_OK;
}

extern "C" int cmbpo_trainer_set_loss(cmbpo_trainer_t *t, int loss) {
  CMBPO_REQUIRE(t != nullptr, "cmbpo_trainer_set_loss: handle is NULL");
  CMBPO_REQUIRE(loss == CMBPO_LOSS_DEFAULT || loss == CMBPO_LOSS_NLL, "cmbpo_trainer_set_loss: unknown loss %d", loss);
  CMBPO_REQUIRE(loss == CMBPO_LOSS_DEFAULT || t->prob, "cmbpo_trainer_set_loss: 'NLL' needs a probabilistic (HEAD_PROB) ensemble");
  t->nll = loss == CMBPO_LOSS_NLL;
  return CMBPO_OK;
}

extern "C" int cmbpo_trainer_reset_optimizer(cmbpo_trainer_t *t, void *stream) {
  CMBPO_REQUIRE(t != nullptr, "cmbpo_trainer_reset_optimizer: handle is NULL");
  hipStream_t s = (hipStream_t)stream;
  for (int l = 0; l < 3; ++l) {
    CMBPO_HIP_CHECK(hipMemsetAsync(t->mW[l], 0, t->wsize[l] * sizeof(float), s));
    CMBPO_HIP_CHECK(hipMemsetAsync(t->vW[l], 0, t->wsize[l] * sizeof(float), s));
    CMBPO_HIP_CHECK(hipMemsetAsync(t->mB[l], 0, t->bsize[l] * sizeof(float), s));
    CMBPO_HIP_CHECK(hipMemsetAsync(t->vB[l], 0, t->bsize[l] * sizeof(float), s));
  }
  t->step = 0;
  return CMBPO_OK;
}

extern "C" int cmbpo_mlp_set_scalers(cmbpo_mlp_t *m, const float *h_in_mu, const float *h_in_var,
                                     const float *h_out_mu, const float *h_out_var, void *stream) {
  CMBPO_REQUIRE(m != nullptr, "cmbpo_mlp_set_scalers: handle is NULL");
  CMBPO_REQUIRE((h_in_mu == nullptr) == (h_in_var == nullptr), "cmbpo_mlp_set_scalers: in scaler needs both mu and var");
  CMBPO_REQUIRE((h_out_mu == nullptr) == (h_out_var == nullptr), "cmbpo_mlp_set_scalers: out scaler needs both mu and var");
  hipStream_t s = (hipStream_t)stream;
  const int I = m->in_dim, D = m->out_dim;
  std::vector<float> tmp(2 * (size_t)I + 3 * (size_t)D);
  // sigma = max(sqrt(var), 1e-2) in float32, as cmbpo_mlp_load (models/pens/utils.py:156,167,187)
  if (h_in_mu) {
    for (int k = 0; k < I; ++k) { tmp[k] = h_in_mu[k]; tmp[I + k] = fmaxf(sqrtf(h_in_var[k]), 1e-2f); }
    CMBPO_HIP_CHECK(hipMemcpyAsync(m->d_blob + m->off_in_mu, tmp.data(), I * sizeof(float), hipMemcpyHostToDevice, s));
    CMBPO_HIP_CHECK(hipMemcpyAsync(m->d_blob + m->off_in_var, tmp.data() + I, I * sizeof(float), hipMemcpyHostToDevice, s));
  }
  if (h_out_mu) {
    float *o = tmp.data() + 2 * (size_t)I;
    for (int k = 0; k < D; ++k) {
      const float sig = fmaxf(sqrtf(h_out_var[k]), 1e-2f);
      o[k] = h_out_mu[k]; o[D + k] = sig; o[2 * D + k] = 2.0f * logf(sig);
    }
    CMBPO_HIP_CHECK(hipMemcpyAsync(m->d_blob + m->off_out_mu, o, D * sizeof(float), hipMemcpyHostToDevice, s));
    CMBPO_HIP_CHECK(hipMemcpyAsync(m->d_blob + m->off_out_var, o + D, D * sizeof(float), hipMemcpyHostToDevice, s));
    CMBPO_HIP_CHECK(hipMemcpyAsync(m->d_blob + m->off_out_lsig2, o + 2 * D, D * sizeof(float), hipMemcpyHostToDevice, s));
  }
  CMBPO_HIP_CHECK(hipStreamSynchronize(s));   // tmp goes out of scope
  m->has_in_scaler = m->has_in_scaler || h_in_mu != nullptr;
  m->has_out_scaler = m->has_out_scaler || h_out_mu != nullptr;
  return CMBPO_OK;
}

extern "C" int cmbpo_trainer_step(cmbpo_trainer_t *t, const float *d_inputs, int in_dim, const float *d_targets,
                                  int target_dim, const int32_t *d_idx, int idx_stride, int batch, void *stream) {
  CMBPO_REQUIRE(t && d_inputs && d_targets, "cmbpo_trainer_step: NULL argument");
  if (!t->m->loaded) { cmbpo_set_error("cmbpo_trainer_step: weights not loaded"); return CMBPO_ESTATE; }
  CMBPO_REQUIRE(in_dim == t->I, "cmbpo_trainer_step: in_dim %d != %d", in_dim, t->I);
  CMBPO_REQUIRE(target_dim == t->D, "cmbpo_trainer_step: target_dim %d != %d", target_dim, t->D);
  CMBPO_REQUIRE(batch >= 1 && batch <= t->max_batch, "cmbpo_trainer_step: batch %d not in [1, %d]", batch, t->max_batch);
  CMBPO_REQUIRE(d_idx == nullptr || idx_stride >= 0, "cmbpo_trainer_step: negative idx_stride");
  hipStream_t s = (hipStream_t)stream;
  const int E = t->E, H = t->H;

  if (t->fused) {
    int rc = launch_fused(t, d_inputs, d_targets, d_idx, idx_stride, batch, s);
    if (rc != CMBPO_OK) return rc;
    t->step += 1;
    const double lr_f = (double)t->lr * sqrt(1.0 - pow((double)t->b2, (double)t->step)) / (1.0 - pow((double)t->b1, (double)t->step));
    return launch_update(t, 1, (float)lr_f, s);
  }

  // training forward (+ per-tile loss statistics) -> backward chain (+ output deltas) -> weight gradients -> Adam
  int rc = run_forward(t, d_inputs, d_idx, idx_stride, batch, true, s, d_targets);
  if (rc != CMBPO_OK) return rc;

  BwdArgs b{};
  b.wpb2 = reinterpret_cast<const f32x4 *>(t->wpb2); b.wpb1 = reinterpret_cast<const f32x4 *>(t->wpb1);
  b.wpb2_stride = (size_t)(H / 32) * (t->OPk / 8) * 64; b.wpb1_stride = (size_t)(H / 32) * (H / 8) * 64;
  b.d3 = t->d3; b.g2 = t->g2; b.g1 = t->g1; b.d2 = t->d2; b.d1 = t->d1;
  b.B = batch; b.OPk = t->OPk;
  {
    cmbpo_mlp *m = t->m;
    b.fuse = 1; b.prob = t->prob ? (t->nll ? 2 : 1) : 0; b.O = t->O; b.D = t->D; b.n_items = E * cmbpo_ceil_div(batch, 32);
    b.o = t->o; b.targets = d_targets; b.idx = d_idx; b.idx_stride = idx_stride;
    b.out_mu = m->has_out_scaler ? m->d_blob + m->off_out_mu : nullptr;
    b.out_sig = m->has_out_scaler ? m->d_blob + m->off_out_var : nullptr;
    b.loss_part = t->loss_part; b.d3_out = t->d3;
    b.opmax = t->opmax;
  }
  const size_t lds = ((size_t)H / 4 * 32 + (size_t)t->OPk / 4 * 32) * sizeof(f32x4);
  const int tiles = cmbpo_ceil_div(batch, 32);
  if (t->b16 && bwd_f16()) rc = launch_bwd_h(t, b, batch, s);
  else rc = (H == 512) ? launch_bwd<512>(b, tiles, E, lds, s) : (H == 256 ? launch_bwd<256>(b, tiles, E, lds, s) : launch_bwd<128>(b, tiles, E, lds, s));
  if (rc != CMBPO_OK) return rc;

  if (H == 512) {
    rc = launch_wgrad_all(t, batch, s);
    if (rc != CMBPO_OK) return rc;
  } else {
    for (int layer = 0; layer < 3; ++layer) {
      rc = launch_wgrad(t, layer, batch, s);
      if (rc != CMBPO_OK) return rc;
    }
  }
  CMBPO_HIP_CHECK(hipGetLastError());

  t->step += 1;
  const double lr_t = (double)t->lr * sqrt(1.0 - pow((double)t->b2, (double)t->step)) / (1.0 - pow((double)t->b1, (double)t->step));
  return launch_update(t, 1, (float)lr_t, s);
}

// One pass over the bootstrap index lists: ceil(n_rows / batch) train_ops enqueued back to back (the inner loop of
// PE.train, models/pens/pe.py:541-563), the last one on the ragged remainder.
extern "C" int cmbpo_trainer_epoch(cmbpo_trainer_t *t, const float *d_inputs, int in_dim, const float *d_targets,
                                   int target_dim, const int32_t *d_idx, int idx_stride, int n_rows, int batch,
                                   void *stream) {
  CMBPO_REQUIRE(t && d_inputs && d_targets && d_idx, "cmbpo_trainer_epoch: NULL argument");
  CMBPO_REQUIRE(n_rows >= 1 && batch >= 1 && batch <= t->max_batch, "cmbpo_trainer_epoch: n_rows %d / batch %d (max %d)",
                n_rows, batch, t->max_batch);
  CMBPO_REQUIRE(idx_stride >= n_rows, "cmbpo_trainer_epoch: idx_stride %d < n_rows %d", idx_stride, n_rows);
  for (int r0 = 0; r0 < n_rows; r0 += batch) {
    const int b = min(batch, n_rows - r0);
    const int rc = cmbpo_trainer_step(t, d_inputs, in_dim, d_targets, target_dim, d_idx + r0, idx_stride, b, stream);
    if (rc != CMBPO_OK) return rc;
  }
  return CMBPO_OK;
}

extern "C" int cmbpo_trainer_losses(cmbpo_trainer_t *t, const float *d_inputs, int in_dim, const float *d_targets,
                                    int target_dim, const int32_t *d_idx, int idx_stride, int n_rows, float *d_losses,
                                    void *stream) {
  CMBPO_REQUIRE(t && d_inputs && d_targets && d_losses, "cmbpo_trainer_losses: NULL argument");
  if (!t->m->loaded) { cmbpo_set_error("cmbpo_trainer_losses: weights not loaded"); return CMBPO_ESTATE; }
  CMBPO_REQUIRE(in_dim == t->I && target_dim == t->D, "cmbpo_trainer_losses: dims (%d, %d) != (%d, %d)", in_dim, target_dim, t->I, t->D);
  CMBPO_REQUIRE(n_rows >= 1, "cmbpo_trainer_losses: n_rows %d", n_rows);
  CMBPO_REQUIRE(d_idx != nullptr || n_rows <= t->max_batch, "cmbpo_trainer_losses: %d direct rows exceed max_batch %d (pass an index list)",
                n_rows, t->max_batch);
  hipStream_t s = (hipStream_t)stream;
  const int E = t->E;
  CMBPO_HIP_CHECK(hipMemsetAsync(t->sums, 0, 3 * (size_t)E * sizeof(double), s));
  for (int r0 = 0; r0 < n_rows; r0 += t->max_batch) {
    const int rows = min(t->max_batch, n_rows - r0);
    const int32_t *idx = d_idx ? d_idx + r0 : nullptr;
    int rc = run_forward(t, d_inputs, idx, idx_stride, rows, false, s);
    if (rc != CMBPO_OK) return rc;
    LossArgs l{};
    fill_loss_args(t, l, d_targets, idx, idx_stride, rows);
    l.prob = 0;   // only the squared error of the mean head is needed
    const int gx = min(cmbpo_ceil_div(rows * t->D, kThreads), 64);
    hipLaunchKernelGGL(loss_sums_kernel, dim3(gx, E), dim3(kThreads), 0, s, l);
  }
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, s, t->sums, E, 1.0 / ((double)n_rows * t->D), d_losses);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

extern "C" long cmbpo_trainer_steps_done(const cmbpo_trainer_t *t) { return t ? t->step : -1; }
