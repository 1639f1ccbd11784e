// Both critics of a rollout step in ONE launch (CPOPolicy.get_v / get_vc, policies/cpo_policy.py:825-835 -> PE.predict,
// models/pens/pe.py:338-343,648-669: the mean over ALL members of a 3-layer swish ensemble with one output), with every
// float32 product as three f16 MFMAs (f16_split.h; arithmetic and scales exactly as in ens_h3.hip).
//
// A 128-wide member is too small to share among waves without paying a barrier and an LDS round trip per layer -- as two
// launches of the general kernel the critics were 2 x 19 us of a 120 us step at 1000 branches, the largest item of it --
// but on the f16 matrix cores a whole member is 120 MFMAs: ONE wave carries one member of one critic for a 32-row tile
// from the input to the output without leaving its registers.  The input fragment is split in registers, layer 0's
// accumulator tiles ARE layer 1's B operand (an accumulator tile's rows are the next product's k index; the W1 image is
// packed in the matching k order), the single output column is a VALU dot product over the h2 registers.  A workgroup is
// the 2 x E waves of a tile (both critics read the same observation rows); the members' values meet in LDS and are
// averaged in member order.  No cross-workgroup traffic, no barrier before the final one.
#include "common.h"
#include "ens_mlp_internal.h"
#include "f16_split.h"
#include "policy_f16_tile.h"

#include <stdlib.h>

namespace {

#ifndef IN_SCALE_RCP
#define IN_SCALE_RCP 1     // input scaler as (x - mu) (1 / sigma) (0: the IEEE division, diagnostic)
#endif
constexpr int HC = 128;          // hidden units
constexpr int NT = HC / 32;      // n-tiles
constexpr int S1 = HC / 16;      // k-slabs of layer 1

struct CfNet {
  const f16x8 *w0, *w1;          // images [member][n-tile][slab][piece][lane]
  size_t w0_stride, w1_stride;   // per member, 16-B units
  const float *b0, *b1, *w2, *b2;   // [E][128], [E][128], packed W2 ([E] x wp2_stride float4), [E][32]
  const float *cst;                 // [E][3][128]: b0 log2(e) | b1 log2(e) | W2[:, 0] ln 2 (the epilogues work on z log2(e): one multiply
                                    // less per activation, cf16_consts_kernel)
  size_t w2_stride;              // floats per member of the packed W2
  const float *stats;            // [E][NSTAT]
  const float *in_mu, *in_sig, *out_mu, *out_sig;   // or NULL
  float *out;                    // [branch slot]
};

struct CfArgs {
  CfNet net[2];
  const float *obs;
  int obs_dim, ensemble;
  const int32_t *row_idx, *n_rows_dev;
  int n_rows;
  // optional rider: one more wave per tile evaluates the actor at the same rows (the NEXT step's action at the observation
  // the critics are evaluated at -- the step after this one then starts without a policy launch)
  int has_pol;
  PfArgs pol;
  int has_store;       // the vector half of the step's store rides along (CmbpoStoreVec, ens_mlp_internal.h)
  CmbpoStoreVec sv;
};

template <int S0>   // k-slabs of the input layer (obs_dim <= 16 S0)
__global__ __launch_bounds__(512) void critic_pair_kernel(const CfArgs a) {
  constexpr int KP = 16 * S0;
  extern __shared__ float sm[];
  float *xraw = sm;                         // [32][KP + 1] raw observation rows of the tile
  float *part = xraw + 32 * (KP + 1);       // [2 E][32] member values
  __shared__ int rows[32];
  __shared__ float s_mu[2][KP], s_sig[2][KP];   // the critics' input scalers (identity where there is none)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int E = a.ensemble;
  const int n_rows = a.n_rows_dev ? *a.n_rows_dev : a.n_rows;
  const int row0 = blockIdx.x * 32;
  if (row0 >= n_rows) return;
  const bool is_pol = wave == 2 * E;              // (only launched with has_pol)
  const int ni = is_pol ? 0 : wave / E, e = is_pol ? 0 : wave - ni * E;     // this wave's critic and member
  const CfNet &N = a.net[ni];
  // first weight fragments of layer 0: requested before anything else
  const f16x8 *w0 = N.w0 + (size_t)e * N.w0_stride + lane;      // + ((tile * S0 + s) * 2 + piece) * 64
  const f16x8 *w1 = N.w1 + (size_t)e * N.w1_stride + lane;      // + ((tile * S1 + s) * 2 + piece) * 64
  // The wave streams its member's weights alone (80 KB from L2) and every slab is 12 MFMAs = 0.2 us of work against a load
  // latency several times that: all of W0 and the first slabs of W1 are requested before the input is even staged, and
  // layer 1 keeps a ring of RW slabs in flight.
  constexpr int RW = S0 <= 2 ? 3 : 2;
  f16x8 A0[S0][NT][2], R[3][NT][2];
  auto load_w = [&](f16x8 (&x)[NT][2], const f16x8 *w, int slabs, int s) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f16x8 *q = w + ((size_t)(t * slabs + s) * 2) * 64;
      x[t][0] = q[0]; x[t][1] = q[64];
    }
  };
  // rows, scalers, raw observation rows: staged by every wave of the workgroup (two barriers)
  auto stage = [&]() {
    if (tid < 32) {
      const int rr = row0 + tid;
      int v = rr < n_rows ? rr : 0;
      if (a.row_idx) v = a.row_idx[v];
      rows[tid] = rr < n_rows ? v : -1;
    }
    if (tid < 2 * KP) {
      const int n2 = tid / KP, k = tid - n2 * KP;
      const CfNet &M = a.net[n2];
      const int kc = k < a.obs_dim ? k : 0;
      s_mu[n2][k] = (M.in_mu && k < a.obs_dim) ? M.in_mu[kc] : 0.0f;
      s_sig[n2][k] = (M.in_mu && k < a.obs_dim) ? (IN_SCALE_RCP ? 1.0f / M.in_sig[kc] : M.in_sig[kc]) : 1.0f;      // 1 / sigma
    }
    __syncthreads();
    if (a.has_store) {
      // the tile's rows of the step's store: requested here, complete behind the next barrier -- long before the rider
      // (this workgroup's own) writes the next step's actions over act / mu / log_std of these rows
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int dim = a.sv.dim[f];
        const float inv = 1.0f / (float)dim;
        for (int i = tid; i < 32 * dim; i += blockDim.x) {
          const int b = (int)(((float)i + 0.5f) * inv);   // i / dim (exact: i < 2^16)
          const int d = i - b * dim;
          const int slot = rows[b];
          if (slot >= 0 && !a.sv.fin_code[slot])
            a.sv.dst[f][(a.sv.col_off + (size_t)slot) * dim + d] = a.sv.src[f][(size_t)slot * dim + d];
        }
      }
    }
    for (int i = tid; i < 32 * KP; i += blockDim.x) {      // unconditional loads (clamped), the selection is on the values
      const int b = i / KP, k = i - b * KP;
      const int rr = rows[b];
      const float x = a.obs[(size_t)(rr >= 0 ? rr : 0) * a.obs_dim + (k < a.obs_dim ? k : 0)];
      xraw[b * (KP + 1) + k] = (rr >= 0 && k < a.obs_dim) ? x : 0.0f;
    }
    __syncthreads();
  };
  if (is_pol) {
    // the actor's wave: its own first weight requests, then the common staging, then its tile
    policy_tile<S0, true>(a.pol, row0, n_rows, xraw, rows, lane, stage);
    __syncthreads();      // (the critics' barrier before the member mean)
    return;
  }
#pragma unroll
  for (int s = 0; s < S0; ++s) load_w(A0[s], w0, S0, s);
#pragma unroll
  for (int s = 0; s < RW; ++s) load_w(R[s], w1, S1, s);
  stage();
  const float *st = N.stats + (size_t)e * NSTAT;

  // ---- input fragment: this critic's scaler, the row's lift, the split -- in registers ---------------------------------
  float xs[S0][8];
  float m0 = 0.0f;
#pragma unroll
  for (int s = 0; s < S0; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * s + 8 * hh + j;
      // TensorStandardScaler.transform, utils.py:156, as (x - mu) (1 / sigma): within an ulp of the division (sixteen IEEE
      // divisions per lane were a fifth of a (tile, member) unit's vector instructions)
      float x = IN_SCALE_RCP ? (xraw[r * (KP + 1) + k] - s_mu[ni][k]) * s_sig[ni][k] : (xraw[r * (KP + 1) + k] - s_mu[ni][k]) / s_sig[ni][k];
      if (k >= a.obs_dim) x = 0.0f;
      xs[s][j] = x;
      m0 = fmaxf(m0, fabsf(x));
    }
  m0 = fmaxf(m0, __shfl_xor(m0, 32, 64));
  const float t0 = pow2_lift(m0);
  const float bound1 = (st[1] * m0 + st[2]) * 1.001f, t1 = pow2_lift(bound1);
  const float inv0 = pow2_rcp(st[0] * t0), inv1 = pow2_rcp(st[4] * t1);     // (products of powers of two)

  f32x16 acc[NT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
  };
  zero_acc();
  // ---- layer 0 -------------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < S0; ++s) {
    f16x8 b1, b2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      _Float16 c1, c2;
      split_h(xs[s][j] * t0, c1, c2);
      b1[j] = c1; b2[j] = c2;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) mm3(acc[t], A0[s][t][0], A0[s][t][1], b1, b2);
  }
  // ---- h1 = swish(. + b0), lifted and split: accumulator registers 8 half .. + 7 of tile t are slab 2 t + half of layer 1
  f16x8 bf[S1][2];
  {
    const float *b0 = N.cst + (size_t)e * 3 * HC;            // b0 log2(e)
    const float inv0l = inv0 * kLog2e, it1 = pow2_rcp(t1) * kLog2e;
    u32x4 bu[S1][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(b0 + 32 * t + 8 * q + 4 * hh);
        Epi4 es;
        epi_all<false, true, true>(es, acc[t], q, inv0l, bv, it1);     // (pre-scaled operands, the lift in the reciprocal: f16_split.h)
        const int S = 2 * t + (q >> 1), o = 2 * (q & 1);
        bu[S][0][o] = es.q1[0]; bu[S][0][o + 1] = es.q1[1];
        bu[S][1][o] = es.q2[0]; bu[S][1][o + 1] = es.q2[1];
      }
#pragma unroll
    for (int S = 0; S < S1; ++S) { bf[S][0] = __builtin_bit_cast(f16x8, bu[S][0]); bf[S][1] = __builtin_bit_cast(f16x8, bu[S][1]); }
  }
  zero_acc();
  // ---- layer 1 (its first fragments arrived behind layer 0 and the epilogue) ------------------------------------------------
  if constexpr (RW < 3) load_w(R[2], w1, S1, 2);      // (the third ring slot: free now that W0 is consumed)
#pragma unroll
  for (int s = 0; s < S1; ++s) {
#pragma unroll
    for (int t = 0; t < NT; ++t) mm3(acc[t], R[s % 3][t][0], R[s % 3][t][1], bf[s][0], bf[s][1]);
    if (s + 3 < S1) load_w(R[s % 3], w1, S1, s + 3);
  }
  // ---- h2 = swish(. + b1); output = h2 . W2[:, 0] + b2; output scaler ----------------------------------------------------
  float dot = 0.0f;
  {
    const float *b1 = N.cst + (size_t)e * 3 * HC + HC, *w2 = b1 + HC;     // b1 log2(e), W2[:, 0] ln 2
    const float inv1l = inv1 * kLog2e;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k0 = 32 * t + 8 * q + 4 * hh;
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(b1 + k0);
        const f32x4 wv = *reinterpret_cast<const f32x4 *>(w2 + k0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          // y = z log2(e);  swish(z) w = y sigma(z) (w ln 2): six instructions per activation
          const float y = __builtin_fmaf(acc[t][4 * q + s], inv1l, bv[s]);
          const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-y));
          dot = __builtin_fmaf(y * sg, wv[s], dot);
        }
      }
  }
  dot += __shfl_xor(dot, 32, 64);
  const float o_sig = N.out_mu ? N.out_sig[0] : 1.0f, o_mu = N.out_mu ? N.out_mu[0] : 0.0f;
  const float val = o_sig * (dot + N.b2[(size_t)e * 32]) + o_mu;       // inverse_transform, models/pens/utils.py:167
  if (hh == 0) part[wave * 32 + r] = val;
  __syncthreads();
  if (tid < 64) {                   // mean over ALL members (pe.py:343), added in member order
    const int n2 = tid >> 5, b = tid & 31;
    float sum = 0.0f;
    for (int m = 0; m < E; ++m) sum += part[(n2 * E + m) * 32 + b];
    const int rr = rows[b];
    if (rr >= 0) a.net[n2].out[rr] = sum / (float)E;
  }
}

// ---- the same evaluation at large batches -----------------------------------------------------------------------------
// critic_pair_kernel spends a tile's 15 us waiting: every wave streams its member's 80 KB of weights from L2 for 32 rows, one
// workgroup per CU -- 190 us at 100 k rows, 12 rounds of tiles deep.  Here a workgroup owns a chunk of up to CH rows and takes
// the 2 E members one after the other: a member's W0 | W1 images go to LDS once per workgroup (80 KB), its eight waves carry
// the chunk's 32-row tiles with the A fragments read from LDS (the images are lane-linear: conflict-free 16-byte reads), the
// member values are added per row in LDS in member order.  Per-member arithmetic exactly as in critic_pair_kernel: same bits.
template <int S0>
struct BigGeo {
  static constexpr int KP = 16 * S0, XS = KP + 1;
  static constexpr int CH = S0 <= 2 ? 512 : (S0 == 3 ? 256 : 192);    // rows per workgroup (LDS: weights + the chunk's rows)
  static constexpr int W0Q = NT * S0 * 2 * 64, W1Q = NT * S1 * 2 * 64;   // f16x8 units
  static constexpr size_t LDS = (size_t)(W0Q + W1Q) * 16 + (size_t)CH * XS * 4 + (size_t)2 * CH * 4 + (size_t)CH * 4 + (size_t)4 * KP * 4;
};

template <int S0>
__global__ __launch_bounds__(512) void critic_big_kernel(const CfArgs a, int ch_rows) {
  using G = BigGeo<S0>;
  constexpr int KP = G::KP, XS = G::XS;
  extern __shared__ f16x8 smem8[];
  f16x8 *w0s = smem8, *w1s = w0s + G::W0Q;
  float *xraw = reinterpret_cast<float *>(w1s + G::W1Q);   // [ch_rows][XS]
  float *vsum = xraw + G::CH * XS;                          // [2][CH]
  int *rows = reinterpret_cast<int *>(vsum + 2 * G::CH);    // [CH]
  float *s_mu = reinterpret_cast<float *>(rows + G::CH);    // [2][KP] | s_sig [2][KP]
  float *s_sig = s_mu + 2 * KP;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int E = a.ensemble;
  const int n_rows = a.n_rows_dev ? *a.n_rows_dev : a.n_rows;
  const int row0 = blockIdx.x * ch_rows;
  if (row0 >= n_rows) return;
  const int n_here = min(ch_rows, n_rows - row0), tiles = (n_here + 31) / 32;
  // ---- the chunk's rows ------------------------------------------------------------------------------------------------
  for (int i = tid; i < tiles * 32; i += 512) {
    const int rr = row0 + i;
    int v = rr < n_rows ? rr : 0;
    if (a.row_idx) v = a.row_idx[v];
    rows[i] = rr < n_rows ? v : -1;
    vsum[i] = 0.0f;
    vsum[G::CH + i] = 0.0f;
  }
  if (tid < 2 * KP) {
    const int n2 = tid / KP, k = tid - n2 * KP;
    const CfNet &M = a.net[n2];
    const int kc = k < a.obs_dim ? k : 0;
    s_mu[n2 * KP + k] = (M.in_mu && k < a.obs_dim) ? M.in_mu[kc] : 0.0f;
    s_sig[n2 * KP + k] = (M.in_mu && k < a.obs_dim) ? (IN_SCALE_RCP ? 1.0f / M.in_sig[kc] : M.in_sig[kc]) : 1.0f;      // 1 / sigma
  }
  __syncthreads();
  for (int base = 0; base < tiles * 32 * KP; base += 512 * 4) {     // four requests in flight per thread
    float xv[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + 512 * u + tid, b = i / KP, k = i - b * KP;
      const int rr = i < tiles * 32 * KP ? rows[b] : -1;
      xv[u] = a.obs[(size_t)(rr >= 0 ? rr : 0) * a.obs_dim + (k < a.obs_dim ? k : 0)];
      ok[u] = rr >= 0 && k < a.obs_dim;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + 512 * u + tid, b = i / KP, k = i - b * KP;
      if (i < tiles * 32 * KP) xraw[b * XS + k] = ok[u] ? xv[u] : 0.0f;
    }
  }
  // ---- member after member ---------------------------------------------------------------------------------------------
  for (int net = 0; net < 2 * E; ++net) {
    const int ni = net / E, e = net - ni * E;
    const CfNet &N = a.net[ni];
    __syncthreads();          // the previous member's fragments are read (and, first time, the rows are staged)
    {
      const f16x8 *g0 = N.w0 + (size_t)e * N.w0_stride, *g1 = N.w1 + (size_t)e * N.w1_stride;
      constexpr int TOT = G::W0Q + G::W1Q;
      for (int base = 0; base < TOT; base += 512 * 5) {             // 80 KB: five requests in flight per thread
        f16x8 q[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int i = base + 512 * u + tid;
          const int ic = i < TOT ? i : 0;
          q[u] = ic < G::W0Q ? g0[ic] : g1[ic - G::W0Q];
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const int i = base + 512 * u + tid;
          if (i < TOT) w0s[i] = q[u];          // (w1s follows w0s)
        }
      }
    }
    __syncthreads();
    const float *st = N.stats + (size_t)e * NSTAT;
    const f16x8 *w0 = w0s + lane, *w1 = w1s + lane;
    for (int t = wave; t < tiles; t += 8) {
      const float *xt = xraw + (size_t)t * 32 * XS;
      // input fragment: this critic's scaler, the row's lift, the split -- in registers
      float xs[S0][8];
      float m0 = 0.0f;
#pragma unroll
      for (int s = 0; s < S0; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 16 * s + 8 * hh + j;
          float x = IN_SCALE_RCP ? (xt[r * XS + k] - s_mu[ni * KP + k]) * s_sig[ni * KP + k] : (xt[r * XS + k] - s_mu[ni * KP + k]) / s_sig[ni * KP + k];   // TensorStandardScaler.transform, utils.py:156 (x 1 / sigma)
          if (k >= a.obs_dim) x = 0.0f;
          xs[s][j] = x;
          m0 = fmaxf(m0, fabsf(x));
        }
      m0 = fmaxf(m0, __shfl_xor(m0, 32, 64));
      const float t0 = pow2_lift(m0);
      const float bound1 = (st[1] * m0 + st[2]) * 1.001f, t1 = pow2_lift(bound1);
      const float inv0 = pow2_rcp(st[0] * t0), inv1 = pow2_rcp(st[4] * t1);     // (products of powers of two)
      f32x16 acc[NT];
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tt][i] = 0.0f;
      // layer 0
#pragma unroll
      for (int s = 0; s < S0; ++s) {
        f16x8 b1, b2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          _Float16 c1, c2;
          split_h(xs[s][j] * t0, c1, c2);
          b1[j] = c1; b2[j] = c2;
        }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const f16x8 *q = w0 + ((size_t)(tt * S0 + s) * 2) * 64;
          mm3(acc[tt], q[0], q[64], b1, b2);
        }
      }
      // h1 = swish(. + b0), lifted and split: accumulator registers 8 half .. + 7 of tile tt are slab 2 tt + half of layer 1
      f16x8 bf[S1][2];
      {
        const float *b0 = N.cst + (size_t)e * 3 * HC;            // b0 log2(e)
        const float inv0l = inv0 * kLog2e, it1 = pow2_rcp(t1) * kLog2e;
        u32x4 bu[S1][2];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(b0 + 32 * tt + 8 * q + 4 * hh);
            Epi4 es;
            epi_all<false, true, true>(es, acc[tt], q, inv0l, bv, it1);
            const int S = 2 * tt + (q >> 1), o = 2 * (q & 1);
            bu[S][0][o] = es.q1[0]; bu[S][0][o + 1] = es.q1[1];
            bu[S][1][o] = es.q2[0]; bu[S][1][o + 1] = es.q2[1];
          }
#pragma unroll
        for (int S = 0; S < S1; ++S) { bf[S][0] = __builtin_bit_cast(f16x8, bu[S][0]); bf[S][1] = __builtin_bit_cast(f16x8, bu[S][1]); }
      }
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tt][i] = 0.0f;
      // layer 1
#pragma unroll
      for (int s = 0; s < S1; ++s) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const f16x8 *q = w1 + ((size_t)(tt * S1 + s) * 2) * 64;
          mm3(acc[tt], q[0], q[64], bf[s][0], bf[s][1]);
        }
      }
      // h2 = swish(. + b1); output = h2 . W2[:, 0] + b2; output scaler
      float dot = 0.0f;
      {
        const float *b1 = N.cst + (size_t)e * 3 * HC + HC, *w2 = b1 + HC;
        const float inv1l = inv1 * kLog2e;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int k0 = 32 * tt + 8 * q + 4 * hh;
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(b1 + k0);
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(w2 + k0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const float y = __builtin_fmaf(acc[tt][4 * q + s], inv1l, bv[s]);
              const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-y));
              dot = __builtin_fmaf(y * sg, wv[s], dot);
            }
          }
      }
      dot += __shfl_xor(dot, 32, 64);
      const float o_sig = N.out_mu ? N.out_sig[0] : 1.0f, o_mu = N.out_mu ? N.out_mu[0] : 0.0f;
      const float val = o_sig * (dot + N.b2[(size_t)e * 32]) + o_mu;       // inverse_transform, models/pens/utils.py:167
      if (hh == 0) vsum[ni * G::CH + t * 32 + r] += val;                  // (this wave owns the tile: members in order)
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * tiles * 32; i += 512) {      // mean over ALL members (pe.py:343)
    const int n2 = i / (tiles * 32), b = i - n2 * (tiles * 32);
    const int rr = rows[b];
    if (rr >= 0) a.net[n2].out[rr] = vsum[n2 * G::CH + b] / (float)E;
  }
}

// the epilogues' constants, pre-scaled once per pack: [e][0] = b0 log2(e), [e][1] = b1 log2(e), [e][2] = W2[:, 0] ln 2 (compact)
__global__ void cf16_consts_kernel(const float *b0, const float *b1, const float *w2, size_t w2_stride, float *cst) {
  const int e = blockIdx.x, t = threadIdx.x, f = t / HC, k = t - f * HC;
  float v;
  if (f == 0) v = b0[(size_t)e * HC + k] * kLog2e;
  else if (f == 1) v = b1[(size_t)e * HC + k] * kLog2e;
  else v = w2[(size_t)e * w2_stride + ((size_t)(k >> 3) * 64 + ((k >> 2) & 1) * 32) * 4 + (k & 3)] * kLn2;     // packed (k, n = 0)
  cst[(size_t)e * 3 * HC + t] = v;
}

// (re)builds the two f16 images and the statistics of a 128-wide single-output ensemble when its packs changed
int ensure_cf16(cmbpo_mlp *m, hipStream_t s) {
  const int E = m->ensemble;
  const int S0 = m->h3_s0;
  if (m->d_h3 == nullptr) {
    m->h3_stride[0] = (size_t)NT * S0 * 2 * 64;
    m->h3_stride[1] = (size_t)NT * S1 * 2 * 64;
    m->h3_stride[2] = 0;
    m->h3_off[0] = 0;
    m->h3_off[1] = m->h3_stride[0] * E;
    m->h3_stats_off = m->h3_off[1] + m->h3_stride[1] * E;
    const size_t bytes = m->h3_stats_off * 16 + (size_t)E * NSTAT * sizeof(float) + (size_t)E * 3 * HC * sizeof(float);   // ... | stats | constants
    if (hipMalloc(&m->d_h3, bytes) != hipSuccess) {
      (void)hipGetLastError();
      m->d_h3 = nullptr;
      cmbpo_set_error("critic_f16: hipMalloc of the f16 weight images failed");
      return CMBPO_ENOMEM;
    }
    m->h3_version = ~0ul;
  }
  if (m->h3_version == m->pack_version) return CMBPO_OK;
  float *stats = reinterpret_cast<float *>(reinterpret_cast<char *>(m->d_h3) + m->h3_stats_off * 16);
  cmbpo_internal_f16_stats(m, stats, s);
  f16x8 *base = reinterpret_cast<f16x8 *>(m->d_h3);
  cmbpo_internal_f16_pack(m, 0, base + m->h3_off[0], m->h3_stride[0], NT, S0, 0, stats, s);
  cmbpo_internal_f16_pack(m, 1, base + m->h3_off[1], m->h3_stride[1], NT, S1, 1, stats, s);
  hipLaunchKernelGGL(cf16_consts_kernel, dim3(E), dim3(3 * HC), 0, s, m->d_blob + m->off_b0, m->d_blob + m->off_b1, m->d_blob + m->off_wp2,
                     (size_t)m->o_tiles * (HC / 8) * 256, stats + (size_t)E * NSTAT);
  CMBPO_HIP_CHECK(hipGetLastError());
  m->h3_version = m->pack_version;
  return CMBPO_OK;
}

bool pair_eligible(const cmbpo_mlp *m) {
  return m->head == CMBPO_HEAD_DETMEAN && m->hidden == HC && m->act == CMBPO_ACT_SWISH && m->o_width == 1 && m->in_pad <= 64 &&
         m->ensemble >= 1 && m->ensemble <= 4 && m->loaded;     // 2 E waves of up to 256 registers
}

}  // namespace

extern "C" int cmbpo_critic_pair_supported(const cmbpo_mlp_t *v, const cmbpo_mlp_t *vc) {
  return v && vc && pair_eligible(v) && pair_eligible(vc) && v->in_dim == vc->in_dim && v->ensemble == vc->ensemble ? 1 : 0;
}

extern "C" int cmbpo_critic_pair_predict(cmbpo_mlp_t *v, cmbpo_mlp_t *vc, const float *d_obs, int obs_dim, const int32_t *d_row_idx,
                                         const int32_t *d_n_rows, int n_rows, float *d_v, float *d_vc, void *stream) {
  return cmbpo_internal_critic_pair_ride(v, vc, d_obs, obs_dim, d_row_idx, d_n_rows, n_rows, d_v, d_vc, nullptr, nullptr, nullptr,
                                         nullptr, nullptr, nullptr, stream);
}

// rows from which the critics run member after member with LDS-resident weights (critic_big_kernel; no rider there)
int cmbpo_internal_critic_big_min() {
  static const int v = getenv("CMBPO_CRITIC_BIG_MIN") ? atoi(getenv("CMBPO_CRITIC_BIG_MIN")) : 24576;
  return v;
}

// can the actor ride along? (its f16 kernel applies, same input width, room for one more wave)
bool cmbpo_internal_critic_pair_can_ride(const cmbpo_mlp *v, const cmbpo_mlp *vc, const cmbpo_mlp *policy) {
  return cmbpo_critic_pair_supported(v, vc) && policy && cmbpo_internal_policy_f16_eligible(policy) && policy->in_dim == v->in_dim &&
         2 * v->ensemble + 1 <= 8 && cmbpo_get_ens_matrix_path() == CMBPO_ENS_SPLIT_F16;
}

// the critics at d_obs and -- if policy != NULL -- the actor at the same rows in the same launch (d_eps, outputs slot indexed)
int cmbpo_internal_critic_pair_ride(cmbpo_mlp *v, cmbpo_mlp *vc, const float *d_obs, int obs_dim, const int32_t *d_row_idx,
                                    const int32_t *d_n_rows, int n_rows, float *d_v, float *d_vc, cmbpo_mlp *policy,
                                    const float *d_eps, float *d_pi, float *d_logp, float *d_mu, float *d_ls, void *stream,
                                    const CmbpoStoreVec *store_vec) {
  CMBPO_REQUIRE(v && vc && d_obs && d_v && d_vc, "cmbpo_critic_pair_predict: NULL argument");
  CMBPO_REQUIRE(cmbpo_critic_pair_supported(v, vc), "cmbpo_critic_pair_predict: needs two loaded 128-wide swish ensembles of equal "
                                                    "size with one output (HEAD_DETMEAN)");
  CMBPO_REQUIRE(obs_dim == v->in_dim, "cmbpo_critic_pair_predict: obs_dim %d, the critics take %d", obs_dim, v->in_dim);
  CMBPO_REQUIRE(n_rows >= 0, "cmbpo_critic_pair_predict: n_rows %d", n_rows);
  if (n_rows == 0) return CMBPO_OK;
  hipStream_t s = (hipStream_t)stream;
  CfArgs a{};
  cmbpo_mlp *ms[2] = {v, vc};
  float *outs[2] = {d_v, d_vc};
  const int s0 = (v->in_pad + 15) / 16;
  for (int i = 0; i < 2; ++i) {
    cmbpo_mlp *m = ms[i];
    if (m->h3_s0 == 0) m->h3_s0 = s0 < 2 ? 2 : s0;
    if (int rc = ensure_cf16(m, s)) return rc;
    const float *blob = m->d_blob;
    CfNet &n = a.net[i];
    const f16x8 *base = reinterpret_cast<const f16x8 *>(m->d_h3);
    n.w0 = base + m->h3_off[0]; n.w1 = base + m->h3_off[1];
    n.w0_stride = m->h3_stride[0]; n.w1_stride = m->h3_stride[1];
    n.b0 = blob + m->off_b0; n.b1 = blob + m->off_b1; n.b2 = blob + m->off_b2;
    n.w2 = blob + m->off_wp2; n.w2_stride = (size_t)m->o_tiles * (HC / 8) * 256;
    n.stats = reinterpret_cast<const float *>(reinterpret_cast<const char *>(m->d_h3) + m->h3_stats_off * 16);
    n.cst = n.stats + (size_t)m->ensemble * NSTAT;
    n.in_mu = m->has_in_scaler ? blob + m->off_in_mu : nullptr;
    n.in_sig = m->has_in_scaler ? blob + m->off_in_var : nullptr;
    n.out_mu = m->has_out_scaler ? blob + m->off_out_mu : nullptr;
    n.out_sig = m->has_out_scaler ? blob + m->off_out_var : nullptr;
    n.out = outs[i];
  }
  a.obs = d_obs; a.obs_dim = obs_dim; a.ensemble = v->ensemble;
  a.row_idx = d_row_idx; a.n_rows_dev = d_n_rows; a.n_rows = n_rows;
  const int S0 = ms[0]->h3_s0;
  a.has_pol = 0;
  if (policy != nullptr) {
    CMBPO_REQUIRE(cmbpo_internal_critic_pair_can_ride(v, vc, policy) && d_eps && d_pi && d_logp && d_mu && d_ls,
                  "critic pair: the actor cannot ride along (shape / path) or a NULL buffer");
    int ps0 = 0;
    if (int rc = cmbpo_internal_policy_f16_args(policy, &a.pol, &ps0, s)) return rc;
    CMBPO_REQUIRE(ps0 == S0, "critic pair: actor and critics disagree on the input slabs (%d, %d)", ps0, S0);
    a.pol.obs = d_obs; a.pol.eps = d_eps; a.pol.row_idx = d_row_idx; a.pol.n_rows_dev = d_n_rows; a.pol.n_rows = n_rows;
    a.pol.pi = d_pi; a.pol.logp = d_logp; a.pol.mu = d_mu; a.pol.ls = d_ls;
    a.has_pol = 1;
  }
  a.has_store = 0;
  if (store_vec != nullptr) {
    CMBPO_REQUIRE(d_row_idx != nullptr && n_rows < cmbpo_internal_critic_big_min() && store_vec->fin_code,
                  "critic pair: the store rides with the one-wave-per-member kernel on a row list only");
    a.sv = *store_vec;
    a.has_store = 1;
  }
  // large batches without a rider: members one after the other with their weights in LDS (critic_big_kernel)
  if (!a.has_pol && n_rows >= cmbpo_internal_critic_big_min()) {
    hipDeviceProp_t prop;
    static int n_cu = 0;
    if (n_cu == 0) {
      int dev = 0;
      n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                 ? prop.multiProcessorCount : 256;
    }
    auto launch_big = [&](auto geo, auto kern) -> int {
      using G = decltype(geo);
      static bool attr = false;
      if (!attr) {
        CMBPO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS));
        attr = true;
      }
      int ch = (cmbpo_ceil_div(n_rows, n_cu) + 31) / 32 * 32;      // one chunk per CU where the batch allows
      if (ch > G::CH) ch = G::CH;
      hipLaunchKernelGGL(kern, dim3(cmbpo_ceil_div(n_rows, ch)), dim3(512), G::LDS, s, a, ch);
      CMBPO_HIP_CHECK(hipGetLastError());
      return CMBPO_OK;
    };
    if (S0 == 2) return launch_big(BigGeo<2>{}, critic_big_kernel<2>);
    if (S0 == 3) return launch_big(BigGeo<3>{}, critic_big_kernel<3>);
    return launch_big(BigGeo<4>{}, critic_big_kernel<4>);
  }
  const int threads = 64 * (2 * v->ensemble + a.has_pol);
  const size_t lds = ((size_t)32 * (16 * S0 + 1) + (size_t)2 * v->ensemble * 32) * sizeof(float);
  const int grid = cmbpo_ceil_div(n_rows, 32);
  if (S0 == 2) hipLaunchKernelGGL(critic_pair_kernel<2>, dim3(grid), dim3(threads), lds, s, a);
  else if (S0 == 3) hipLaunchKernelGGL(critic_pair_kernel<3>, dim3(grid), dim3(threads), lds, s, a);
  else hipLaunchKernelGGL(critic_pair_kernel<4>, dim3(grid), dim3(threads), lds, s, a);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}
