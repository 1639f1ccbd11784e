// The actor's forward pass of one 32-row tile on the three-term f16 path: shared by policy_f16.hip (its own launch) and
// critic_f16.hip (as one more wave of the critics' launch: the next step's action at the observation the critics are
// evaluated at).  Same code, same bits, whichever launch carries it.
#pragma once
#include "common.h"
#include "f16_split.h"

namespace {


constexpr int HP = 128;          // hidden units
constexpr int NTP = HP / 32;     // n-tiles
constexpr int SP1 = HP / 16;     // k-slabs of the hidden-width products
constexpr float T_H = 16384.0f;  // lift of a tanh output

struct PfArgs {
  const f16x8 *w0, *w1, *w2;     // images [n-tile][slab][piece][lane]
  const float *b0, *b1, *b2, *log_std, *stats;
  const float *obs, *eps;
  int obs_dim, act_dim;
  const int32_t *row_idx, *n_rows_dev;
  int n_rows;
  float *pi, *logp, *mu, *ls;
};

// four accumulator values -> tanh(d inv + b) -> lifted, split, packed (two dwords per piece)
__device__ __forceinline__ void tanh_split4(const f32x16 &d, int q, float inv, const f32x4 &bv, unsigned (&q1)[2], unsigned (&q2)[2]) {
  float h[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) h[i] = cmbpo_fast_tanh(__builtin_fmaf(d[4 * q + i], inv, bv[i]));
  split2<true>(h[0], h[1], T_H, q1[0], q2[0]);
  split2<true>(h[2], h[3], T_H, q1[1], q2[1]);
}

// One wave, one 32-row tile.  STAGED: the tile's raw rows ([32][16 S0 + 1] floats) and slot ids (rows[32], -1 past the end) are
// put into LDS by stage() (the critics' launch stages them with all its waves, barriers included: called here, behind this
// wave's first weight requests); otherwise this wave stages them itself.
struct NoStage { __device__ void operator()() const {} };
template <int S0, bool STAGED, class STAGE = NoStage>   // S0: k-slabs of the input layer (obs_dim <= 16 S0)
__device__ __forceinline__ void policy_tile(const PfArgs &a, int row0, int n_rows, float *xraw, int *rows, int lane,
                                            STAGE &&stage = STAGE()) {
  constexpr int KP = 16 * S0;
  const int r = lane & 31, hh = lane >> 5;
  const f16x8 *w0 = a.w0 + lane, *w1 = a.w1 + lane, *w2 = a.w2 + lane;
  // all of W0 and the first slabs of W1 are requested before the rows are staged (critic_f16.hip)
  constexpr int RW = S0 <= 2 ? 3 : 2;
  f16x8 A0[S0][NTP][2], R[3][NTP][2];
  auto load_w = [&](f16x8 (&x)[NTP][2], const f16x8 *w, int slabs, int s) {
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
      const f16x8 *q = w + ((size_t)(t * slabs + s) * 2) * 64;
      x[t][0] = q[0]; x[t][1] = q[64];
    }
  };
#pragma unroll
  for (int s = 0; s < S0; ++s) load_w(A0[s], w0, S0, s);
#pragma unroll
  for (int s = 0; s < RW; ++s) load_w(R[s], w1, SP1, s);
  if constexpr (!STAGED) {
    if (lane < 32) {
      const int rr = row0 + lane;
      int v = rr < n_rows ? rr : 0;
      if (a.row_idx) v = a.row_idx[v];
      rows[lane] = rr < n_rows ? v : -1;
    }
    __builtin_amdgcn_wave_barrier();
    // unconditional loads (clamped), the selection is on the values; eight requests in flight per lane before the first
    // LDS write (one at a time, the loop is sixteen global round trips long)
    for (int base = 0; base < 32 * KP; base += 64 * 8) {
      float xv[8];
      bool ok[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + 64 * u + lane, b = i / KP, k = i - b * KP;
        const int rr = rows[b];
        xv[u] = a.obs[(size_t)(rr >= 0 ? rr : 0) * a.obs_dim + (k < a.obs_dim ? k : 0)];
        ok[u] = rr >= 0 && k < a.obs_dim;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + 64 * u + lane, b = i / KP, k = i - b * KP;
        xraw[b * (KP + 1) + k] = ok[u] ? xv[u] : 0.0f;
      }
    }
    __builtin_amdgcn_wave_barrier();
  } else {
    stage();
  }
  const float *st = a.stats;

  // ---- input fragment: the row's lift, the split -- in registers -------------------------------------------------------
  float xs[S0][8];
  float m0 = 0.0f;
#pragma unroll
  for (int s = 0; s < S0; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = xraw[r * (KP + 1) + 16 * s + 8 * hh + j];
      xs[s][j] = x;
      m0 = fmaxf(m0, fabsf(x));
    }
  m0 = fmaxf(m0, __shfl_xor(m0, 32, 64));
  const float t0 = pow2_lift(m0);
  const float inv0 = pow2_rcp(st[0] * t0), inv1 = pow2_rcp(st[4] * T_H), inv2 = pow2_rcp(st[8] * T_H);     // (powers of two)

  f32x16 acc[NTP];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < NTP; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
  };
  zero_acc();
  // ---- layer 0 -------------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < S0; ++s) {
    unsigned c1[4], c2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2<true>(xs[s][2 * j], xs[s][2 * j + 1], t0, c1[j], c2[j]);
    const u32x4 u1 = {c1[0], c1[1], c1[2], c1[3]}, u2 = {c2[0], c2[1], c2[2], c2[3]};
    const f16x8 b1 = __builtin_bit_cast(f16x8, u1), b2 = __builtin_bit_cast(f16x8, u2);
#pragma unroll
    for (int t = 0; t < NTP; ++t) mm3(acc[t], A0[s][t][0], A0[s][t][1], b1, b2);
  }
  // ---- a hidden layer's accumulators -> tanh -> the next product's B fragments: registers 8 half .. + 7 of tile t are slab
  // 2 t + half
  f16x8 bf[SP1][2];
  auto to_frags = [&](const float *bias, float inv) {
    u32x4 bu[SP1][2];
#pragma unroll
    for (int t = 0; t < NTP; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + 32 * t + 8 * q + 4 * hh);
        unsigned q1[2], q2[2];
        tanh_split4(acc[t], q, inv, bv, q1, q2);
        const int S = 2 * t + (q >> 1), o = 2 * (q & 1);
        bu[S][0][o] = q1[0]; bu[S][0][o + 1] = q1[1];
        bu[S][1][o] = q2[0]; bu[S][1][o + 1] = q2[1];
      }
#pragma unroll
    for (int S = 0; S < SP1; ++S) { bf[S][0] = __builtin_bit_cast(f16x8, bu[S][0]); bf[S][1] = __builtin_bit_cast(f16x8, bu[S][1]); }
  };
  to_frags(a.b0, inv0);
  zero_acc();
  // ---- layer 1 (its first fragments arrived behind layer 0 and the epilogue) ------------------------------------------------
  if constexpr (RW < 3) load_w(R[2], w1, SP1, 2);      // (the third ring slot: free now that W0 is consumed)
#pragma unroll
  for (int s = 0; s < SP1; ++s) {
#pragma unroll
    for (int t = 0; t < NTP; ++t) mm3(acc[t], R[s % 3][t][0], R[s % 3][t][1], bf[s][0], bf[s][1]);
    if (s + 3 < SP1) load_w(R[s % 3], w1, SP1, s + 3);
  }
  // the output layer is one n-tile: its eight slabs (two pieces each) are requested while the epilogue runs
  f16x8 W2[SP1][2];
#pragma unroll
  for (int s = 0; s < SP1; ++s) { W2[s][0] = w2[(size_t)(2 * s) * 64]; W2[s][1] = w2[(size_t)(2 * s + 1) * 64]; }
  to_frags(a.b1, inv1);
  // ---- mu = h2 W2 + b2 ----------------------------------------------------------------------------------------------------
  f32x16 o;
#pragma unroll
  for (int i = 0; i < 16; ++i) o[i] = 0.0f;
#pragma unroll
  for (int s = 0; s < SP1; ++s) mm3(o, W2[s][0], W2[s][1], bf[s][0], bf[s][1]);
  // ---- head: accumulator register 4 q + e of lane (r, hh) is action 8 q + 4 hh + e of row r -----------------------------
  const int rr = rows[r];
  const int A = a.act_dim;
  const int nq = (A + 7) >> 3;            // groups of 8 actions that exist: wave-uniform
  const size_t rbase = (size_t)(rr >= 0 ? rr : 0) * A;
  float lp = 0.0f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q < nq) {
      // every load of the group is requested before anything is computed, unconditionally (clamped): a load under a
      // per-lane condition makes hipcc wait for each one in turn, and the head took five times as long as the rest of the tile
      float ev[4], bv[4], lv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ac = 8 * q + 4 * hh + e, acc = ac < A ? ac : A - 1;
        ev[e] = a.eps[rbase + acc];
        bv[e] = a.b2[acc];
        lv[e] = a.log_std[acc];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ac = 8 * q + 4 * hh + e;
        const bool valid = ac < A && rr >= 0;
        const float mu = __builtin_fmaf(o[4 * q + e], inv2, bv[e]);
        const float ls = lv[e];
        const float sd = expf(ls);
        const float pi = mu + ev[e] * sd;
        const float z = (pi - mu) / (sd + 1e-8f);
        const float term = -0.5f * (z * z + 2.0f * ls + 1.8378770664093453f);   // gaussian_likelihood, network/ac_network.py:46-48
        lp += valid ? term : 0.0f;
        if (valid) {
          const size_t at = rbase + ac;
          a.pi[at] = pi;
          a.mu[at] = mu;
          a.ls[at] = ls;
        }
      }
    }
  }
  lp += __shfl_xor(lp, 32, 64);
  if (hh == 0 && rr >= 0) a.logp[rr] = lp;
}


}  // namespace
