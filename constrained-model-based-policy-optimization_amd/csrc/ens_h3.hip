// Ensemble forward (HEAD_PROB, 512-wide, swish) with every float32 product carried by THREE f16 MFMAs.
//
// Same function as ens_mlp_kernel<512, *, swish, prob> / ens_split_kernel (models/pens/pe.py:688-697,789-838,
// fc.py:74-95, models/pens/utils.py:156-187): one item = 128 rows of one member through
//     x -> swish(x W0 + b0) -> swish(. W1 + b1) -> . W2 + b2 -> (mean, var).
//
// Arithmetic.  A float32 number a, lifted by a power of two s into the top of the f16 range, splits into two f16
// pieces a1 = f16(a s), a2 = f16(a s - a1) (the difference is exact in float32): 11 + 1 + 11 significant bits, i.e.
// a s = (a1 + a2)(1 + d), |d| <= 2^-24 -- a float32 rounding.  Then
//     a . b  =  [a2 b1 + a1 b2 + a1 b1] / (s t)  +  O(2^-23 |ab|)
// as three v_mfma_f32_32x32x16_f16 with fp32 accumulation (every partial product exact, smallest first; the dropped
// a2 b2 <= 2^-24 |ab|).  Measured (tools/split_f16_probe.hip): error 3.2e-7 of sum|a_k b_k| at K = 512 -- the fp32 MFMA
// chain 7.6e-7, the six-term bf16 split 6.2e-7 -- at half the matrix instructions of the latter.
// Scales (all powers of two, so scaling and unscaling are exact):
//   * weights: one per (member, layer), max |W| s in [2^13, 2^14)  (h3_stats_kernel, whenever the packs change);
//   * activations: one per (row, layer), from a BOUND known before the layer runs: the row's max |x| is taken when the
//     input is staged, and |h1| <= |z1| <= L1 m0 + B, |h2| <= L2 bound1 + B' with L = the member's largest column 1-norm
//     of W and B = max |b| (swish(z) <= |z|).  bound t in [2^13, 2^14): no piece can overflow for any finite input, no
//     cross-wave reduction is needed, and a loose bound only moves the SMALLEST elements of a row towards the f16
//     subnormals: an element keeps full relative precision down to 2^-17 of the bound, below that its absolute error is
//     <= 2^-39 of the bound.  Non-finite inputs give non-finite outputs for their own row only (rows are MFMA columns).
//
// Work decomposition (what round 1's stamps asked for): 512 threads = 8 waves = two per SIMD, so the non-matrix
// phases issue VALU at the SIMD's full rate and one wave's waits hide behind its partner's MFMAs; 128-row items, so a
// weight fragment fetched from L2 feeds four 32-row tiles (the probe's slab loop is L2 -> CU bound at 64 rows: 489 ->
// 680 ns per slab with the loads); every activation is split ONCE, by the wave that produced it, into two f16 images in
// LDS which all waves read as ready MFMA operands (no VALU in the slab loop).  A 128-row h1 image would need 266 KB, so
// layers 0 and 1 are fused over k: chunk c of h1 (64 hidden units x 128 rows, 8 (n-tile, row-tile) pairs = one per
// wave) is produced into a double-buffered 36 KB image while the layer-1 MFMAs consume chunk c - 1.  Wave w owns hidden
// n-tiles 2w, 2w+1 x four row tiles of layer 1 (128 accumulator registers).  h2 leaves the accumulators in two 64-row
// halves through a [64][512] image; the output layer is split over (output tile, row tile, k half) units, its results
// pass through an LDS staging tile so that rows are stored as contiguous runs.
#include "common.h"
#include "ens_mlp_internal.h"
#include "f16_split.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int kThreadsH = 512;
constexpr int kWavesH = 8;
constexpr int HIDH = 512;

// Geometry of an item of RT 32-row tiles (4: the throughput shape; 2 / 1: the same kernel for rollout batches too small
// to give every CU a 128-row item -- an item's latency is the step's there).  Whatever RT, a chunk of h1 is 8 (n-tile,
// row-tile) pairs = one per wave, and a step of the fused layer-0/1 loop is 16 (slab, row tile) positions of 6 MFMAs.
template <int RT>
struct Geo {
  static constexpr int ROWS = 32 * RT;
  static constexpr int NTC = 8 / RT;           // hidden n-tiles per chunk
  static constexpr int CK = 32 * NTC;          // hidden units per chunk
  static constexpr int NCH = HIDH / CK;        // chunks
  static constexpr int SLC = CK / 16;          // 16-deep slabs per chunk
  static constexpr int CSTR = CK + 8;          // row stride (halves) of a chunk image: 16-B slots per row odd -> conflict-free
  static constexpr int TPR = kThreadsH / ROWS; // threads staging one input row
  static constexpr int KPT = 64 / TPR;         // input elements per staging thread
  static constexpr int DA = RT == 4 ? 2 : 4;   // ring of layer-1 weight fragments (slabs): DA - 1 slabs of lookahead
  static constexpr int CBUF_BYTES = 2 * ROWS * CSTR * 2;          // one chunk image (both pieces)
  static constexpr bool W0_LDS = RT == 4;      // W0 fragments shared by several waves pass through LDS
};

// ---- LDS map (bytes) ---------------------------------------------------------------------------------------------
// [0, ...)            layers 0 / 1: chunk images [2][2 pieces][ROWS][CSTR] | x image [2][ROWS][XSTR] | W0 chunk [2][4 S0 KB]
//                     tail (aliases the above): partial outputs [2][8 waves][2 tiles][16][64] f32 | staging [2][32][SWS]
// [OFF_CONST, ...)    bias0 | bias1 | head constants | per-row scales | input scaler
constexpr int PBUF_BYTES = kWavesH * 2 * 16 * 64 * 4;     // partial outputs of one unit: [8 waves][2 tiles][4][64 lanes] x 16 B
constexpr int SWS = 65;                                   // row stride of a staging tile (floats): odd
constexpr int STG_BYTES = 32 * SWS * 4;
__host__ __device__ constexpr int ximg_bytes(int S0, int RT) { return 2 * 32 * RT * (16 * S0 + 8) * 2; }
__host__ __device__ constexpr int w0buf_bytes(int S0, int RT) { return RT == 4 ? 2 * S0 * 2 * 1024 : 0; }   // one chunk
__host__ __device__ constexpr int off_const(int S0, int RT) {
  const int cb = RT == 4 ? Geo<4>::CBUF_BYTES : (RT == 2 ? Geo<2>::CBUF_BYTES : Geo<1>::CBUF_BYTES);
  const int a = 2 * cb + ximg_bytes(S0, RT) + 2 * w0buf_bytes(S0, RT), b = 2 * PBUF_BYTES + 2 * STG_BYTES;
  return ((a > b ? a : b) + 15) / 16 * 16;
}
constexpr int CONST_FLOATS = 2 * HIDH + 2 * 128 + 6 * 128 + 2 * 64;
__host__ __device__ constexpr int lds_bytes(int S0, int RT) { return off_const(S0, RT) + CONST_FLOATS * 4; }

#ifdef CMBPO_STAMPS
#define H3_STAMP(k)                                                         \
  do {                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                      \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();          \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                     \
    t_acc[k] += t_now - t_last;                                             \
    t_last = t_now;                                                         \
    __builtin_amdgcn_sched_barrier(0);                                      \
  } while (0)
#else
#define H3_STAMP(k) do { } while (0)
#endif

// ---- per-member statistics of the fp32 packs: max |W|, largest column 1-norm, max |b| -----------------------------------
// pack layout [n-tile][k-group][lane (r, h)][4]: W[k = 8 g + 4 h + s][n = 32 tile + r].  grid (E, 3), 512 threads.
__global__ void h3_stats_kernel(const float *blob, size_t off0, size_t off1, size_t off2, size_t offb0, size_t offb1, size_t offb2,
                                int kg0, int o_tiles, int hidden, float *stats) {
  const int e = blockIdx.x, l = blockIdx.y, tid = threadIdx.x;
  const int tiles = l == 2 ? o_tiles : hidden / 32, kg = l == 0 ? kg0 : hidden / 8;
  const size_t woff = l == 0 ? off0 : (l == 1 ? off1 : off2), boff = l == 0 ? offb0 : (l == 1 ? offb1 : offb2);
  const float *w = blob + woff + (size_t)e * tiles * kg * 256;
  const float *b = blob + boff + (size_t)e * tiles * 32;
  float wmax = 0.0f, l1 = 0.0f, bmax = 0.0f;
  if (tid < tiles * 32) {
    const int tile = tid >> 5, r = tid & 31;
    for (int g = 0; g < kg; ++g)
      for (int h = 0; h < 2; ++h) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(w + (((size_t)tile * kg + g) * 64 + h * 32 + r) * 4);
        for (int s = 0; s < 4; ++s) { const float a = fabsf(v[s]); wmax = fmaxf(wmax, a); l1 += a; }
      }
    bmax = fabsf(b[tid]);
  }
  __shared__ float red[3][kThreadsH];
  red[0][tid] = wmax; red[1][tid] = l1; red[2][tid] = bmax;
  __syncthreads();
  for (int st = kThreadsH / 2; st > 0; st >>= 1) {
    if (tid < st)
      for (int q = 0; q < 3; ++q) red[q][tid] = fmaxf(red[q][tid], red[q][tid + st]);
    __syncthreads();
  }
  if (tid == 0) {
    float *o = stats + (size_t)e * NSTAT + 4 * l;
    o[0] = pow2_lift(red[0][0]); o[1] = red[1][0]; o[2] = red[2][0]; o[3] = red[0][0];
  }
}

// ---- fp32 pack -> two f16 images [n-tile][k-slab 16][piece 2][lane 64][8 halves] -----------------------------------------
// lane (r, h), element j of slab s holds W[n = 32 tile + r][k = 16 s + 8 h + j] * scale; tiles >= src_tiles and k beyond the
// pack are zero.  perm (output layer: its B operand is an accumulator tile, whose register i of lane half h is hidden
// unit (i & 3) + 8 (i >> 2) + 4 h of the tile): slab s covers registers 8 (s & 1) .. + 7 of hidden n-tile s >> 1, i.e.
// k = 32 (s >> 1) + (i & 3) + 8 (i >> 2) + 4 h with i = 8 (s & 1) + j.
__global__ void h3_pack_kernel(const float *src, size_t src_stride, int kg, int src_tiles, f16x8 *dst, size_t dst_stride,
                               int n_tiles, int slabs, int members, const float *stats, int layer, int perm) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per = (long)n_tiles * slabs * 64;
  if (idx >= per * members) return;
  const int e = (int)(idx / per);
  const int rem = (int)(idx - (long)e * per);
  const int lane = rem & 63, s = (rem >> 6) % slabs, tile = (rem >> 6) / slabs;
  const int r = lane & 31, h = lane >> 5;
  const float scale = stats[(size_t)e * NSTAT + 4 * layer];
  const float *sp = src + (size_t)e * src_stride;
  f16x8 p1, p2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int k = 16 * s + 8 * h + j;
    if (perm) {
      const int i = 8 * (s & 1) + j;
      k = 32 * (s >> 1) + (i & 3) + 8 * (i >> 2) + 4 * h;
    }
    float v = 0.0f;
    if (tile < src_tiles && (k >> 3) < kg) v = sp[(((size_t)tile * kg + (k >> 3)) * 64 + ((k >> 2) & 1) * 32 + r) * 4 + (k & 3)];
    _Float16 q1, q2;
    split_h(v * scale, q1, q2);
    p1[j] = q1; p2[j] = q2;
  }
  f16x8 *d = dst + (size_t)e * dst_stride + ((size_t)tile * slabs + s) * 2 * 64 + lane;
  d[0] = p1; d[64] = p2;
}

// compile-time loop: f(integral_constant<int, I>) for I in [I0, N) -- indices of register arrays must be constants
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// The item loop's barriers order LDS traffic only (images, partial sums, staging tiles; nothing stored to global memory is read
// back inside the kernel).  __syncthreads() is a workgroup fence + s_barrier, in front of which hipcc waits vmcnt(0): for
// every global access in flight -- the output stores of the previous unit (a write acknowledgement from HBM), the next
// item's input rows and row indices, the weight fragments requested a slab ahead.  Draining the LDS counter is enough.
#ifdef H3_FULL_BARRIERS      // diagnostic: round 2's barriers
#define H3_BARRIER() __syncthreads()
#else
#define H3_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

#ifndef H3_EPI_FUSE
#define H3_EPI_FUSE 1       // diagnostic: 0 = round 2's epilogue (separate product z sigma(z), the lift in the split's FMAs)
#endif
#ifndef IN_SCALE_RCP
#define IN_SCALE_RCP 1       // input scaler as (x - mu) (1 / sigma) (0: the IEEE division, diagnostic)
#endif
#ifndef H3_PEEL_LAST
#define H3_PEEL_LAST 1      // the fused loop's last step as an instance of its own, without the production of a chunk nobody reads
                            // (0: round 2's branch-free loop; measured 1.212 / 1.215 -> 1.199 / 1.198 ms per 100 k-row forward)
#endif
#ifndef H3_STORE8
#define H3_STORE8 1         // diagnostic: 0 = the outputs as 4-byte lane stores (round 2)
#endif
#ifndef H3_TAIL_RING
#define H3_TAIL_RING 0      // diagnostic: 1 = W2 fragments through the two-slab ring for every shape (round 2's tail)
#endif

struct H3Args {
  MlpKernelArgs m;
  const f16x8 *w0, *w1, *w2;
  size_t w0_stride, w1_stride, w2_stride;   // per member, in 16-B units
  const float *stats;                        // [E][NSTAT]
  int item0;                                 // first item of this launch (a launch may carry a suffix of the item list)
};

// S0: k-slabs of the input layer (in_pad <= 16 S0); OTP: output n-tiles, 2 or 4 (2 out_dim <= 32 OTP); RT: 32-row tiles per item
template <int S0, int OTP, int RT>
__global__ __launch_bounds__(kThreadsH, 2) void ens_h3_kernel(const H3Args a) {
  using G = Geo<RT>;
  constexpr int ROWSH = G::ROWS, NTC = G::NTC, CK = G::CK, NCH = G::NCH, SLC = G::SLC, CSTR = G::CSTR, TPR = G::TPR,
                KPT = G::KPT, DA = G::DA, CBUF_BYTES = G::CBUF_BYTES;
  constexpr int XSTR = 16 * S0 + 8;
  constexpr int NPASS = OTP / 2;         // output tiles are taken two at a time
  constexpr int NUNIT = RT * NPASS;      // (row tile, pass) units of the tail
  constexpr int O_PAD = 32 * OTP;
  constexpr int W0P = NTC * S0 * 2;      // 1-KB pieces of one W0 chunk
  const MlpKernelArgs &p = a.m;
  extern __shared__ f32x4 smem4[];
  char *smem = reinterpret_cast<char *>(smem4);
  _Float16 *cbuf = reinterpret_cast<_Float16 *>(smem);                               // [2][2][128][CSTR]
  _Float16 *ximg = reinterpret_cast<_Float16 *>(smem + 2 * CBUF_BYTES);              // [2][128][XSTR]
  f16x8 *w0buf = reinterpret_cast<f16x8 *>(smem + 2 * CBUF_BYTES + ximg_bytes(S0, RT));   // [2][W0P][64] (RT == 4 only)
  f32x4 *pbuf = reinterpret_cast<f32x4 *>(smem);                                     // [2][8 waves][2 tiles][4][64] x 16 B
  float *stg = reinterpret_cast<float *>(smem + 2 * PBUF_BYTES);                     // [2][32][SWS]
  float *cst = reinterpret_cast<float *>(smem + off_const(S0, RT));
  float *bias0 = cst, *bias1 = cst + HIDH, *hc_a = cst + 2 * HIDH, *hc_c = hc_a + 128;
  int *rowidx = reinterpret_cast<int *>(hc_c + 128);
  float *r_inv0 = reinterpret_cast<float *>(rowidx) + 128, *r_t1 = r_inv0 + 128, *r_inv1 = r_t1 + 128, *r_t2 = r_inv1 + 128,
        *r_inv2 = r_t2 + 128;
  float *in_mu_l = r_inv2 + 128, *in_sig_l = in_mu_l + 64;

  const int n_rows = p.n_rows_dev ? *p.n_rows_dev : p.n_rows;
  const int out = p.out_dim;
  if (threadIdx.x < 64) {   // input scaler, once per workgroup (TensorStandardScaler.transform, models/pens/utils.py:156)
    const int k = threadIdx.x;
    in_mu_l[k] = (p.in_mu && k < p.in_dim) ? p.in_mu[k] : 0.0f;
    in_sig_l[k] = (p.in_mu && k < p.in_dim) ? (IN_SCALE_RCP ? 1.0f / p.in_sig[k] : p.in_sig[k]) : 1.0f;      // 1 / sigma: see the stage
  }
  __syncthreads();

  // ---- prefetch registers: the NEXT item's raw input rows, biases and output bias travel behind the current item's tail.
  // Every load is unconditional (clamped addresses, the selection happens on the values): a load inside a per-element
  // branch makes hipcc wait for it there, one memory round trip per element.
  float xpre[KPT];
  float bpre[2], b2pre = 0.0f;
  int rr_pre = -1;
  auto fetch_row = [&](int it, int tid) {
    const int xb = tid / TPR;
    const int e2 = it / p.tiles;
    const int rr = (it - e2 * p.tiles) * ROWSH + xb;
    const bool ok = it < p.n_items && rr < n_rows;
    int v = ok ? rr : 0;
    if (p.row_idx) v = p.row_idx[v];
    rr_pre = ok ? v : -1;
  };
  auto fetch_x = [&](int tid) {
    const int xc = tid % TPR;
    const int rr = rr_pre >= 0 ? rr_pre : 0;
    const float *orow = p.obs + (size_t)rr * p.obs_dim;
    const float *arow = p.act_dim > 0 ? p.act + (size_t)rr * p.act_dim - p.obs_dim : orow;
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
      const int k = KPT * xc + u;
      const float *q = (k < p.obs_dim) ? orow + k : ((k < p.in_dim) ? arow + k : orow);
      xpre[u] = *q;
    }
  };
  auto fetch_bias = [&](int it, int tid) {
    const int e2 = (it < p.n_items) ? it / p.tiles : 0;
    bpre[0] = p.b0[(size_t)e2 * HIDH + tid];
    bpre[1] = p.b1[(size_t)e2 * HIDH + tid];
    const int ob = p.o_tiles * 32;
    b2pre = p.b2[(size_t)e2 * ob + (tid < ob ? tid : 0)];
  };
  fetch_row(a.item0 + blockIdx.x, threadIdx.x);
  fetch_x(threadIdx.x);
  fetch_bias(a.item0 + blockIdx.x, threadIdx.x);
#ifdef CMBPO_STAMPS
  unsigned long long t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
  const unsigned long long t_rt0 = __builtin_amdgcn_s_memrealtime();
#endif

#ifdef H3_EXP_PRIO
  if (threadIdx.x >= 256) __builtin_amdgcn_s_setprio(1);     // the second-dispatched half loses every arbitration otherwise
#endif
  for (int item = a.item0 + blockIdx.x; item < p.n_items; item += gridDim.x) {
    // the thread index, re-read inside the loop through an opaque move: everything derived from it is recomputed per
    // item instead of being hoisted out of the loop and parked in scratch (the loop body needs every register)
    int tid = threadIdx.x;
    asm volatile("v_mov_b32 %0, %0" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int xb = tid / TPR, xc = tid % TPR;   // stage: row of the item, KPT-wide k part
    const int e = item / p.tiles;
    const int row0 = (item - e * p.tiles) * ROWSH;
    if (row0 >= n_rows) {    // (uniform) nothing alive in this tile; keep the prefetch chain going
      fetch_row(item + gridDim.x, tid);
      fetch_x(tid);
      fetch_bias(item + gridDim.x, tid);
      continue;
    }
    const float *st = a.stats + (size_t)e * NSTAT;
    // ---- stage: constants, per-row scales, the split input image, the first two W0 chunks --------------------------
    const f16x8 *w0e = a.w0 + (size_t)e * a.w0_stride;
    {
      constexpr int W0R = G::W0_LDS ? (2 * W0P + kWavesH - 1) / kWavesH : 1;
      f16x8 w0r[W0R];
      if constexpr (G::W0_LDS) {
#pragma unroll
        for (int u = 0; u < W0R; ++u) {
          const int j = wave + kWavesH * u;
          w0r[u] = w0e[(size_t)(j < 2 * W0P ? j : 0) * 64 + lane];
        }
      }
      bias0[tid] = bpre[0] * kLog2e;      // (the epilogues work on z log2(e): epi_stage<.., PRE>)
      bias1[tid] = bpre[1] * kLog2e;
      if (tid < O_PAD) {
        // y = A_n z + B_n with z = o + b2_n; n < out: mean = sig z + mu; out <= n < 2 out: var = exp(z + 2 log sig)
        // (models/pens/pe.py:815-835)
        float A = 0.0f, Bc = 0.0f;
        if (tid < out) { A = p.out_mu ? p.out_sig[tid] : 1.0f; Bc = p.out_mu ? p.out_mu[tid] : 0.0f; }
        else if (tid < 2 * out) { A = 1.0f; Bc = p.out_mu ? p.out_lsig2[tid - out] : 0.0f; }
        hc_a[tid] = A;
        hc_c[tid] = A * b2pre + Bc;
      }
      float xs[KPT];
      float m = 0.0f;
#pragma unroll
      for (int u = 0; u < KPT; ++u) {
        const int k = KPT * xc + u;
        // TensorStandardScaler.transform (models/pens/utils.py:156) as (x - mu) (1 / sigma): within an ulp of the division, a
        // tenth of its instructions (sixteen IEEE divisions per thread were a third of the stage)
        float x = IN_SCALE_RCP ? (xpre[u] - in_mu_l[k & 63]) * in_sig_l[k & 63] : (xpre[u] - in_mu_l[k & 63]) / in_sig_l[k & 63];
        x = (k < p.in_dim && rr_pre >= 0) ? x : 0.0f;
        xs[u] = x;
        m = fmaxf(m, fabsf(x));
      }
#pragma unroll
      for (int o = 1; o < TPR; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      const float t0 = pow2_lift(m);
      if (xc == 0) {
        const float bound1 = (st[1] * m + st[2]) * 1.001f, t1 = pow2_lift(bound1);
        const float bound2 = (st[5] * bound1 + st[6]) * 1.001f, t2 = pow2_lift(bound2);
        rowidx[xb] = rr_pre;
        r_inv0[xb] = 1.0f / (st[0] * t0);
        r_t1[xb] = t1;
        r_inv1[xb] = 1.0f / (st[4] * t1);
        r_t2[xb] = t2;
        r_inv2[xb] = 1.0f / (st[8] * t2);
      }
      if (KPT * xc < 16 * S0) {
        _Float16 q1[KPT], q2[KPT];
#pragma unroll
        for (int u = 0; u < KPT; ++u) split_h(xs[u] * t0, q1[u], q2[u]);
        _Float16 *d1 = ximg + (size_t)xb * XSTR + KPT * xc, *d2 = d1 + (size_t)ROWSH * XSTR;
        if constexpr (KPT >= 8) {
#pragma unroll
          for (int v = 0; v < KPT / 8; ++v) {
            f16x8 w1, w2;
#pragma unroll
            for (int u = 0; u < 8; ++u) { w1[u] = q1[8 * v + u]; w2[u] = q2[8 * v + u]; }
            reinterpret_cast<f16x8 *>(d1)[v] = w1; reinterpret_cast<f16x8 *>(d2)[v] = w2;
          }
        } else {
          f16x4 w1, w2;
#pragma unroll
          for (int u = 0; u < 4; ++u) { w1[u] = q1[u]; w2[u] = q2[u]; }
          *reinterpret_cast<f16x4 *>(d1) = w1; *reinterpret_cast<f16x4 *>(d2) = w2;
        }
      }
      if constexpr (G::W0_LDS) {
#pragma unroll
        for (int u = 0; u < W0R; ++u) {
          const int j = wave + kWavesH * u;
          if (j < 2 * W0P) w0buf[(size_t)j * 64 + lane] = w0r[u];     // chunks 0, 1
        }
      }
    }
    fetch_row(item + gridDim.x, tid);     // the next item's row index: lands during the layers
    H3_STAMP(0);
    H3_BARRIER();
    H3_STAMP(1);

    // ---- layers 0 + 1, fused over the 8 chunks of h1 -------------------------------------------------------------------
    const int l0_tn = wave & (NTC - 1), l0_bt = wave / NTC;     // this wave's (n-tile, row-tile) pair of every chunk
    const float inv0_l = r_inv0[32 * l0_bt + r] * kLog2e;
    const float t1_l = H3_EPI_FUSE ? pow2_rcp(r_t1[32 * l0_bt + r]) * kLog2e : r_t1[32 * l0_bt + r] * kLn2;     // (FUSE: the epilogue takes 1 / lift)
    const _Float16 *xb0 = ximg + (size_t)(32 * l0_bt + r) * XSTR + 8 * hh;
    // layer-0 operands of one 16-deep slab: W0 fragments of the chunk (LDS copy) and this wave's rows of the x image
    struct L0Ops { f16x8 a1, a2, b1, b2; };
    auto l0_read = [&](L0Ops &o, int cc, int s) {
      // W0 fragments: the chunk's LDS copy when several waves share an n-tile (RT == 4), straight from L2 otherwise (the
      // step after the last chunk computes a chunk nobody reads: any valid address will do)
      const f16x8 *wa = G::W0_LDS ? w0buf + ((size_t)(cc & 1) * W0P + (size_t)l0_tn * S0 * 2 + 2 * s) * 64 + lane
                                  : w0e + (((size_t)(cc < NCH ? cc : NCH - 1) * NTC + l0_tn) * S0 * 2 + 2 * s) * 64 + lane;
      o.a1 = wa[0]; o.a2 = wa[64];
      o.b1 = *reinterpret_cast<const f16x8 *>(xb0 + 16 * s);
      o.b2 = *reinterpret_cast<const f16x8 *>(xb0 + (size_t)ROWSH * XSTR + 16 * s);
    };
    auto l0_store = [&](const Epi4 &s, int cc, int q) {
      _Float16 *c1 = cbuf + (size_t)(cc & 1) * (CBUF_BYTES / 2) + (size_t)(32 * l0_bt + r) * CSTR + 32 * l0_tn + 4 * hh + 8 * q;
      *reinterpret_cast<uint2 *>(c1) = make_uint2(s.q1[0], s.q1[1]);
      *reinterpret_cast<uint2 *>(c1 + (size_t)ROWSH * CSTR) = make_uint2(s.q2[0], s.q2[1]);
    };
    auto l0_bias = [&](int cc, int q) {
      return *reinterpret_cast<const f32x4 *>(bias0 + cc * CK + 32 * l0_tn + 8 * q + 4 * hh);
    };

    f32x16 acc[2][RT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int bt = 0; bt < RT; ++bt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][bt][i] = 0.0f;
    const f16x8 *w1a = a.w1 + (size_t)e * a.w1_stride + (size_t)(2 * wave) * 32 * 128 + lane;   // [tile][slab 32][piece][lane]
    f16x8 A[DA][2][2];     // ring over the layer-1 slabs: [slab % DA][n-tile][piece], DA - 1 slabs of lookahead
    f16x8 Bt[3][2];        // rolling window over the (slab, row tile) sequence: current, +1, +2 (two LDS reads in flight)
    auto load_a = [&](f16x8 (&x)[2][2], int s) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f16x8 *q = w1a + ((size_t)t * 32 + s) * 128;
        x[t][0] = q[0]; x[t][1] = q[64];
      }
    };
    auto read_bt = [&](f16x8 (&x)[2], const _Float16 *img, int bt, int sl) {
      const _Float16 *q = img + (size_t)(32 * bt + r) * CSTR + 16 * sl + 8 * hh;
      x[0] = *reinterpret_cast<const f16x8 *>(q);
      x[1] = *reinterpret_cast<const f16x8 *>(q + (size_t)ROWSH * CSTR);
    };
#pragma unroll
    for (int s = 0; s < DA - 1; ++s) load_a(A[s], s);
    {   // chunk 0 of h1: nothing to overlap it with yet
      f32x16 d;
#pragma unroll
      for (int i = 0; i < 16; ++i) d[i] = 0.0f;
#pragma unroll
      for (int s = 0; s < S0; ++s) {
        L0Ops o;
        l0_read(o, 0, s);
        mm3(d, o.a1, o.a2, o.b1, o.b2);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        Epi4 es;
        const f32x4 bv = l0_bias(0, q);
        epi_all<false, true, H3_EPI_FUSE>(es, d, q, inv0_l, bv, t1_l);
        l0_store(es, 0, q);
      }
    }
    H3_BARRIER();
    H3_STAMP(2);

    // One step = the 96 layer-1 MFMAs of chunk c -- 16 (slab, row tile) positions of 6, in 8 groups of 12 -- with the
    // production of chunk c + 1 dealt out between them: a wave issues in order, so what stands between two MFMAs runs in the
    // shadow of the first.  Groups 0 .. S0-1 carry the layer-0 MFMAs of one input slab each (operands read one group
    // ahead), groups 4 .. 7 one quarter of the swish / lift / split epilogue each, one piece behind every MFMA.  The last
    // step has no chunk to produce: it is an instance of its own (H3_PEEL_LAST).
    // H3_PEEL_LAST: the last step as an instance of its own without the production of a chunk nobody reads (9 MFMAs, a
    // quarter-chunk epilogue and its LDS stores per wave)
    auto step = [&](auto LASTC, const int c) {
      constexpr bool LAST = decltype(LASTC)::value;
      f16x8 wst;
      const bool stage_w0 = c + 2 < NCH;
      const int cw = stage_w0 ? c + 2 : NCH - 1;
      const _Float16 *img = cbuf + (size_t)(c & 1) * (CBUF_BYTES / 2);
      f32x16 d;
#pragma unroll
      for (int i = 0; i < 16; ++i) d[i] = 0.0f;
      L0Ops l0;
      Epi4 es;
      f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
      read_bt(Bt[0], img, 0, 0);
      read_bt(Bt[1], img, 1 % RT, 1 / RT);
      if constexpr (!LAST) l0_read(l0, c + 1, 0);
#pragma unroll
      for (int slot = 0; slot < 8; ++slot) {
        if (!LAST && slot >= 4) bv = l0_bias(c + 1, slot - 4);
        if constexpr (G::W0_LDS && !LAST) {
          // W0 fragments of chunk c + 2 pass through four registers, one 1-KB piece at a time
          if (slot == 4) wst = w0e[((size_t)cw * W0P + wave) * 64 + lane];
          if (slot == 5 && W0P > kWavesH) {
            if (stage_w0) w0buf[((size_t)(c & 1) * W0P + wave) * 64 + lane] = wst;
            wst = w0e[((size_t)cw * W0P + (wave + kWavesH < W0P ? wave + kWavesH : wave)) * 64 + lane];
          }
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int pos = 2 * slot + i / 6;                       // position in the (slab, row tile) sequence
          const int sl = pos / RT, bt = pos % RT;
          const int t = (i / 3) % 2, term = i % 3;
          if (i % 6 == 0) {
#ifndef H3_DIAG_NOA
            if (bt == 0) {                                         // a new slab: request the one DA - 1 ahead
              const int sn = c * SLC + sl + DA - 1;
              load_a(A[(sl + DA - 1) % DA], sn < 32 ? sn : 31);
            }
#endif
#ifndef H3_DIAG_NOBT
            if (pos + 2 < 16) read_bt(Bt[(pos + 2) % 3], img, (pos + 2) % RT, (pos + 2) / RT);
#endif
            if (i == 0) __builtin_amdgcn_sched_barrier(0);
          }
          f16x8(&Ac)[2][2] = A[sl % DA];
          f32x16 &ac = acc[t][bt];
          const f16x8 &b1 = Bt[pos % 3][0], &b2 = Bt[pos % 3][1];
          if (term == 0) ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ac[t][1], b1, ac, 0, 0, 0);
          else if (term == 1) ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ac[t][0], b2, ac, 0, 0, 0);
          else ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ac[t][0], b1, ac, 0, 0, 0);
#ifndef H3_DIAG_NOL0
          if (!LAST && slot < S0 && i == 5) {          // layer-0 MFMAs of input slab `slot`, then the next slab's operands
            mm3(d, l0.a1, l0.a2, l0.b1, l0.b2);
            if (slot + 1 < S0) l0_read(l0, c + 1, slot + 1);
          }
#endif
#ifdef H3_DIAG_NOEPI_LOOP
          if (false) {
#else
          if (!LAST && slot >= 4) {
#endif
            const int q = slot - 4;
            if (i == 0) epi_stage<0, true, true, H3_EPI_FUSE>(es, d, q, inv0_l, bv, t1_l);
            if (i == 1) epi_stage<1, true, true, H3_EPI_FUSE>(es, d, q, inv0_l, bv, t1_l);
            if (i == 2) epi_stage<2, true, true, H3_EPI_FUSE>(es, d, q, inv0_l, bv, t1_l);
            if (i == 3) epi_stage<3, true, true, H3_EPI_FUSE>(es, d, q, inv0_l, bv, t1_l);
            if (i == 4) epi_stage<4, true, true, H3_EPI_FUSE>(es, d, q, inv0_l, bv, t1_l);
            if (i == 5) epi_stage<5, true, true, H3_EPI_FUSE>(es, d, q, inv0_l, bv, t1_l);
            if (i == 6) epi_stage<6, true, true, H3_EPI_FUSE>(es, d, q, inv0_l, bv, t1_l);
            if (i == 7) epi_stage<7, true, true, H3_EPI_FUSE>(es, d, q, inv0_l, bv, t1_l);
            if (i == 8) epi_stage<8, true, true, H3_EPI_FUSE>(es, d, q, inv0_l, bv, t1_l);
            if (i == 9) epi_stage<9, true, true, H3_EPI_FUSE>(es, d, q, inv0_l, bv, t1_l);
            if (i == 10) l0_store(es, c + 1, q);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      H3_STAMP(4);
      if constexpr (G::W0_LDS && !LAST) {
        if (stage_w0) {
          const int j = W0P > kWavesH ? wave + kWavesH : wave;
          if (j < W0P) w0buf[((size_t)(c & 1) * W0P + j) * 64 + lane] = wst;
        }
      }
      H3_BARRIER();
      H3_STAMP(5);
    };
#if H3_PEEL_LAST
#pragma unroll 1
    for (int c = 0; c < NCH - 1; ++c) step(std::false_type{}, c);
    step(std::true_type{}, NCH - 1);
#else
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) step(std::false_type{}, c);
#endif

    // the next item's rows, biases and output bias: in flight behind the tail
    fetch_x(tid);
    fetch_bias(item + gridDim.x, tid);
    H3_STAMP(6);
#ifdef H3_DIAG_NOTAIL
    {
      float keep = 0.0f;      // every accumulator register stays live: the loop's MFMAs must all execute
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int bt = 0; bt < RT; ++bt)
#pragma unroll
          for (int i = 0; i < 16; ++i) keep += acc[t][bt][i];
      if (keep == 123.456f) p.out0[0] = keep;
    }
    H3_BARRIER();
    continue;
#endif

    // ---- tail: h2 -> output layer -> head -> stores, one (32-row tile, pair of output tiles) unit at a time.
    // h2 never leaves the registers: an accumulator tile's rows are the next product's k index, so the wave's own 64 hidden
    // units (4 slabs) are its slice of the output layer's K (W2 images packed in the matching k order), and the eight
    // waves' partial outputs are added up through LDS in a fixed order.
    const f16x8 *w2w = a.w2 + (size_t)e * a.w2_stride + lane + (size_t)(4 * wave) * 128;   // + (tile * 32 + S) * 128 + piece * 64
    auto reduce_unit = [&](int u) {
      // wave w adds up registers 4q .. 4q+3 (q = w & 3) of output tile tt = w >> 2 of this unit over the eight waves,
      // applies the head (models/pens/pe.py:815-835) and leaves the values in the staging tile [row][column of the pair]
      const int rt = u / NPASS, pass = u % NPASS;
      const int tt = wave >> 2, q = wave & 3;
      const f32x4 *pp = pbuf + ((size_t)(u & 1) * 64 + (size_t)tt * 4 + q) * 64 + lane;
      f32x4 v = pp[0];
#pragma unroll
      for (int wv = 1; wv < kWavesH; ++wv) v += pp[(size_t)wv * 8 * 64];
      const int nl = 32 * tt + 8 * q + 4 * hh, n = 64 * pass + nl;
      const f32x4 ha = *reinterpret_cast<const f32x4 *>(hc_a + n), hc = *reinterpret_cast<const f32x4 *>(hc_c + n);
      const float inv2_l = r_inv2[32 * rt + r];
      float *sg = stg + (size_t)(u & 1) * 32 * SWS + r * SWS + nl;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float y = ha[s] * (v[s] * inv2_l) + hc[s];
        if (n + s >= out) y = __expf(y);
        sg[s] = y;
      }
    };
    auto store_unit = [&](int u) {
      // a row's means / variances are contiguous runs of `out` floats; lane = column of the pair
      const int rt = u / NPASS, pass = u % NPASS;
      if (H3_STORE8 && NPASS == 1 && (out & 1) == 0) {
        // even widths (every shipped task): 8-byte stores, two rows per instruction -- lanes 0 .. out - 1 carry row A's
        // (mean | var) as `out` float pairs, lanes out .. 2 out - 1 row B's (rows are 8-byte aligned: out 4 bytes a row)
        const int half = lane >= out ? 1 : 0, c = 2 * (lane - half * out);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int rl = wave + kWavesH * (2 * j + half);
          const int rr = rowidx[32 * rt + rl];
          const float *sp = stg + (size_t)(u & 1) * 32 * SWS + rl * SWS + c;
          const float2 y = make_float2(sp[0], sp[1]);
          if (rr >= 0 && lane < 2 * out) {
            const size_t obase = ((size_t)e * p.ld_rows + rr) * out;
            float *dst = c < out ? p.out0 + obase + c : p.out1 + obase + (c - out);
            *reinterpret_cast<float2 *>(dst) = y;
          }
        }
        return;
      }
      const int n = 64 * pass + lane;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rl = wave + kWavesH * j;
        const int rr = rowidx[32 * rt + rl];
        const float y = stg[(size_t)(u & 1) * 32 * SWS + rl * SWS + lane];
        if (rr >= 0 && n < 2 * out) {
          const size_t obase = ((size_t)e * p.ld_rows + rr) * out;
          if (n < out) p.out0[obase + n] = y;
          else p.out1[obase + (n - out)] = y;
        }
      }
    };
    if constexpr (NPASS == 1 && !H3_TAIL_RING) {
      // Two output tiles (every shipped task but Humanoid): the wave's 16 W2 fragments (its 4 slabs x 2 tiles x 2 pieces)
      // stay in 64 registers for the four row tiles of the item -- requested once, behind the first row tile's epilogue,
      // instead of once per row tile (a third of the item's L2 -> CU traffic, and an L2 round trip in front of every slab's
      // MFMAs: the chip holds its clock by power, and weight fragments from L2 are what costs most of it beside the MFMAs).
      f16x8 w2r[4][2][2];      // [slab][output tile][piece]
      static_for<0, 4>([&](auto SI) {
        constexpr int S = decltype(SI)::value;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const f16x8 *q = w2w + ((size_t)tt * 32 + S) * 128;
          w2r[S][tt][0] = q[0]; w2r[S][tt][1] = q[64];
        }
      });
      // The tail as four stages per row tile, A: swish / lift / split of h2 (VALU) -> B: output-layer MFMAs + partial sums to
      // LDS -> [barrier] -> C: sum of the eight waves' partials + head -> D: stores.  Between two barriers a wave runs
      // D(k-2), C(k-1), B(k), A(k+1).  Measured (profiles/r03/h3_variants_5.log, _6.log): each stage costs about what its
      // instructions cost alone (A 5.7 %, B 5.1 %, C + D 4.5 % of a forward); giving the two waves of a SIMD opposite orders
      // inside an interval (-DH3_TAIL_STAGGER: waves 4-7 run A(k+1) first) or dealing A(k+1) out between B(k)'s MFMAs
      // changed nothing (1.264 -> 1.269 ms), so the plain order stands.
      u32x4 bfu[2][4][2];        // [row tile & 1][slab of this wave's K slice][piece]
      auto stA = [&](auto RTI) {
        constexpr int rt = decltype(RTI)::value;
        const float inv1_l = r_inv1[32 * rt + r] * kLog2e;
        const float t2_l = H3_EPI_FUSE ? pow2_rcp(r_t2[32 * rt + r]) * kLog2e : r_t2[32 * rt + r] * kLn2;
        static_for<0, 8>([&](auto QD) {
          constexpr int quad = decltype(QD)::value, S = quad >> 1, jq = quad & 1, q = 2 * (S & 1) + jq;
          const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias1 + 64 * wave + 32 * (S >> 1) + 8 * q + 4 * hh);
          Epi4 es;
#ifdef H3_DIAG_TAIL_NOA
          es.q1[0] = __float_as_uint(acc[S >> 1][rt][4 * q] + bv[0]) & 0x3fff3fffu; es.q1[1] = __float_as_uint(acc[S >> 1][rt][4 * q + 1] + inv1_l) & 0x3fff3fffu;
          es.q2[0] = __float_as_uint(acc[S >> 1][rt][4 * q + 2] + t2_l) & 0x3fff3fffu; es.q2[1] = __float_as_uint(acc[S >> 1][rt][4 * q + 3]) & 0x3fff3fffu;
#else
          epi_all<false, true, H3_EPI_FUSE>(es, acc[S >> 1][rt], q, inv1_l, bv, t2_l);
#endif
          bfu[rt & 1][S][0][2 * jq] = es.q1[0]; bfu[rt & 1][S][0][2 * jq + 1] = es.q1[1];
          bfu[rt & 1][S][1][2 * jq] = es.q2[0]; bfu[rt & 1][S][1][2 * jq + 1] = es.q2[1];
        });
      };
      auto stB = [&](auto RTI) {
        constexpr int rt = decltype(RTI)::value;
        f32x16 o[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[tt][i] = 0.0f;
        static_for<0, 4>([&](auto SI) {
          constexpr int S = decltype(SI)::value;
          const f16x8 b1 = __builtin_bit_cast(f16x8, bfu[rt & 1][S][0]), b2 = __builtin_bit_cast(f16x8, bfu[rt & 1][S][1]);
#ifdef H3_DIAG_TAIL_NOB
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) { o[tt][S] += (float)b1[tt] + (float)w2r[S][tt][0][0]; o[tt][S + 4] += (float)b2[tt] + (float)w2r[S][tt][1][1]; }
#else
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) mm3(o[tt], w2r[S][tt][0], w2r[S][tt][1], b1, b2);
#endif
        });
        f32x4 *pw = pbuf + ((size_t)(rt & 1) * 64 + (size_t)wave * 8) * 64 + lane;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = {o[tt][4 * q], o[tt][4 * q + 1], o[tt][4 * q + 2], o[tt][4 * q + 3]};
            pw[(size_t)(tt * 4 + q) * 64] = v;
          }
      };
      stA(std::integral_constant<int, 0>{});
      static_for<0, RT + 2>([&](auto KI) {
        constexpr int k = decltype(KI)::value;      // interval k: D(k - 2), C(k - 1), B(k), A(k + 1)
        auto matrix_side = [&]() {
#ifndef H3_DIAG_TAIL_NOD
          if constexpr (k >= 2 && k - 2 < RT) store_unit(k - 2);
#endif
#ifndef H3_DIAG_TAIL_NOC
          if constexpr (k >= 1 && k - 1 < RT) reduce_unit(k - 1);
#endif
          if constexpr (k < RT) stB(std::integral_constant<int, (k < RT ? k : 0)>{});
        };
        auto valu_side = [&]() {
          if constexpr (k + 1 < RT) stA(std::integral_constant<int, (k + 1 < RT ? k + 1 : 0)>{});
        };
#ifdef H3_TAIL_STAGGER      // diagnostic (from the second interval on: in the first it spills 45 registers)
        if (k == 0 || wave < 4) { matrix_side(); valu_side(); }
        else { valu_side(); matrix_side(); }
#else
        matrix_side(); valu_side();
#endif
        H3_STAMP(7);
        if constexpr (k + 1 < RT + 2) H3_BARRIER();
        H3_STAMP(8);
      });
    } else {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f16x8 wf[2][2][2];       // [ping-pong][output tile of the pair][piece]
      auto load_w2 = [&](f16x8 (&x)[2][2], int pass, int S) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const f16x8 *q = w2w + ((size_t)(2 * pass + tt) * 32 + S) * 128;
          x[tt][0] = q[0]; x[tt][1] = q[64];
        }
      };
      load_w2(wf[0], 0, 0);
      // this row tile of h2: swish, lift, split -- straight into B fragments
      f16x8 bf[4][2];
      {
        u32x4 bfu[4][2];
        const float inv1_l = r_inv1[32 * rt + r] * kLog2e;
        const float t2_l = H3_EPI_FUSE ? pow2_rcp(r_t2[32 * rt + r]) * kLog2e : r_t2[32 * rt + r] * kLn2;
#pragma unroll
        for (int S = 0; S < 4; ++S)
#pragma unroll
          for (int jq = 0; jq < 2; ++jq) {
            const int q = 2 * (S & 1) + jq;
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias1 + 64 * wave + 32 * (S >> 1) + 8 * q + 4 * hh);
            Epi4 es;
            epi_all<false, true, H3_EPI_FUSE>(es, acc[S >> 1][rt], q, inv1_l, bv, t2_l);
            bfu[S][0][2 * jq] = es.q1[0]; bfu[S][0][2 * jq + 1] = es.q1[1];
            bfu[S][1][2 * jq] = es.q2[0]; bfu[S][1][2 * jq + 1] = es.q2[1];
          }
#pragma unroll
        for (int S = 0; S < 4; ++S)
#pragma unroll
          for (int pc = 0; pc < 2; ++pc) bf[S][pc] = __builtin_bit_cast(f16x8, bfu[S][pc]);
      }
#pragma unroll
      for (int pass = 0; pass < NPASS; ++pass) {
        const int u = rt * NPASS + pass;
        f32x16 o[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[tt][i] = 0.0f;
#pragma unroll
        for (int S = 0; S < 4; ++S) {
          const int nS = (S + 1) & 3, npass = (S == 3) ? pass + 1 : pass;
          if (S < 3 || pass + 1 < NPASS) load_w2(wf[(S & 1) ^ 1], npass, nS);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) mm3(o[tt], wf[S & 1][tt][0], wf[S & 1][tt][1], bf[S][0], bf[S][1]);
        }
        f32x4 *pw = pbuf + ((size_t)(u & 1) * 64 + (size_t)wave * 8) * 64 + lane;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = {o[tt][4 * q], o[tt][4 * q + 1], o[tt][4 * q + 2], o[tt][4 * q + 3]};
            pw[(size_t)(tt * 4 + q) * 64] = v;
          }
        H3_STAMP(7);
        H3_BARRIER();       // unit u's partials are complete; unit u - 1's staging tile too
        H3_STAMP(8);
        if (u > 0) store_unit(u - 1);
        reduce_unit(u);
        H3_STAMP(9);
      }
    }
    H3_BARRIER();
    store_unit(NUNIT - 1);
    }
    H3_STAMP(10);
    H3_BARRIER();     // the LDS regions are rewritten by the next item's stage
    H3_STAMP(11);
  }  // persistent item loop
#ifdef CMBPO_STAMPS
  if (p.stamps && threadIdx.x == 0) {
    for (int k = 0; k < 12; ++k) p.stamps[(size_t)blockIdx.x * 16 + k] = t_acc[k];
    p.stamps[(size_t)blockIdx.x * 16 + 12] = __builtin_amdgcn_s_memrealtime() - t_rt0;
  }
#endif
}

}  // namespace

// ---- host side -------------------------------------------------------------------------------------------------------
// (re)builds the statistics and the two f16 images from the handle's fp32 packs when they changed since the last build
static int ensure_h3(cmbpo_mlp *m, hipStream_t s) {
  const int H = m->hidden, E = m->ensemble;
  const int S0 = m->h3_s0, OTP = m->h3_otp;
  const int slabs[3] = {S0, H / 16, H / 16};
  const int tiles[3] = {H / 32, H / 32, OTP};
  if (m->d_h3 == nullptr) {
    size_t off = 0;
    for (int l = 0; l < 3; ++l) {
      m->h3_stride[l] = (size_t)tiles[l] * slabs[l] * 2 * 64;
      m->h3_off[l] = off;
      off += m->h3_stride[l] * E;
    }
    m->h3_stats_off = off;    // in 16-B units
    const size_t bytes = off * 16 + (size_t)E * NSTAT * sizeof(float);
    if (hipMalloc(&m->d_h3, bytes) != hipSuccess) {
      (void)hipGetLastError();
      m->d_h3 = nullptr;
      cmbpo_set_error("ens_h3: hipMalloc of the f16 weight images failed");
      return CMBPO_ENOMEM;
    }
    m->h3_version = ~0ul;
  }
  if (m->h3_version == m->pack_version) return CMBPO_OK;
  float *stats = reinterpret_cast<float *>(reinterpret_cast<char *>(m->d_h3) + m->h3_stats_off * 16);
  hipLaunchKernelGGL(h3_stats_kernel, dim3(E, 3), dim3(kThreadsH), 0, s, m->d_blob, m->off_wp0, m->off_wp1, m->off_wp2,
                     m->off_b0, m->off_b1, m->off_b2, m->in_pad / 8, m->o_tiles, H, stats);
  const size_t src_off[3] = {m->off_wp0, m->off_wp1, m->off_wp2};
  const int kg[3] = {m->in_pad / 8, H / 8, H / 8};
  const int src_tiles[3] = {H / 32, H / 32, m->o_tiles};
  for (int l = 0; l < 3; ++l) {
    const size_t src_stride = (size_t)src_tiles[l] * kg[l] * 256;
    const long total = (long)tiles[l] * slabs[l] * 64 * E;
    hipLaunchKernelGGL(h3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, m->d_blob + src_off[l], src_stride,
                       kg[l], src_tiles[l], reinterpret_cast<f16x8 *>(m->d_h3) + m->h3_off[l], m->h3_stride[l], tiles[l], slabs[l],
                       E, stats, l, l == 2 ? 1 : 0);
  }
  CMBPO_HIP_CHECK(hipGetLastError());
  m->h3_version = m->pack_version;
  return CMBPO_OK;
}

// shared with critic_f16.hip
void cmbpo_internal_f16_stats(const cmbpo_mlp *m, float *stats, hipStream_t s) {
  hipLaunchKernelGGL(h3_stats_kernel, dim3(m->ensemble, 3), dim3(kThreadsH), 0, s, m->d_blob, m->off_wp0, m->off_wp1, m->off_wp2,
                     m->off_b0, m->off_b1, m->off_b2, m->in_pad / 8, m->o_tiles, m->hidden, stats);
}
void cmbpo_internal_f16_pack(const cmbpo_mlp *m, int layer, void *dst, size_t dst_stride, int n_tiles, int slabs, int perm,
                             const float *stats, hipStream_t s) {
  const int H = m->hidden;
  const size_t src_off[3] = {m->off_wp0, m->off_wp1, m->off_wp2};
  const int kg[3] = {m->in_pad / 8, H / 8, H / 8};
  const int src_tiles[3] = {H / 32, H / 32, m->o_tiles};
  const size_t src_stride = (size_t)src_tiles[layer] * kg[layer] * 256;
  const long total = (long)n_tiles * slabs * 64 * m->ensemble;
  hipLaunchKernelGGL(h3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, m->d_blob + src_off[layer], src_stride,
                     kg[layer], src_tiles[layer], reinterpret_cast<f16x8 *>(dst), dst_stride, n_tiles, slabs, m->ensemble, stats, layer,
                     perm);
}

static int g_h3_rt = getenv("CMBPO_ENS_H3_RT") ? atoi(getenv("CMBPO_ENS_H3_RT")) : 0;   // 0: by row count; 1 / 2 / 4 forces it
extern "C" int cmbpo_set_ens_f16_row_tiles(int rt) {
  CMBPO_REQUIRE(rt == 0 || rt == 1 || rt == 2 || rt == 4, "cmbpo_set_ens_f16_row_tiles: 0 (by row count), 1, 2 or 4");
  g_h3_rt = rt;
  return CMBPO_OK;
}

bool cmbpo_internal_h3_eligible(const cmbpo_mlp *m) {
  return m->head == CMBPO_HEAD_PROB && m->hidden == 512 && m->act == CMBPO_ACT_SWISH && m->o_tiles <= 4 && m->in_pad <= 64 &&
         2 * m->out_dim == m->o_width;
}

int cmbpo_internal_launch_h3(cmbpo_mlp *m, MlpKernelArgs &a, hipStream_t s) {
  if (m->h3_s0 == 0) {
    const int s0 = (m->in_pad + 15) / 16;
    m->h3_s0 = s0 < 2 ? 2 : s0;
    m->h3_otp = m->o_tiles <= 2 ? 2 : 4;
  }
  if (int rc = ensure_h3(m, s)) return rc;
  H3Args k{};
  k.m = a;
  const f16x8 *base = reinterpret_cast<const f16x8 *>(m->d_h3);
  k.w0 = base + m->h3_off[0]; k.w1 = base + m->h3_off[1]; k.w2 = base + m->h3_off[2];
  k.w0_stride = m->h3_stride[0]; k.w1_stride = m->h3_stride[1]; k.w2_stride = m->h3_stride[2];
  k.stats = reinterpret_cast<const float *>(reinterpret_cast<const char *>(m->d_h3) + m->h3_stats_off * 16);
  static int n_cu = 0;
  if (n_cu == 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  // Rows per item: the cheapest plan by a two-number model per item size -- the first round of a launch costs an item's
  // latency, every further round its steady-state time (tenths of a microsecond, measured by forcing each size over 17 row
  // counts, profiles/r03/h3_split_plan_sweep.log: 32 / 64 / 128 rows = 22.0 / 31.0 / 55.0 us then 17.5 / 30.0 / 54.0 us per
  // round at two output tiles; 25.5 / 36.0 / 72.0 then 19.8 / 34.3 / 70.5 at four -- Humanoid's 2 x 46 outputs take two
  // passes over the output layer, and 64-row items are its better main size).  The full rounds of one item size may be
  // followed by the leftovers at a SMALLER size in a launch of their own: a last round that would leave most CUs idle
  // becomes one or two rounds of short items on all of them (100 000 rows x 7 members = 5474 items of 128 rows on 256 CUs:
  // 21 full rounds + 98 items -> 196 items of 64 rows; 10 000 rows: 2 rounds of 128-row items + 41 items -> 164 of 32 rows,
  // 135 us by this model -- 134.5 measured -- against 147 for five rounds of 64-row items).  An item is (member, row tile) in
  // member-major order in every list, so a suffix of one list is a suffix of the others; a row's arithmetic does not depend
  // on the item it travels in (tests: bitwise against the forced sizes).
  const int E = m->ensemble;
  static const int split_tail = getenv("CMBPO_ENS_H3_SPLIT_TAIL") ? atoi(getenv("CMBPO_ENS_H3_SPLIT_TAIL")) : 1;
  static const int split_gap = getenv("CMBPO_ENS_H3_SPLIT_GAP") ? 10 * atoi(getenv("CMBPO_ENS_H3_SPLIT_GAP")) : 40;   // launch boundary
  int RT = g_h3_rt, RT_tail = 0, full = 0, tail_start = 0;
  if (RT == 0) {
    const int first_two[3] = {220, 310, 550}, steady_two[3] = {175, 300, 540};
    const int first_four[3] = {255, 360, 720}, steady_four[3] = {198, 343, 705};
    const int *first = m->h3_otp > 2 ? first_four : first_two, *steady = m->h3_otp > 2 ? steady_four : steady_two;
    const int rts[3] = {1, 2, 4};
    auto cost_of = [&](int i, long rounds) -> long { return rounds <= 0 ? 0 : first[i] + (rounds - 1) * steady[i]; };
    long best = -1;
    for (int i = 0; i < 3; ++i) {
      const long cost = cost_of(i, cmbpo_ceil_div(cmbpo_ceil_div(a.n_rows, 32 * rts[i]) * E, n_cu));
      if (best < 0 || cost <= best) { best = cost; RT = rts[i]; }
    }
    if (split_tail)
      for (int i = 1; i < 3; ++i) {
        const int tiles_m = cmbpo_ceil_div(a.n_rows, 32 * rts[i]), n_m = tiles_m * E;
        const int fl = n_m / n_cu * n_cu, left = n_m - fl;
        if (fl == 0 || left == 0) continue;
        const int e0 = fl / tiles_m, t0 = fl - e0 * tiles_m;
        for (int j = 0; j < i; ++j) {
          const int tiles_t = cmbpo_ceil_div(a.n_rows, 32 * rts[j]);
          const int start = e0 * tiles_t + (rts[i] / rts[j]) * t0, n_tail = tiles_t * E - start;
          const long cost = cost_of(i, fl / n_cu) + cost_of(j, cmbpo_ceil_div(n_tail, n_cu)) + split_gap;
          if (cost < best) { best = cost; RT = rts[i]; RT_tail = rts[j]; full = fl; tail_start = start; }
        }
      }
  }
  const int S0 = m->h3_s0, OTP = m->h3_otp;
  static bool attr_done[5][5][5] = {};
  // one launch: items [item0, end) of the (member-major) item list at rt 32-row tiles per item; end < 0: all of it
  auto launch = [&](int rt, int item0, int end) -> int {
    const int tiles = cmbpo_ceil_div(a.n_rows, 32 * rt);
    k.m.tiles = tiles;
    k.m.n_items = end < 0 ? tiles * E : end;     // (the kernel's loop bound)
    k.item0 = item0;
    const size_t lds = (size_t)lds_bytes(S0, rt);
    CMBPO_REQUIRE(lds <= 160 * 1024, "ens_h3: LDS budget exceeded (%zu B)", lds);
    const int n_here = k.m.n_items - item0;
    const int grid = n_here < n_cu ? n_here : n_cu;
#define CMBPO_H3_CASE(S0_, OTP_, RT_)                                                                                  \
  if (S0 == S0_ && OTP == OTP_ && rt == RT_) {                                                                         \
    if (!attr_done[S0_][OTP_][RT_]) {                                                                                  \
      CMBPO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ens_h3_kernel<S0_, OTP_, RT_>),               \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(S0_, RT_)));      \
      attr_done[S0_][OTP_][RT_] = true;                                                                                \
    }                                                                                                                  \
    hipLaunchKernelGGL((ens_h3_kernel<S0_, OTP_, RT_>), dim3(grid), dim3(kThreadsH), lds, s, k);                      \
  }
#define CMBPO_H3_CASES(RT_)                                                                                            \
  CMBPO_H3_CASE(2, 2, RT_) CMBPO_H3_CASE(3, 2, RT_) CMBPO_H3_CASE(4, 2, RT_)                                           \
  CMBPO_H3_CASE(2, 4, RT_) CMBPO_H3_CASE(3, 4, RT_) CMBPO_H3_CASE(4, 4, RT_)
    CMBPO_H3_CASES(4) CMBPO_H3_CASES(2) CMBPO_H3_CASES(1)
#undef CMBPO_H3_CASES
#undef CMBPO_H3_CASE
    CMBPO_HIP_CHECK(hipGetLastError());
    return CMBPO_OK;
  };
  if (RT_tail) {
    if (int rc = launch(RT, 0, full)) return rc;
    return launch(RT_tail, tail_start, -1);
  }
  return launch(RT, 0, -1);
}

