// Ensemble forward (HEAD_PROB, 512-wide, swish) with the fp32 GEMMs carried by the bf16 matrix cores.
//
// Same function as ens_mlp_kernel<512, *, swish, prob> (models/pens/pe.py:688-697,789-838, fc.py:74-95): one item =
// 64 rows of one member through  x -> swish(x W0 + b0) -> swish(. W1 + b1) -> . W2 + b2 -> (mean, var).
// An fp32 product a.b is computed as sum_{i+j<=4} a_i b_j with a = a1 + a2 + a3, a_i the successive bf16 roundings of
// the remainder (3 x 8 = 24 mantissa bits): six v_mfma_f32_32x32x16_bf16 with fp32 accumulation, every partial
// product exact, the dropped terms <= 2^-24 |ab|.  Measured (tools/split_bf16_probe.hip): error 6.2e-7 of sum|a_k b_k|
// at K = 512 (the fp32 MFMA chain: 7.6e-7), 2.08x the fp32-MFMA rate.
//
// Layout: weights are split once per weight change into three bf16 images in MFMA fragment order
//   [n-tile 32][k-slab 16][image 3][lane 64][8 bf16]   (lane (r, h) holds W[n = r][k = 16 s + 8 h + j]),
// streamed from L2 (1.9 MB per member); 64 rows per item halve that stream per row against 32 (L2 -> CU bandwidth is
// the next bound after the matrix pipe).  Activations stay fp32 in LDS in the row layout [row][516] (conflict-free
// 16-B reads: 516 / 4 odd) and are split into their three images when a wave reads its B fragment -- three images of a
// 512-wide layer for 64 rows do not fit LDS, and the split (converts / subtracts) issues in the shadow of the MFMAs.
// 4 waves, one per SIMD: wave w owns hidden n-tiles 4w .. 4w+3 for both 32-row halves (48 MFMAs per 16 split values per
// lane: the split issues in the MFMA gaps of the same wave).  The output layer takes its B operand straight from the
// accumulators (an accumulator tile's rows are the next product's k index, so the wave's own h2 slice is split once,
// in registers, with the matching permutation baked into the W2 images), K split over the waves, partial outputs
// reduced through LDS -- the structure of ens_mlp_kernel's output layer.
#include "common.h"
#include "ens_mlp_internal.h"

#include <stdlib.h>

namespace {

constexpr int kThreadsS = 256;
constexpr int HIDS = 512;
constexpr int HS = HIDS + 4;      // row stride of the activation image

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Split3 {
  bf16x8 p1, p2, p3;
};

__device__ __forceinline__ Split3 split8(const f32x4 lo, const f32x4 hi) {
  Split3 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float a = j < 4 ? lo[j] : hi[j - 4];
    const __bf16 q1 = (__bf16)a;
    const float r1 = a - (float)q1;      // exact
    const __bf16 q2 = (__bf16)r1;
    const float r2 = r1 - (float)q2;     // exact
    o.p1[j] = q1; o.p2[j] = q2; o.p3[j] = (__bf16)r2;
  }
  return o;
}

// six-term product, smallest terms first
__device__ __forceinline__ void mfma6(f32x16 &acc, const Split3 &a, const Split3 &b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p3, b.p1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p1, b.p3, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p2, b.p2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p2, b.p1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p1, b.p2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p1, b.p1, acc, 0, 0, 0);
}

__device__ __forceinline__ float swishf(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ---- fp32 pack [n-tile][k-group 8][lane][4] -> three bf16 images [n-tile][k-slab 16][image][lane][8] ---------------
// perm == 0: lane (r, h), element j of slab s holds W[n = 32 tile + r][k = 16 s + 8 h + j].
// perm == 1 (output layer, B operand = accumulator registers): slab S = 8 w + 2 t + half covers the hidden units
// wave w holds in registers 8 half .. 8 half + 7 of its tile t: k = 128 w + 32 t + (i & 3) + 8 (i >> 2) + 4 h, i = 8 half + j.
__global__ void split_pack_kernel(const float *src, size_t src_stride, int kg, bf16x8 *dst, size_t dst_stride, int n_tiles,
                                  int slabs, int members, int perm) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per = (long)n_tiles * slabs * 64;
  if (idx >= per * members) return;
  const int e = (int)(idx / per);
  const int rem = (int)(idx - (long)e * per);
  const int lane = rem & 63, s = (rem >> 6) % slabs, tile = (rem >> 6) / slabs;
  const int r = lane & 31, h = lane >> 5;
  const float *sp = src + (size_t)e * src_stride;
  f32x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int k = 16 * s + 8 * h + j;
    if (perm) {
      const int i = 8 * (s & 1) + j;
      k = 128 * (s >> 3) + 32 * ((s >> 1) & 3) + (i & 3) + 8 * (i >> 2) + 4 * h;
    }
    float v = 0.0f;
    if ((k >> 3) < kg) v = sp[(((size_t)tile * kg + (k >> 3)) * 64 + ((k >> 2) & 1) * 32 + r) * 4 + (k & 3)];
    if (j < 4) lo[j] = v; else hi[j - 4] = v;
  }
  const Split3 sp3 = split8(lo, hi);
  bf16x8 *d = dst + (size_t)e * dst_stride + ((size_t)tile * slabs + s) * 3 * 64 + lane;
  d[0] = sp3.p1; d[64] = sp3.p2; d[128] = sp3.p3;
}

struct SplitArgs {
  MlpKernelArgs m;
  const bf16x8 *sp0, *sp1, *sp2;
  size_t sp0_stride, sp1_stride, sp2_stride;   // per member, in 16-B units
  int slabs0;                                  // k-slabs of the input layer (in_pad rounded up to 16)
};

constexpr int NTS = 4;   // hidden n-tiles per wave

struct RawB {
  f32x4 x0, x1, y0, y1;
};

// one hidden layer: acc[t][bt] += W[n-tiles 4w + t] . B[rows 32 bt ..] over `slabs` k-slabs; B from the fp32 row image.
// Software pipeline, one slab deep, pinned with scheduling barriers (left alone the compiler sinks the weight loads
// into the iteration that consumes them and every slab waits out an L2 round trip): while the 48 MFMAs of slab s issue,
// the weight fragments of slab s + 1 are in flight, the B rows of slab s + 1 are split and those of s + 2 are read.
template <int BT>
__device__ __forceinline__ void split_layer(f32x16 (&acc)[NTS][BT], const bf16x8 *wp, int slabs, const float *img, int stride,
                                            int wave, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const bf16x8 *wa = wp + (size_t)(NTS * wave) * slabs * 3 * 64 + lane;
  const size_t tstep = (size_t)slabs * 3 * 64;
  const float *b0 = img + (size_t)r * stride + 8 * h, *b1 = b0 + (size_t)32 * stride;
  auto load_a = [&](Split3 (&a)[NTS], int s) {
#pragma unroll
    for (int t = 0; t < NTS; ++t) {
      const bf16x8 *q = wa + t * tstep + (size_t)s * 192;
      a[t].p1 = q[0]; a[t].p2 = q[64]; a[t].p3 = q[128];
    }
  };
  auto read_b = [&](int s) {
    RawB v;
    v.x0 = *reinterpret_cast<const f32x4 *>(b0 + 16 * s); v.x1 = *reinterpret_cast<const f32x4 *>(b0 + 16 * s + 4);
    if constexpr (BT == 2) {
      v.y0 = *reinterpret_cast<const f32x4 *>(b1 + 16 * s); v.y1 = *reinterpret_cast<const f32x4 *>(b1 + 16 * s + 4);
    } else {
      v.y0 = v.x0; v.y1 = v.x1;
    }
    return v;
  };
  // two register sets in ping-pong (a copy "current = next" per slab is 32 moves in front of the MFMAs)
  Split3 A[2][NTS], BX[2], BY[2];
  RawB R[2];
  load_a(A[0], 0);
  R[0] = read_b(0);
  BX[0] = split8(R[0].x0, R[0].x1);
  if constexpr (BT == 2) BY[0] = split8(R[0].y0, R[0].y1);
  R[1] = read_b(slabs > 1 ? 1 : 0);
  auto step = [&](int s, Split3 (&a_use)[NTS], Split3 (&a_ld)[NTS], const Split3 &bx_use, const Split3 &by_use, Split3 &bx_mk,
                  Split3 &by_mk, const RawB &raw_use, RawB &raw_ld) {
    const int s1 = (s + 1 < slabs) ? s + 1 : s, s2 = (s + 2 < slabs) ? s + 2 : s1;
    __builtin_amdgcn_sched_barrier(0);
    load_a(a_ld, s1);
    const RawB nn = read_b(s2);
    bx_mk = split8(raw_use.x0, raw_use.x1);
    if constexpr (BT == 2) by_mk = split8(raw_use.y0, raw_use.y1);
#pragma unroll
    for (int t = 0; t < NTS; ++t) {
      mfma6(acc[t][0], a_use[t], bx_use);
      if constexpr (BT == 2) mfma6(acc[t][1], a_use[t], by_use);
    }
    // A wave issues in order: whatever stands between two MFMAs in the instruction stream runs in the shadow of the
    // first (its 32 cycles in the pipe, 8 of them holding the issue port), whatever stands in front of a run of MFMAs
    // delays all of them.  So the slab's other work is dealt out one gap at a time: a weight-fragment load (~27 cycles
    // of issue) every fourth gap, the four LDS reads in the first gaps, three of the split's VALU instructions in
    // every other gap.
#pragma unroll
    for (int i = 0; i < 6 * BT * NTS; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // 1 MFMA
      if ((i & 3) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // 1 VMEM read
      else __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                  // 3 VALU
      if (i < 8 && (i & 1)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // 1 DS read
    }
    __builtin_amdgcn_sched_barrier(0);
    raw_ld = nn;
  };
  int s = 0;
  for (; s + 1 < slabs; s += 2) {
    step(s, A[0], A[1], BX[0], BY[0], BX[1], BY[1], R[1], R[0]);
    step(s + 1, A[1], A[0], BX[1], BY[1], BX[0], BY[0], R[0], R[1]);
  }
  if (s < slabs) step(s, A[0], A[1], BX[0], BY[0], BX[1], BY[1], R[1], R[0]);
}

template <int OT, int BT>   // output n-tiles (2 out_dim <= 32 OT); 32-row halves per item: 2 (64 rows), or 1 when 64-row items
                            // would leave CUs idle (small rollout batches: an item's latency is the step's)
__global__ __launch_bounds__(kThreadsS, (BT == 1 ? 2 : 1)) void ens_split_kernel(const SplitArgs a) {
  constexpr int ROWS = 32 * BT, RED_LDS = ROWS + 1, TPR = kThreadsS / ROWS;   // TPR threads stage one row
  const MlpKernelArgs &p = a.m;
  extern __shared__ f32x4 smem4[];
  float *hbuf = reinterpret_cast<float *>(smem4);          // [ROWS][HS]; later the partial outputs [4][64][RED_LDS]
  const int kpad0 = a.slabs0 * 16, XS = kpad0 + 4;
  float *xs = hbuf + ROWS * HS;                            // [ROWS][XS] scaled input, zero padded
  float *bias_l = xs + ROWS * XS;                          // [HID | HID | 64]
  float *oconst = bias_l + 2 * HIDS + OT * 32;             // [sig | 2 log sig | mu] x out_dim
  int *rows = reinterpret_cast<int *>(oconst + 3 * p.out_dim);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n_rows = p.n_rows_dev ? *p.n_rows_dev : p.n_rows;
  constexpr int o_pad = OT * 32;
  if (tid < p.out_dim) {   // output scaler constants, once per workgroup (models/pens/utils.py:167,187)
    oconst[tid] = p.out_mu ? p.out_sig[tid] : 1.0f;
    oconst[p.out_dim + tid] = p.out_mu ? p.out_lsig2[tid] : 0.0f;
    oconst[2 * p.out_dim + tid] = p.out_mu ? p.out_mu[tid] : 0.0f;
  }
  // Persistent workgroups (one per CU: 150 KB of LDS): with a single workgroup per CU nothing else covers an item's
  // prologue, so the NEXT item's input rows are gathered into registers behind the 512 x 512 layer of the current one
  // (row index first, then the dependent row gather), and the dispatcher's turnaround between items disappears.
  constexpr int XPRE = 64 / TPR;      // >= kpad0 / TPR (in_pad <= 64)
  const int xc = tid % TPR, xb = tid / TPR;
  float xpre[XPRE];
  int rr_pre = -1;
  auto item_rows = [&](int it, int &e_out, int &row0_out) {
    e_out = it / p.tiles;
    row0_out = (it - e_out * p.tiles) * ROWS;
  };
  auto fetch_row = [&](int it) {
    int e2, r2;
    item_rows(it, e2, r2);
    const int rr = r2 + xb;
    rr_pre = (it < p.n_items && rr < n_rows) ? (p.row_idx ? p.row_idx[rr] : rr) : -1;
  };
  auto fetch_x = [&]() {
#pragma unroll
    for (int u = 0; u < XPRE; ++u) {
      const int k = xc + TPR * u;
      float x = 0.0f;
      if (k < p.in_dim && rr_pre >= 0) {
        x = (k < p.obs_dim) ? p.obs[(size_t)rr_pre * p.obs_dim + k] : p.act[(size_t)rr_pre * p.act_dim + (k - p.obs_dim)];
        if (p.in_mu) x = (x - p.in_mu[k]) / p.in_sig[k];   // TensorStandardScaler.transform, models/pens/utils.py:156
      }
      xpre[u] = x;
    }
  };
  // this member's biases travel the same way (1088 floats: five registers per thread)
  constexpr int BPRE = (2 * HIDS + o_pad + kThreadsS - 1) / kThreadsS;
  float bpre[BPRE];
  auto fetch_bias = [&](int it) {
    const int e2 = (it < p.n_items) ? it / p.tiles : 0;
#pragma unroll
    for (int u = 0; u < BPRE; ++u) {
      const int i = tid + u * kThreadsS;
      bpre[u] = 0.0f;
      if (i < 2 * HIDS + o_pad)
        bpre[u] = (i < HIDS) ? p.b0[(size_t)e2 * HIDS + i]
                             : (i < 2 * HIDS) ? p.b1[(size_t)e2 * HIDS + (i - HIDS)] : p.b2[(size_t)e2 * o_pad + (i - 2 * HIDS)];
    }
  };
  fetch_row(blockIdx.x);
  fetch_x();
  fetch_bias(blockIdx.x);

  for (int item = blockIdx.x; item < p.n_items; item += gridDim.x) {
  int e, row0;
  item_rows(item, e, row0);
  if (row0 >= n_rows) {    // (uniform) nothing alive in this tile; keep the prefetch chain going
    fetch_row(item + gridDim.x);
    fetch_x();
    fetch_bias(item + gridDim.x);
    continue;
  }
  // ---- biases of this member, input tile from the prefetch registers ------------------------------------------------
#pragma unroll
  for (int u = 0; u < BPRE; ++u) {
    const int i = tid + u * kThreadsS;
    if (i < 2 * HIDS + o_pad) bias_l[i] = bpre[u];
  }
  if (xc == 0) rows[xb] = rr_pre;
#pragma unroll
  for (int u = 0; u < XPRE; ++u) {
    const int k = xc + TPR * u;
    if (k < kpad0) xs[xb * XS + k] = xpre[u];
  }
  fetch_row(item + gridDim.x);     // the next item's row index: lands during layer 0
  __syncthreads();

  f32x16 acc[NTS][BT];
  auto init_bias = [&](const float *bias) {
#pragma unroll
    for (int t = 0; t < NTS; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + (NTS * wave + t) * 32 + 8 * q + 4 * h);
#pragma unroll
        for (int bt = 0; bt < BT; ++bt)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[t][bt][4 * q + s] = bv[s];
      }
  };
  auto swish_acc = [&]() {
#pragma unroll
    for (int t = 0; t < NTS; ++t)
#pragma unroll
      for (int bt = 0; bt < BT; ++bt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][bt][i] = swishf(acc[t][bt][i]);
  };

  // ---- layer 0: in -> 512 -----------------------------------------------------------------------------------
  init_bias(bias_l);
  split_layer<BT>(acc, a.sp0 + (size_t)e * a.sp0_stride, a.slabs0, xs, XS, wave, lane);
  swish_acc();
#pragma unroll
  for (int t = 0; t < NTS; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int bt = 0; bt < BT; ++bt) {
        f32x4 v;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = acc[t][bt][4 * q + s];
        *reinterpret_cast<f32x4 *>(hbuf + (size_t)(32 * bt + r) * HS + (NTS * wave + t) * 32 + 8 * q + 4 * h) = v;
      }
  __syncthreads();
  fetch_x();               // the next item's rows and biases: in flight behind layer 1
  fetch_bias(item + gridDim.x);
  // ---- layer 1: 512 -> 512 ----------------------------------------------------------------------------------
  init_bias(bias_l + HIDS);
  split_layer<BT>(acc, a.sp1 + (size_t)e * a.sp1_stride, HIDS / 16, hbuf, HS, wave, lane);
  swish_acc();
  // ---- layer 2: 512 -> 2 out (64 padded), K split over the waves, B operand = this wave's h2 registers ----------
  f32x16 o[OT][BT];
#pragma unroll
  for (int t2 = 0; t2 < OT; ++t2)
#pragma unroll
    for (int bt = 0; bt < BT; ++bt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[t2][bt][i] = 0.0f;
  {
    const bf16x8 *w2 = a.sp2 + (size_t)e * a.sp2_stride + lane;
    constexpr int S2 = HIDS / 16;   // slabs per output tile
#pragma unroll
    for (int t = 0; t < NTS; ++t)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int S = 8 * wave + 2 * t + half;
        Split3 wa[OT];
#pragma unroll
        for (int t2 = 0; t2 < OT; ++t2) {
          const bf16x8 *q = w2 + ((size_t)t2 * S2 + S) * 192;
          wa[t2].p1 = q[0]; wa[t2].p2 = q[64]; wa[t2].p3 = q[128];
        }
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
          f32x4 lo, hi;
#pragma unroll
          for (int j = 0; j < 4; ++j) { lo[j] = acc[t][bt][8 * half + j]; hi[j] = acc[t][bt][8 * half + 4 + j]; }
          const Split3 b = split8(lo, hi);
#pragma unroll
          for (int t2 = 0; t2 < OT; ++t2) mfma6(o[t2][bt], wa[t2], b);
        }
      }
  }
  __syncthreads();         // every wave has finished reading h1: hbuf becomes the partial-output image
  float *red = hbuf;       // [wave][n o_pad][RED_LDS] (133 KB at four output tiles: runs over into the dead input image)
#pragma unroll
  for (int t2 = 0; t2 < OT; ++t2)
#pragma unroll
    for (int bt = 0; bt < BT; ++bt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = t2 * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        red[(wave * o_pad + n) * RED_LDS + 32 * bt + r] = o[t2][bt][i];
      }
  __syncthreads();
  // ---- head: mean = sig * o + mu ; var = exp(2 log sig + o')   (models/pens/pe.py:815-835) -----------------------
  auto reduced = [&](int b, int n) -> float {
    float v = red[(0 * o_pad + n) * RED_LDS + b];
    v += red[(1 * o_pad + n) * RED_LDS + b];
    v += red[(2 * o_pad + n) * RED_LDS + b];
    v += red[(3 * o_pad + n) * RED_LDS + b];
    return v + bias_l[2 * HIDS + n];
  };
  // thread (row b = tid / TPR, q = tid % TPR) writes outputs n = q, q + TPR, ...: the row index is read once and the
  // iterations are independent, so their LDS reads are in flight together (one wave per SIMD: nobody else hides them)
  const int out = p.out_dim;
  {
    const int b = tid / TPR, q = tid % TPR;
    const int rr = rows[b];
    const size_t obase = ((size_t)e * p.ld_rows + (rr >= 0 ? rr : 0)) * out;
#pragma unroll
    for (int u = 0; u < 16 * OT / TPR; ++u) {    // out <= 16 OT
      const int n = q + TPR * u;
      if (n < out) {
        const float m = oconst[n] * reduced(b, n) + oconst[2 * out + n];
        const float lv = oconst[out + n] + reduced(b, out + n);
        if (rr >= 0) {
          p.out0[obase + n] = m;
          p.out1[obase + n] = __expf(lv);
        }
      }
    }
  }
  __syncthreads();         // the next item overwrites rows / bias / the images
  }  // persistent item loop
}

// ---------------------------------------------------------------------------------------------------------------------
// The critics (HEAD_DETMEAN, 128 hidden units, one output; models/pens/pe.py:338-343,648-669) on the same matrix path.
// At 128 hidden units a wave that owned one n-tile would split 8 values per 6 MFMAs -- VALU-bound -- so here waves own
// ROWS: wave w of the 2-wave workgroup carries rows 32 w .. 32 w + 31 of the 64-row item through all 128 hidden units
// (4 n-tiles: 24 MFMAs per 8 split values, the ratio of the 512-wide kernel).  A wave's rows are private to it from the
// input image to the output, so the members' layers follow each other without a workgroup barrier; the single output
// column is a VALU dot product on the h2 registers, the members' values are averaged in a register.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kThreadsC = 128;
constexpr int HC = 128;
constexpr int HSC = HC + 4;

// first weight fragments (slab 0) of a layer: the caller requests them ahead of the previous layer's epilogue, so the short
// layers of a 128-wide member (2 and 8 slabs) do not start with an exposed L2 round trip each
__device__ __forceinline__ void load_first_rows(Split3 (&a)[4], const bf16x8 *wp, int slabs, int lane) {
  const size_t tstep = (size_t)slabs * 3 * 64;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const bf16x8 *q = wp + lane + t * tstep;
    a[t].p1 = q[0]; a[t].p2 = q[64]; a[t].p3 = q[128];
  }
}

// A0 holds the fragments of slab 0 on entry (load_first_rows) and is dead on return
__device__ __forceinline__ void split_layer_rows(f32x16 (&acc)[4], Split3 (&A0)[4], const bf16x8 *wp, int slabs,
                                                 const float *rows_img, int stride, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const bf16x8 *wa = wp + lane;
  const size_t tstep = (size_t)slabs * 3 * 64;
  const float *b0 = rows_img + (size_t)r * stride + 8 * h;
  auto load_a = [&](Split3 (&a)[4], int s) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bf16x8 *q = wa + t * tstep + (size_t)s * 192;
      a[t].p1 = q[0]; a[t].p2 = q[64]; a[t].p3 = q[128];
    }
  };
  Split3 A1[4], B[2];
  f32x4 R[2][2];
  R[0][0] = *reinterpret_cast<const f32x4 *>(b0); R[0][1] = *reinterpret_cast<const f32x4 *>(b0 + 4);
  B[0] = split8(R[0][0], R[0][1]);
  const int sb1 = slabs > 1 ? 1 : 0;
  R[1][0] = *reinterpret_cast<const f32x4 *>(b0 + 16 * sb1); R[1][1] = *reinterpret_cast<const f32x4 *>(b0 + 16 * sb1 + 4);
  auto step = [&](int s, Split3 (&a_use)[4], Split3 (&a_ld)[4], const Split3 &b_use, Split3 &b_mk, const f32x4 (&raw_use)[2],
                  f32x4 (&raw_ld)[2]) {
    const int s1 = (s + 1 < slabs) ? s + 1 : s, s2 = (s + 2 < slabs) ? s + 2 : s1;
    if (s + 1 < slabs) load_a(a_ld, s1);
    const f32x4 n0 = *reinterpret_cast<const f32x4 *>(b0 + 16 * s2), n1 = *reinterpret_cast<const f32x4 *>(b0 + 16 * s2 + 4);
    __builtin_amdgcn_sched_barrier(0);
    b_mk = split8(raw_use[0], raw_use[1]);
#pragma unroll
    for (int t = 0; t < 4; ++t) mfma6(acc[t], a_use[t], b_use);
    __builtin_amdgcn_sched_barrier(0);
    raw_ld[0] = n0; raw_ld[1] = n1;
  };
  int s = 0;
  for (; s + 1 < slabs; s += 2) {
    step(s, A0, A1, B[0], B[1], R[1], R[0]);
    step(s + 1, A1, A0, B[1], B[0], R[0], R[1]);
  }
  if (s < slabs) step(s, A0, A1, B[0], B[1], R[1], R[0]);
}

__global__ __launch_bounds__(kThreadsC) void critic_split_kernel(const SplitArgs a) {
  const MlpKernelArgs &p = a.m;
  extern __shared__ f32x4 smem4[];
  const int kpad0 = a.slabs0 * 16, XS = kpad0 + 4;
  float *xs = reinterpret_cast<float *>(smem4);       // [64][XS] scaled input, zero padded
  float *h1 = xs + 64 * XS;                           // [64][HSC]
  float *cst = h1 + 64 * HSC;                         // per member: b0[128] | b1[128] | W2 column [128] | b2, pad
  constexpr int CST = 3 * HC + 4;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n_rows = p.n_rows_dev ? *p.n_rows_dev : p.n_rows;
  const int row0 = blockIdx.x * 64;
  if (row0 >= n_rows) return;
  const int E = p.ensemble;
  // ---- constants of all members (once per workgroup) ----------------------------------------------------------------
  for (int i = tid; i < E * CST; i += kThreadsC) {
    const int e = i / CST, k = i - e * CST;
    float v = 0.0f;
    if (k < HC) v = p.b0[(size_t)e * HC + k];
    else if (k < 2 * HC) v = p.b1[(size_t)e * HC + (k - HC)];
    else if (k < 3 * HC) {
      const int kk = k - 2 * HC;      // column 0 of the packed W2: pack index of (k = kk, n = 0)
      v = reinterpret_cast<const float *>(p.wp2 + (size_t)e * p.wp2_stride)[(((size_t)(kk >> 3)) * 64 + ((kk >> 2) & 1) * 32) * 4 + (kk & 3)];
    } else if (k == 3 * HC) v = p.b2[(size_t)e * p.o_tiles * 32];
    cst[i] = v;
  }
  // ---- this wave's 32 rows of the input image ------------------------------------------------------------------------
  int my_row = -1;    // global row of batch row r of this wave (same in both lane halves)
  {
    const int rr0 = row0 + 32 * wave + r;
    my_row = (rr0 < n_rows) ? (p.row_idx ? p.row_idx[rr0] : rr0) : -1;
    const int c = lane & 7;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int b = 8 * pass + (lane >> 3);
      const int rr = __shfl(my_row, b, 64);
      for (int k = c; k < kpad0; k += 8) {
        float x = 0.0f;
        if (k < p.in_dim && rr >= 0) {
          x = (k < p.obs_dim) ? p.obs[(size_t)rr * p.obs_dim + k] : p.act[(size_t)rr * p.act_dim + (k - p.obs_dim)];
          if (p.in_mu) x = (x - p.in_mu[k]) / p.in_sig[k];   // TensorStandardScaler.transform, models/pens/utils.py:156
        }
        xs[(32 * wave + b) * XS + k] = x;
      }
    }
  }
  __syncthreads();   // the members' constants (shared); the input rows are this wave's own

  const float *xrow = xs + (size_t)(32 * wave) * XS;
  float *hrow = h1 + (size_t)(32 * wave) * HSC;
  float member_sum = 0.0f;
  const float o_sig = p.out_mu ? p.out_sig[0] : 1.0f, o_mu = p.out_mu ? p.out_mu[0] : 0.0f;
  Split3 F[4];
  load_first_rows(F, a.sp0, a.slabs0, lane);
  for (int e = 0; e < E; ++e) {
    const float *ce = cst + e * CST;
    f32x16 acc[4];
    auto init_bias = [&](const float *bias) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + t * 32 + 8 * q + 4 * h);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[t][4 * q + s] = bv[s];
        }
    };
    init_bias(ce);
    split_layer_rows(acc, F, a.sp0 + (size_t)e * a.sp0_stride, a.slabs0, xrow, XS, lane);
    load_first_rows(F, a.sp1 + (size_t)e * a.sp1_stride, HC / 16, lane);      // in flight behind the swish / LDS image
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = swishf(acc[t][4 * q + s]);
        *reinterpret_cast<f32x4 *>(hrow + (size_t)r * HSC + t * 32 + 8 * q + 4 * h) = v;   // read back by this wave only
      }
    init_bias(ce + HC);
    split_layer_rows(acc, F, a.sp1 + (size_t)e * a.sp1_stride, HC / 16, hrow, HSC, lane);
    if (e + 1 < E) load_first_rows(F, a.sp0 + (size_t)(e + 1) * a.sp0_stride, a.slabs0, lane);   // next member, behind the output dot
    float partial = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w = *reinterpret_cast<const f32x4 *>(ce + 2 * HC + t * 32 + 8 * q + 4 * h);
#pragma unroll
        for (int s = 0; s < 4; ++s) partial = fmaf(swishf(acc[t][4 * q + s]), w[s], partial);
      }
    partial += __shfl_xor(partial, 32, 64);
    member_sum += o_sig * (partial + ce[3 * HC]) + o_mu;
  }
  if (h == 0 && my_row >= 0) p.out0[my_row] = member_sum / (float)E;
}

}  // namespace

// ---- host side -------------------------------------------------------------------------------------------------
// (re)builds the three bf16 images from the handle's fp32 packs when they changed since the last build
static int ensure_split(cmbpo_mlp *m, hipStream_t s) {
  const int H = m->hidden, E = m->ensemble;
  const int slabs[3] = {(m->in_pad + 15) / 16, H / 16, H / 16};
  const int tiles[3] = {H / 32, H / 32, m->o_tiles};
  if (m->d_split == nullptr) {
    size_t off = 0;
    for (int l = 0; l < 3; ++l) {
      m->sp_stride[l] = (size_t)tiles[l] * slabs[l] * 3 * 64;
      m->sp_off[l] = off;
      off += m->sp_stride[l] * E;
    }
    if (hipMalloc(&m->d_split, off * 16) != hipSuccess) {
      (void)hipGetLastError();
      m->d_split = nullptr;
      cmbpo_set_error("ens_split: hipMalloc of the bf16 weight images failed");
      return CMBPO_ENOMEM;
    }
    m->split_version = ~0ul;
  }
  if (m->split_version == m->pack_version) return CMBPO_OK;
  const size_t src_off[3] = {m->off_wp0, m->off_wp1, m->off_wp2};
  const int kg[3] = {m->in_pad / 8, H / 8, H / 8};
  for (int l = 0; l < 3; ++l) {
    const size_t src_stride = (size_t)tiles[l] * kg[l] * 256;
    const long total = (long)tiles[l] * slabs[l] * 64 * E;
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, m->d_blob + src_off[l],
                       src_stride, kg[l], reinterpret_cast<bf16x8 *>(m->d_split) + m->sp_off[l], m->sp_stride[l], tiles[l],
                       slabs[l], E, l == 2 ? 1 : 0);
  }
  CMBPO_HIP_CHECK(hipGetLastError());
  m->split_version = m->pack_version;
  return CMBPO_OK;
}

int cmbpo_internal_launch_split(cmbpo_mlp *m, MlpKernelArgs &a, hipStream_t s) {
  if (int rc = ensure_split(m, s)) return rc;
  SplitArgs k{};
  k.m = a;
  const bf16x8 *base = reinterpret_cast<const bf16x8 *>(m->d_split);
  k.sp0 = base + m->sp_off[0]; k.sp1 = base + m->sp_off[1]; k.sp2 = base + m->sp_off[2];
  k.sp0_stride = m->sp_stride[0]; k.sp1_stride = m->sp_stride[1]; k.sp2_stride = m->sp_stride[2];
  k.slabs0 = (m->in_pad + 15) / 16;
  const int OT = m->o_tiles;
  CMBPO_REQUIRE(OT >= 1 && OT <= 4, "ens_split: %d output tiles", OT);
  static int n_cu = 0;
  if (n_cu == 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  // 64-row items halve the weight stream per row; while 32-row items still find a CU each (small rollout batches, where
  // an item's latency is the step's) they take 0.8 of a 64-row item's time.  Measured, AntSafe shapes: 42 / 42 / 45 us
  // at 256 / 512 / 1000 rows against 51 / 52 / 54 us; at 2000 rows 32-row items would share CUs: 74 us against 62.
  const int BT = (cmbpo_ceil_div(a.n_rows, 32) * m->ensemble <= n_cu) ? 1 : 2;
  const int rows = 32 * BT;
  const int tiles = cmbpo_ceil_div(a.n_rows, rows);
  k.m.tiles = tiles;
  k.m.n_items = tiles * m->ensemble;
  const size_t img = (size_t)rows * HS + (size_t)rows * (k.slabs0 * 16 + 4);
  const size_t lds = (img + 2 * HIDS + OT * 32 + 3 * m->out_dim + rows) * sizeof(float);
  CMBPO_REQUIRE(lds <= 160 * 1024, "ens_split: LDS budget exceeded (%zu B)", lds);
  CMBPO_REQUIRE((size_t)4 * OT * 32 * (rows + 1) <= img, "ens_split: partial-output image does not fit");
  const int resident = (BT == 1 ? 2 : 1) * n_cu;
  const int grid = k.m.n_items < resident ? k.m.n_items : resident;
  static size_t attr_bytes[5][3] = {};
#define CMBPO_SPLIT_CASE(OT_, BT_)                                                                                   \
  if (OT == OT_ && BT == BT_) {                                                                                      \
    if (lds > attr_bytes[OT_][BT_]) {                                                                                \
      CMBPO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ens_split_kernel<OT_, BT_>),                \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                    \
      attr_bytes[OT_][BT_] = lds;                                                                                    \
    }                                                                                                                \
    hipLaunchKernelGGL((ens_split_kernel<OT_, BT_>), dim3(grid), dim3(kThreadsS), lds, s, k);                       \
  }
  CMBPO_SPLIT_CASE(1, 1) CMBPO_SPLIT_CASE(2, 1) CMBPO_SPLIT_CASE(3, 1) CMBPO_SPLIT_CASE(4, 1)
  CMBPO_SPLIT_CASE(1, 2) CMBPO_SPLIT_CASE(2, 2) CMBPO_SPLIT_CASE(3, 2) CMBPO_SPLIT_CASE(4, 2)
#undef CMBPO_SPLIT_CASE
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

int cmbpo_internal_launch_critic_split(cmbpo_mlp *m, MlpKernelArgs &a, hipStream_t s) {
  if (int rc = ensure_split(m, s)) return rc;
  SplitArgs k{};
  k.m = a;
  const bf16x8 *base = reinterpret_cast<const bf16x8 *>(m->d_split);
  k.sp0 = base + m->sp_off[0]; k.sp1 = base + m->sp_off[1]; k.sp2 = base + m->sp_off[2];
  k.sp0_stride = m->sp_stride[0]; k.sp1_stride = m->sp_stride[1]; k.sp2_stride = m->sp_stride[2];
  k.slabs0 = (m->in_pad + 15) / 16;
  const int items = cmbpo_ceil_div(a.n_rows, 64);
  const size_t lds = ((size_t)64 * (k.slabs0 * 16 + 4) + (size_t)64 * HSC + (size_t)m->ensemble * (3 * HC + 4)) * sizeof(float);
  static size_t attr_bytes = 0;
  if (lds > attr_bytes) {
    CMBPO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(critic_split_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_bytes = lds;
  }
  hipLaunchKernelGGL(critic_split_kernel, dim3(items), dim3(kThreadsC), lds, s, k);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}
