// Fused 3-layer ensemble-MLP forward for gfx950 (MI355X), fp32-exact MFMA.
//
// Replaces, on the rollout path of the reference:
//   models/pens/fc.py:74-95      FC.compute_output_tensor (einsum / matmul + bias + act)
//   models/pens/pe.py:789-838    PE._compile_outputs (scalers, mean|logvar split, exp)
//   models/pens/pe.py:338-343    critic mean over members
//   network/ac_network.py:99-123 Gaussian policy head
//
// Formulation (transposed GEMM chain, activations never leave the CU):
//   hT[n][b] = act( sum_k W[k][n] * xT[k][b] + bias[n] )
// one workgroup = 4 waves = one tile of BB branches of one member (or a loop
// over members).  v_mfma_f32_32x32x2_f32 computes D[i][j] += A[i][k]*B[k][j]
// with A = W^T (rows n) and B = hT (columns b = branches on the lanes):
//   * A fragments come straight from global/L2 as one coalesced 16-B load per
//     lane per 8 k: the host packs W into [n-tile][k-group][lane][4] so that
//     lane (i = lane&31, h = lane>>5) holds W[8g+4h+s][n0+i], s = 0..3.
//   * B fragments come from LDS, layout [k/4][b] of float4, one ds_read_b128
//     per lane per 8 k (conflict-free: consecutive lanes, consecutive 16 B).
//   * the accumulator tile (row n = (r&3)+8(r>>2)+4h, col b = lane&31) is
//     written back to the same LDS layout with one ds_write_b128 per 4
//     registers, so the next layer reads it without any transpose.
// Hidden layers split N over the 4 waves, the (narrow) output layer splits K
// over the 4 waves and reduces through LDS.
#include "common.h"
#include "ens_mlp_internal.h"
#include "mfma_tile.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#ifdef CMBPO_STAMPS
// Diagnostic build only (tools/probe_stamps.py): per-workgroup phase timestamps of wave 0.
static unsigned long long *g_stamps = nullptr;
extern "C" void cmbpo_debug_set_stamps(unsigned long long *p) { g_stamps = p; }
#define STAMP(k)                                                                                   \
  do {                                                                                             \
    if (p.stamps && threadIdx.x == 0)                                                              \
      p.stamps[(size_t)stamp_item * 12 + (k)] =                                  \
          ((k) >= 8 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime());                 \
  } while (0)
#else
#define STAMP(k)
#endif

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = 4;


template <int ACT>
__device__ __forceinline__ float activate(float x) {
  if constexpr (ACT == CMBPO_ACT_SWISH) {
    // x * sigmoid(x), models/pens/fc.py:19.  v_exp + v_rcp (1 ulp each) instead of an IEEE divide:
    // the epilogue is VALU work that competes with the partner wave's MFMA issue.
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
  } else {
    return cmbpo_fast_tanh(x);
  }
}

// swish and its derivative from one sigmoid: h = z s, dh/dz = s (1 + z (1 - s)); v_exp + v_rcp like the rollout
// forward (an IEEE divide is ~10 more VALU instructions per value, next to the partner wave's MFMAs).
__device__ __forceinline__ void swish_with_grad(float z, float &h, float &g) {
  const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
  h = z * sg;
  g = sg * (1.0f + z * (1.0f - sg));
}

// Accumulators start from the bias (read from LDS as the same float4 groups the tile rows form), so the
// epilogues have no bias add and no zero-initialising moves: VALU work next to a partner wave's fp32 MFMA
// stream is the scarce resource of this kernel.
template <int NT, int BT>
__device__ __forceinline__ void init_acc_bias(f32x16 (&acc)[NT][BT], const float *bias, int n_base, int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + n_base + t * 32 + 8 * q + 4 * h);
#pragma unroll
      for (int bt = 0; bt < BT; ++bt)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[t][bt][4 * q + s] = bv[s];
    }
}

// activation, accumulator tile -> LDS image [n/4][BB] of float4.
template <int NT, int BT, int ACT>
__device__ __forceinline__ void store_hidden(const f32x16 (&acc)[NT][BT],
                                             int n_base, f32x4 *lds_out, int lane) {
  constexpr int BB = 32 * BT;
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n0 = n_base + t * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + 8 * q + 4 * h;
#pragma unroll
      for (int bt = 0; bt < BT; ++bt) {
        f32x4 v;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = activate<ACT>(acc[t][bt][4 * q + s]);
        lds_out[(n >> 2) * BB + bt * 32 + j] = v;
      }
    }
  }
}

// Training forward: as store_hidden / activate_regs (swish), and the activation h and its derivative g of the
// wave's [n-slice x BB rows] tile go to the export arrays [row][HID] as float4 groups of four consecutive n.
// `grow` = first export row of the tile (member offset included), rows at or beyond `n_valid` are not written.
template <int NT, int BT, int HID, bool TO_LDS>
__device__ __forceinline__ float hidden_train(f32x16 (&acc)[NT][BT], int n_base, f32x4 *lds_out, int lane,
                                              float *__restrict__ eh, float *__restrict__ eg, size_t grow, int n_valid) {
  constexpr int BB = 32 * BT;
  const int j = lane & 31, h = lane >> 5;
  float hmax = 0.0f;      // largest |h| among the rows this lane exports
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n0 = n_base + t * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + 8 * q + 4 * h;
#pragma unroll
      for (int bt = 0; bt < BT; ++bt) {
        f32x4 hv, gv;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          float a, b;
          swish_with_grad(acc[t][bt][4 * q + s], a, b);
          hv[s] = a; gv[s] = b;
          acc[t][bt][4 * q + s] = a;
        }
        if constexpr (TO_LDS) lds_out[(n >> 2) * BB + bt * 32 + j] = hv;
        const int b = bt * 32 + j;
        if (eh && b < n_valid) {
          const size_t o = (grow + b) * HID + n;
          *reinterpret_cast<f32x4 *>(eh + o) = hv;
          *reinterpret_cast<f32x4 *>(eg + o) = gv;
          hmax = fmaxf(fmaxf(hmax, fmaxf(fabsf(hv[0]), fabsf(hv[1]))), fmaxf(fabsf(hv[2]), fabsf(hv[3])));
        }
      }
    }
  }
  return hmax;
}

// activation in place: the accumulator tile becomes the B operand of the output layer.
template <int NT, int BT, int ACT>
__device__ __forceinline__ void activate_regs(f32x16 (&acc)[NT][BT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int bt = 0; bt < BT; ++bt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][bt][r] = activate<ACT>(acc[t][bt][r]);
}

// Output-layer slice straight from registers.  The accumulator rows a wave owns after a hidden layer
// (n = n_base + 32 t + (r & 3) + 8 (r >> 2) + 4 h) are exactly the k indices the packed A layout assigns to
// k-group g = n_base / 8 + 4 t + (r >> 2), element s = r & 3 of lane-half h -- so hreg[t][bt][r] IS the
// B fragment of MFMA step (g, s) and the output layer's K split over the waves needs no LDS round trip.
template <int NT, int BT>
__device__ __forceinline__ void mfma_from_regs(const f32x4 *__restrict__ wp, int g_base, const f32x16 (&hreg)[NT][BT],
                                               int lane, f32x16 (&out)[BT]) {
  f32x4 a_cur = wp[(size_t)g_base * 64 + lane], a_nxt;
#pragma unroll
  for (int i = 0; i < NT * 4; ++i) {
    const int t = i >> 2, q = i & 3;
    const int inext = (i + 1 < NT * 4) ? i + 1 : i;
    a_nxt = wp[(size_t)(g_base + inext) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int bt = 0; bt < BT; ++bt)
        out[bt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s], hreg[t][bt][4 * q + s], out[bt], 0, 0, 0);
    a_cur = a_nxt;
  }
}

template <int HID, int BT, int ACT, int HEAD>
__global__ __launch_bounds__(kThreads, (BT == 1 ? 2 : 1)) void ens_mlp_kernel(
    const MlpKernelArgs p) {
  constexpr int BB = 32 * BT;
  constexpr int NT = HID / 128;       // n-tiles per wave in the hidden layers
  constexpr int KG_H = HID / 8;       // k-groups of a hidden-width K
  constexpr int RED_LD = BB + 1;      // padded b-stride of the reduction image

  extern __shared__ f32x4 smem[];
  const int hbuf_f4 = HID / 4 * BB;
  const int red_f4 = (kWaves * p.o_tiles * 32 * RED_LD + 3) / 4;
  f32x4 *hbuf = smem;
  float *red = reinterpret_cast<float *>(smem);  // aliases hbuf (used after it is dead)
  f32x4 *xbuf = smem + (hbuf_f4 > red_f4 ? hbuf_f4 : red_f4);
  float *bias_l = reinterpret_cast<float *>(xbuf + p.in_pad / 4 * BB);   // [HID | HID | o_tiles*32] of one member
  float *oconst = bias_l + 2 * HID + p.o_tiles * 32;   // [sig | 2 log sig | mu] x out_dim (output scaler)
  int *rows = reinterpret_cast<int *>(oconst + 3 * p.out_dim);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_rows = p.n_rows_dev ? *p.n_rows_dev : p.n_rows;
  // Stagger: the workgroups that share a CU run the same program and would otherwise stay in lockstep (all in
  // their latency-bound prologue / epilogue at once, then all fighting for the MFMA pipe).  The dispatcher
  // deals the first n_cu workgroups one per CU and the next ones on top of them, so delaying that second
  // batch once by about half an item puts the pairs in anti-phase.  Placement is only a speed assumption: a
  // wrong guess costs the delay, never correctness.
  if (p.stagger_sleeps > 0 && (int)blockIdx.x >= p.n_cu && (int)blockIdx.x < 2 * p.n_cu) {
    for (int i = 0; i < p.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
  }
  if constexpr (HEAD == CMBPO_HEAD_PROB || HEAD == CMBPO_HEAD_DETMEAN) {
    // output scaler constants, once per workgroup (models/pens/utils.py:167,187)
    if (tid < p.out_dim) {
      oconst[tid] = p.out_mu ? p.out_sig[tid] : 1.0f;
      oconst[p.out_dim + tid] = p.out_mu ? p.out_lsig2[tid] : 0.0f;
      oconst[2 * p.out_dim + tid] = p.out_mu ? p.out_mu[tid] : 0.0f;
    }
  }
  // Persistent workgroups: the grid is sized to what is co-resident (2 per CU for the 512-wide ensemble) and
  // each workgroup strides over the (member chunk, row tile) items in member-major order, so the hardware
  // dispatcher never has to back-fill a CU (measured: 20-36 us gaps per slot on half of the CUs with one
  // workgroup per item) and concurrently running workgroups share one member's weights in L2.
  __shared__ int s_item;
  for (int item = blockIdx.x;;) {
  if (p.work_counter) {
    // dynamic claiming: one returning device-scope atomic per item (MI355X_MICROARCH "dequeue": 0.3-1 us)
    if (tid == 0) s_item = atomicAdd(p.work_counter, 1);
    __syncthreads();
    item = s_item;
  }
  if (item >= p.n_items) break;
  const int chunk = item / p.tiles;
  const int row0 = (item - chunk * p.tiles) * BB;
  if (row0 >= n_rows) {
    if (!p.work_counter) item += gridDim.x;
    else __syncthreads();
    continue;
  }
#ifdef CMBPO_STAMPS
  const int stamp_item = item;
#endif
  STAMP(0);
  STAMP(8);
#ifdef CMBPO_STAMPS
  if (p.stamps && threadIdx.x == 0) {
    p.stamps[(size_t)stamp_item * 12 + 10] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    p.stamps[(size_t)stamp_item * 12 + 11] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
  }
#endif

  // biases of the item's first member: requested before the input tile, so that their latency passes together with
  // the row-index -> row-gather chain below instead of after it
  constexpr int kMaxB = (2 * HID + 128 + kThreads - 1) / kThreads;
  float bias_tmp[kMaxB];
  auto fetch_bias = [&](int e) {
    const int o_pad = p.o_tiles * 32;
#pragma unroll
    for (int u = 0; u < kMaxB; ++u) {
      const int i = tid + u * kThreads;
      bias_tmp[u] = 0.0f;
      if (i < 2 * HID + o_pad)
        bias_tmp[u] = (i < HID) ? p.b0[(size_t)e * HID + i]
                                : (i < 2 * HID) ? p.b1[(size_t)e * HID + (i - HID)] : p.b2[(size_t)e * o_pad + (i - 2 * HID)];
    }
  };
  fetch_bias(chunk * p.e_chunk);

  // ---- stage the (scaled) input tile: xT[k][b], k = [obs | act] ------------
  // 8 threads per branch row; every thread issues all of its (independent) global loads before it
  // touches LDS, so the prologue pays the memory latency once instead of once per element.
  __shared__ float s_opmax[3][kWaves];     // HEAD_TRAIN: per-wave maxima of |x|, |h1|, |h2| of this tile
  float xmax_t = 0.0f;
  {
    float *xf = reinterpret_cast<float *>(xbuf);
    const int c = tid & 7;
#pragma unroll
    for (int rb = 0; rb < BT; ++rb) {
      const int b = (tid >> 3) + 32 * rb;
      int r = row0 + b;
      if constexpr (HEAD == CMBPO_HEAD_TRAIN) {
        // per-member bootstrap rows: member (= chunk, e_chunk is 1) reads its own index list
        r = (r < n_rows) ? (p.row_idx ? p.row_idx[(size_t)chunk * p.row_idx_stride + r] : r) : -1;
      } else {
        r = (r < n_rows) ? (p.row_idx ? p.row_idx[r] : r) : -1;
      }
      if (c == 0) rows[b] = r;
      for (int k0 = 0; k0 < p.in_pad; k0 += 64) {
        float v[8], mu[8], sig[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + c + 8 * u;
          v[u] = 0.0f; mu[u] = 0.0f; sig[u] = 1.0f;
          if (k < p.in_dim) {
            if (r >= 0)
              v[u] = (k < p.obs_dim) ? p.obs[(size_t)r * p.obs_dim + k] : p.act[(size_t)r * p.act_dim + (k - p.obs_dim)];
            if (p.in_mu) { mu[u] = p.in_mu[k]; sig[u] = p.in_sig[k]; }
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + c + 8 * u;
          if (k < p.in_pad) {
            float x = v[u];
            if (p.in_mu && k < p.in_dim && r >= 0) {
              // TensorStandardScaler.transform, models/pens/utils.py:156 (sig precomputed at load time)
              x = (x - mu[u]) / sig[u];
            }
            xf[((k >> 2) * BB + b) * 4 + (k & 3)] = x;   // k >= in_dim: zero padding
            if constexpr (HEAD == CMBPO_HEAD_TRAIN) {
              if (p.tr_x && r >= 0) {
                p.tr_x[((size_t)chunk * n_rows + row0 + b) * p.in_pad + k] = x;
                xmax_t = fmaxf(xmax_t, fabsf(x));
              }
            }
          }
        }
      }
    }
  }

  if constexpr (HEAD == CMBPO_HEAD_TRAIN) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xmax_t = fmaxf(xmax_t, __shfl_xor(xmax_t, o, 64));
    if (lane == 0) s_opmax[0][wave] = xmax_t;      // (read in the head epilogue, several barriers on)
  }
  const int e_begin = chunk * p.e_chunk;
  const int e_end = min(p.ensemble, e_begin + p.e_chunk);
  float member_sum = 0.0f;  // HEAD_DETMEAN: running sum over members

  for (int e = e_begin; e < e_end; ++e) {
    const int kg0 = p.in_pad / 8;
    const int o_pad = p.o_tiles * 32;
    // biases of this member -> LDS (read back as float4 in the epilogues; a global load there would
    // expose its latency 16 times per layer)
    STAMP(1);
    if (e > e_begin) __syncthreads();   // previous member's epilogue still reads bias_l / red (= hbuf)
    {
      if (e > e_begin) fetch_bias(e);
#pragma unroll
      for (int u = 0; u < kMaxB; ++u) {
        const int i = tid + u * kThreads;
        if (i < 2 * HID + o_pad) bias_l[i] = bias_tmp[u];
      }
    }
    __syncthreads();   // x tile (first member) and this member's biases are in LDS
    // ---- layer 0: in -> HID ------------------------------------------------
    {
      f32x16 acc[NT][BT];
      init_acc_bias<NT, BT>(acc, bias_l, wave * NT * 32, lane);
      const f32x4 *wp = p.wp0 + e * p.wp0_stride + (size_t)(wave * NT) * kg0 * 64;
      mfma_layer<NT, BT>(wp, (size_t)kg0 * 64, 0, kg0, xbuf, lane, acc);
      STAMP(2);
      if constexpr (HEAD == CMBPO_HEAD_TRAIN) {
        float hm = hidden_train<NT, BT, HID, true>(acc, wave * NT * 32, hbuf, lane, p.tr_h1, p.tr_g1,
                                                   (size_t)e * n_rows + row0, n_rows - row0);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) hm = fmaxf(hm, __shfl_xor(hm, o, 64));
        if (lane == 0) s_opmax[1][wave] = hm;
      } else
        store_hidden<NT, BT, ACT>(acc, wave * NT * 32, hbuf, lane);
    }
    __syncthreads();
    STAMP(3);
    // ---- layer 1: HID -> HID, then layer 2: HID -> o_width from registers --------------------
    {
      f32x16 acc[NT][BT];
      init_acc_bias<NT, BT>(acc, bias_l + HID, wave * NT * 32, lane);
      const f32x4 *wp = p.wp1 + e * p.wp1_stride + (size_t)(wave * NT) * KG_H * 64;
      // (a two-group-deep ring, mfma_layer<..., DEPTH = 2>, measured no faster: the lone-wave loss is the
      // ~27-cycle issue cost of each global_load_dwordx4 next to the MFMAs, not exposed latency)
      mfma_layer<NT, BT>(wp, (size_t)KG_H * 64, 0, KG_H, hbuf, lane, acc);
      STAMP(4);
      if constexpr (HEAD == CMBPO_HEAD_TRAIN) {
        float hm = hidden_train<NT, BT, HID, false>(acc, wave * NT * 32, nullptr, lane, p.tr_h2, p.tr_g2,
                                                    (size_t)e * n_rows + row0, n_rows - row0);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) hm = fmaxf(hm, __shfl_xor(hm, o, 64));
        if (lane == 0) s_opmax[2][wave] = hm;
      } else
        activate_regs<NT, BT, ACT>(acc);   // h2 slice of this wave
      STAMP(5);
      __syncthreads();  // every wave has finished reading h1: hbuf becomes the reduction image
      STAMP(6);
      // K of the output layer is split over the waves: wave w contributes k in [128 w, 128 w + 128) for
      // HID = 512 ([32 w, 32 w + 32) for 128) -- its own h2 slice, straight from registers.  One output tile at
      // a time keeps a single 16-register partial live.
      const int j = lane & 31, h = lane >> 5;
      bool valu_out = false;
      if constexpr (HEAD == CMBPO_HEAD_DETMEAN && BT == 1) {
        // One output column (the critics): 16 NT fused multiply-adds per lane on the h2 registers instead of a
        // 32-column MFMA tile of which one column is used (16 NT MFMAs = 1/6 of a 128-wide member's MFMA time).
        // Column 0 of the packed W2 sits in lanes 0 / 32 of every k-group: a broadcast 16-B load per (t, q).
        if (p.o_width == 1) {
          valu_out = true;
          const f32x4 *w2c = p.wp2 + e * p.wp2_stride + (size_t)(wave * NT * 4) * 64 + h * 32;
          float partial = 0.0f;
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 w = w2c[(size_t)(4 * t + q) * 64];
#pragma unroll
              for (int s = 0; s < 4; ++s) partial = fmaf(acc[t][0][4 * q + s], w[s], partial);
            }
          partial += __shfl_xor(partial, 32, 64);
          if (h == 0) red[(wave * p.o_tiles * 32) * RED_LD + j] = partial;
        }
      }
      for (int ot = 0; ot < (valu_out ? 0 : p.o_tiles); ++ot) {
        f32x16 part[BT];
#pragma unroll
        for (int bt = 0; bt < BT; ++bt)
#pragma unroll
          for (int r = 0; r < 16; ++r) part[bt][r] = 0.0f;
        const f32x4 *wp2 = p.wp2 + e * p.wp2_stride + (size_t)ot * KG_H * 64;
        mfma_from_regs<NT, BT>(wp2, wave * NT * 4, acc, lane, part);
#pragma unroll
        for (int bt = 0; bt < BT; ++bt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            red[(wave * p.o_tiles * 32 + n) * RED_LD + bt * 32 + j] = part[bt][r];
          }
      }
    }
    __syncthreads();

    // ---- head epilogue: all threads over (b, n), n fastest -------------------
    const int o_ld = p.o_tiles * 32;
    auto reduced = [&](int b, int n) -> float {
      float v = red[(0 * o_ld + n) * RED_LD + b];
      v += red[(1 * o_ld + n) * RED_LD + b];
      v += red[(2 * o_ld + n) * RED_LD + b];
      v += red[(3 * o_ld + n) * RED_LD + b];
      return v + bias_l[2 * HID + n];
    };
    if constexpr (HEAD == CMBPO_HEAD_PROB) {
      // mean = sig*o + mu ; var = exp(2*log(sig) + o'), models/pens/pe.py:815-835
      const int out = p.out_dim;
      const float inv_out = 1.0f / (float)out;
      for (int i = tid; i < BB * out; i += kThreads) {
        const int b = (int)(((float)i + 0.5f) * inv_out);   // exact for i < 2^16, out <= 128
        const int n = i - b * out;
        const int r = rows[b];
        if (r < 0) continue;
        const float m = oconst[n] * reduced(b, n) + oconst[2 * out + n];
        const float lv = oconst[out + n] + reduced(b, out + n);
        const size_t o = ((size_t)e * p.ld_rows + r) * out + n;
        p.out0[o] = m;
        p.out1[o] = __expf(lv);
      }
    } else if constexpr (HEAD == CMBPO_HEAD_TRAIN) {
      // raw network output (no output scaler, no exp): models/pens/pe.py:803-812 with ret_log_var
      if (p.tr_opmax && tid < 3)
        p.tr_opmax[(size_t)item * 8 + tid] = fmaxf(fmaxf(s_opmax[tid][0], s_opmax[tid][1]), fmaxf(s_opmax[tid][2], s_opmax[tid][3]));
      const int ow = p.o_width;
      for (int i = tid; i < BB * ow; i += kThreads) {
        const int b = i / ow, n = i - b * ow;
        if (rows[b] < 0) continue;
        p.out0[((size_t)e * n_rows + row0 + b) * ow + n] = reduced(b, n);
      }
      if (p.tr_loss_part) {
        // this tile's share of the loss statistics (sum (m - t)^2, sum (var - mse)^2, sum lv^2): the backward kernel
        // adds the tiles' partials in a fixed order, so the training step needs no separate reduction kernels
        const int D = p.tr_tdim;
        const bool prob = ow == 2 * D;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        for (int i = tid; i < BB * D; i += kThreads) {
          const int b = i / D, d = i - b * D;
          const int r = rows[b];
          if (r < 0) continue;
          float t = p.tr_targets[(size_t)r * D + d];
          if (p.out_mu) t = (t - p.out_mu[d]) / p.out_sig[d];
          const float diff = reduced(b, d) - t;
          const float mse = diff * diff;
          s0 += (double)mse;
          if (prob) {
            const float lv = reduced(b, D + d);
            const float dv = expf(lv) - mse;
            s1 += (double)(dv * dv);
            s2 += (double)(lv * lv);
          }
        }
        __shared__ double s_loss[3][kWaves];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          s0 += __shfl_down(s0, o, 64);
          s1 += __shfl_down(s1, o, 64);
          s2 += __shfl_down(s2, o, 64);
        }
        if (lane == 0) { s_loss[0][wave] = s0; s_loss[1][wave] = s1; s_loss[2][wave] = s2; }
        __syncthreads();
        if (tid < 3)
          p.tr_loss_part[(size_t)item * 3 + tid] = (s_loss[tid][0] + s_loss[tid][1]) + (s_loss[tid][2] + s_loss[tid][3]);
      }
    } else if constexpr (HEAD == CMBPO_HEAD_DETMEAN) {
      const int out = p.out_dim;
      if (tid < BB * out) {
        const int b = tid / out, n = tid - b * out;
        member_sum += oconst[n] * reduced(b, n) + oconst[2 * out + n];
      }
    } else {  // CMBPO_HEAD_GAUSS_PI
      const int A = p.out_dim;
      float *term = reinterpret_cast<float *>(xbuf);  // x tile is dead (E == 1)
      for (int i = tid; i < BB * A; i += kThreads) {
        const int b = i / A, a = i - b * A;
        const int r = rows[b];
        float t = 0.0f;
        if (r >= 0) {
          const float mu = reduced(b, a);
          const float ls = p.log_std[a];
          const float sd = expf(ls);
          const float pi = mu + p.eps[(size_t)r * A + a] * sd;
          const float z = (pi - mu) / (sd + 1e-8f);
          // gaussian_likelihood, network/ac_network.py:46-48
          t = -0.5f * (z * z + 2.0f * ls + 1.8378770664093453f);
          const size_t o = (size_t)r * A + a;
          p.out0[o] = pi;
          p.out2[o] = mu;
          p.out3[o] = ls;
        }
        term[b * A + a] = t;
      }
      __syncthreads();
      if (tid < BB && rows[tid] >= 0) {
        float s = 0.0f;
        for (int a = 0; a < A; ++a) s += term[tid * A + a];
        p.out1[rows[tid]] = s;
      }
    }
  }
  if constexpr (HEAD == CMBPO_HEAD_DETMEAN) {
    const int out = p.out_dim;
    if (tid < BB * out) {
      const int b = tid / out, n = tid - b * out;
      const int r = rows[b];
      if (r >= 0) p.out0[(size_t)r * out + n] = member_sum / (float)p.ensemble;
    }
  }
  STAMP(7);
  STAMP(9);
  __syncthreads();   // the next item re-uses xbuf / rows / red / s_item
  if (!p.work_counter) item += gridDim.x;
  }  // persistent item loop
}

// Host-side packing: W[K][N] row-major -> [n-tile][k-group][lane][4].
void pack_weights(const float *w, int K, int N, int k_pad, int n_tiles, float *dst) {
  const int kg = k_pad / 8;
  for (int nt = 0; nt < n_tiles; ++nt)
    for (int g = 0; g < kg; ++g)
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 31, h = lane >> 5;
        const int n = nt * 32 + i;
        for (int s = 0; s < 4; ++s) {
          const int k = 8 * g + 4 * h + s;
          dst[(((size_t)nt * kg + g) * 64 + lane) * 4 + s] =
              (k < K && n < N) ? w[(size_t)k * N + n] : 0.0f;
        }
      }
}

}  // namespace


extern "C" int cmbpo_mlp_create(cmbpo_mlp_t **out, int ensemble, int in_dim, int hidden,
                                int out_width, int activation, int head) {
  CMBPO_REQUIRE(out != nullptr, "cmbpo_mlp_create: out is NULL");
  CMBPO_REQUIRE(ensemble >= 1 && ensemble <= 64, "cmbpo_mlp_create: ensemble %d out of range", ensemble);
  CMBPO_REQUIRE(in_dim >= 1 && in_dim <= 256, "cmbpo_mlp_create: in_dim %d out of range", in_dim);
  CMBPO_REQUIRE(hidden == 128 || hidden == 256 || hidden == 512, "cmbpo_mlp_create: hidden must be 128, 256 or 512 (got %d)", hidden);
  CMBPO_REQUIRE(out_width >= 1 && out_width <= 128, "cmbpo_mlp_create: out_width %d out of range", out_width);
  CMBPO_REQUIRE(activation == CMBPO_ACT_SWISH || activation == CMBPO_ACT_TANH, "cmbpo_mlp_create: bad activation %d", activation);
  CMBPO_REQUIRE(head >= CMBPO_HEAD_PROB && head <= CMBPO_HEAD_GAUSS_PI, "cmbpo_mlp_create: bad head %d", head);
  if (head == CMBPO_HEAD_PROB)
    CMBPO_REQUIRE(out_width % 2 == 0, "cmbpo_mlp_create: HEAD_PROB needs an even out_width");
  if (head == CMBPO_HEAD_GAUSS_PI)
    CMBPO_REQUIRE(ensemble == 1 && out_width <= 32, "cmbpo_mlp_create: policy head needs ensemble 1, act_dim <= 32");
  if (head == CMBPO_HEAD_DETMEAN)
    CMBPO_REQUIRE(out_width <= 8, "cmbpo_mlp_create: HEAD_DETMEAN needs out_width <= 8");

  cmbpo_mlp *m = new (std::nothrow) cmbpo_mlp();
  if (!m) { cmbpo_set_error("cmbpo_mlp_create: out of host memory"); return CMBPO_ENOMEM; }
  m->ensemble = ensemble; m->in_dim = in_dim; m->in_pad = (in_dim + 7) / 8 * 8;
  m->hidden = hidden; m->o_width = out_width; m->o_tiles = (out_width + 31) / 32;
  m->out_dim = (head == CMBPO_HEAD_PROB) ? out_width / 2 : out_width;
  m->act = activation; m->head = head; m->loaded = false;
  m->has_in_scaler = m->has_out_scaler = false;

  const int E = ensemble, H = hidden;
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += (n + 3) / 4 * 4; return o; };
  m->off_wp0 = take((size_t)E * (H / 32) * (m->in_pad / 8) * 256);
  m->off_wp1 = take((size_t)E * (H / 32) * (H / 8) * 256);
  m->off_wp2 = take((size_t)E * m->o_tiles * (H / 8) * 256);
  m->off_b0 = take((size_t)E * H);
  m->off_b1 = take((size_t)E * H);
  m->off_b2 = take((size_t)E * m->o_tiles * 32);
  m->off_in_mu = take(in_dim); m->off_in_var = take(in_dim);
  m->off_out_mu = take(m->out_dim); m->off_out_var = take(m->out_dim); m->off_out_lsig2 = take(m->out_dim);
  m->off_log_std = take(m->out_dim);
  m->blob_floats = off;
  m->d_blob = nullptr;
  hipError_t err = hipMalloc(reinterpret_cast<void **>(&m->d_blob), off * sizeof(float));
  if (err != hipSuccess) {
    cmbpo_set_error("cmbpo_mlp_create: hipMalloc(%zu) failed: %s", off * sizeof(float), hipGetErrorString(err));
    delete m;
    return CMBPO_ENOMEM;
  }
  *out = m;
  return CMBPO_OK;
}

extern "C" void cmbpo_mlp_destroy(cmbpo_mlp_t *m) {
  if (!m) return;
  if (m->d_blob) (void)hipFree(m->d_blob);
  if (m->d_split) (void)hipFree(m->d_split);
  if (m->d_h3) (void)hipFree(m->d_h3);
  delete m;
}

extern "C" int cmbpo_mlp_load(cmbpo_mlp_t *m, const float *h_w0, const float *h_b0,
                              const float *h_w1, const float *h_b1, const float *h_w2,
                              const float *h_b2, const float *h_in_mu, const float *h_in_var,
                              const float *h_out_mu, const float *h_out_var,
                              const float *h_log_std, void *stream) {
  CMBPO_REQUIRE(m != nullptr, "cmbpo_mlp_load: handle is NULL");
  CMBPO_REQUIRE(h_w0 && h_b0 && h_w1 && h_b1 && h_w2 && h_b2, "cmbpo_mlp_load: weight pointer is NULL");
  CMBPO_REQUIRE((h_in_mu == nullptr) == (h_in_var == nullptr), "cmbpo_mlp_load: in scaler needs both mu and var");
  CMBPO_REQUIRE((h_out_mu == nullptr) == (h_out_var == nullptr), "cmbpo_mlp_load: out scaler needs both mu and var");
  if (m->head == CMBPO_HEAD_GAUSS_PI)
    CMBPO_REQUIRE(h_log_std != nullptr, "cmbpo_mlp_load: policy head needs log_std");
  const int E = m->ensemble, H = m->hidden, I = m->in_dim, O = m->o_width;
  m->h_blob.assign(m->blob_floats, 0.0f);
  float *hb = m->h_blob.data();
  const size_t s0 = (size_t)(H / 32) * (m->in_pad / 8) * 256;
  const size_t s1 = (size_t)(H / 32) * (H / 8) * 256;
  const size_t s2 = (size_t)m->o_tiles * (H / 8) * 256;
  for (int e = 0; e < E; ++e) {
    pack_weights(h_w0 + (size_t)e * I * H, I, H, m->in_pad, H / 32, hb + m->off_wp0 + e * s0);
    pack_weights(h_w1 + (size_t)e * H * H, H, H, H, H / 32, hb + m->off_wp1 + e * s1);
    pack_weights(h_w2 + (size_t)e * H * O, H, O, H, m->o_tiles, hb + m->off_wp2 + e * s2);
    memcpy(hb + m->off_b0 + (size_t)e * H, h_b0 + (size_t)e * H, H * sizeof(float));
    memcpy(hb + m->off_b1 + (size_t)e * H, h_b1 + (size_t)e * H, H * sizeof(float));
    memcpy(hb + m->off_b2 + (size_t)e * m->o_tiles * 32, h_b2 + (size_t)e * O, O * sizeof(float));
  }
  m->has_in_scaler = h_in_mu != nullptr;
  m->has_out_scaler = h_out_mu != nullptr;
  // scaler sigmas in float32, max(sqrt(var), 1e-2) (models/pens/utils.py:156,167,187); sqrtf is correctly
  // rounded on the host, the device only divides / multiplies by them
  if (h_in_mu) {
    memcpy(hb + m->off_in_mu, h_in_mu, I * sizeof(float));
    for (int k = 0; k < I; ++k) hb[m->off_in_var + k] = fmaxf(sqrtf(h_in_var[k]), 1e-2f);
  }
  if (h_out_mu) {
    memcpy(hb + m->off_out_mu, h_out_mu, m->out_dim * sizeof(float));
    for (int k = 0; k < m->out_dim; ++k) {
      const float sig = fmaxf(sqrtf(h_out_var[k]), 1e-2f);
      hb[m->off_out_var + k] = sig;
      hb[m->off_out_lsig2 + k] = 2.0f * logf(sig);
    }
  }
  if (h_log_std) memcpy(hb + m->off_log_std, h_log_std, m->out_dim * sizeof(float));
  // h_blob stays alive in the handle until the next load, so the async copy is safe.
  ++m->pack_version;
  CMBPO_HIP_CHECK(hipMemcpyAsync(m->d_blob, hb, m->blob_floats * sizeof(float),
                                 hipMemcpyHostToDevice, (hipStream_t)stream));
  m->loaded = true;
  return CMBPO_OK;
}

namespace {
// One flat parameter vector of a Gaussian policy on the device -> the handle's packs (the layout of pack_weights):
// flat = W0[in,H] | b0[H] | W1[H,H] | b1[H] | W2[H,O] | b2[O] | log_std[O]  (get_vars('pi'), network/ac_network.py:35-36)
struct FlatLoadArgs {
  int I, H, O, in_pad, o_tiles;
  size_t off_wp0, off_wp1, off_wp2, off_b0, off_b1, off_b2, off_log_std;
  int n0, n1, n2;        // floats of the three packs
};
__global__ void policy_flat_load_kernel(const float *flat, float *blob, FlatLoadArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int oW0 = 0, ob0 = a.I * a.H, oW1 = ob0 + a.H, ob1 = oW1 + a.H * a.H, oW2 = ob1 + a.H, ob2 = oW2 + a.H * a.O,
            ols = ob2 + a.O;
  auto pack = [&](int i, int K, int N, int k_pad, int src_off, size_t dst_off) {
    const int s = i & 3, lane = (i >> 2) & 63, kg = k_pad / 8, g = (i >> 8) % kg, nt = (i >> 8) / kg;
    const int n = nt * 32 + (lane & 31), k = 8 * g + 4 * (lane >> 5) + s;
    blob[dst_off + i] = (k < K && n < N) ? flat[src_off + (size_t)k * N + n] : 0.0f;
  };
  int i = idx;
  if (i < a.n0) { pack(i, a.I, a.H, a.in_pad, oW0, a.off_wp0); return; }
  i -= a.n0;
  if (i < a.n1) { pack(i, a.H, a.H, a.H, oW1, a.off_wp1); return; }
  i -= a.n1;
  if (i < a.n2) { pack(i, a.H, a.O, a.H, oW2, a.off_wp2); return; }
  i -= a.n2;
  if (i < a.H) { blob[a.off_b0 + i] = flat[ob0 + i]; return; }
  i -= a.H;
  if (i < a.H) { blob[a.off_b1 + i] = flat[ob1 + i]; return; }
  i -= a.H;
  if (i < a.o_tiles * 32) { blob[a.off_b2 + i] = i < a.O ? flat[ob2 + i] : 0.0f; return; }
  i -= a.o_tiles * 32;
  if (i < a.O) blob[a.off_log_std + i] = flat[ols + i];
}
}  // namespace

extern "C" int cmbpo_mlp_load_policy_flat(cmbpo_mlp_t *m, const float *d_flat, void *stream) {
  CMBPO_REQUIRE(m != nullptr && d_flat != nullptr, "cmbpo_mlp_load_policy_flat: NULL argument");
  CMBPO_REQUIRE(m->head == CMBPO_HEAD_GAUSS_PI && m->ensemble == 1,
                "cmbpo_mlp_load_policy_flat: a single Gaussian-policy network (head %d, ensemble %d)", m->head, m->ensemble);
  if (!m->loaded) { cmbpo_set_error("cmbpo_mlp_load_policy_flat: load the handle once through cmbpo_mlp_load first"); return CMBPO_ESTATE; }
  CMBPO_REQUIRE(!m->has_in_scaler && !m->has_out_scaler, "cmbpo_mlp_load_policy_flat: the handle has scalers");
  FlatLoadArgs a;
  a.I = m->in_dim; a.H = m->hidden; a.O = m->o_width; a.in_pad = m->in_pad; a.o_tiles = m->o_tiles;
  a.off_wp0 = m->off_wp0; a.off_wp1 = m->off_wp1; a.off_wp2 = m->off_wp2;
  a.off_b0 = m->off_b0; a.off_b1 = m->off_b1; a.off_b2 = m->off_b2; a.off_log_std = m->off_log_std;
  a.n0 = (a.H / 32) * (a.in_pad / 8) * 256; a.n1 = (a.H / 32) * (a.H / 8) * 256; a.n2 = a.o_tiles * (a.H / 8) * 256;
  const int total = a.n0 + a.n1 + a.n2 + 2 * a.H + a.o_tiles * 32 + a.O;
  ++m->pack_version;        // (the f16 images of the rollout actor follow the packs' version)
  hipLaunchKernelGGL(policy_flat_load_kernel, dim3(cmbpo_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, d_flat,
                     m->d_blob, a);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

namespace {

int g_dispatch_mode = 0;   // 0: one workgroup per item (hardware dispatch), 1: persistent static, 2: persistent dynamic
int *g_work_counter = nullptr;

int device_cu_count() {
  static int n_cu = 0;
  if (n_cu == 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  return n_cu;
}

template <int HID, int BT, int ACT, int HEAD>
int launch_one(MlpKernelArgs &a, int tiles, int chunks, size_t lds, hipStream_t s) {
  auto kern = ens_mlp_kernel<HID, BT, ACT, HEAD>;
  static size_t attr_bytes = 0;   // the kernel also has a few bytes of static LDS: ask for what is needed
  static size_t occ_lds = ~(size_t)0;
  static int per_cu = 1;
  if (lds > attr_bytes) {
    CMBPO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_bytes = lds;
  }
  if (occ_lds != lds) {   // co-resident workgroups per CU for this LDS footprint
    int nb = 0;
    CMBPO_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, kThreads, lds));
    per_cu = nb < 1 ? 1 : (nb > 4 ? 4 : nb);
    occ_lds = lds;
  }
  a.tiles = tiles;
  a.n_items = tiles * chunks;
  a.n_cu = device_cu_count();
  const int resident = per_cu * a.n_cu;
  int grid = a.n_items < resident ? a.n_items : resident;
  a.work_counter = nullptr;
  if (g_dispatch_mode == 0) {
    grid = a.n_items;                       // every workgroup runs exactly one item
  } else if (g_dispatch_mode == 2) {
    if (!g_work_counter) CMBPO_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&g_work_counter), sizeof(int)));
    CMBPO_HIP_CHECK(hipMemsetAsync(g_work_counter, 0, sizeof(int), s));
    a.work_counter = g_work_counter;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, s, a);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

int g_block_rows = 32;  // 32 (2 workgroups / CU) or 64 (1 workgroup / CU)
// matrix path of the 512-wide PROB forward: 0 fp32 MFMAs, 1 ens_split.hip (six bf16 terms), 2 ens_h3.hip (three f16 terms)
int g_split_path = getenv("CMBPO_ENS_SPLIT") ? atoi(getenv("CMBPO_ENS_SPLIT")) : CMBPO_ENS_SPLIT_F16;
// rows below which the 512-wide forward falls back to the bf16 kernel: none since the f16 kernel has 32- and 64-row items
// (22 us against the bf16 kernel's 43 at 1000 rows, AntSafe shapes); the knob stays for experiments
int g_h3_min_rows = getenv("CMBPO_ENS_H3_MIN_ROWS") ? atoi(getenv("CMBPO_ENS_H3_MIN_ROWS")) : 0;
int g_stagger = 10;     // x s_sleep(127) (~8k cycles each) for the second dispatch batch
int g_lds_pad = 0;      // diagnostic: extra dynamic LDS bytes (forces one workgroup per CU)

int launch_mlp(cmbpo_mlp *m, MlpKernelArgs &a, hipStream_t s, int head_override = -1) {
  const int head = head_override >= 0 ? head_override : m->head;
  const float *blob = m->d_blob;
  const int E = m->ensemble, H = m->hidden;
  a.wp0 = reinterpret_cast<const f32x4 *>(blob + m->off_wp0);
  a.wp1 = reinterpret_cast<const f32x4 *>(blob + m->off_wp1);
  a.wp2 = reinterpret_cast<const f32x4 *>(blob + m->off_wp2);
  a.wp0_stride = (size_t)(H / 32) * (m->in_pad / 8) * 64;
  a.wp1_stride = (size_t)(H / 32) * (H / 8) * 64;
  a.wp2_stride = (size_t)m->o_tiles * (H / 8) * 64;
  a.b0 = blob + m->off_b0; a.b1 = blob + m->off_b1; a.b2 = blob + m->off_b2;
  a.in_mu = m->has_in_scaler ? blob + m->off_in_mu : nullptr;
  a.in_sig = m->has_in_scaler ? blob + m->off_in_var : nullptr;
  a.out_mu = m->has_out_scaler ? blob + m->off_out_mu : nullptr;
  a.out_sig = m->has_out_scaler ? blob + m->off_out_var : nullptr;
  a.out_lsig2 = m->has_out_scaler ? blob + m->off_out_lsig2 : nullptr;
  a.log_std = blob + m->off_log_std;
#ifdef CMBPO_STAMPS
  a.stamps = g_stamps;
#endif
  a.stagger_sleeps = (H == 512 && head != CMBPO_HEAD_TRAIN) ? g_stagger : 0;
  a.ensemble = E;
  a.e_chunk = (head == CMBPO_HEAD_PROB || head == CMBPO_HEAD_TRAIN) ? 1 : E;
  a.in_dim = m->in_dim; a.in_pad = m->in_pad;
  a.o_width = m->o_width; a.o_tiles = m->o_tiles; a.out_dim = m->out_dim;
  if (a.n_rows <= 0) return CMBPO_OK;

  if (g_split_path == CMBPO_ENS_SPLIT_F16 && head == CMBPO_HEAD_PROB && cmbpo_internal_h3_eligible(m) && a.n_rows >= g_h3_min_rows)
    return cmbpo_internal_launch_h3(m, a, s);
  if (g_split_path == CMBPO_ENS_SPLIT_F16 && head == CMBPO_HEAD_GAUSS_PI && cmbpo_internal_policy_f16_eligible(m))
    return cmbpo_internal_launch_policy_f16(m, a, s);      // the actor on the same arithmetic (policy_f16.hip)
  if (head == CMBPO_HEAD_PROB && H == 512 && m->act == CMBPO_ACT_SWISH && m->o_tiles <= 4 && m->in_pad <= 64 && g_split_path)
    return cmbpo_internal_launch_split(m, a, s);
  // (the critics' split kernel pays from ~30 k rows: below that a launch is one round of items and an item's latency
  // counts -- 31 us for its three members in sequence on two waves against 19 us for the fp32 kernel's 32-row items)
  if (head == CMBPO_HEAD_DETMEAN && H == 128 && m->act == CMBPO_ACT_SWISH && m->o_width == 1 && m->in_pad <= 64 && E <= 8 &&
      g_split_path && a.n_rows >= 32768)
    return cmbpo_internal_launch_critic_split(m, a, s);
  const int BT = (H == 512 && g_block_rows == 64 && head != CMBPO_HEAD_TRAIN) ? 2 : 1;
  const int BB = 32 * BT;
  const int tiles = cmbpo_ceil_div(a.n_rows, BB);
  const int grid_y = cmbpo_ceil_div(E, a.e_chunk);
  const size_t hbuf = (size_t)H * BB * 4;
  const size_t red = ((size_t)kWaves * m->o_tiles * 32 * (BB + 1) * 4 + 15) / 16 * 16;
  const size_t lds = (hbuf > red ? hbuf : red) + (size_t)m->in_pad * BB * 4 + (2 * H + m->o_tiles * 32 + 3 * m->out_dim) * 4 + BB * 4;
  CMBPO_REQUIRE(lds + g_lds_pad <= 160 * 1024, "ens_mlp: LDS budget exceeded (%zu B)", lds + g_lds_pad);

#define CMBPO_LAUNCH(HID_, BT_, ACT_, HEAD_) \
  return launch_one<HID_, BT_, ACT_, HEAD_>(a, tiles, grid_y, lds + g_lds_pad, s)
  if (head == CMBPO_HEAD_TRAIN && m->act == CMBPO_ACT_SWISH) {
    if (H == 512) CMBPO_LAUNCH(512, 1, CMBPO_ACT_SWISH, CMBPO_HEAD_TRAIN);
    if (H == 128) CMBPO_LAUNCH(128, 1, CMBPO_ACT_SWISH, CMBPO_HEAD_TRAIN);
    if (H == 256) CMBPO_LAUNCH(256, 1, CMBPO_ACT_SWISH, CMBPO_HEAD_TRAIN);
  } else if (head == CMBPO_HEAD_PROB && m->act == CMBPO_ACT_SWISH) {
    if (H == 512 && BT == 1) CMBPO_LAUNCH(512, 1, CMBPO_ACT_SWISH, CMBPO_HEAD_PROB);
    if (H == 512 && BT == 2) CMBPO_LAUNCH(512, 2, CMBPO_ACT_SWISH, CMBPO_HEAD_PROB);
    if (H == 128) CMBPO_LAUNCH(128, 1, CMBPO_ACT_SWISH, CMBPO_HEAD_PROB);
    if (H == 256) CMBPO_LAUNCH(256, 1, CMBPO_ACT_SWISH, CMBPO_HEAD_PROB);
  } else if (head == CMBPO_HEAD_DETMEAN && m->act == CMBPO_ACT_SWISH) {
    if (H == 128) CMBPO_LAUNCH(128, 1, CMBPO_ACT_SWISH, CMBPO_HEAD_DETMEAN);
    if (H == 256) CMBPO_LAUNCH(256, 1, CMBPO_ACT_SWISH, CMBPO_HEAD_DETMEAN);   // (configs/baseconfig/base.py:16,19: the default critic width)
    if (H == 512) CMBPO_LAUNCH(512, 1, CMBPO_ACT_SWISH, CMBPO_HEAD_DETMEAN);
  } else if (head == CMBPO_HEAD_GAUSS_PI && m->act == CMBPO_ACT_TANH) {
    if (H == 128) CMBPO_LAUNCH(128, 1, CMBPO_ACT_TANH, CMBPO_HEAD_GAUSS_PI);
    if (H == 256) CMBPO_LAUNCH(256, 1, CMBPO_ACT_TANH, CMBPO_HEAD_GAUSS_PI);    // (base.py:7: the default policy width)
    if (H == 512) CMBPO_LAUNCH(512, 1, CMBPO_ACT_TANH, CMBPO_HEAD_GAUSS_PI);
  }
#undef CMBPO_LAUNCH
  cmbpo_set_error("ens_mlp: no kernel for hidden=%d act=%d head=%d", H, m->act, head);
  return CMBPO_EINVAL;
}

}  // namespace

int cmbpo_internal_launch_mlp(cmbpo_mlp *m, MlpKernelArgs &a, hipStream_t s, int head_override) {
  return launch_mlp(m, a, s, head_override);
}

extern "C" int cmbpo_set_ens_matrix_path(int path) {
  CMBPO_REQUIRE(path == CMBPO_ENS_FP32 || path == CMBPO_ENS_SPLIT_BF16 || path == CMBPO_ENS_SPLIT_F16,
                "cmbpo_set_ens_matrix_path: 0 (fp32 MFMA), 1 (six bf16 terms) or 2 (three f16 terms)");
  g_split_path = path;
  return CMBPO_OK;
}

extern "C" int cmbpo_set_ens_f16_min_rows(int rows) {
  CMBPO_REQUIRE(rows >= 0, "cmbpo_set_ens_f16_min_rows: rows >= 0");
  g_h3_min_rows = rows;
  return CMBPO_OK;
}

extern "C" int cmbpo_get_ens_matrix_path(void) { return g_split_path; }

extern "C" int cmbpo_set_dispatch_mode(int mode) {
  CMBPO_REQUIRE(mode >= 0 && mode <= 2, "cmbpo_set_dispatch_mode: 0 (per-item), 1 (persistent static), 2 (persistent dynamic)");
  g_dispatch_mode = mode;
  return CMBPO_OK;
}

extern "C" int cmbpo_debug_set_lds_pad(int bytes) {
  g_lds_pad = bytes;
  return CMBPO_OK;
}

extern "C" int cmbpo_set_stagger(int sleeps) {
  CMBPO_REQUIRE(sleeps >= 0 && sleeps <= 64, "cmbpo_set_stagger: 0..64");
  g_stagger = sleeps;
  return CMBPO_OK;
}

extern "C" int cmbpo_set_block_rows(int rows) {
  CMBPO_REQUIRE(rows == 32 || rows == 64, "cmbpo_set_block_rows: 32 or 64");
  g_block_rows = rows;
  return CMBPO_OK;
}

extern "C" int cmbpo_ens_forward(cmbpo_mlp_t *m, const float *d_obs, int obs_dim,
                                 const float *d_act, int act_dim, const int32_t *d_row_idx,
                                 const int32_t *d_n_rows, int n_rows, int ld_rows,
                                 float *d_mean, float *d_var, void *stream) {
  CMBPO_REQUIRE(m != nullptr, "cmbpo_ens_forward: handle is NULL");
  if (!m->loaded) { cmbpo_set_error("cmbpo_ens_forward: weights not loaded"); return CMBPO_ESTATE; }
  CMBPO_REQUIRE(m->head == CMBPO_HEAD_PROB, "cmbpo_ens_forward: handle is not HEAD_PROB");
  CMBPO_REQUIRE(obs_dim >= 1 && act_dim >= 0 && obs_dim + act_dim == m->in_dim,
                "cmbpo_ens_forward: obs_dim %d + act_dim %d != in_dim %d", obs_dim, act_dim, m->in_dim);
  CMBPO_REQUIRE(d_obs && (act_dim == 0 || d_act) && d_mean && d_var, "cmbpo_ens_forward: NULL buffer");
  CMBPO_REQUIRE(n_rows >= 0 && ld_rows >= n_rows, "cmbpo_ens_forward: n_rows %d / ld_rows %d", n_rows, ld_rows);
  MlpKernelArgs a{};
  a.obs = d_obs; a.obs_dim = obs_dim; a.act = d_act; a.act_dim = act_dim;
  a.row_idx = d_row_idx; a.n_rows_dev = d_n_rows; a.n_rows = n_rows; a.ld_rows = ld_rows;
  a.out0 = d_mean; a.out1 = d_var;
  return launch_mlp(m, a, (hipStream_t)stream);
}

extern "C" int cmbpo_ens_predict_mean(cmbpo_mlp_t *m, const float *d_obs, int obs_dim,
                                      const int32_t *d_row_idx, const int32_t *d_n_rows,
                                      int n_rows, float *d_out, void *stream) {
  CMBPO_REQUIRE(m != nullptr, "cmbpo_ens_predict_mean: handle is NULL");
  if (!m->loaded) { cmbpo_set_error("cmbpo_ens_predict_mean: weights not loaded"); return CMBPO_ESTATE; }
  CMBPO_REQUIRE(m->head == CMBPO_HEAD_DETMEAN, "cmbpo_ens_predict_mean: handle is not HEAD_DETMEAN");
  CMBPO_REQUIRE(obs_dim == m->in_dim, "cmbpo_ens_predict_mean: obs_dim %d != in_dim %d", obs_dim, m->in_dim);
  CMBPO_REQUIRE(d_obs && d_out && n_rows >= 0, "cmbpo_ens_predict_mean: bad buffer / n_rows");
  MlpKernelArgs a{};
  a.obs = d_obs; a.obs_dim = obs_dim; a.act = nullptr; a.act_dim = 0;
  a.row_idx = d_row_idx; a.n_rows_dev = d_n_rows; a.n_rows = n_rows; a.ld_rows = n_rows;
  a.out0 = d_out;
  return launch_mlp(m, a, (hipStream_t)stream);
}

extern "C" int cmbpo_policy_forward(cmbpo_mlp_t *m, const float *d_obs, int obs_dim,
                                    const float *d_eps, const int32_t *d_row_idx,
                                    const int32_t *d_n_rows, int n_rows, float *d_pi,
                                    float *d_logp, float *d_mu, float *d_logstd, void *stream) {
  CMBPO_REQUIRE(m != nullptr, "cmbpo_policy_forward: handle is NULL");
  if (!m->loaded) { cmbpo_set_error("cmbpo_policy_forward: weights not loaded"); return CMBPO_ESTATE; }
  CMBPO_REQUIRE(m->head == CMBPO_HEAD_GAUSS_PI, "cmbpo_policy_forward: handle is not HEAD_GAUSS_PI");
  CMBPO_REQUIRE(obs_dim == m->in_dim, "cmbpo_policy_forward: obs_dim %d != in_dim %d", obs_dim, m->in_dim);
  CMBPO_REQUIRE(d_obs && d_eps && d_pi && d_logp && d_mu && d_logstd && n_rows >= 0,
                "cmbpo_policy_forward: bad buffer / n_rows");
  MlpKernelArgs a{};
  a.obs = d_obs; a.obs_dim = obs_dim; a.act = nullptr; a.act_dim = 0; a.eps = d_eps;
  a.row_idx = d_row_idx; a.n_rows_dev = d_n_rows; a.n_rows = n_rows; a.ld_rows = n_rows;
  a.out0 = d_pi; a.out1 = d_logp; a.out2 = d_mu; a.out3 = d_logstd;
  return launch_mlp(m, a, (hipStream_t)stream);
}
