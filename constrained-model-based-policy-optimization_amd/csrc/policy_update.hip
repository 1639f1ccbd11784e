// Fused policy-MLP kernels for the CPO trust-region update (gfx950, fp32-exact MFMA).
//
// Replaces the TF graph fragments of the reference that every CG iteration / line-search trial
// re-evaluates with a full feed of the batch:
//   policies/cpo_policy.py:522-543   ratio, surr_adv, surr_cost, pi_loss, cur_cret_avg
//   policies/cpo_policy.py:549,555   flat_g = grad(pi_loss), flat_b = grad(surr_cost)     (MODE_GRAD)
//   utilities/trust_region.py:15-19  hessian_vector_product(d_kl, pi_params)             (MODE_FVP)
//   policies/cpo_policy.py:278-280   [d_kl, pi_loss, surr_cost] at trial parameters       (MODE_EVAL)
//   network/ac_network.py:26-55,99-123  tanh MLP, gaussian_likelihood, gaussian_kl
//
// One persistent workgroup per CU walks tiles of 32 samples (samples on the MFMA lanes).  Per tile:
// forward chain (as ens_mlp.hip), then either the JVP chain + Fisher cotangent (FVP) or the loss
// cotangent (GRAD), then the backward chain with the transposed-packed weights, and finally the
// weight-gradient products sum_b x[b]^T d[b] as MFMAs whose K dimension is the sample index.  Weight
// gradients stay in registers across all tiles of the workgroup and are written once, as the workgroup's
// partial vector, which a small kernel adds up in workgroup order (no float atomics: results are bitwise
// reproducible); nothing but the batch is read from HBM.
//
// The FVP is the Gauss-Newton / Fisher form  mean_n J^T diag(1/(var_old+eps)) J v  (+ the log_std
// diagonal): it equals TF's double back-prop of d_kl whenever mu_old == mu(theta), which holds during
// CG because the buffer's mu / log_std were produced by the current policy.
#include "common.h"
#include "f16_split.h"
#include "mfma_tile.h"
#include "row_tile.h"

#include <math.h>

namespace {

constexpr int kThreads = 256;
constexpr int HID = 128;
constexpr int RS = HID + 4;     // row stride of the [b][n] images (conflict-free ds_write_b128 / ds_read_b32)
constexpr int RED_LD = BB + 1;
// row stride of the images at hidden width H (RS / 4 odd; at 256 the split-K reduction image -- 8 waves x 32 x 33 floats -- must
// still fit one image: 32 x 268)
__host__ __device__ constexpr int pi_rs(int H) { return H == 128 ? 132 : H + 12; }

enum { MODE_EVAL = 0, MODE_GRAD = 1, MODE_FVP = 2 };

#ifdef CMBPO_STAMPS   // diagnostic build only (tools/probe_pi_stamps.py): where wave 0 of each workgroup spends its cycles
#define PI_STAMP(k)                                                         \
  do {                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                      \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();          \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                     \
    t_acc[k] += t_now - t_last;                                             \
    t_last = t_now;                                                         \
    __builtin_amdgcn_sched_barrier(0);                                      \
  } while (0)
unsigned long long *g_pi_stamps = nullptr;
#else
#define PI_STAMP(k) do { } while (0)
#endif

struct PiDims {
  int D, A, in_pad, kg0, a_kpad, kga, n_it;  // n_it = row tiles of W0 (1 or 2)
  int oW0, ob0, oW1, ob1, oW2, ob2, ols, P;
  int H;         // hidden width: 128 (both matrix paths) or 256 (fp32 MFMAs)
};

struct PiPack {          // device pointers into the handle's blob
  const f32x4 *F0, *F1, *F2, *B1, *B2;
  const float *b0, *b1, *b2, *ls;
  // the three-term f16 path (see "f16 images" below): every matrix as the A operand of its forward (F) and backward (B)
  // product, lifted by one power of two per matrix: lift[L_*] and, 8 floats on, its inverse
  const u32x4 *F0h, *F1h, *F2h, *B1h, *B2h;
  const float *lift;
  const float *mx;     // {max |W0|, max |b0|, max |W2|, 0}
};

struct PiArgs {
  PiDims d;
  PiPack w;   // parameters
  PiPack v;   // packed direction (FVP only)
  int n;
  const float *obs, *act, *adv, *cadv, *logp_old, *cost, *mu_old, *ls_old;
  int which;        // GRAD: 0 -> cotangent of pi_loss (-adv), 1 -> cotangent of surr_cost (+cadv)
  float *vec;       // [P]  raw sums (not divided by n), written by reduce_parts_kernel
  float *part;      // [grid][part_ld] per-workgroup partial sums
  int part_ld;
  double *sums;     // [8]  n, sum ratio*adv, sum ratio*cadv, sum kl, sum cost
  const f32x4 *cache_r;   // FVP: saved hidden activations of this batch ([tile][h1 | h2][32][128]) or NULL (recompute)
  f32x4 *cache_w;         // GRAD: where to save them, or NULL
  unsigned long long *stamps;
};

// ---- packing (device side, so set_params / FVP directions never visit the host) -------------------
// One launch packs every piece of a flat parameter vector.  Matrix pieces:
//   dst[off + ((nt*kg + g)*64 + lane)*4 + s] = src[n*sn + k*sk], n = nt*32 + (lane&31) < n_lim, k = 8g + 4(lane>>5) + s < k_lim
// vector pieces (kg == 0): dst[off + i] = i < n_lim ? src[i] : 0.
struct PackPiece {
  int dst_off, src_off, n_lim, k_lim, sn, sk, kg, count;   // count = floats written (incl. zero padding)
};
// ---- f16 images ------------------------------------------------------------------------------------------
// pi_kernel_h runs every matrix product as three f16 MFMAs on two-piece operands (f16_split.h).  The matrix side is split
// once per pack: image[n-tile][k-slab][piece 2][lane 64] of 8 halves, row n = 32 nt + (lane & 31), k = 16 slab +
// 8 (lane >> 5) + e, every element lifted by S = pow2_lift(max |A|) (one power of two per matrix: an element keeps its
// full 22 bits while it is within 2^-12 of the matrix maximum, and loses them one by one below); rows / columns past the
// matrix are zero.
enum { L_F0 = 0, L_F1, L_F2, L_B1, L_B2 };
struct H16Img {
  int src_off, sn, sk, n_lim, k_lim;   // A[n][k] = flat[src_off + n sn + k sk], n < n_lim, k < k_lim
  int n_tiles, slabs;
  int dst_off;                         // u32x4 from the pack's f16 block
  int lift_idx;                        // L_*
};
// layout of a pack's f16 block (u32x4 units); the input layer is laid out for its largest width (64 inputs, 4 slabs)
constexpr int H16_F0 = 0, H16_F1 = H16_F0 + 4 * 4 * 128, H16_B1 = H16_F1 + 4 * 8 * 128, H16_F2 = H16_B1 + 4 * 8 * 128,
              H16_B2 = H16_F2 + 8 * 128, H16_INV = H16_B2 + 4 * 2 * 128;
constexpr int INV_LIFT = 0, INV_MX = 16, INV_FLOATS = 20;   // lift[8] | 1 / lift[8] | {max |W0|, max |b0|, max |W2|, 0}
constexpr int H16_PACK = H16_INV + INV_FLOATS / 4;

struct PackPlan {
  PackPiece piece[9];
  int total;
  int nb32;      // workgroups of the fp32 part; then one for the maxima, then the f16 images
  int oW0, nW0, ob0, nb0, oW2, nW2;   // the pieces whose largest magnitudes bound dh1 and delta2
  H16Img img[5];
  int img_blk[6];                // first workgroup of each image, relative to nb32 + 1
  int n_img;
};

__device__ __forceinline__ void pack_f16_part(u32x4 *dst16, const float *flat, const H16Img &im, int blk) {
  // every workgroup of an image first finds the matrix's largest magnitude (16 K elements from L2: cheaper than a launch)
  // (the matrix is one contiguous run of the flat vector whichever way the image walks it, and a maximum has no order)
  __shared__ float red[4];
  const float *src = flat + im.src_off;
  const int n4 = im.n_lim * im.k_lim / 4;     // (every matrix of the policy is a multiple of 128 floats, 16-byte aligned)
  const f32x4 *src4 = reinterpret_cast<const f32x4 *>(src);
  float mx = 0.0f;
#pragma unroll 16                             // all of a thread's loads in flight at once (<= 16 for 128 x 128)
  for (int e = threadIdx.x; e < n4; e += 256) {
    const f32x4 q = src4[e];
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(q[0]), fabsf(q[1]))), fmaxf(fabsf(q[2]), fabsf(q[3])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  const float S = pow2_lift(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
  if (blk == 0 && threadIdx.x == 0) {
    float *lift = reinterpret_cast<float *>(dst16 + H16_INV) + INV_LIFT;
    lift[im.lift_idx] = S;
    lift[8 + im.lift_idx] = 1.0f / S;
  }
  const int t = blk * 256 + threadIdx.x;
  const int lane = t & 63, s = (t >> 6) % im.slabs, nt = (t >> 6) / im.slabs;
  if (nt >= im.n_tiles) return;
  const int n = nt * 32 + (lane & 31), k0 = 16 * s + 8 * (lane >> 5);
  const float *row = src + (size_t)n * im.sn;
  union { _Float16 hv[8]; u32x4 q; } p1, p2;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = (n < im.n_lim && k0 + e < im.k_lim) ? row[(size_t)(k0 + e) * im.sk] * S : 0.0f;
    split_h(v, p1.hv[e], p2.hv[e]);
  }
  u32x4 *img = dst16 + im.dst_off + (size_t)((nt * im.slabs + s) * 2) * 64 + lane;
  img[0] = p1.q;
  img[64] = p2.q;
}

__device__ __forceinline__ void pack_max_part(u32x4 *dst16, const float *flat, const PackPlan &plan) {
  __shared__ float red[3][256];
  float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f;
  for (int i = threadIdx.x; i < plan.nW0; i += 256) m0 = fmaxf(m0, fabsf(flat[plan.oW0 + i]));
  for (int i = threadIdx.x; i < plan.nb0; i += 256) m1 = fmaxf(m1, fabsf(flat[plan.ob0 + i]));
  for (int i = threadIdx.x; i < plan.nW2; i += 256) m2 = fmaxf(m2, fabsf(flat[plan.oW2 + i]));
  red[0][threadIdx.x] = m0; red[1][threadIdx.x] = m1; red[2][threadIdx.x] = m2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = fmaxf(red[k][threadIdx.x], red[k][threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x < 4)
    reinterpret_cast<float *>(dst16 + H16_INV)[INV_MX + threadIdx.x] = threadIdx.x < 3 ? red[threadIdx.x][0] : 0.0f;
}

__global__ void pack_all_kernel(float *dst, const float *flat, const PackPlan plan, u32x4 *dst16) {
  if ((int)blockIdx.x == plan.nb32) {
    pack_max_part(dst16, flat, plan);
    return;
  }
  if ((int)blockIdx.x > plan.nb32) {
    const int b = blockIdx.x - plan.nb32 - 1;
    int m = 0;
    while (m + 1 < plan.n_img && b >= plan.img_blk[m + 1]) ++m;
    pack_f16_part(dst16, flat, plan.img[m], b - plan.img_blk[m]);
    return;
  }
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= plan.total) return;
  int q = 0;
  while (q < 8 && idx >= plan.piece[q].count) {
    idx -= plan.piece[q].count;
    ++q;
  }
  const PackPiece pc = plan.piece[q];
  const float *src = flat + pc.src_off;
  float v;
  if (pc.kg == 0) {
    v = (idx < pc.n_lim) ? src[idx] : 0.0f;
  } else {
    const int s = idx & 3, lane = (idx >> 2) & 63, g = (idx >> 8) % pc.kg, nt = (idx >> 8) / pc.kg;
    const int n = nt * 32 + (lane & 31), k = 8 * g + 4 * (lane >> 5) + s;
    v = (n < pc.n_lim && k < pc.k_lim) ? src[(size_t)n * pc.sn + (size_t)k * pc.sk] : 0.0f;
  }
  dst[pc.dst_off + idx] = v;
}

// Every activation image lives in LDS ONCE, in the row layout [sample][RS] (RS / 4 odd): it is the B operand of
// the forward / JVP / backward chains (conflict-free 16-B reads, mfma_layer<.., ROWS = true>) and the A / B
// operand of the weight-gradient MFMAs (conflict-free 4-B reads).  76.8 KB per workgroup -> two per CU.
template <int MODE, int N_IT, int HIDT>
__global__ __launch_bounds__(2 * HIDT, HIDT == 128 ? 2 : 1) void pi_kernel(const PiArgs p) {
  // HIDT = the hidden width (128: the shipped policies; 256: configs/baseconfig/base.py's default): a wave per 32 hidden
  // units, 2 HIDT threads, images of HIDT + pad floats a row
  constexpr int kThreads = 2 * HIDT, HID = HIDT, KGH = HIDT / 8, RS = pi_rs(HIDT), NW = HIDT / 32;
  constexpr int AS = kThreads / 32, NITE = 32 / AS;     // element phase: a = tid / 32 + AS it, it < NITE
  extern __shared__ f32x4 smem4[];
  const PiDims d = p.d;
  float *sm = reinterpret_cast<float *>(smem4);
  const int XS = d.in_pad + 4;
  float *xR = sm;                       // [BB][XS]
  float *h1R = xR + BB * XS;            // [BB][RS]
  float *h2R = h1R + BB * RS;
  float *u1R = h2R + BB * RS;           // dh1, then the split-K reduction image (4*32*33 floats), then delta1
  float *u2R = u1R + BB * RS;           // dh2, then delta2
  float *wR = u2R + BB * RS;            // [BB][36] cotangent on mu; GRAD / EVAL scratch before that
  float *red = u1R;
  float *d1R = u1R;
  float *d2R = u2R;
  const f32x4 *xR4 = reinterpret_cast<const f32x4 *>(xR), *h1R4 = reinterpret_cast<const f32x4 *>(h1R),
              *h2R4 = reinterpret_cast<const f32x4 *>(h2R), *u1R4 = reinterpret_cast<const f32x4 *>(u1R),
              *u2R4 = reinterpret_cast<const f32x4 *>(u2R), *wR4 = reinterpret_cast<const f32x4 *>(wR);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int n_tiles = (p.n + BB - 1) / BB;
  constexpr int IMG4 = BB * HID / 4;    // float4s of one dense [32][128] activation image
  const bool cached = (MODE == MODE_FVP) && p.cache_r != nullptr;

  // persistent accumulators
  f32x16 gW1[NW], gW0[N_IT], gW2;
#pragma unroll
  for (int t = 0; t < NW; ++t) zero(gW1[t]);
#pragma unroll
  for (int t = 0; t < N_IT; ++t) zero(gW0[t]);
  zero(gW2);
  float gbias = 0.0f;   // thread n < 128: d/d b1[n]; thread 128 + n: d/d b0[n] (column sums of the delta images)
  float gb2p[NITE] = {}, glsp[NITE] = {};   // per (a = tid/32 + AS*it) partials over this thread's b
  double s_n = 0, s_ra = 0, s_rc = 0, s_kl = 0, s_cost = 0;

  // zero the padded cotangent columns (a in [A, 36)) and the padded input columns (k in [D, in_pad)) once: nothing
  // else ever writes them
  for (int i = tid; i < BB * 36; i += kThreads) wR[i] = 0.0f;
  for (int i = tid; i < BB * XS; i += kThreads) xR[i] = 0.0f;
  __syncthreads();

  // The tile's observations are one contiguous block of 32 D floats: each thread keeps its <= 4 N_IT of them in
  // registers, fetched one tile ahead (behind the weight-gradient MFMAs), so that staging never waits for HBM.
  // (The 33..64-wide input variant has no registers to spare for it and stages straight from HBM.)
  constexpr bool PRE = (N_IT == 1);
  float xpre[4 * N_IT];
  auto fetch_x = [&](int t) {
    if constexpr (!PRE) return;
    const size_t base = (size_t)t * BB * d.D;
    const int lim = min(BB, p.n - t * BB) * d.D;      // rows past the batch end read as zeros
#pragma unroll
    for (int q = 0; q < 4 * N_IT; ++q) {
      const int i = tid + q * kThreads;
      xpre[q] = (i < lim) ? p.obs[base + i] : 0.0f;
    }
  };
  if ((int)blockIdx.x < n_tiles) fetch_x(blockIdx.x);
  // Fisher cotangent: its tile-invariant factors per action column, computed once (LDS: no registers to spare)
  __shared__ float c_b2[32], c_e2[32];
  if constexpr (MODE == MODE_FVP) {
    if (tid < 32) {
      c_b2[tid] = (tid < d.A) ? p.v.b2[tid] : 0.0f;
      c_e2[tid] = (tid < d.A) ? 2.0f * expf(2.0f * p.w.ls[tid]) : 0.0f;
    }
    __syncthreads();
  }

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int row0 = tile * BB;
    // ---- stage x in the row layout ---------------------------------------------------------------------
    if constexpr (PRE) {
#pragma unroll
      for (int q = 0; q < 4 * N_IT; ++q) {
        const int i = tid + q * kThreads;
        if (i < BB * d.D) {
          const int b = i / d.D, k = i - b * d.D;
          xR[b * XS + k] = xpre[q];
        }
      }
    } else {
      for (int i = tid; i < BB * d.D; i += kThreads) {
        const int b = i / d.D, k = i - b * d.D;
        xR[b * XS + k] = (row0 + b < p.n) ? p.obs[(size_t)row0 * d.D + i] : 0.0f;
      }
    }
    if constexpr (MODE == MODE_EVAL) {   // (the other modes fetch ahead later, next to their weight-gradient MFMAs)
      if (tile + (int)gridDim.x < n_tiles) fetch_x(tile + gridDim.x);
    }
    // the Fisher cotangent's log_std_old (first 8 action columns): requested now, used in the element phase, so
    // that its HBM latency passes behind the JVP chain instead of between two barriers
    float lso0 = 0.0f;
    float e_act0 = 0.0f, e_mu0 = 0.0f, e_logp = 0.0f, e_adv = 0.0f, e_cadv = 0.0f;   // same for the loss terms
    {
      const int er0 = row0 + (tid & 31), a0 = tid >> 5;
      if (er0 < p.n) {
        if constexpr (MODE != MODE_FVP) {
          e_logp = p.logp_old[er0];
          e_adv = p.adv[er0];
          e_cadv = p.cadv[er0];
        }
        if (a0 < d.A) {
          if constexpr (MODE != MODE_GRAD) lso0 = p.ls_old[(size_t)er0 * d.A + a0];
          if constexpr (MODE != MODE_FVP) e_act0 = p.act[(size_t)er0 * d.A + a0];
          if constexpr (MODE == MODE_EVAL) e_mu0 = p.mu_old[(size_t)er0 * d.A + a0];
        }
      }
    }
    f32x16 acc[1][1];
    if (cached) {
      // ---- h1, h2 as cmbpo_pi_loss_grad left them (same parameters, same batch): identical bits to recomputing
      // them, without the forward chain's 80 MFMAs and 8 K tanh per tile
      const f32x4 *src = p.cache_r + (size_t)tile * (2 * IMG4);
      f32x4 t1[4], t2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        t1[i] = src[tid + kThreads * i];
        t2[i] = src[IMG4 + tid + kThreads * i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + kThreads * i, b = idx / (HID / 4), c = idx % (HID / 4);
        reinterpret_cast<f32x4 *>(h1R)[b * (RS / 4) + c] = t1[i];
        reinterpret_cast<f32x4 *>(h2R)[b * (RS / 4) + c] = t2[i];
      }
      __syncthreads();
    } else {
      __syncthreads();
      // ---- forward ----------------------------------------------------------------------------------
      acc[0][0] = load_bias(p.w.b0, wave * 32, lane);
      mfma_layer<1, 1, true>(p.w.F0 + (size_t)wave * d.kg0 * 64, 0, 0, d.kg0, xR4, lane, acc, XS);
      {
        f32x16 hv;
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = cmbpo_fast_tanh(acc[0][0][r]);
        store_tile_R(hv, wave * 32, h1R, RS, lane);
      }
      __syncthreads();
      acc[0][0] = load_bias(p.w.b1, wave * 32, lane);
      mfma_layer<1, 1, true>(p.w.F1 + (size_t)wave * KGH * 64, 0, 0, KGH, h1R4, lane, acc, RS);
      {
        f32x16 hv;
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = cmbpo_fast_tanh(acc[0][0][r]);
        store_tile_R(hv, wave * 32, h2R, RS, lane);
      }
      __syncthreads();
      if constexpr (MODE == MODE_GRAD) {
        if (p.cache_w != nullptr) {
          f32x4 *dst = p.cache_w + (size_t)tile * (2 * IMG4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int idx = tid + kThreads * i, b = idx / (HID / 4), c = idx % (HID / 4);
            dst[idx] = h1R4[b * (RS / 4) + c];
            dst[IMG4 + idx] = h2R4[b * (RS / 4) + c];
          }
        }
      }
    }

    if constexpr (MODE == MODE_FVP) {
      // ---- JVP chain: dh1 = (1-h1^2)(x dW0 + db0) ; dh2 = (1-h2^2)(dh1 W1 + h1 dW1 + db1) -----------
      acc[0][0] = load_bias(p.v.b0, wave * 32, lane);
      mfma_layer<1, 1, true>(p.v.F0 + (size_t)wave * d.kg0 * 64, 0, 0, d.kg0, xR4, lane, acc, XS);
      {
        const f32x16 hh = load_tile_R(h1R, RS, wave * 32, lane);
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = (1.0f - hh[r] * hh[r]) * acc[0][0][r];
        store_tile_R(o, wave * 32, u1R, RS, lane);
      }
      // the h1 dW1 half does not need dh1: it runs ahead of the barrier and absorbs the waves' skew
      acc[0][0] = load_bias(p.v.b1, wave * 32, lane);
      mfma_layer<1, 1, true>(p.v.F1 + (size_t)wave * KGH * 64, 0, 0, KGH, h1R4, lane, acc, RS);
      __syncthreads();
      mfma_layer<1, 1, true>(p.w.F1 + (size_t)wave * KGH * 64, 0, 0, KGH, u1R4, lane, acc, RS);
      {
        const f32x16 hh = load_tile_R(h2R, RS, wave * 32, lane);
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = (1.0f - hh[r] * hh[r]) * acc[0][0][r];
        store_tile_R(o, wave * 32, u2R, RS, lane);
      }
      // dmu = dh2 W2 + h2 dW2 (+ db2): K split over the 4 waves; the h2 dW2 half ahead of the barrier
      zero(acc[0][0]);
      mfma_layer<1, 1, true>(p.v.F2, 0, wave * 4, wave * 4 + 4, h2R4, lane, acc, RS);
      __syncthreads();   // dh2 complete; every wave is done reading dh1 (u1R becomes the reduction image)
      mfma_layer<1, 1, true>(p.w.F2, 0, wave * 4, wave * 4 + 4, u2R4, lane, acc, RS);
    } else {
      // mu = h2 W2 (+ b2): K split over the 4 waves
      zero(acc[0][0]);
      mfma_layer<1, 1, true>(p.w.F2, 0, wave * 4, wave * 4 + 4, h2R4, lane, acc, RS);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int a = (r & 3) + 8 * (r >> 2) + 4 * h;
      red[(wave * 32 + a) * RED_LD + j] = acc[0][0][r];
    }
    __syncthreads();

    // ---- element phase over (a, b): this thread owns b = tid & 31, a = tid/32 + 8*it ------------------
    const int eb = tid & 31, er = row0 + eb;
    const bool valid = er < p.n;
    float z_[NITE], mu_[NITE];
    float logp_part = 0.0f;
#pragma unroll
    for (int it = 0; it < NITE; ++it) {
      const int a = (tid >> 5) + AS * it;
      z_[it] = mu_[it] = 0.0f;
      if (a < d.A) {
        float m = red[(0 * 32 + a) * RED_LD + eb];
#pragma unroll
        for (int wv = 1; wv < NW; ++wv) m += red[(wv * 32 + a) * RED_LD + eb];      // (in wave order)
        if constexpr (MODE == MODE_FVP) {
          m += c_b2[a];
          float cot = 0.0f;
          if (valid) {
            // d2 KL / d mu^2 = 1 / (exp(2 ls_old) + eps)   (network/ac_network.py:52-53)
            const float lso = (it == 0) ? lso0 : p.ls_old[(size_t)er * d.A + a];
            const float v1 = expf(2.0f * lso) + 1e-8f;
            cot = m / v1;
            glsp[it] += c_e2[a] / v1;                           // d2 KL / d log_std^2 = 2 exp(2 ls) / (...)
          }
          wR[eb * 36 + a] = cot;
          gb2p[it] += cot;
        } else {
          m += p.w.b2[a];
          mu_[it] = m;
          float term = 0.0f;
          if (valid) {
            const float ls = p.w.ls[a];
            const float sd = expf(ls) + 1e-8f;
            const float z = (((it == 0) ? e_act0 : p.act[(size_t)er * d.A + a]) - m) / sd;
            z_[it] = z;
            term = -0.5f * (z * z + 2.0f * ls + 1.8378770664093453f);   // gaussian_likelihood :46-48
          }
          wR[eb * 36 + a] = term;   // scratch: per-(b, a) log-likelihood terms
        }
      }
    }
    if constexpr (MODE != MODE_FVP) {
      __syncthreads();
      float logp = logp_part;
      for (int a = 0; a < d.A; ++a) logp += wR[eb * 36 + a];
      const float ratio = valid ? expf(logp - e_logp) : 0.0f;                // cpo_policy.py:522
      const float adv = e_adv, cadv = e_cadv;                                // (zero past the batch end)
      if (tid < 32 && valid) {
        s_n += 1.0;
        s_ra += (double)(ratio * adv);
        s_rc += (double)(ratio * cadv);
        s_cost += (double)p.cost[er];
      }
      __syncthreads();   // every thread has read the scratch terms before they are overwritten
#pragma unroll
      for (int it = 0; it < NITE; ++it) {
        const int a = (tid >> 5) + AS * it;
        if (a < d.A) {
          if constexpr (MODE == MODE_EVAL) {
            if (valid) {
              // gaussian_kl(mu, log_std, mu_old, log_std_old), ac_network.py:50-55
              const float ls = p.w.ls[a], lso = (it == 0) ? lso0 : p.ls_old[(size_t)er * d.A + a];
              const float dm = ((it == 0) ? e_mu0 : p.mu_old[(size_t)er * d.A + a]) - mu_[it];
              const float pre = 0.5f * ((dm * dm + expf(2.0f * ls)) / (expf(2.0f * lso) + 1e-8f) - 1.0f) + lso - ls;
              s_kl += (double)pre;
            }
          } else {
            const float wgt = (p.which == 0) ? -adv : cadv;
            const float ls = p.w.ls[a];
            const float sd = expf(ls);
            const float inv = 1.0f / (sd + 1e-8f);
            const float cot = wgt * ratio * z_[it] * inv;                    // d logp / d mu = z / (sd + eps)
            wR[eb * 36 + a] = cot;
            gb2p[it] += cot;
            glsp[it] += wgt * ratio * (z_[it] * z_[it] * sd * inv - 1.0f);  // d logp / d log_std
          }
        }
      }
    }
    __syncthreads();
    if constexpr (MODE == MODE_EVAL) continue;

    // ---- backward: delta2 = (W2 cot) (1-h2^2) ; delta1 = (W1 delta2) (1-h1^2) --------------------------
    zero(acc[0][0]);
    mfma_layer<1, 1, true>(p.w.B2 + (size_t)wave * d.kga * 64, 0, 0, d.kga, wR4, lane, acc, 36);
    {
      const f32x16 hh = load_tile_R(h2R, RS, wave * 32, lane);
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = (1.0f - hh[r] * hh[r]) * acc[0][0][r];
      store_tile_R(o, wave * 32, d2R, RS, lane);     // dh2 is dead (the barrier after the reduction image)
    }
    __syncthreads();
    zero(acc[0][0]);
    mfma_layer<1, 1, true>(p.w.B1 + (size_t)wave * KGH * 64, 0, 0, KGH, u2R4, lane, acc, RS);
    {
      const f32x16 hh = load_tile_R(h1R, RS, wave * 32, lane);
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = (1.0f - hh[r] * hh[r]) * acc[0][0][r];
      store_tile_R(o, wave * 32, d1R, RS, lane);     // the reduction image is dead (element phase barrier)
    }
    // ---- weight gradients: K = the tile's samples ---------------------------------------------------
    float warm = 0.0f;
    if constexpr (MODE != MODE_EVAL) {
      if (tile + (int)gridDim.x < n_tiles) fetch_x(tile + gridDim.x);
    }
    if constexpr (MODE == MODE_FVP) {
      // touch one dword of each 128-B line of the NEXT tile's saved activations: they travel HBM -> L2 behind the
      // 96 MFMAs below, and the loads at the top of the next iteration hit L2 (one VGPR instead of a 32-VGPR prefetch)
      const int nxt = tile + gridDim.x;
      if (cached && nxt < n_tiles) warm = reinterpret_cast<const float *>(p.cache_r + (size_t)nxt * (2 * IMG4))[tid * 32];
    }
    // dW1 and dW2 need delta2 / the cotangent only (complete since the previous barrier): they run while the slower
    // waves still write delta1
#pragma unroll
    for (int J = 0; J < NW; ++J) wgrad_tile(gW1[J], h1R, RS, wave * 32, d2R, RS, J * 32, lane);
    wgrad_tile(gW2, h2R, RS, wave * 32, wR, 36, 0, lane);
    __syncthreads();   // delta1 complete
#pragma unroll
    for (int t = 0; t < N_IT; ++t) wgrad_tile(gW0[t], xR, XS, 32 * t, d1R, RS, wave * 32, lane);
    {
      // bias gradients: column sums over the tile's samples (conflict-free: adjacent threads, adjacent columns)
      const float *img = (tid < HID) ? d2R : d1R;
      const int n = tid & (HID - 1);
      float sb = 0.0f;
#pragma unroll 8
      for (int b = 0; b < BB; ++b) sb += img[b * RS + n];
      gbias += sb;
    }
    asm volatile("" ::"v"(warm));   // the warm-up load must be issued, its value is not used
    __syncthreads();
  }

  // ---- flush: this workgroup's partial vector (plain stores; reduce_parts_kernel adds the partials in a fixed order)
  if constexpr (MODE != MODE_EVAL) {
    float *part = p.part + (size_t)blockIdx.x * p.part_ld;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
      for (int J = 0; J < NW; ++J) part[d.oW1 + (wave * 32 + row) * HID + J * 32 + j] = gW1[J][r];
#pragma unroll
      for (int t = 0; t < N_IT; ++t)
        if (32 * t + row < d.D) part[d.oW0 + (32 * t + row) * HID + wave * 32 + j] = gW0[t][r];
      if (j < d.A) part[d.oW2 + (wave * 32 + row) * d.A + j] = gW2[r];
    }
    part[(tid < HID ? d.ob1 : d.ob0) + (tid & (HID - 1))] = gbias;
#pragma unroll
    for (int it = 0; it < NITE; ++it) {
      const int a = (tid >> 5) + AS * it;
      const float sb = half_sum(gb2p[it]), sl = half_sum(glsp[it]);
      if (a < d.A && (tid & 31) == 0) {
        part[d.ob2 + a] = sb;
        if constexpr (MODE == MODE_GRAD) part[d.ols + a] = sl;
        else part[d.ols + a] = sl * p.v.ls[a];
      }
    }
  }
  if constexpr (MODE != MODE_FVP) {
    // tid < 32 hold the per-sample sums; s_kl is spread over every thread
    __shared__ double sd[NW];
    const double kl = wave_sum_d(s_kl);
    if (lane == 0) sd[wave] = kl;
    __syncthreads();
    if (wave == 0) {
      const double n = wave_sum_d(s_n), ra = wave_sum_d(s_ra), rc = wave_sum_d(s_rc), c = wave_sum_d(s_cost);
      if (lane == 0) {
        atomicAdd(&p.sums[0], n);
        atomicAdd(&p.sums[1], ra);
        atomicAdd(&p.sums[2], rc);
        double klt = sd[0];
#pragma unroll
        for (int wv = 1; wv < NW; ++wv) klt += sd[wv];
        atomicAdd(&p.sums[3], klt);
        atomicAdd(&p.sums[4], c);
      }
    }
  }
}

#include "pi_kernel_f16.h"

// vec[i] = sum over workgroups of part[w][i], in workgroup order: the gradient / Fisher-vector product is bitwise
// reproducible, and cheaper than 2 n_cu workgroups x P float atomics on the same P addresses (11 M atomics for the
// 128-wide policy, which dominated the kernel at N = 50 000).
// One workgroup = 64 columns x 16 partial groups: thread (c, g) adds partials g, g + 16, ... of column c, the 16 group
// sums are combined through LDS in a fixed tree.
__global__ __launch_bounds__(1024) void reduce_parts_kernel(const float *part, int part_ld, int n_parts, int P, float *vec) {
  __shared__ float sm[16][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + c;
  float s = 0.0f;
  if (i < P)
    // (eight partials in flight per thread, added in the order of the plain loop: same bits)
    for (int w = g; w < n_parts; w += 16 * 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (w + 16 * k < n_parts) ? part[(size_t)(w + 16 * k) * part_ld + i] : 0.0f;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (w + 16 * k < n_parts) s += v[k];
    }
  sm[g][c] = s;
  __syncthreads();
  if (g == 0 && i < P) {
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = sm[k][c];
#pragma unroll
    for (int st = 1; st < 16; st <<= 1)
#pragma unroll
      for (int k = 0; k + st < 16; k += 2 * st) t[k] += t[k + st];
    vec[i] = t[0];
  }
}

// ---- one CG iteration's vector algebra on many workgroups (utilities/trust_region.py:37-44) ----------------------
// k1: hv = sum of the Fisher-vector product's per-workgroup partials; z = hv / N + damping p; per-workgroup p.z
// k2: alpha = rr / (p.z + EPS); x += alpha p; r -= alpha z; per-workgroup r.r
// k3: rr_new = r.r; p = r + (rr_new / rr) p
// Dot products are float64 sums of per-workgroup partials that EVERY workgroup adds up in the same order (no atomics,
// reproducible).  scal[0] = r.r of the current residual; k3 parks the new value in scal[1] and the next k1 (or the
// final commit) moves it, so no workgroup reads scal[0] while another one updates it.
__device__ __forceinline__ double ordered_sum(const double *part, int n, double *sm) {
  // every workgroup: the same strided partial sums, the same tree
  double s = 0.0;
  for (int w = threadIdx.x; w < n; w += blockDim.x) s += part[w];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wv] = s;
  __syncthreads();
  double t = 0.0;
  for (int k = 0; k < nw; ++k) t += sm[k];
  return t;
}

__global__ __launch_bounds__(1024) void cg_k1_kernel(const float *part, int part_ld, int n_parts, int P, float inv_n,
                                                     float damping, const float *p, float *z, double *pzp, double *scal) {
  __shared__ float sm[16][64];
  __shared__ double smd[64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + c;
  if (blockIdx.x == 0 && threadIdx.x == 0 && scal[1] >= 0.0) {
    scal[0] = scal[1];
    scal[1] = -1.0;
  }
  float s = 0.0f;
  if (i < P)
    // (eight partials in flight per thread, added in the order of the plain loop: same bits)
    for (int w = g; w < n_parts; w += 16 * 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (w + 16 * k < n_parts) ? part[(size_t)(w + 16 * k) * part_ld + i] : 0.0f;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (w + 16 * k < n_parts) s += v[k];
    }
  sm[g][c] = s;
  __syncthreads();
  double pz = 0.0;
  if (g == 0) {
    if (i < P) {
      float t[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) t[k] = sm[k][c];
#pragma unroll
      for (int st = 1; st < 16; st <<= 1)
#pragma unroll
        for (int k = 0; k + st < 16; k += 2 * st) t[k] += t[k + st];
      const float pi = p[i];
      const float zi = t[0] * inv_n + damping * pi;
      z[i] = zi;
      pz = (double)pi * (double)zi;
    }
    smd[c] = pz;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < 64; ++k) t += smd[k];
    pzp[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(256) void cg_k2_kernel(int P, int n1, const double *pzp, const float *p, const float *z,
                                                    float *x, float *r, double *rrp, const double *scal) {
  __shared__ double sm[4];
  const double pz = ordered_sum(pzp, n1, sm);
  const float rr_old = (float)scal[0];
  const float alpha = rr_old / ((float)pz + 1e-8f);
  const int i = blockIdx.x * 256 + threadIdx.x;
  double rr = 0.0;
  if (i < P) {
    x[i] += alpha * p[i];
    const float ri = r[i] - alpha * z[i];
    r[i] = ri;
    rr = (double)ri * (double)ri;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) rr += __shfl_down(rr, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = rr;
  __syncthreads();
  if (threadIdx.x == 0) rrp[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void cg_k3_kernel(int P, int n2, const double *rrp, const float *r, float *p, double *scal) {
  __shared__ double sm[4];
  const double rr = ordered_sum(rrp, n2, sm);
  const float rr_new = (float)rr, rr_old = (float)scal[0];
  const float beta = rr_new / rr_old;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < P) p[i] = r[i] + beta * p[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) scal[1] = (double)rr_new;
}

__global__ void cg_commit_kernel(double *scal) {
  if (scal[1] >= 0.0) {
    scal[0] = scal[1];
    scal[1] = -1.0;
  }
}

}  // namespace

struct cmbpo_pi {
  PiDims d;
  float *blob;       // parameters pack | direction pack
  u32x4 *blob16;     // f16 images of W1 of the two packs (H16_PACK each)
  float *parts;      // [2 n_cu][part_ld] per-workgroup partial gradients
  int part_ld;
  int last_grid;     // workgroups (= partial vectors) of the last gradient / FVP launch
  float *cg_z;       // [part_ld] z of the current CG iteration
  double *cg_part;   // [P / 64 + P / 256 + 2] per-workgroup partial dot products
  size_t pack_floats;
  size_t off_F0, off_F1, off_F2, off_B1, off_B2, off_b0, off_b1, off_b2, off_ls;
  int n_cu;
  bool has_params;
  // hidden activations saved by cmbpo_pi_loss_grad for the Fisher-vector products of the same update
  // (cmbpo_pi_keep_activations): [tile][h1 | h2][32][128] floats, grow-only
  float *act;
  size_t act_tiles;       // capacity
  bool act_keep, act_valid;
  int act_path;           // the matrix path that wrote them
  const float *act_obs;   // the batch the saved activations belong to
  int act_n;
  long act_hits;          // Fisher-vector products that read the saved activations (diagnostics)
};

namespace {

PiPack pack_ptrs(const cmbpo_pi *h, const float *base) {
  PiPack k;
  k.F0 = reinterpret_cast<const f32x4 *>(base + h->off_F0);
  k.F1 = reinterpret_cast<const f32x4 *>(base + h->off_F1);
  k.F2 = reinterpret_cast<const f32x4 *>(base + h->off_F2);
  k.B1 = reinterpret_cast<const f32x4 *>(base + h->off_B1);
  k.B2 = reinterpret_cast<const f32x4 *>(base + h->off_B2);
  k.b0 = base + h->off_b0; k.b1 = base + h->off_b1; k.b2 = base + h->off_b2; k.ls = base + h->off_ls;
  const u32x4 *b16 = h->blob16 + (base == h->blob ? 0 : H16_PACK);
  k.F0h = b16 + H16_F0; k.F1h = b16 + H16_F1; k.F2h = b16 + H16_F2; k.B1h = b16 + H16_B1; k.B2h = b16 + H16_B2;
  const float *inv = reinterpret_cast<const float *>(b16 + H16_INV);
  k.lift = inv + INV_LIFT;
  k.mx = inv + INV_MX;
  return k;
}

// 0: every product as fp32 MFMAs (round 1); 1: the 128 x 128 products as three f16 MFMAs on split operands
int g_pi_path = -1;
int pi_path() {
  if (g_pi_path < 0) {
    const char *e = getenv("CMBPO_PI_F16");
    g_pi_path = (e && e[0] == '0') ? 0 : 1;
  }
  return g_pi_path;
}

int do_pack(const cmbpo_pi *h, float *dst, const float *flat, hipStream_t s) {
  const PiDims &d = h->d;
  CMBPO_REQUIRE((reinterpret_cast<uintptr_t>(flat) & 15u) == 0, "policy pack: the flat vector must be 16-byte aligned");
  PackPlan plan;
  int q = 0, total = 0;
  auto add = [&](size_t off, int src_off, int n_lim, int k_lim, int sn, int sk, int kg, int count) {
    plan.piece[q++] = PackPiece{(int)off, src_off, n_lim, k_lim, sn, sk, kg, count};
    total += count;
  };
  // forward packs: A[i = out unit][k = in unit] = W[k][i]
  const int H = d.H, NT = H / 32, KH = H / 8;
  add(h->off_F0, d.oW0, H, d.D, 1, H, d.kg0, NT * d.kg0 * 256);
  add(h->off_F1, d.oW1, H, H, 1, H, KH, NT * KH * 256);
  add(h->off_F2, d.oW2, d.A, H, 1, d.A, KH, KH * 256);
  // backward packs: A[i = in unit][k = out unit] = W[i][k]
  add(h->off_B1, d.oW1, H, H, H, 1, KH, NT * KH * 256);
  add(h->off_B2, d.oW2, H, d.A, d.A, 1, d.kga, NT * d.kga * 256);
  add(h->off_b0, d.ob0, H, 0, 0, 0, 0, H);
  add(h->off_b1, d.ob1, H, 0, 0, 0, 0, H);
  add(h->off_b2, d.ob2, d.A, 0, 0, 0, 0, 32);
  add(h->off_ls, d.ols, d.A, 0, 0, 0, 0, 32);
  plan.total = total;
  plan.nb32 = cmbpo_ceil_div(total, 256);
  plan.oW0 = d.oW0; plan.nW0 = d.D * H; plan.ob0 = d.ob0; plan.nb0 = H; plan.oW2 = d.oW2; plan.nW2 = H * d.A;
  // f16 images: the direction pack needs the forward ones only
  const bool params = dst == h->blob;
  const int s0 = 2 * d.n_it;
  int ni = 0, blk = 0;
  auto add16 = [&](int src_off, int sn, int sk, int n_lim, int k_lim, int n_tiles, int slabs, int dst_off, int lift_idx) {
    plan.img[ni] = H16Img{src_off, sn, sk, n_lim, k_lim, n_tiles, slabs, dst_off, lift_idx};
    plan.img_blk[ni++] = blk;
    blk += cmbpo_ceil_div(n_tiles * slabs * 64, 256);
  };
  if (H == HID) {      // (the f16 kernels are written for 128 hidden units: a 256-wide policy runs the fp32 MFMAs)
  add16(d.oW0, 1, HID, HID, d.D, 4, s0, H16_F0, L_F0);          // A[unit][input] = W0[input][unit]
  add16(d.oW1, 1, HID, HID, HID, 4, 8, H16_F1, L_F1);           // A[unit][k] = W1[k][unit]
  add16(d.oW2, 1, d.A, d.A, HID, 1, 8, H16_F2, L_F2);           // A[action][k] = W2[k][action]
  }
  if (params && H == HID) {
    add16(d.oW1, HID, 1, HID, HID, 4, 8, H16_B1, L_B1);         // A[unit][k] = W1[unit][k]
    add16(d.oW2, d.A, 1, HID, d.A, 4, 2, H16_B2, L_B2);         // A[unit][action] = W2[unit][action]
  }
  plan.img_blk[ni] = blk;
  plan.n_img = ni;
  hipLaunchKernelGGL(pack_all_kernel, dim3(plan.nb32 + 1 + blk), dim3(256), 0, s, dst, flat, plan,
                     h->blob16 + (params ? 0 : H16_PACK));
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

size_t lds_bytes(const PiDims &d, bool f16) {
  const size_t f = (size_t)BB * ((f16 ? 32 * d.n_it : d.in_pad) + 4) + 4 * BB * pi_rs(d.H) + BB * 36;
  return f * sizeof(float);
}

template <class KERN>
int launch_pi_k(cmbpo_pi *h, PiArgs &a, hipStream_t s, KERN kern, size_t &attr_bytes, bool f16, bool reduce) {
  const size_t lds = lds_bytes(h->d, f16);   // (the kernels also have a few bytes of static LDS: ask for what is needed)
  if (lds > attr_bytes) {
    CMBPO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_bytes = lds;
  }
  const int tiles = cmbpo_ceil_div(a.n, BB);
  const int resident = (h->d.H == HID ? 2 : 1) * h->n_cu;   // two 76.8 KB workgroups per CU (one of 150 KB at 256 hidden units)
  const int grid = tiles < resident ? tiles : resident;
  a.part = h->parts; a.part_ld = h->part_ld;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(2 * h->d.H), lds, s, a);
  h->last_grid = grid;
  if (reduce && a.vec != nullptr)
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(cmbpo_ceil_div(h->d.P, 64)), dim3(1024), 0, s, h->parts, h->part_ld, grid,
                       h->d.P, a.vec);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

template <int MODE, int N_IT>
int launch_pi_n(cmbpo_pi *h, PiArgs &a, hipStream_t s) {
  static size_t attr[4] = {0, 0, 0, 0};   // per kernel: the dynamic LDS already granted
  if (h->d.H == 256) return launch_pi_k(h, a, s, pi_kernel<MODE, N_IT, 256>, attr[3], false, MODE != MODE_EVAL);
  if (pi_path() == 0) return launch_pi_k(h, a, s, pi_kernel<MODE, N_IT, 128>, attr[0], false, MODE != MODE_EVAL);
  if (MODE == MODE_FVP && a.cache_r != nullptr)
    return launch_pi_k(h, a, s, pi_kernel_h<MODE, N_IT, true>, attr[1], true, MODE != MODE_EVAL);
  return launch_pi_k(h, a, s, pi_kernel_h<MODE, N_IT, false>, attr[2], true, MODE != MODE_EVAL);
}

template <int MODE>
int launch_pi(cmbpo_pi *h, PiArgs &a, hipStream_t s) {
  return h->d.n_it > 1 ? launch_pi_n<MODE, 2>(h, a, s) : launch_pi_n<MODE, 1>(h, a, s);
}

int fill_args(cmbpo_pi *h, const cmbpo_pi_batch_t *b, PiArgs &a, const char *who) {
  CMBPO_REQUIRE(h != nullptr && b != nullptr, "%s: NULL handle / batch", who);
  if (!h->has_params) { cmbpo_set_error("%s: parameters not set", who); return CMBPO_ESTATE; }
  CMBPO_REQUIRE(b->obs_dim == h->d.D && b->act_dim == h->d.A, "%s: batch dims (%d,%d) != policy dims (%d,%d)", who,
                b->obs_dim, b->act_dim, h->d.D, h->d.A);
  CMBPO_REQUIRE(b->n >= 1, "%s: empty batch", who);
  CMBPO_REQUIRE(b->obs != nullptr, "%s: obs is NULL", who);
  a.d = h->d;
  a.w = pack_ptrs(h, h->blob);
  a.v = pack_ptrs(h, h->blob + h->pack_floats);
  a.n = b->n;
  a.obs = b->obs; a.act = b->act; a.adv = b->adv; a.cadv = b->cadv; a.logp_old = b->logp_old; a.cost = b->cost;
  a.mu_old = b->mu_old; a.ls_old = b->logstd_old;
  a.cache_r = nullptr; a.cache_w = nullptr;
#ifdef CMBPO_STAMPS
  a.stamps = g_pi_stamps;
#else
  a.stamps = nullptr;
#endif
  return CMBPO_OK;
}

// the saved activations a Fisher-vector product on batch b may read, or NULL
const f32x4 *act_for(const cmbpo_pi *h, const cmbpo_pi_batch_t *b) {
  if (!h->act_keep || !h->act_valid || h->act == nullptr || b->obs != h->act_obs || b->n != h->act_n ||
      h->act_path != (h->d.H == HID ? pi_path() : 0))      // (the two matrix paths keep different images)
    return nullptr;
  return reinterpret_cast<const f32x4 *>(h->act);
}

// make room for the activations of n samples; on failure the update simply recomputes them
bool act_reserve(cmbpo_pi *h, int n) {
  const size_t tiles = (size_t)cmbpo_ceil_div(n, BB);
  if (h->act != nullptr && h->act_tiles >= tiles) return true;
  h->act_valid = false;
  if (h->act) (void)hipFree(h->act);      // synchronises: nothing in flight still reads the old block
  h->act = nullptr; h->act_tiles = 0;
  // (per tile: the fp32 kernels keep two dense [32][128] images, the f16 kernels the padded LDS block -- room for the larger,
  //  plus a KiB: the last LDS-DMA piece of a block reads past its end)
  if (hipMalloc(reinterpret_cast<void **>(&h->act), tiles * 2 * BB * pi_rs(h->d.H) * sizeof(float) + 1024) != hipSuccess) {
    (void)hipGetLastError();
    h->act = nullptr;
    return false;
  }
  h->act_tiles = tiles;
  return true;
}

}  // namespace

extern "C" int cmbpo_pi_create(cmbpo_pi_t **out, int obs_dim, int hidden, int act_dim) {
  CMBPO_REQUIRE(out != nullptr, "cmbpo_pi_create: out is NULL");
  CMBPO_REQUIRE(hidden == HID || hidden == 256, "cmbpo_pi_create: hidden must be 128 or 256 (got %d)", hidden);
  CMBPO_REQUIRE(obs_dim >= 1 && obs_dim <= 64, "cmbpo_pi_create: obs_dim %d not in [1, 64]", obs_dim);
  CMBPO_REQUIRE(act_dim >= 1 && act_dim <= 32, "cmbpo_pi_create: act_dim %d not in [1, 32]", act_dim);
  cmbpo_pi *h = new (std::nothrow) cmbpo_pi();
  if (!h) { cmbpo_set_error("cmbpo_pi_create: out of host memory"); return CMBPO_ENOMEM; }
  PiDims &d = h->d;
  d.D = obs_dim; d.A = act_dim;
  d.in_pad = (obs_dim + 7) / 8 * 8; d.kg0 = d.in_pad / 8;
  d.a_kpad = (act_dim + 7) / 8 * 8; d.kga = d.a_kpad / 8;
  d.n_it = (obs_dim + 31) / 32;
  d.H = hidden;
  const int H = hidden, NT = H / 32, KH = H / 8;
  d.oW0 = 0; d.ob0 = obs_dim * H; d.oW1 = d.ob0 + H; d.ob1 = d.oW1 + H * H; d.oW2 = d.ob1 + H;
  d.ob2 = d.oW2 + H * act_dim; d.ols = d.ob2 + act_dim; d.P = d.ols + act_dim;
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += (n + 3) / 4 * 4; return o; };
  h->off_F0 = take((size_t)NT * d.kg0 * 256); h->off_F1 = take((size_t)NT * KH * 256); h->off_F2 = take((size_t)KH * 256);
  h->off_B1 = take((size_t)NT * KH * 256); h->off_B2 = take((size_t)NT * d.kga * 256);
  h->off_b0 = take(H); h->off_b1 = take(H); h->off_b2 = take(32); h->off_ls = take(32);
  h->pack_floats = off;
  h->has_params = false;
  h->act = nullptr; h->act_tiles = 0; h->act_keep = false; h->act_valid = false; h->act_obs = nullptr; h->act_n = 0; h->act_hits = 0;
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    cmbpo_set_error("cmbpo_pi_create: cannot query the device");
    delete h;
    return CMBPO_EHIP;
  }
  h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  h->part_ld = (d.P + 63) / 64 * 64;
  h->parts = nullptr;
  h->blob16 = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&h->blob), 2 * off * sizeof(float)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&h->blob16), (size_t)2 * H16_PACK * sizeof(u32x4)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&h->parts), (size_t)2 * h->n_cu * h->part_ld * sizeof(float)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&h->cg_z), (size_t)h->part_ld * sizeof(float)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&h->cg_part), (size_t)(h->part_ld / 64 + h->part_ld / 256 + 4) * sizeof(double)) != hipSuccess) {
    cmbpo_set_error("cmbpo_pi_create: hipMalloc failed");
    if (h->blob) (void)hipFree(h->blob);
    if (h->blob16) (void)hipFree(h->blob16);
    delete h;
    return CMBPO_ENOMEM;
  }
  *out = h;
  return CMBPO_OK;
}

extern "C" void cmbpo_pi_destroy(cmbpo_pi_t *h) {
  if (!h) return;
  cmbpo_pi_cg_release(h);   // a cached CG graph holds this handle's buffers in its kernel arguments
  if (h->blob) (void)hipFree(h->blob);
  if (h->blob16) (void)hipFree(h->blob16);
  if (h->parts) (void)hipFree(h->parts);
  if (h->cg_z) (void)hipFree(h->cg_z);
  if (h->cg_part) (void)hipFree(h->cg_part);
  if (h->act) (void)hipFree(h->act);
  delete h;
}

#ifdef CMBPO_STAMPS
extern "C" void cmbpo_debug_set_pi_stamps(unsigned long long *p) { g_pi_stamps = p; }
#endif
extern "C" void cmbpo_set_pi_matrix_path(int path) { g_pi_path = path != 0 ? 1 : 0; }
extern "C" int cmbpo_get_pi_matrix_path(void) { return pi_path(); }

extern "C" int cmbpo_pi_num_params(const cmbpo_pi_t *h) { return h ? h->d.P : -1; }

extern "C" int cmbpo_pi_set_params(cmbpo_pi_t *h, const float *d_flat, void *stream) {
  CMBPO_REQUIRE(h != nullptr && d_flat != nullptr, "cmbpo_pi_set_params: NULL argument");
  if (int rc = do_pack(h, h->blob, d_flat, (hipStream_t)stream)) return rc;
  h->has_params = true;
  h->act_valid = false;       // saved activations belong to the previous parameters
  return CMBPO_OK;
}

extern "C" int cmbpo_pi_keep_activations(cmbpo_pi_t *h, int enable) {
  CMBPO_REQUIRE(h != nullptr, "cmbpo_pi_keep_activations: NULL handle");
  h->act_keep = enable != 0;
  h->act_valid = false;
  return CMBPO_OK;
}

extern "C" long cmbpo_pi_saved_activation_uses(const cmbpo_pi_t *h) { return h ? h->act_hits : -1; }

const void *cmbpo_pi_act_token(const cmbpo_pi_t *h, const cmbpo_pi_batch_t *b) { return (h && b) ? act_for(h, b) : nullptr; }

extern "C" int cmbpo_pi_loss_grad(cmbpo_pi_t *h, const cmbpo_pi_batch_t *b, int which, float *d_vec, double *d_sums,
                                  void *stream) {
  PiArgs a{};
  if (int rc = fill_args(h, b, a, "cmbpo_pi_loss_grad")) return rc;
  CMBPO_REQUIRE(which == 0 || which == 1, "cmbpo_pi_loss_grad: which must be 0 (pi_loss) or 1 (surr_cost)");
  CMBPO_REQUIRE(b->act && b->adv && b->cadv && b->logp_old && b->cost && d_vec && d_sums, "cmbpo_pi_loss_grad: NULL buffer");
  hipStream_t s = (hipStream_t)stream;
  CMBPO_HIP_CHECK(hipMemsetAsync(d_sums, 0, 8 * sizeof(double), s));
  a.which = which; a.vec = d_vec; a.sums = d_sums;
  const bool save = h->act_keep && act_for(h, b) == nullptr && act_reserve(h, b->n);
  if (save) a.cache_w = reinterpret_cast<f32x4 *>(h->act);
  if (int rc = launch_pi<MODE_GRAD>(h, a, s)) return rc;
  if (save) { h->act_valid = true; h->act_obs = b->obs; h->act_n = b->n; h->act_path = h->d.H == HID ? pi_path() : 0; }
  return CMBPO_OK;
}

extern "C" int cmbpo_pi_fvp(cmbpo_pi_t *h, const cmbpo_pi_batch_t *b, const float *d_v, float *d_vec, void *stream) {
  PiArgs a{};
  if (int rc = fill_args(h, b, a, "cmbpo_pi_fvp")) return rc;
  CMBPO_REQUIRE(b->logstd_old && d_v && d_vec, "cmbpo_pi_fvp: NULL buffer");
  hipStream_t s = (hipStream_t)stream;
  if (int rc = do_pack(h, h->blob + h->pack_floats, d_v, s)) return rc;
  a.vec = d_vec; a.sums = nullptr;
  a.cache_r = act_for(h, b);
  h->act_hits += a.cache_r != nullptr;
  return launch_pi<MODE_FVP>(h, a, s);
}

// One CG iteration on the handle's own partial vectors (single-GPU path of cmbpo_pi_cg_solve): Fisher-vector product of
// direction d_p, then the three vector kernels; d_scal[0] carries r.r (see cg_k1_kernel), d_scal[1] must be < 0 on entry
// to the first iteration (cmbpo_cg_init leaves it there) and cmbpo_pi_cg_commit finishes the hand-over after the last.
extern "C" int cmbpo_pi_cg_iter(cmbpo_pi_t *h, const cmbpo_pi_batch_t *b, double inv_n, float damping, float *d_x,
                                float *d_r, float *d_p, double *d_scal, void *stream) {
  PiArgs a{};
  if (int rc = fill_args(h, b, a, "cmbpo_pi_cg_iter")) return rc;
  CMBPO_REQUIRE(b->logstd_old && d_x && d_r && d_p && d_scal, "cmbpo_pi_cg_iter: NULL buffer");
  hipStream_t s = (hipStream_t)stream;
  if (int rc = do_pack(h, h->blob + h->pack_floats, d_p, s)) return rc;
  a.vec = nullptr; a.sums = nullptr;              // partial vectors only: cg_k1 adds them up
  a.cache_r = act_for(h, b);
  h->act_hits += a.cache_r != nullptr;            // (a captured replay counts once, at capture)
  if (int rc = launch_pi<MODE_FVP>(h, a, s)) return rc;
  const int P = h->d.P, n1 = cmbpo_ceil_div(P, 64), n2 = cmbpo_ceil_div(P, 256);
  double *pzp = h->cg_part, *rrp = h->cg_part + n1;
  hipLaunchKernelGGL(cg_k1_kernel, dim3(n1), dim3(1024), 0, s, h->parts, h->part_ld, h->last_grid, P, (float)inv_n, damping,
                     d_p, h->cg_z, pzp, d_scal);
  hipLaunchKernelGGL(cg_k2_kernel, dim3(n2), dim3(256), 0, s, P, n1, pzp, d_p, h->cg_z, d_x, d_r, rrp, d_scal);
  hipLaunchKernelGGL(cg_k3_kernel, dim3(n2), dim3(256), 0, s, P, n2, rrp, d_r, d_p, d_scal);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

extern "C" int cmbpo_pi_cg_commit(double *d_scal, void *stream) {
  CMBPO_REQUIRE(d_scal != nullptr, "cmbpo_pi_cg_commit: NULL scalars");
  hipLaunchKernelGGL(cg_commit_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d_scal);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

extern "C" int cmbpo_pi_eval(cmbpo_pi_t *h, const cmbpo_pi_batch_t *b, double *d_sums, void *stream) {
  PiArgs a{};
  if (int rc = fill_args(h, b, a, "cmbpo_pi_eval")) return rc;
  CMBPO_REQUIRE(b->act && b->adv && b->cadv && b->logp_old && b->cost && b->mu_old && b->logstd_old && d_sums,
                "cmbpo_pi_eval: NULL buffer");
  hipStream_t s = (hipStream_t)stream;
  CMBPO_HIP_CHECK(hipMemsetAsync(d_sums, 0, 8 * sizeof(double), s));
  a.vec = nullptr; a.sums = d_sums;
  return launch_pi<MODE_EVAL>(h, a, s);
}
