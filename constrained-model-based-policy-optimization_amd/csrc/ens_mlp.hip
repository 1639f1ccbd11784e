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
#include "mfma_tile.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = 4;

struct MlpKernelArgs {
  // packed weights, float4 units
  const f32x4 *wp0, *wp1, *wp2;
  size_t wp0_stride, wp1_stride, wp2_stride;  // per member, in float4
  const float *b0, *b1, *b2;                  // [E][HID],[E][HID],[E][o_pad]
  const float *in_mu, *in_var;                // [in_dim] or nullptr
  const float *out_mu, *out_var;              // [out_dim] or nullptr
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
};

template <int ACT>
__device__ __forceinline__ float activate(float x) {
  if constexpr (ACT == CMBPO_ACT_SWISH) {
    // x * sigmoid(x), models/pens/fc.py:19
    return x / (1.0f + __expf(-x));
  } else {
    return tanhf(x);
  }
}

// bias + activation, accumulator tile -> LDS image [n/4][BB] of float4.
template <int NT, int BT, int ACT>
__device__ __forceinline__ void store_hidden(const f32x16 (&acc)[NT][BT],
                                             const float *__restrict__ bias,
                                             int n_base, f32x4 *lds_out, int lane) {
  constexpr int BB = 32 * BT;
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n0 = n_base + t * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + 8 * q + 4 * h;
      const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + n);
#pragma unroll
      for (int bt = 0; bt < BT; ++bt) {
        f32x4 v;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = activate<ACT>(acc[t][bt][4 * q + s] + bv[s]);
        lds_out[(n >> 2) * BB + bt * 32 + j] = v;
      }
    }
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
  int *rows = reinterpret_cast<int *>(xbuf + p.in_pad / 4 * BB);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_rows = p.n_rows_dev ? *p.n_rows_dev : p.n_rows;
  const int row0 = blockIdx.x * BB;
  if (row0 >= n_rows) return;

  // ---- stage the (scaled) input tile: xT[k][b], k = [obs | act] ------------
  if (tid < BB) {
    const int r = row0 + tid;
    rows[tid] = (r < n_rows) ? (p.row_idx ? p.row_idx[r] : r) : -1;
  }
  {
    float *xf = reinterpret_cast<float *>(xbuf);
    // zero the k padding [in_dim, in_pad)
    const int npad = (p.in_pad - p.in_dim) * BB;
    for (int i = tid; i < npad; i += kThreads) {
      const int b = i % BB, k = p.in_dim + i / BB;
      xf[((k >> 2) * BB + b) * 4 + (k & 3)] = 0.0f;
    }
  }
  __syncthreads();
  {
    float *xf = reinterpret_cast<float *>(xbuf);
    const int n_in = p.in_dim * BB;
    for (int i = tid; i < n_in; i += kThreads) {
      const int b = i / p.in_dim, k = i - b * p.in_dim;
      const int r = rows[b];
      float v = 0.0f;
      if (r >= 0) {
        v = (k < p.obs_dim) ? p.obs[(size_t)r * p.obs_dim + k]
                            : p.act[(size_t)r * p.act_dim + (k - p.obs_dim)];
        if (p.in_mu) {
          // TensorStandardScaler.transform, models/pens/utils.py:156
          const float sig = fmaxf(sqrtf(p.in_var[k]), 1e-2f);
          v = (v - p.in_mu[k]) / sig;
        }
      }
      xf[((k >> 2) * BB + b) * 4 + (k & 3)] = v;
    }
  }
  __syncthreads();

  const int e_begin = blockIdx.y * p.e_chunk;
  const int e_end = min(p.ensemble, e_begin + p.e_chunk);
  float member_sum = 0.0f;  // HEAD_DETMEAN: running sum over members

  for (int e = e_begin; e < e_end; ++e) {
    const int kg0 = p.in_pad / 8;
    // ---- layer 0: in -> HID ------------------------------------------------
    {
      f32x16 acc[NT][BT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int bt = 0; bt < BT; ++bt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][bt][r] = 0.0f;
      const f32x4 *wp = p.wp0 + e * p.wp0_stride + (size_t)(wave * NT) * kg0 * 64;
      mfma_layer<NT, BT>(wp, (size_t)kg0 * 64, 0, kg0, xbuf, lane, acc);
      // hbuf may still be read (as `red`) by the previous member's epilogue
      __syncthreads();
      store_hidden<NT, BT, ACT>(acc, p.b0 + (size_t)e * HID, wave * NT * 32, hbuf, lane);
    }
    __syncthreads();
    // ---- layer 1: HID -> HID -----------------------------------------------
    {
      f32x16 acc[NT][BT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int bt = 0; bt < BT; ++bt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][bt][r] = 0.0f;
      const f32x4 *wp = p.wp1 + e * p.wp1_stride + (size_t)(wave * NT) * KG_H * 64;
      mfma_layer<NT, BT>(wp, (size_t)KG_H * 64, 0, KG_H, hbuf, lane, acc);
      __syncthreads();  // every wave has finished reading h1
      store_hidden<NT, BT, ACT>(acc, p.b1 + (size_t)e * HID, wave * NT * 32, hbuf, lane);
    }
    __syncthreads();
    // ---- layer 2: HID -> o_width, K split over the 4 waves -------------------
    {
      constexpr int KGQ = KG_H / kWaves;
      f32x16 part[4][BT];  // o_tiles <= 4, statically indexed below
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) {
        if (ot < p.o_tiles) {
          f32x16 acc[1][BT];
#pragma unroll
          for (int bt = 0; bt < BT; ++bt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][bt][r] = 0.0f;
          const f32x4 *wp = p.wp2 + e * p.wp2_stride + (size_t)ot * KG_H * 64;
          mfma_layer<1, BT>(wp, 0, wave * KGQ, (wave + 1) * KGQ, hbuf, lane, acc);
#pragma unroll
          for (int bt = 0; bt < BT; ++bt) part[ot][bt] = acc[0][bt];
        }
      }
      __syncthreads();  // h2 dead: hbuf becomes the reduction image
      const int j = lane & 31, h = lane >> 5;
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) {
        if (ot < p.o_tiles) {
#pragma unroll
          for (int bt = 0; bt < BT; ++bt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int n = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
              red[(wave * p.o_tiles * 32 + n) * RED_LD + bt * 32 + j] = part[ot][bt][r];
            }
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
      return v + p.b2[(size_t)e * o_ld + n];
    };
    if constexpr (HEAD == CMBPO_HEAD_PROB) {
      // mean = sig*o + mu ; var = exp(2*log(sig) + o'), models/pens/pe.py:815-835
      const int out = p.out_dim;
      for (int i = tid; i < BB * out; i += kThreads) {
        const int b = i / out, n = i - b * out;
        const int r = rows[b];
        if (r < 0) continue;
        float m = reduced(b, n);
        float lv = reduced(b, out + n);
        if (p.out_mu) {
          const float sig = fmaxf(sqrtf(p.out_var[n]), 1e-2f);
          m = sig * m + p.out_mu[n];
          lv = 2.0f * logf(sig) + lv;
        }
        const size_t o = ((size_t)e * p.ld_rows + r) * out + n;
        p.out0[o] = m;
        p.out1[o] = expf(lv);
      }
    } else if constexpr (HEAD == CMBPO_HEAD_DETMEAN) {
      const int out = p.out_dim;
      if (tid < BB * out) {
        const int b = tid / out, n = tid - b * out;
        float m = reduced(b, n);
        if (p.out_mu) {
          const float sig = fmaxf(sqrtf(p.out_var[n]), 1e-2f);
          m = sig * m + p.out_mu[n];
        }
        member_sum += m;
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

struct cmbpo_mlp {
  int ensemble, in_dim, in_pad, hidden, o_width, o_tiles, out_dim, act, head;
  bool loaded, has_in_scaler, has_out_scaler;
  float *d_blob;  // one allocation holding everything below
  size_t blob_floats;
  // offsets (in floats) into the blob
  size_t off_wp0, off_wp1, off_wp2, off_b0, off_b1, off_b2;
  size_t off_in_mu, off_in_var, off_out_mu, off_out_var, off_log_std;
  std::vector<float> h_blob;
};

extern "C" int cmbpo_mlp_create(cmbpo_mlp_t **out, int ensemble, int in_dim, int hidden,
                                int out_width, int activation, int head) {
  CMBPO_REQUIRE(out != nullptr, "cmbpo_mlp_create: out is NULL");
  CMBPO_REQUIRE(ensemble >= 1 && ensemble <= 64, "cmbpo_mlp_create: ensemble %d out of range", ensemble);
  CMBPO_REQUIRE(in_dim >= 1 && in_dim <= 256, "cmbpo_mlp_create: in_dim %d out of range", in_dim);
  CMBPO_REQUIRE(hidden == 128 || hidden == 512, "cmbpo_mlp_create: hidden must be 128 or 512 (got %d)", hidden);
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
  m->off_out_mu = take(m->out_dim); m->off_out_var = take(m->out_dim);
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
  if (h_in_mu) {
    memcpy(hb + m->off_in_mu, h_in_mu, I * sizeof(float));
    memcpy(hb + m->off_in_var, h_in_var, I * sizeof(float));
  }
  if (h_out_mu) {
    memcpy(hb + m->off_out_mu, h_out_mu, m->out_dim * sizeof(float));
    memcpy(hb + m->off_out_var, h_out_var, m->out_dim * sizeof(float));
  }
  if (h_log_std) memcpy(hb + m->off_log_std, h_log_std, m->out_dim * sizeof(float));
  // h_blob stays alive in the handle until the next load, so the async copy is safe.
  CMBPO_HIP_CHECK(hipMemcpyAsync(m->d_blob, hb, m->blob_floats * sizeof(float),
                                 hipMemcpyHostToDevice, (hipStream_t)stream));
  m->loaded = true;
  return CMBPO_OK;
}

namespace {

template <int HID, int BT, int ACT, int HEAD>
int launch_one(const MlpKernelArgs &a, int tiles, int grid_y, size_t lds, hipStream_t s) {
  auto kern = ens_mlp_kernel<HID, BT, ACT, HEAD>;
  static bool attr_set = false;
  if (!attr_set) {
    CMBPO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles, grid_y), dim3(kThreads), lds, s, a);
  CMBPO_HIP_CHECK(hipGetLastError());
  return CMBPO_OK;
}

int g_block_rows = 32;  // 32 (2 workgroups / CU) or 64 (1 workgroup / CU)

int launch_mlp(cmbpo_mlp *m, MlpKernelArgs &a, hipStream_t s) {
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
  a.in_var = m->has_in_scaler ? blob + m->off_in_var : nullptr;
  a.out_mu = m->has_out_scaler ? blob + m->off_out_mu : nullptr;
  a.out_var = m->has_out_scaler ? blob + m->off_out_var : nullptr;
  a.log_std = blob + m->off_log_std;
  a.ensemble = E;
  a.e_chunk = (m->head == CMBPO_HEAD_PROB) ? 1 : E;
  a.in_dim = m->in_dim; a.in_pad = m->in_pad;
  a.o_width = m->o_width; a.o_tiles = m->o_tiles; a.out_dim = m->out_dim;
  if (a.n_rows <= 0) return CMBPO_OK;

  const int BT = (H == 512 && g_block_rows == 64) ? 2 : 1;
  const int BB = 32 * BT;
  const int tiles = cmbpo_ceil_div(a.n_rows, BB);
  const int grid_y = cmbpo_ceil_div(E, a.e_chunk);
  const size_t hbuf = (size_t)H * BB * 4;
  const size_t red = ((size_t)kWaves * m->o_tiles * 32 * (BB + 1) * 4 + 15) / 16 * 16;
  const size_t lds = (hbuf > red ? hbuf : red) + (size_t)m->in_pad * BB * 4 + BB * 4;
  CMBPO_REQUIRE(lds <= 160 * 1024, "ens_mlp: LDS budget exceeded (%zu B)", lds);

#define CMBPO_LAUNCH(HID_, BT_, ACT_, HEAD_) \
  return launch_one<HID_, BT_, ACT_, HEAD_>(a, tiles, grid_y, lds, s)
  if (m->head == CMBPO_HEAD_PROB && m->act == CMBPO_ACT_SWISH) {
    if (H == 512 && BT == 1) CMBPO_LAUNCH(512, 1, CMBPO_ACT_SWISH, CMBPO_HEAD_PROB);
    if (H == 512 && BT == 2) CMBPO_LAUNCH(512, 2, CMBPO_ACT_SWISH, CMBPO_HEAD_PROB);
    if (H == 128) CMBPO_LAUNCH(128, 1, CMBPO_ACT_SWISH, CMBPO_HEAD_PROB);
  } else if (m->head == CMBPO_HEAD_DETMEAN && m->act == CMBPO_ACT_SWISH) {
    if (H == 128) CMBPO_LAUNCH(128, 1, CMBPO_ACT_SWISH, CMBPO_HEAD_DETMEAN);
    if (H == 512) CMBPO_LAUNCH(512, 1, CMBPO_ACT_SWISH, CMBPO_HEAD_DETMEAN);
  } else if (m->head == CMBPO_HEAD_GAUSS_PI && m->act == CMBPO_ACT_TANH) {
    if (H == 128) CMBPO_LAUNCH(128, 1, CMBPO_ACT_TANH, CMBPO_HEAD_GAUSS_PI);
    if (H == 512) CMBPO_LAUNCH(512, 1, CMBPO_ACT_TANH, CMBPO_HEAD_GAUSS_PI);
  }
#undef CMBPO_LAUNCH
  cmbpo_set_error("ens_mlp: no kernel for hidden=%d act=%d head=%d", H, m->act, m->head);
  return CMBPO_EINVAL;
}

}  // namespace

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
