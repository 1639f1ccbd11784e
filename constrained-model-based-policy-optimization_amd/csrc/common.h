// Shared helpers for the gfx950 C-ABI library (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/cmbpo_hip.h"

void cmbpo_set_error(const char *fmt, ...);
// the saved-activation block a Fisher-vector product on this batch would read (NULL: recompute); part of the key of a
// captured CG graph, whose kernel arguments contain it
const void *cmbpo_pi_act_token(const cmbpo_pi_t *h, const cmbpo_pi_batch_t *b);

#define CMBPO_HIP_CHECK(expr)                                                  \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      cmbpo_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                      __FILE__, __LINE__);                                     \
      return CMBPO_EHIP;                                                       \
    }                                                                          \
  } while (0)

#define CMBPO_REQUIRE(cond, ...)                                               \
  do {                                                                         \
    if (!(cond)) {                                                             \
      cmbpo_set_error(__VA_ARGS__);                                            \
      return CMBPO_EINVAL;                                                     \
    }                                                                          \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int cmbpo_ceil_div(int a, int b) { return (a + b - 1) / b; }

// tanh as 1 - 2 / (1 + e^{2x}) with the hardware exp / rcp (absolute error ~1e-7, saturates correctly at +-1; the libm
// tanhf is ~30 VALU instructions per value, and VALU work next to a partner wave's fp32 MFMAs is the scarce resource of
// the 128-wide kernels' epilogues)
__device__ __forceinline__ float cmbpo_fast_tanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
}
